// Training-side HBM-bound kernels of libw2l_hip.so: BatchNorm in batch-statistics mode (forward + backward),
// activation backward, channel sums (bias gradients), loss forward/backward pairs and a fused multi-tensor Adam.
//
// Replaces, on the reference's training loops (wav2lip_train.py:201-262, color_syncnet_train.py:140-190,
// hq_wav2lip_train.py:204-310), the torch autograd nodes of: nn.BatchNorm2d in train mode (models/conv.py:10,40),
// ReLU / LeakyReLU / Sigmoid (models/conv.py:12,27,43; models/wav2lip.py:85,152), nn.L1Loss (wav2lip_train.py:191),
// cosine_similarity + BCELoss (wav2lip_train.py:179-184), F.normalize (models/syncnet.py:62-63),
// F.binary_cross_entropy (models/wav2lip.py:171) and optim.Adam (wav2lip_train.py:359).
//
// All tensors are NHWC fp32 "[rows][cs]" views (rows = N*H*W pixels, C valid channels, cs channel stride).
// Column reductions accumulate in fp64 per thread, combine per workgroup through LDS and finish in a second
// single-workgroup-per-channel-block kernel in a fixed order: deterministic, no atomics.
#include <math.h>
#include <mutex>
#include <new>
#include <vector>

#include "w2l_common.h"

namespace w2l {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_cap(long long work, int block, int cap) {
    long long g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

template <int ACT>
__device__ __forceinline__ float act_grad_from_y(float y) {
    if (ACT == W2L_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (ACT == W2L_ACT_LEAKY) return y > 0.f ? 1.f : 0.01f;
    if (ACT == W2L_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}
// Branch-free forms for the row loops (see train_bf16.hip: a switch per element on the wave-uniform `act` compiles to a scalar
// branch per element).  The gradient is a select of constants (or y(1-y) for the sigmoid); the forward is act_leaky
// (w2l_common.h): a select on v < 0, equal to the switch forms on every input including NaN and +-inf.
struct ActK {
    float neg;
    unsigned sigmask;
};
__device__ __forceinline__ ActK act_consts(int act) {
    ActK k;
    k.neg = act == W2L_ACT_RELU ? 0.f : (act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    k.sigmask = act == W2L_ACT_SIGMOID ? 0xffffffffu : 0u;
    return k;
}
__device__ __forceinline__ float act_grad_k(const ActK k, float y) {
    const float gr = y > 0.f ? 1.f : k.neg;
    const float gs = y * (1.f - y);
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, gs) & k.sigmask) | (__builtin_bit_cast(unsigned, gr) & ~k.sigmask));
}
template <int NE>
__device__ __forceinline__ void act_fwd_n(const ActK k, float* v) {
    if (k.sigmask) {
#pragma unroll
        for (int e = 0; e < NE; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) v[e] = act_leaky(v[e], k.neg);
    }
}

// ---------------------------------------------------------------- column reductions
enum ColMode { kColStats = 0, kColBnBwd = 1, kColSum = 2 };

struct ColArgs {
    const float* a;     // stats: z;  bn_bwd: dy;  sum: the tensor
    const float* y;     // bn_bwd: block output (activation mask)
    const float* z;     // bn_bwd: pre-BN conv output
    const float* mean;  // bn_bwd
    const float* rstd;  // bn_bwd
    double* partial;    // [nblocks][2][C]
    long long rows;
    int C, a_cs, y_cs, z_cs, act;
    int rows_per_block;
};

// thread -> (float4 column group c4 = t % CG, row lane t / CG); needs C % 4 == 0 and C <= 1024
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const ColArgs a) {
    __shared__ double red[256][8];
    const int CG = a.C >> 2;
    const int RPP = 256 / CG;
    const int t = threadIdx.x;
    const int c4 = t % CG;
    const int rl = t / CG;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (rl < RPP) {
        const long long r0 = (long long)blockIdx.x * a.rows_per_block;
        const long long r1 = r0 + a.rows_per_block < a.rows ? r0 + a.rows_per_block : a.rows;
        f32x4 mu = {0, 0, 0, 0}, rs = {0, 0, 0, 0};
        if (MODE == kColBnBwd) {
            mu = *reinterpret_cast<const f32x4*>(a.mean + c4 * 4);
            rs = *reinterpret_cast<const f32x4*>(a.rstd + c4 * 4);
        }
        for (long long r = r0 + rl; r < r1; r += RPP) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.a + r * a.a_cs + c4 * 4);
            if (MODE == kColStats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += (double)v[e]; s1[e] += (double)v[e] * (double)v[e]; }
            } else if (MODE == kColSum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s0[e] += (double)v[e];
            } else {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(a.y + r * a.y_cs + c4 * 4);
                const f32x4 zv = *reinterpret_cast<const f32x4*>(a.z + r * a.z_cs + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = v[e] * act_grad_k(act_consts(a.act), yv[e]);
                    const float zh = (zv[e] - mu[e]) * rs[e];
                    s0[e] += (double)g;
                    s1[e] += (double)g * (double)zh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[t][e] = s0[e]; red[t][4 + e] = s1[e]; }
    __syncthreads();
    if (t < CG) {
        double o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < RPP; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += red[t + j * CG][e];
        double* dst = a.partial + (long long)blockIdx.x * 2 * a.C;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dst[t * 4 + e] = o[e]; dst[a.C + t * 4 + e] = o[4 + e]; }
    }
}

struct ColFinalArgs {
    const double* partial;
    int nblocks, C;
    long long rows;
    // stats
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* mean;
    float* rstd;
    float* scale;       // gamma*rstd
    float* shift;       // beta - mean*gamma*rstd
    float* running_mean;
    float* running_var;
    // bn_bwd / sum
    float* out0;        // sum of g (d beta) / column sum
    float* out1;        // sum of g*zhat (d gamma)
};

// 64 channels per workgroup; the partial blocks are split 4 ways across the waves and summed with 4 independent loads in
// flight per thread (a serial walk over ~1000 partials costs >100 us of pure latency), then combined through LDS in a
// fixed order
template <int MODE>
__global__ __launch_bounds__(256) void col_final_kernel(const ColFinalArgs a) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s0 = 0, s1 = 0;
    if (c < a.C) {
        const long long st = 2ll * a.C;
        const double* p = a.partial + c;
        double t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
        int b = part;
        for (; b + 12 < a.nblocks; b += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t0[u] += p[(long long)(b + 4 * u) * st];
                if (MODE != kColSum) t1[u] += p[(long long)(b + 4 * u) * st + a.C];
            }
        }
        for (; b < a.nblocks; b += 4) {
            t0[0] += p[(long long)b * st];
            if (MODE != kColSum) t1[0] += p[(long long)b * st + a.C];
        }
        s0 = (t0[0] + t0[1]) + (t0[2] + t0[3]);
        s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
    }
    red[0][part][cl] = s0;
    red[1][part][cl] = s1;
    __syncthreads();
    if (part != 0 || c >= a.C) return;
    s0 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    s1 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    if (MODE == kColStats) {
        const double m = s0 / (double)a.rows;
        double var = s1 / (double)a.rows - m * m;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float mf = (float)m;
        a.mean[c] = mf;
        a.rstd[c] = rstd;
        const float sc = (a.gamma ? a.gamma[c] : 1.f) * rstd;
        a.scale[c] = sc;
        a.shift[c] = (a.beta ? a.beta[c] : 0.f) - mf * sc;
        if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mf;
        if (a.running_var) {
            const double unb = a.rows > 1 ? var * (double)a.rows / (double)(a.rows - 1) : var;
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
        }
    } else {
        if (a.out0) a.out0[c] = (float)s0;
        if (a.out1) a.out1[c] = (float)s1;
    }
}

// fp64 scratch for the reduction partials, one fixed 16 MiB buffer PER STREAM (covers 512 workgroups x 1024 channels x 2
// sums and the 1024 L1 partials): stream-ordered reuse, no sharing between streams
struct PartialWs {
    hipStream_t stream;
    double* ptr;
};
static std::mutex g_partial_mutex;
static std::vector<PartialWs> g_partial_table;
constexpr size_t kPartialBytes = (size_t)16 << 20;
static double* partial_ws(hipStream_t stream, size_t bytes) {
    if (bytes > kPartialBytes) { set_error("reduction scratch request of %zu bytes exceeds the fixed buffer", bytes); return nullptr; }
    std::lock_guard<std::mutex> lock(g_partial_mutex);
    for (const PartialWs& w : g_partial_table)
        if (w.stream == stream) return w.ptr;
    double* p = nullptr;
    if (hipMalloc(&p, kPartialBytes) != hipSuccess) { set_error("hipMalloc(reduction scratch) failed"); return nullptr; }
    g_partial_table.push_back(PartialWs{stream, p});
    return p;
}

static int col_check(long long rows, int C, const float* p, int cs, const char* what) {
    W2L_REQUIRE(rows >= 1 && C >= 4 && (C & 3) == 0 && C <= 1024, "%s: C=%d must be a multiple of 4 in [4, 1024]", what, C);
    W2L_REQUIRE(p && cs >= C && (cs & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0,
                "%s: tensor must be 16-byte aligned with a channel stride that is a multiple of 4 (cs=%d)", what, cs);
    return W2L_OK;
}

template <int MODE>
static int col_reduce_launch(ColArgs a, ColFinalArgs f, hipStream_t s) {
    const int CG = a.C >> 2;
    const int RPP = 256 / CG;
    long long per = (a.rows + 511) / 512;            // at most 512 workgroups (2 per CU)
    const long long min_rows = (long long)RPP * 16;  // at least 16 rows per thread
    if (per < min_rows) per = min_rows;
    a.rows_per_block = (int)per;
    const int nblocks = (int)((a.rows + per - 1) / per);
    a.partial = partial_ws(s, (size_t)nblocks * 2 * a.C * sizeof(double));
    if (!a.partial) return W2L_ERR_NOMEM;
    hipLaunchKernelGGL(col_reduce_kernel<MODE>, dim3(nblocks), dim3(256), 0, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    f.partial = a.partial;
    f.nblocks = nblocks;
    f.C = a.C;
    f.rows = a.rows;
    hipLaunchKernelGGL(col_final_kernel<MODE>, dim3(ceil_div(a.C, 64)), dim3(256), 0, s, f);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---------------------------------------------------------------- elementwise over [rows][C]
struct EwArgs {
    const float* a;      // affine: z;            bn_bwd_apply: dy;      act_bwd: dy
    const float* b;      // affine: res or NULL;  bn_bwd_apply: y;       act_bwd: y
    const float* c;      //                       bn_bwd_apply: z
    float* out;          // affine: y;            bn_bwd_apply: dz;      act_bwd: dz
    float* out2;         //                       in-place g (= masked dy) or NULL
    const float* v0;     // per-channel: affine scale;   bwd: scale_eff;   act_bwd: scale or NULL
    const float* v1;     // per-channel: affine shift;   bwd: mean
    const float* v2;     //                              bwd: rstd
    const float* v3;     //                              bwd: sum_g
    const float* v4;     //                              bwd: sum_gz
    long long rows;
    int C, a_cs, b_cs, c_cs, out_cs, out2_cs, act;
    float inv_rows;
};

enum EwMode { kEwAffine = 0, kEwBnBwd = 1, kEwActBwd = 2, kEwAdd = 3 };

// thread -> (float4 column group t % CG, row lane t / CG): the per-channel vectors are loaded once per thread
template <int MODE>
__global__ __launch_bounds__(256) void ew_kernel(const EwArgs a) {
    const int CG = a.C >> 2;
    const int RPP = 256 / CG;
    const int c = (threadIdx.x % CG) * 4;
    const int rl = threadIdx.x / CG;
    if (rl >= RPP) return;
    f32x4 v0 = {1, 1, 1, 1}, v1 = {0, 0, 0, 0}, v2 = v1, v3 = v1, v4 = v1;
    if (a.v0) v0 = *reinterpret_cast<const f32x4*>(a.v0 + c);
    if (a.v1) v1 = *reinterpret_cast<const f32x4*>(a.v1 + c);
    if (a.v2) v2 = *reinterpret_cast<const f32x4*>(a.v2 + c);
    if (a.v3) v3 = *reinterpret_cast<const f32x4*>(a.v3 + c);
    if (a.v4) v4 = *reinterpret_cast<const f32x4*>(a.v4 + c);
    for (long long r = (long long)blockIdx.x * RPP + rl; r < a.rows; r += (long long)gridDim.x * RPP) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a.a + r * a.a_cs + c);
        f32x4 o;
        if (MODE == kEwAffine) {       // v0 scale, v1 shift
            f32x4 rv = {0, 0, 0, 0};
            if (a.b) rv = *reinterpret_cast<const f32x4*>(a.b + r * a.b_cs + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = av[e] * v0[e] + v1[e] + rv[e];
            {
                float ov[4] = {o[0], o[1], o[2], o[3]};
                act_fwd_n<4>(act_consts(a.act), ov);
                o = f32x4{ov[0], ov[1], ov[2], ov[3]};
            }
        } else if (MODE == kEwBnBwd) {  // v0 gamma*rstd, v1 mean, v2 rstd, v3 sum g, v4 sum g*zhat
            const f32x4 yv = *reinterpret_cast<const f32x4*>(a.b + r * a.b_cs + c);
            const f32x4 zv = *reinterpret_cast<const f32x4*>(a.c + r * a.c_cs + c);
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[e] = av[e] * act_grad_k(act_consts(a.act), yv[e]);
                const float zh = (zv[e] - v1[e]) * v2[e];
                o[e] = v0[e] * (g[e] - v3[e] * a.inv_rows - zh * (v4[e] * a.inv_rows));
            }
            if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + r * a.out2_cs + c) = g;
        } else if (MODE == kEwActBwd) {  // v0 scale (or ones)
            f32x4 yv = {1, 1, 1, 1};
            if (a.b) yv = *reinterpret_cast<const f32x4*>(a.b + r * a.b_cs + c);
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[e] = av[e] * act_grad_k(act_consts(a.act), yv[e]);
                o[e] = g[e] * v0[e];
            }
            if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + r * a.out2_cs + c) = g;
        } else {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b + r * a.b_cs + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = av[e] + bv[e];
        }
        *reinterpret_cast<f32x4*>(a.out + r * a.out_cs + c) = o;
    }
}

template <int MODE>
static int ew_launch(const EwArgs& a, hipStream_t s) {
    const int RPP = 256 / (a.C >> 2);
    hipLaunchKernelGGL(ew_kernel<MODE>, dim3(grid_cap(a.rows, RPP * 4, 16384)), dim3(256), 0, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---------------------------------------------------------------- losses
// stage 1 of a deterministic mean: per-workgroup fp64 partial of sum |a-b|
__global__ __launch_bounds__(256) void l1_partial_kernel(long long n, const float* __restrict__ a,
                                                         const float* __restrict__ b, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s += (double)fabsf(a[i] - b[i]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one wave: lane l sums partials l, l + 64, ... (four in flight), then a fixed-order xor fold - deterministic, and 3 us instead of the
// 57 us one thread needed to walk ~2 000 partials (it sits on the critical path between the generator's forward and backward)
__global__ void mean_final_kernel(int nblocks, const double* __restrict__ partial, double inv_n, float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int l = threadIdx.x;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int i = l;
    for (; i + 192 < nblocks; i += 256) {
        t0 += partial[i];
        t1 += partial[i + 64];
        t2 += partial[i + 128];
        t3 += partial[i + 192];
    }
    for (; i < nblocks; i += 64) t0 += partial[i];
    double s = (t0 + t1) + (t2 + t3);
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (l == 0) out[0] = (float)(s * inv_n);
}
// d/da mean|a-b| = sign(a-b)/n, scaled by the upstream gradient gout[0] (device scalar)
__global__ void l1_bwd_kernel(long long n, const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ gout, float inv_n, float* __restrict__ da) {
    const float gsc = (gout ? gout[0] : 1.f) * inv_n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        da[i] = d > 0.f ? gsc : (d < 0.f ? -gsc : 0.f);
    }
}

// one wave per row: gradient of mean BCE(cos(a,v), y) w.r.t. a and v
__global__ void cosine_bce_bwd_kernel(int N, int C, const float* __restrict__ a, const float* __restrict__ v,
                                      const float* __restrict__ y, const float* __restrict__ gout,
                                      float* __restrict__ da, float* __restrict__ dv) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* pa = a + (long long)row * C;
    const float* pv = v + (long long)row * C;
    float dot = 0.f, na = 0.f, nv = 0.f;
    for (int c = lane; c < C; c += 64) {
        dot += pa[c] * pv[c];
        na += pa[c] * pa[c];
        nv += pv[c] * pv[c];
    }
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o);
        na += __shfl_xor(na, o);
        nv += __shfl_xor(nv, o);
    }
    const float prod = na * nv;
    const bool clamped = prod < 1e-16f;
    const float den = sqrtf(fmaxf(prod, 1e-16f));
    const float cs = dot / den;
    // BCE backward (ATen): (p - y) / max((1 - p) * p, 1e-12) * grad / N
    const float dcos = (gout ? gout[0] : 1.f) / (float)N * (cs - y[row]) / fmaxf((1.f - cs) * cs, 1e-12f);
    // cos = dot / sqrt(na*nv):  d/da = v/den - dot * nv * a / den^3   (the second term vanishes under the eps clamp)
    const float k1 = dcos / den;
    const float ka = clamped ? 0.f : dcos * dot * nv / (den * den * den);
    const float kv = clamped ? 0.f : dcos * dot * na / (den * den * den);
    for (int c = lane; c < C; c += 64) {
        da[(long long)row * C + c] = k1 * pv[c] - ka * pa[c];
        dv[(long long)row * C + c] = k1 * pa[c] - kv * pv[c];
    }
}

// y = x / max(|x|, 1e-12):  dx = (dy - y * <y, dy>) / max(|x|, 1e-12)
__global__ void l2norm_bwd_kernel(int N, int C, const float* __restrict__ x, int x_cs, const float* __restrict__ dy,
                                  float* __restrict__ dx, int dx_cs) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* px = x + (long long)row * x_cs;
    const float* pd = dy + (long long)row * C;
    float ss = 0.f, xd = 0.f;
    for (int c = lane; c < C; c += 64) {
        ss += px[c] * px[c];
        xd += px[c] * pd[c];
    }
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o);
        xd += __shfl_xor(xd, o);
    }
    const float nrm = sqrtf(ss);
    const float d = fmaxf(nrm, 1e-12f);
    const float k = nrm > 1e-12f ? xd / (d * d * d) : 0.f;   // under the clamp the norm is a constant
    for (int c = lane; c < C; c += 64) dx[(long long)row * dx_cs + c] = pd[c] / d - px[c] * k;
}

__global__ void bce_bwd_kernel(int N, const float* __restrict__ p, const float* __restrict__ y,
                               const float* __restrict__ gout, float* __restrict__ dp) {
    const float gsc = (gout ? gout[0] : 1.f) / (float)N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        dp[i] = gsc * (p[i] - y[i]) / fmaxf((1.f - p[i]) * p[i], 1e-12f);
}

// ---------------------------------------------------------------- fused multi-tensor Adam
struct AdamChunk {
    int tensor;
    int start;   // element offset of the chunk inside the tensor (chunks are kAdamChunk elements)
};
constexpr int kAdamChunk = 16384;

__global__ __launch_bounds__(256) void adam_kernel(const w2l_adam_tensor* __restrict__ tensors,
                                                   const AdamChunk* __restrict__ chunks, float lr, float beta1,
                                                   float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt) {
    const AdamChunk ch = chunks[blockIdx.x];
    const w2l_adam_tensor tt = tensors[ch.tensor];
    const long long end = (long long)ch.start + kAdamChunk < tt.n ? (long long)ch.start + kAdamChunk : tt.n;
    const float step_size = lr / bc1;
    for (long long i = ch.start + threadIdx.x; i < end; i += blockDim.x) {
        float g = tt.grad[i];
        const float p = tt.param[i];
        if (weight_decay != 0.f) g += weight_decay * p;
        // torch.optim.Adam (single-tensor form): exp_avg.lerp_(grad, 1-beta1); exp_avg_sq = beta2*v + (1-beta2) g^2;
        // denom = sqrt(v)/sqrt(bias_correction2) + eps; p -= (lr/bias_correction1) * m / denom
        const float m = tt.exp_avg[i] + (g - tt.exp_avg[i]) * (1.f - beta1);
        const float v = beta2 * tt.exp_avg_sq[i] + (1.f - beta2) * g * g;
        tt.exp_avg[i] = m;
        tt.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        tt.param[i] = p - step_size * (m / denom);
    }
}

}  // namespace w2l

using namespace w2l;

struct w2l_adam {
    int ntensors = 0;
    int nchunks = 0;
    w2l_adam_tensor* tensors_dev = nullptr;
    AdamChunk* chunks_dev = nullptr;
};

extern "C" {

int w2l_bn_train_stats(void* stream, long long rows, int C, const float* z, int z_cs, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* mean, float* rstd, float* scale, float* shift) {
    if (col_check(rows, C, z, z_cs, "bn_train_stats") != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(mean && rstd && scale && shift, "bn_train_stats: NULL output");
    ColArgs a = {};
    a.a = z; a.a_cs = z_cs; a.rows = rows; a.C = C;
    ColFinalArgs f = {};
    f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.mean = mean; f.rstd = rstd; f.scale = scale; f.shift = shift;
    f.running_mean = running_mean; f.running_var = running_var;
    return col_reduce_launch<kColStats>(a, f, static_cast<hipStream_t>(stream));
}

int w2l_affine_act(void* stream, long long rows, int C, const float* z, int z_cs, const float* scale,
                   const float* shift, const float* res, int res_cs, int act, float* y, int y_cs) {
    if (col_check(rows, C, z, z_cs, "affine_act z") != W2L_OK || col_check(rows, C, y, y_cs, "affine_act y") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(scale && shift, "affine_act: NULL scale/shift");
    W2L_REQUIRE(res == nullptr || col_check(rows, C, res, res_cs, "affine_act res") == W2L_OK, "affine_act: bad residual");
    EwArgs a = {};
    a.a = z; a.a_cs = z_cs; a.b = res; a.b_cs = res_cs; a.out = y; a.out_cs = y_cs;
    a.v0 = scale; a.v1 = shift; a.rows = rows; a.C = C; a.act = act;
    return ew_launch<kEwAffine>(a, static_cast<hipStream_t>(stream));
}

int w2l_bn_train_bwd(void* stream, long long rows, int C, const float* dy, int dy_cs, const float* y, int y_cs,
                     const float* z, int z_cs, int act, const float* mean, const float* rstd, const float* scale,
                     float* dgamma, float* dbeta, float* dz, int dz_cs, float* g_out, int g_cs) {
    if (col_check(rows, C, dy, dy_cs, "bn_train_bwd dy") != W2L_OK || col_check(rows, C, y, y_cs, "bn_train_bwd y") != W2L_OK ||
        col_check(rows, C, z, z_cs, "bn_train_bwd z") != W2L_OK || col_check(rows, C, dz, dz_cs, "bn_train_bwd dz") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(mean && rstd && scale && dgamma && dbeta, "bn_train_bwd: NULL argument");
    W2L_REQUIRE(g_out == nullptr || col_check(rows, C, g_out, g_cs, "bn_train_bwd g") == W2L_OK, "bn_train_bwd: bad g_out");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a = {};
    a.a = dy; a.a_cs = dy_cs; a.y = y; a.y_cs = y_cs; a.z = z; a.z_cs = z_cs; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.C = C; a.act = act;
    ColFinalArgs f = {};
    f.out0 = dbeta; f.out1 = dgamma;
    const int rc = col_reduce_launch<kColBnBwd>(a, f, s);
    if (rc != W2L_OK) return rc;
    EwArgs e = {};
    e.a = dy; e.a_cs = dy_cs; e.b = y; e.b_cs = y_cs; e.c = z; e.c_cs = z_cs; e.out = dz; e.out_cs = dz_cs;
    e.out2 = g_out; e.out2_cs = g_cs;
    e.v0 = scale; e.v1 = mean; e.v2 = rstd; e.v3 = dbeta; e.v4 = dgamma;
    e.rows = rows; e.C = C; e.act = act; e.inv_rows = (float)(1.0 / (double)rows);
    return ew_launch<kEwBnBwd>(e, s);
}

int w2l_act_bwd(void* stream, long long rows, int C, const float* dy, int dy_cs, const float* y, int y_cs, int act,
                const float* scale, float* dz, int dz_cs, float* g_out, int g_cs) {
    if (col_check(rows, C, dy, dy_cs, "act_bwd dy") != W2L_OK || col_check(rows, C, dz, dz_cs, "act_bwd dz") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(act == W2L_ACT_NONE || col_check(rows, C, y, y_cs, "act_bwd y") == W2L_OK, "act_bwd: bad y");
    W2L_REQUIRE(g_out == nullptr || col_check(rows, C, g_out, g_cs, "act_bwd g") == W2L_OK, "act_bwd: bad g_out");
    EwArgs e = {};
    e.a = dy; e.a_cs = dy_cs; e.b = act == W2L_ACT_NONE ? nullptr : y; e.b_cs = y_cs; e.out = dz; e.out_cs = dz_cs;
    e.out2 = g_out; e.out2_cs = g_cs; e.v0 = scale; e.rows = rows; e.C = C; e.act = act;
    return ew_launch<kEwActBwd>(e, static_cast<hipStream_t>(stream));
}

int w2l_add_rows(void* stream, long long rows, int C, const float* a, int a_cs, const float* b, int b_cs, float* out,
                 int out_cs) {
    if (col_check(rows, C, a, a_cs, "add_rows a") != W2L_OK || col_check(rows, C, b, b_cs, "add_rows b") != W2L_OK ||
        col_check(rows, C, out, out_cs, "add_rows out") != W2L_OK)
        return W2L_ERR_ARG;
    EwArgs e = {};
    e.a = a; e.a_cs = a_cs; e.b = b; e.b_cs = b_cs; e.out = out; e.out_cs = out_cs; e.rows = rows; e.C = C;
    return ew_launch<kEwAdd>(e, static_cast<hipStream_t>(stream));
}

int w2l_col_sum(void* stream, long long rows, int C, const float* x, int x_cs, float* out) {
    if (col_check(rows, C, x, x_cs, "col_sum") != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(out, "col_sum: NULL output");
    ColArgs a = {};
    a.a = x; a.a_cs = x_cs; a.rows = rows; a.C = C;
    ColFinalArgs f = {};
    f.out0 = out;
    return col_reduce_launch<kColSum>(a, f, static_cast<hipStream_t>(stream));
}

int w2l_l1_mean(void* stream, long long n, const float* a, const float* b, float* loss_out) {
    W2L_REQUIRE(a && b && loss_out && n >= 1, "bad l1_mean arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = grid_cap(n, 256 * 16, 1024);
    double* partial = partial_ws(s, (size_t)nb * sizeof(double));
    if (!partial) return W2L_ERR_NOMEM;
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, s, n, a, b, partial);
    W2L_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, s, nb, partial, 1.0 / (double)n, loss_out);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_l1_bwd(void* stream, long long n, const float* a, const float* b, const float* gout, float* da) {
    W2L_REQUIRE(a && b && da && n >= 1, "bad l1_bwd arguments");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_cap(n, 256, 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), n, a,
                       b, gout, (float)(1.0 / (double)n), da);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_cosine_bce_bwd(void* stream, int N, int C, const float* a, const float* v, const float* y, const float* gout,
                       float* da, float* dv) {
    W2L_REQUIRE(a && v && y && da && dv && N >= 1 && C >= 1, "bad cosine_bce_bwd arguments");
    hipLaunchKernelGGL(cosine_bce_bwd_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), N, C, a,
                       v, y, gout, da, dv);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_l2norm_bwd(void* stream, int N, int C, const float* x, int x_cs, const float* dy, float* dx, int dx_cs) {
    W2L_REQUIRE(x && dy && dx && N >= 1 && C >= 1 && x_cs >= C && dx_cs >= C, "bad l2norm_bwd arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), N, C, x,
                       x_cs, dy, dx, dx_cs);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_bce_bwd(void* stream, int N, const float* p, const float* y, const float* gout, float* dp) {
    W2L_REQUIRE(p && y && dp && N >= 1, "bad bce_bwd arguments");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_cap(N, 256, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), N, p, y,
                       gout, dp);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_adam_create(int ntensors, const long long* sizes_host, w2l_adam_t** out) {
    W2L_REQUIRE(ntensors >= 1 && sizes_host && out, "bad adam_create arguments");
    w2l_adam* h = new (std::nothrow) w2l_adam();
    if (!h) { set_error("out of host memory"); return W2L_ERR_NOMEM; }
    std::vector<AdamChunk> chunks;
    for (int i = 0; i < ntensors; ++i) {
        if (sizes_host[i] < 0 || sizes_host[i] >= (1ll << 31)) { delete h; set_error("adam: tensor %d has %lld elements", i, sizes_host[i]); return W2L_ERR_ARG; }
        for (long long s = 0; s < sizes_host[i]; s += kAdamChunk) chunks.push_back(AdamChunk{i, (int)s});
    }
    h->ntensors = ntensors;
    h->nchunks = (int)chunks.size();
    if (hipMalloc(&h->tensors_dev, sizeof(w2l_adam_tensor) * ntensors) != hipSuccess ||
        hipMalloc(&h->chunks_dev, sizeof(AdamChunk) * (chunks.empty() ? 1 : chunks.size())) != hipSuccess) {
        set_error("hipMalloc(adam tables) failed");
        w2l_adam_destroy(h);
        return W2L_ERR_NOMEM;
    }
    if (!chunks.empty() &&
        hipMemcpy(h->chunks_dev, chunks.data(), sizeof(AdamChunk) * chunks.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("upload of the adam chunk table failed");
        w2l_adam_destroy(h);
        return W2L_ERR_HIP;
    }
    *out = h;
    return W2L_OK;
}

int w2l_adam_destroy(w2l_adam_t* h) {
    if (!h) return W2L_OK;
    if (h->tensors_dev) (void)hipFree(h->tensors_dev);
    if (h->chunks_dev) (void)hipFree(h->chunks_dev);
    delete h;
    return W2L_OK;
}

int w2l_adam_step(w2l_adam_t* h, void* stream, const w2l_adam_tensor* tensors_host, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step) {
    W2L_REQUIRE(h && tensors_host && step >= 1, "bad adam_step arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    W2L_HIP_CHECK(hipMemcpyAsync(h->tensors_dev, tensors_host, sizeof(w2l_adam_tensor) * h->ntensors, hipMemcpyHostToDevice, s));
    if (h->nchunks == 0) return W2L_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(h->nchunks), dim3(256), 0, s, h->tensors_dev, h->chunks_dev, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2));
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- LSE-style sync scoring (evaluation/scores_LSE)
namespace w2l {
// calc_pdist of evaluation/scores_LSE/SyncNetInstance_calc_scores.py:19-31: f2 is zero-padded by `vshift` rows on both
// sides; out[i][j] = || f1[i] - f2p[i+j] + 1e-6 ||_2 (F.pairwise_distance adds its eps to the difference), j < 2*vshift+1.
// One wave per (i, j).
__global__ void shifted_pdist_kernel(int T, int C, int vshift, const float* __restrict__ f1, const float* __restrict__ f2,
                                     float* __restrict__ out) {
    const int win = 2 * vshift + 1;
    const int item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= T * win) return;
    const int i = item / win, j = item - i * win;
    const int r = i + j - vshift;                 // row of the unpadded f2
    const bool inside = r >= 0 && r < T;
    const float* a = f1 + (long long)i * C;
    const float* b = f2 + (long long)(inside ? r : 0) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = a[c] - (inside ? b[c] : 0.f) + 1e-6f;
        s += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[item] = sqrtf(s);
}
}  // namespace w2l

extern "C" int w2l_shifted_pdist(void* stream, int T, int C, int vshift, const float* f1, const float* f2, float* out) {
    W2L_REQUIRE(f1 && f2 && out && T >= 1 && C >= 1 && vshift >= 0, "bad shifted_pdist arguments");
    const int items = T * (2 * vshift + 1);
    hipLaunchKernelGGL(w2l::shifted_pdist_kernel, dim3(w2l::ceil_div(items, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       T, C, vshift, f1, f2, out);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

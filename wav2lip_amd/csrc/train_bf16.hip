// HBM-bound kernels of the bf16-STORAGE training path: BatchNorm in batch-statistics mode (forward + backward), activation
// backward, channel sums, row adds and the layout changes at the graph boundary - the bf16 twins of train.hip / api.hip.
// Tensors are NHWC bf16 "[rows][cs]" views (cs in elements, multiple of 8, 16-byte aligned rows); every thread moves 8 channels
// (16 bytes) per row and tensor, so a pass costs half the bytes of its fp32 twin.  Statistics, per-channel vectors and every
// intermediate are fp32 / fp64 exactly as in the fp32 kernels: column reductions accumulate in fp64 per thread, combine per
// workgroup through LDS and finish in a fixed order (deterministic, no atomics); one rounding to bf16 per stored element.
//
// Replaces, in bf16 mode, the torch autograd nodes of nn.BatchNorm2d in train mode (models/conv.py:10,40), ReLU / LeakyReLU /
// Sigmoid (models/conv.py:12,27,43; models/wav2lip.py:85,152) and the residual add (models/conv.py:17-18).
#include <math.h>
#include <mutex>
#include <vector>

#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int gridb_cap(long long work, int block, int cap) {
    long long g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// Branch-free forms for the row loops: `act` is wave-uniform, and a switch per ELEMENT compiles to a scalar branch per element
// (62 branches in the BatchNorm-backward apply kernel, 165 in its reduction: these bandwidth kernels ran at 2-3 TB/s).
// Gradient: slope `neg` on the non-positive side (1 = none, 0 = ReLU, 0.01 = LeakyReLU), y(1-y) bit-selected for the sigmoid.
struct ActK {
    float neg;
    unsigned sigmask;     // all ones for the sigmoid, else 0
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
// forward on 8 values: one wave-uniform branch per row (the sigmoid needs expf), none per element
__device__ __forceinline__ void act_fwd8(const ActK k, float* v) {
    if (k.sigmask) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_leaky(v[e], k.neg);
    }
}
__device__ __forceinline__ void ld8(const __bf16* p, float* v) {
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)b[e];
}
__device__ __forceinline__ void st8(__bf16* p, const float* v) {
    bf16x8 b;
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = b;
}
__device__ __forceinline__ void ldv8(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}

// ---------------------------------------------------------------- column reductions
enum ColModeB { kColStatsB = 0, kColBnBwdB = 1, kColSumB = 2 };

struct ColArgsB {
    const __bf16* a;    // stats: z;  bn_bwd: dy;  sum: the tensor
    const __bf16* y;    // bn_bwd: block output (activation mask)
    const __bf16* z;    // bn_bwd: pre-BN conv output
    const float* mean;  // bn_bwd, padded to C
    const float* rstd;
    const float* scale; // bn_bwd with y == NULL (no residual, ReLU): the mask is recomputed as z*scale + shift > 0
    const float* shift;
    double* partial;    // [nblocks][2][C]
    long long rows;
    int C, a_cs, y_cs, z_cs, act;
    int rows_per_block;
};

// thread -> (8-channel group c8 = t % CG, row lane t / CG); needs C % 8 == 0 and C <= 1024
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_bf16_kernel(const ColArgsB a) {
    __shared__ double red[256][9];     // [..][8] + 1 pad column: the final pass walks rows CG apart
    const int CG = a.C >> 3;
    const int RPP = 256 / CG;
    const int t = threadIdx.x;
    const int c8 = t % CG;
    const int rl = t / CG;
    double s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0; s1[e] = 0; }
    if (rl < RPP) {
        const long long r0 = (long long)blockIdx.x * a.rows_per_block;
        const long long r1 = r0 + a.rows_per_block < a.rows ? r0 + a.rows_per_block : a.rows;
        float mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = 0.f; rs[e] = 0.f; }
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
        const bool no_y = (MODE == kColBnBwdB) && a.y == nullptr;
        const ActK ak = act_consts(a.act);
        if (MODE == kColBnBwdB) { ldv8(a.mean + c8 * 8, mu); ldv8(a.rstd + c8 * 8, rs); }
        if (no_y) { ldv8(a.scale + c8 * 8, sc); ldv8(a.shift + c8 * 8, sh); }
        // two rows per iteration: all their loads are issued before the first is consumed (a column reduction with one load in
        // flight per thread ran at 1.5 - 2.5 TB/s: latency, not bandwidth)
        auto accum = [&](const float* v, const float* yv, const float* zv) {
            if (MODE == kColStatsB) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s0[e] += (double)v[e]; s1[e] += (double)v[e] * (double)v[e]; }
            } else if (MODE == kColSumB) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s0[e] += (double)v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float yy = no_y ? zv[e] * sc[e] + sh[e] : yv[e];      // the forward's own expression (affine_act_bf16)
                    const float g = v[e] * act_grad_k(ak, yy);
                    const float zh = (zv[e] - mu[e]) * rs[e];
                    s0[e] += (double)g;
                    s1[e] += (double)g * (double)zh;
                }
            }
        };
        long long r = r0 + rl;
        for (; r + RPP < r1; r += 2 * RPP) {
            float va[8], vb[8], ya[8], yb[8], za[8], zb[8];
            ld8(a.a + r * a.a_cs + c8 * 8, va);
            ld8(a.a + (r + RPP) * a.a_cs + c8 * 8, vb);
            if (MODE == kColBnBwdB) {
                if (!no_y) {
                    ld8(a.y + r * a.y_cs + c8 * 8, ya);
                    ld8(a.y + (r + RPP) * a.y_cs + c8 * 8, yb);
                }
                ld8(a.z + r * a.z_cs + c8 * 8, za);
                ld8(a.z + (r + RPP) * a.z_cs + c8 * 8, zb);
            }
            accum(va, ya, za);
            accum(vb, yb, zb);
        }
        for (; r < r1; r += RPP) {
            float v[8], yv[8], zv[8];
            ld8(a.a + r * a.a_cs + c8 * 8, v);
            if (MODE == kColBnBwdB) {
                if (!no_y) ld8(a.y + r * a.y_cs + c8 * 8, yv);
                ld8(a.z + r * a.z_cs + c8 * 8, zv);
            }
            accum(v, yv, zv);
        }
    }
    // two rounds through LDS (sum, then sum of squares / products): 256 x 9 doubles = 18 KB
    double o0[8], o1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[t][e] = s0[e];
    __syncthreads();
    if (t < CG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o0[e] = 0;
        for (int j = 0; j < RPP; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) o0[e] += red[t + j * CG][e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[t][e] = s1[e];
    __syncthreads();
    if (t < CG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o1[e] = 0;
        for (int j = 0; j < RPP; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) o1[e] += red[t + j * CG][e];
        double* dst = a.partial + (long long)blockIdx.x * 2 * a.C;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dst[t * 8 + e] = o0[e]; dst[a.C + t * 8 + e] = o1[e]; }
    }
}

struct ColFinalArgsB {
    const double* partial;
    int nblocks, C, Cvalid;   // C: padded channels of the partials; Cvalid: channels that exist (outputs beyond are skipped)
    long long rows;
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* mean;
    float* rstd;
    float* scale;
    float* shift;
    float* running_mean;
    float* running_var;
    float* out0;
    float* out1;
};

template <int MODE>
__global__ __launch_bounds__(256) void col_final_bf16_kernel(const ColFinalArgsB a) {
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
                if (MODE != kColSumB) t1[u] += p[(long long)(b + 4 * u) * st + a.C];
            }
        }
        for (; b < a.nblocks; b += 4) {
            t0[0] += p[(long long)b * st];
            if (MODE != kColSumB) t1[0] += p[(long long)b * st + a.C];
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
    const bool live = c < a.Cvalid;
    if (MODE == kColStatsB) {
        const double m = s0 / (double)a.rows;
        double var = s1 / (double)a.rows - m * m;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float mf = (float)m;
        // the per-channel vectors are padded to C (multiple of 8) so that the elementwise kernels can load them as vectors:
        // pad channels get the identity of a zero tensor (mean 0, scale 0, shift 0)
        a.mean[c] = live ? mf : 0.f;
        a.rstd[c] = live ? rstd : 0.f;
        const float sc = live ? (a.gamma ? a.gamma[c] : 1.f) * rstd : 0.f;
        a.scale[c] = sc;
        a.shift[c] = live ? (a.beta ? a.beta[c] : 0.f) - mf * sc : 0.f;
        if (live && a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mf;
        if (live && a.running_var) {
            const double unb = a.rows > 1 ? var * (double)a.rows / (double)(a.rows - 1) : var;
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
        }
    } else {
        if (a.out0) a.out0[c] = live ? (float)s0 : 0.f;
        if (a.out1) a.out1[c] = live ? (float)s1 : 0.f;
    }
}

struct PartialWsB {
    hipStream_t stream;
    double* ptr;
};
static std::mutex g_partialb_mutex;
static std::vector<PartialWsB> g_partialb_table;
constexpr size_t kPartialBytesB = (size_t)16 << 20;
static double* partialb_ws(hipStream_t stream, size_t bytes) {
    if (bytes > kPartialBytesB) { set_error("reduction scratch request of %zu bytes exceeds the fixed buffer", bytes); return nullptr; }
    std::lock_guard<std::mutex> lock(g_partialb_mutex);
    for (const PartialWsB& w : g_partialb_table)
        if (w.stream == stream) return w.ptr;
    double* p = nullptr;
    if (hipMalloc(&p, kPartialBytesB) != hipSuccess) { set_error("hipMalloc(reduction scratch) failed"); return nullptr; }
    g_partialb_table.push_back(PartialWsB{stream, p});
    return p;
}

static int colb_check(long long rows, int C, const void* p, int cs, const char* what) {
    W2L_REQUIRE(rows >= 1 && C >= 8 && (C & 7) == 0 && C <= 1024, "%s: C=%d must be a multiple of 8 in [8, 1024]", what, C);
    W2L_REQUIRE(p && cs >= C && (cs & 7) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0,
                "%s: tensor must be 16-byte aligned with a channel stride that is a multiple of 8 (cs=%d)", what, cs);
    return W2L_OK;
}

template <int MODE>
static int colb_launch(ColArgsB a, ColFinalArgsB f, hipStream_t s) {
    const int CG = a.C >> 3;
    const int RPP = 256 / CG;
    // at most 512 workgroups: the finalize pass walks every partial with C / 64 workgroups, so 1 024 partials cost it more than
    // the reduction gains from them (cfg4 26.3 -> 26.1 ms at 512, 26.6 at 256; session r03ze)
    long long per = (a.rows + 511) / 512;
    const long long min_rows = (long long)RPP * 8;
    if (per < min_rows) per = min_rows;
    a.rows_per_block = (int)per;
    const int nblocks = (int)((a.rows + per - 1) / per);
    a.partial = partialb_ws(s, (size_t)nblocks * 2 * a.C * sizeof(double));
    if (!a.partial) return W2L_ERR_NOMEM;
    hipLaunchKernelGGL(col_reduce_bf16_kernel<MODE>, dim3(nblocks), dim3(256), 0, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    f.partial = a.partial;
    f.nblocks = nblocks;
    f.C = a.C;
    f.rows = a.rows;
    hipLaunchKernelGGL(col_final_bf16_kernel<MODE>, dim3(ceil_div(a.C, 64)), dim3(256), 0, s, f);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---- BatchNorm statistics from the conv epilogue's per-wave column partials (conv_bf16.hip): fp32 [npart][2][cout_p] ->
// fp64 [R][2][C] by R <= 256 workgroups (64 channels x 4 row lanes, four loads in flight per thread, fixed order), then the
// same finalize as the stand-alone reduction
__global__ __launch_bounds__(256) void stats_partial_reduce_kernel(const float* __restrict__ part, int npart, int cout_p, int C,
                                                                   int rows_per_block, double* __restrict__ out) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(npart, r0 + rows_per_block);
    double s0 = 0, s1 = 0;
    if (c < C) {
        const long long st = 2ll * cout_p;
        const float* p = part + c;
        double t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
        int r = r0 + q;
        for (; r + 12 < r1; r += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t0[u] += (double)p[(long long)(r + 4 * u) * st];
                t1[u] += (double)p[(long long)(r + 4 * u) * st + cout_p];
            }
        }
        for (; r < r1; r += 4) {
            t0[0] += (double)p[(long long)r * st];
            t1[0] += (double)p[(long long)r * st + cout_p];
        }
        s0 = (t0[0] + t0[1]) + (t0[2] + t0[3]);
        s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
    }
    red[0][q][cl] = s0;
    red[1][q][cl] = s1;
    __syncthreads();
    if (q == 0 && c < C) {
        double* dst = out + (long long)blockIdx.x * 2 * C;
        dst[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        dst[C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

int bn_stats_from_partials(hipStream_t s, const float* part, int npart, int cout_p, long long rows, int C, int Cvalid,
                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* rstd, float* scale, float* shift) {
    int R = ceil_div(npart, 64);
    if (R > 256) R = 256;
    if (R < 1) R = 1;
    const int rpb = ceil_div(npart, R);
    R = ceil_div(npart, rpb);
    double* out = partialb_ws(s, (size_t)R * 2 * C * sizeof(double));
    if (!out) return W2L_ERR_NOMEM;
    hipLaunchKernelGGL(stats_partial_reduce_kernel, dim3(R, ceil_div(C, 64)), dim3(256), 0, s, part, npart, cout_p, C, rpb, out);
    W2L_HIP_CHECK(hipGetLastError());
    ColFinalArgsB f = {};
    f.partial = out; f.nblocks = R; f.C = C; f.Cvalid = Cvalid; f.rows = rows;
    f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.mean = mean; f.rstd = rstd; f.scale = scale; f.shift = shift;
    f.running_mean = running_mean; f.running_var = running_var;
    hipLaunchKernelGGL(col_final_bf16_kernel<kColStatsB>, dim3(ceil_div(C, 64)), dim3(256), 0, s, f);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// the two BatchNorm-backward column sums from a data-gradient conv's epilogue partials (w2l_convb_forward_bnbwd): the same two
// levels, finished as the stand-alone reduction finishes them (out0 = sum g -> dbeta, out1 = sum g * zhat -> dgamma)
int bn_bwd_sums_from_partials(hipStream_t s, const float* part, int npart, int cout_p, int C, int Cvalid, float* dgamma, float* dbeta) {
    int R = ceil_div(npart, 64);
    if (R > 256) R = 256;
    if (R < 1) R = 1;
    const int rpb = ceil_div(npart, R);
    R = ceil_div(npart, rpb);
    double* out = partialb_ws(s, (size_t)R * 2 * C * sizeof(double));
    if (!out) return W2L_ERR_NOMEM;
    hipLaunchKernelGGL(stats_partial_reduce_kernel, dim3(R, ceil_div(C, 64)), dim3(256), 0, s, part, npart, cout_p, C, rpb, out);
    W2L_HIP_CHECK(hipGetLastError());
    ColFinalArgsB f = {};
    f.partial = out; f.nblocks = R; f.C = C; f.Cvalid = Cvalid; f.rows = 1;
    f.out0 = dbeta; f.out1 = dgamma;
    hipLaunchKernelGGL(col_final_bf16_kernel<kColBnBwdB>, dim3(ceil_div(C, 64)), dim3(256), 0, s, f);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---------------------------------------------------------------- elementwise over [rows][C]
struct EwArgsB {
    const __bf16* a;     // affine: z;            bn_bwd_apply: dy;      act_bwd: dy
    const __bf16* b;     // affine: res or NULL;  bn_bwd_apply: y;       act_bwd: y
    const __bf16* c;     //                       bn_bwd_apply: z
    __bf16* out;         // affine: y;            bn_bwd_apply: dz;      act_bwd: dz
    __bf16* out2;        //                       in-place g (= masked dy) or NULL
    const float* v0;     // per-channel vectors, padded to C
    const float* v1;
    const float* v2;
    const float* v3;
    const float* v4;
    const float* v5;     // bn_bwd without y: the forward shift (beta - mean * gamma * rstd); the mask is z*v0 + v5 > 0
    long long rows;
    int C, a_cs, b_cs, c_cs, out_cs, out2_cs, act;
    float inv_rows;
};

enum EwModeB { kEwAffineB = 0, kEwBnBwdB = 1, kEwActBwdB = 2, kEwAddB = 3 };

template <int MODE>
__global__ __launch_bounds__(256) void ew_bf16_kernel(const EwArgsB a) {
    const int CG = a.C >> 3;
    const int RPP = 256 / CG;
    const int c = (threadIdx.x % CG) * 8;
    const int rl = threadIdx.x / CG;
    if (rl >= RPP) return;
    float v0[8], v1[8], v2[8], v3[8], v4[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { v0[e] = 1.f; v1[e] = 0.f; v2[e] = 0.f; v3[e] = 0.f; v4[e] = 0.f; }
    if (a.v0) ldv8(a.v0 + c, v0);
    if (a.v1) ldv8(a.v1 + c, v1);
    if (a.v2) ldv8(a.v2 + c, v2);
    if (a.v3) ldv8(a.v3 + c, v3);
    if (a.v4) ldv8(a.v4 + c, v4);
    float v5[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v5[e] = 0.f;
    if (MODE == kEwBnBwdB && a.v5) ldv8(a.v5 + c, v5);
    const ActK ak = act_consts(a.act);
    for (long long r = (long long)blockIdx.x * RPP + rl; r < a.rows; r += (long long)gridDim.x * RPP) {
        float av[8], o[8];
        ld8(a.a + r * a.a_cs + c, av);
        if (MODE == kEwAffineB) {
            float rv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = 0.f;
            if (a.b) ld8(a.b + r * a.b_cs + c, rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = av[e] * v0[e] + v1[e] + rv[e];
            act_fwd8(ak, o);
        } else if (MODE == kEwBnBwdB) {   // v0 gamma*rstd, v1 mean, v2 rstd, v3 sum g, v4 sum g*zhat
            float yv[8], zv[8], g[8];
            ld8(a.c + r * a.c_cs + c, zv);
            if (a.b) ld8(a.b + r * a.b_cs + c, yv);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) yv[e] = zv[e] * v0[e] + v5[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                g[e] = av[e] * act_grad_k(ak, yv[e]);
                const float zh = (zv[e] - v1[e]) * v2[e];
                o[e] = v0[e] * (g[e] - v3[e] * a.inv_rows - zh * (v4[e] * a.inv_rows));
            }
            if (a.out2) st8(a.out2 + r * a.out2_cs + c, g);
        } else if (MODE == kEwActBwdB) {
            float yv[8], g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) yv[e] = 1.f;
            if (a.b) ld8(a.b + r * a.b_cs + c, yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                g[e] = av[e] * act_grad_k(ak, yv[e]);
                o[e] = g[e] * v0[e];
            }
            if (a.out2) st8(a.out2 + r * a.out2_cs + c, g);
        } else {
            float bv[8];
            ld8(a.b + r * a.b_cs + c, bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = av[e] + bv[e];
        }
        st8(a.out + r * a.out_cs + c, o);
    }
}

template <int MODE>
static int ewb_launch(const EwArgsB& a, hipStream_t s) {
    const int RPP = 256 / (a.C >> 3);
    hipLaunchKernelGGL(ew_bf16_kernel<MODE>, dim3(gridb_cap(a.rows, RPP * 4, 16384)), dim3(256), 0, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---------------------------------------------------------------- layout at the graph boundary
// fp32 NCHW -> bf16 NHWC through a 32x33 LDS tile over (C, HW); channels [C, c_zero_to) zero-filled
__global__ void nchw_to_nhwc_bf16_kernel(int C, int HW, const float* __restrict__ x, __bf16* __restrict__ y, int y_cs, int c_zero_to) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? x[((long long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < c_zero_to) y[((long long)n * HW + p) * y_cs + c] = (__bf16)tile[tx][j];
    }
}
// bf16 NHWC (first C channels) -> fp32 NCHW
__global__ void nhwc_bf16_to_nchw_kernel(int C, int HW, const __bf16* __restrict__ x, int x_cs, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        tile[j][tx] = (p < HW && c < C) ? (float)x[((long long)n * HW + p) * x_cs + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (c < C && p < HW) y[((long long)n * C + c) * HW + p] = tile[tx][j];
    }
}

}  // namespace w2l

using namespace w2l;

extern "C" {

int w2l_bn_train_stats_bf16(void* stream, long long rows, int C, int Cvalid, const void* z, int z_cs, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                            float* rstd, float* scale, float* shift) {
    if (colb_check(rows, C, z, z_cs, "bn_train_stats_bf16") != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(mean && rstd && scale && shift && Cvalid >= 1 && Cvalid <= C, "bn_train_stats_bf16: bad argument");
    ColArgsB a = {};
    a.a = static_cast<const __bf16*>(z); a.a_cs = z_cs; a.rows = rows; a.C = C;
    ColFinalArgsB f = {};
    f.Cvalid = Cvalid; f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum;
    f.mean = mean; f.rstd = rstd; f.scale = scale; f.shift = shift;
    f.running_mean = running_mean; f.running_var = running_var;
    return colb_launch<kColStatsB>(a, f, static_cast<hipStream_t>(stream));
}

int w2l_affine_act_bf16(void* stream, long long rows, int C, const void* z, int z_cs, const float* scale, const float* shift,
                        const void* res, int res_cs, int act, void* y, int y_cs) {
    if (colb_check(rows, C, z, z_cs, "affine_act_bf16 z") != W2L_OK || colb_check(rows, C, y, y_cs, "affine_act_bf16 y") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(scale && shift, "affine_act_bf16: NULL scale/shift");
    W2L_REQUIRE(res == nullptr || colb_check(rows, C, res, res_cs, "affine_act_bf16 res") == W2L_OK, "affine_act_bf16: bad residual");
    EwArgsB a = {};
    a.a = static_cast<const __bf16*>(z); a.a_cs = z_cs; a.b = static_cast<const __bf16*>(res); a.b_cs = res_cs;
    a.out = static_cast<__bf16*>(y); a.out_cs = y_cs; a.v0 = scale; a.v1 = shift; a.rows = rows; a.C = C; a.act = act;
    return ewb_launch<kEwAffineB>(a, static_cast<hipStream_t>(stream));
}

int w2l_bn_train_bwd_bf16(void* stream, long long rows, int C, int Cvalid, const void* dy, int dy_cs, const void* y, int y_cs,
                          const void* z, int z_cs, int act, const float* mean, const float* rstd, const float* scale,
                          const float* shift, float* dgamma, float* dbeta, void* dz, int dz_cs, void* g_out, int g_cs) {
    if (colb_check(rows, C, dy, dy_cs, "bn_train_bwd_bf16 dy") != W2L_OK ||
        (y != nullptr && colb_check(rows, C, y, y_cs, "bn_train_bwd_bf16 y") != W2L_OK) ||
        colb_check(rows, C, z, z_cs, "bn_train_bwd_bf16 z") != W2L_OK || colb_check(rows, C, dz, dz_cs, "bn_train_bwd_bf16 dz") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(mean && rstd && scale && dgamma && dbeta && Cvalid >= 1 && Cvalid <= C, "bn_train_bwd_bf16: bad argument");
    W2L_REQUIRE(y != nullptr || (shift != nullptr && act == W2L_ACT_RELU && g_out == nullptr),
                "bn_train_bwd_bf16: y may be omitted only for a ReLU block without residual, with the forward shift given");
    W2L_REQUIRE(g_out == nullptr || colb_check(rows, C, g_out, g_cs, "bn_train_bwd_bf16 g") == W2L_OK, "bn_train_bwd_bf16: bad g_out");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgsB a = {};
    a.a = static_cast<const __bf16*>(dy); a.a_cs = dy_cs; a.y = static_cast<const __bf16*>(y); a.y_cs = y_cs;
    a.z = static_cast<const __bf16*>(z); a.z_cs = z_cs; a.mean = mean; a.rstd = rstd; a.scale = scale; a.shift = shift;
    a.rows = rows; a.C = C; a.act = act;
    ColFinalArgsB f = {};
    f.Cvalid = Cvalid; f.out0 = dbeta; f.out1 = dgamma;
    const int rc = colb_launch<kColBnBwdB>(a, f, s);
    if (rc != W2L_OK) return rc;
    EwArgsB e = {};
    e.a = a.a; e.a_cs = dy_cs; e.b = a.y; e.b_cs = y_cs; e.c = a.z; e.c_cs = z_cs;
    e.out = static_cast<__bf16*>(dz); e.out_cs = dz_cs; e.out2 = static_cast<__bf16*>(g_out); e.out2_cs = g_cs;
    e.v0 = scale; e.v1 = mean; e.v2 = rstd; e.v3 = dbeta; e.v4 = dgamma; e.v5 = shift;
    e.rows = rows; e.C = C; e.act = act; e.inv_rows = (float)(1.0 / (double)rows);
    return ewb_launch<kEwBnBwdB>(e, s);
}

int w2l_bn_train_bwd_apply_bf16(void* stream, long long rows, int C, const void* dy, int dy_cs, const void* y, int y_cs,
                                const void* z, int z_cs, int act, const float* mean, const float* rstd, const float* scale,
                                const float* shift, const float* dgamma, const float* dbeta, void* dz, int dz_cs, void* g_out,
                                int g_cs) {
    if (colb_check(rows, C, dy, dy_cs, "bn_train_bwd_apply_bf16 dy") != W2L_OK ||
        (y != nullptr && colb_check(rows, C, y, y_cs, "bn_train_bwd_apply_bf16 y") != W2L_OK) ||
        colb_check(rows, C, z, z_cs, "bn_train_bwd_apply_bf16 z") != W2L_OK ||
        colb_check(rows, C, dz, dz_cs, "bn_train_bwd_apply_bf16 dz") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(mean && rstd && scale && dgamma && dbeta, "bn_train_bwd_apply_bf16: bad argument");
    W2L_REQUIRE(y != nullptr || act == W2L_ACT_NONE || (shift != nullptr && act == W2L_ACT_RELU && g_out == nullptr),
                "bn_train_bwd_apply_bf16: y may be omitted only for a ReLU block without residual, with the forward shift given, or "
                "with act = none when dy already is the masked gradient");
    W2L_REQUIRE(g_out == nullptr || colb_check(rows, C, g_out, g_cs, "bn_train_bwd_apply_bf16 g") == W2L_OK, "bn_train_bwd_apply_bf16: bad g_out");
    EwArgsB e = {};
    e.a = static_cast<const __bf16*>(dy); e.a_cs = dy_cs; e.b = static_cast<const __bf16*>(y); e.b_cs = y_cs;
    e.c = static_cast<const __bf16*>(z); e.c_cs = z_cs;
    e.out = static_cast<__bf16*>(dz); e.out_cs = dz_cs; e.out2 = static_cast<__bf16*>(g_out); e.out2_cs = g_cs;
    e.v0 = scale; e.v1 = mean; e.v2 = rstd; e.v3 = dbeta; e.v4 = dgamma; e.v5 = shift;
    e.rows = rows; e.C = C; e.act = act; e.inv_rows = (float)(1.0 / (double)rows);
    return ewb_launch<kEwBnBwdB>(e, static_cast<hipStream_t>(stream));
}

int w2l_act_bwd_bf16(void* stream, long long rows, int C, const void* dy, int dy_cs, const void* y, int y_cs, int act,
                     const float* scale, void* dz, int dz_cs, void* g_out, int g_cs) {
    if (colb_check(rows, C, dy, dy_cs, "act_bwd_bf16 dy") != W2L_OK || colb_check(rows, C, dz, dz_cs, "act_bwd_bf16 dz") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(act == W2L_ACT_NONE || colb_check(rows, C, y, y_cs, "act_bwd_bf16 y") == W2L_OK, "act_bwd_bf16: bad y");
    W2L_REQUIRE(g_out == nullptr || colb_check(rows, C, g_out, g_cs, "act_bwd_bf16 g") == W2L_OK, "act_bwd_bf16: bad g_out");
    EwArgsB e = {};
    e.a = static_cast<const __bf16*>(dy); e.a_cs = dy_cs; e.b = act == W2L_ACT_NONE ? nullptr : static_cast<const __bf16*>(y);
    e.b_cs = y_cs; e.out = static_cast<__bf16*>(dz); e.out_cs = dz_cs; e.out2 = static_cast<__bf16*>(g_out); e.out2_cs = g_cs;
    e.v0 = scale; e.rows = rows; e.C = C; e.act = act;
    return ewb_launch<kEwActBwdB>(e, static_cast<hipStream_t>(stream));
}

int w2l_add_rows_bf16(void* stream, long long rows, int C, const void* a, int a_cs, const void* b, int b_cs, void* out, int out_cs) {
    if (colb_check(rows, C, a, a_cs, "add_rows_bf16 a") != W2L_OK || colb_check(rows, C, b, b_cs, "add_rows_bf16 b") != W2L_OK ||
        colb_check(rows, C, out, out_cs, "add_rows_bf16 out") != W2L_OK)
        return W2L_ERR_ARG;
    EwArgsB e = {};
    e.a = static_cast<const __bf16*>(a); e.a_cs = a_cs; e.b = static_cast<const __bf16*>(b); e.b_cs = b_cs;
    e.out = static_cast<__bf16*>(out); e.out_cs = out_cs; e.rows = rows; e.C = C;
    return ewb_launch<kEwAddB>(e, static_cast<hipStream_t>(stream));
}

int w2l_col_sum_bf16(void* stream, long long rows, int C, const void* x, int x_cs, float* out) {
    if (colb_check(rows, C, x, x_cs, "col_sum_bf16") != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(out, "col_sum_bf16: NULL output");
    ColArgsB a = {};
    a.a = static_cast<const __bf16*>(x); a.a_cs = x_cs; a.rows = rows; a.C = C;
    ColFinalArgsB f = {};
    f.Cvalid = C; f.out0 = out;
    return colb_launch<kColSumB>(a, f, static_cast<hipStream_t>(stream));
}

int w2l_nchw_to_nhwc_bf16(void* stream, int N, int C, int H, int W, const float* x, void* y, int y_cs, int c_zero_to) {
    W2L_REQUIRE(x && y && N >= 1 && C >= 1 && H >= 1 && W >= 1, "bad nchw_to_nhwc_bf16 arguments");
    if (c_zero_to < C) c_zero_to = C;
    W2L_REQUIRE(y_cs >= c_zero_to, "y_cs=%d < %d", y_cs, c_zero_to);
    W2L_REQUIRE(N <= 65535, "N too large for one launch");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(ceil_div(HW, 32), ceil_div(c_zero_to, 32), N), dim3(256), 0,
                       static_cast<hipStream_t>(stream), C, HW, x, static_cast<__bf16*>(y), y_cs, c_zero_to);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_nhwc_bf16_to_nchw(void* stream, int N, int C, int H, int W, const void* x, int x_cs, float* y) {
    W2L_REQUIRE(x && y && N >= 1 && C >= 1 && H >= 1 && W >= 1 && x_cs >= C, "bad nhwc_bf16_to_nchw arguments");
    W2L_REQUIRE(N <= 65535, "N too large for one launch");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), N), dim3(256), 0,
                       static_cast<hipStream_t>(stream), C, HW, static_cast<const __bf16*>(x), x_cs, y);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // extern "C"

// The THIN 1x1 convolution of the bf16-storage training path: cin <= 32, cout <= 4 over millions of pixels - the generator's
// output layer `nn.Conv2d(32, 3, 1)` + Sigmoid (models/wav2lip.py:83-85) inside wav2lip_train.py:220-231 /
// hq_wav2lip_train.py:221-256.  On the implicit GEMM it is a 128x32 tile multiplying 29 / 32 padding: forward 0.18 ms, data
// gradient 0.29 ms, weight gradient 0.45 ms per cfg4 step (profiles/r04/q_train_bf16_nodes.log) for 0.57 GFLOP each, while
// the three passes move 236 MB each - 0.05 ms at the rate the elementwise kernels reach.  These are HBM-bound row kernels:
// one pixel per thread and iteration, 16-byte loads / stores, the 96 weights in LDS (every lane reads the same address:
// a broadcast), fp32 accumulation of bf16 x bf16 products (weights rounded to bf16 on the way in, as the GEMM path rounds
// them), one rounding per stored element; the weight gradient keeps the cout x cin partial sums of a thread in registers,
// folds them over the wave with shuffles and finishes in a fixed order (deterministic, no atomics).
#include <math.h>
#include <mutex>
#include <vector>

#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThinMaxCin = 32;    // channels of x (padded to a multiple of 8 in the tensor)
constexpr int kThinMaxCout = 4;

struct ThinArgs {
    const __bf16* x;       // [npix][x_cs]
    const __bf16* dz;      // dgrad / wgrad: [npix][dz_cs], cout valid channels (pad channels zero)
    const __bf16* res;     // dgrad: optional [npix][res_cs] added to dx (may alias dx: accumulate)
    __bf16* out;           // forward: y [npix][out_cs] (8 channels written, pad zero);  dgrad: dx [npix][out_cs] (cin8 written)
    const float* w;        // fp32 [cout][cin] (torch layout of a 1x1 conv)
    const float* bias;     // forward: [cout] or NULL
    float* partial;        // wgrad: [nblocks][cout * cin8 + cout]
    long long npix;
    int cin, cin8, cout, x_cs, dz_cs, res_cs, out_cs, act;
};

__device__ __forceinline__ void thin_load_w(const ThinArgs& a, float* ws) {     // ws[cout][cin8] in LDS, bf16-rounded, pad zero
    for (int i = threadIdx.x; i < kThinMaxCout * kThinMaxCin; i += blockDim.x) {
        const int o = i / kThinMaxCin, c = i - o * kThinMaxCin;
        ws[i] = (o < a.cout && c < a.cin) ? (float)(__bf16)a.w[o * a.cin + c] : 0.f;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void thin1x1_forward_bf16_kernel(const ThinArgs a) {
    __shared__ __attribute__((aligned(16))) float ws[kThinMaxCout * kThinMaxCin];
    thin_load_w(a, ws);
    float b[kThinMaxCout];
#pragma unroll
    for (int o = 0; o < kThinMaxCout; ++o) b[o] = (a.bias && o < a.cout) ? a.bias[o] : 0.f;
    const int ng = a.cin8 >> 3;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < a.npix; p += (long long)gridDim.x * blockDim.x) {
        float acc[kThinMaxCout];
#pragma unroll
        for (int o = 0; o < kThinMaxCout; ++o) acc[o] = b[o];
        const __bf16* xr = a.x + p * a.x_cs;
#pragma unroll
        for (int g = 0; g < kThinMaxCin / 8; ++g) {
            if (g < ng) {
                const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xr + g * 8);
#pragma unroll
                for (int o = 0; o < kThinMaxCout; ++o) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + o * kThinMaxCin + g * 8);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws + o * kThinMaxCin + g * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o] = fmaf((float)xv[e], w0[e], fmaf((float)xv[4 + e], w1[e], acc[o]));
                }
            }
        }
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (__bf16)0.f;
#pragma unroll
        for (int o = 0; o < kThinMaxCout; ++o) {
            float v = acc[o];
            if (a.act == W2L_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            else if (a.act == W2L_ACT_RELU) v = act_leaky(v, 0.f);
            else if (a.act == W2L_ACT_LEAKY) v = act_leaky(v, 0.01f);
            if (o < a.cout) y[o] = (__bf16)v;
        }
        *reinterpret_cast<bf16x8*>(a.out + p * a.out_cs) = y;
    }
}

__global__ __launch_bounds__(256) void thin1x1_dgrad_bf16_kernel(const ThinArgs a) {
    __shared__ __attribute__((aligned(16))) float ws[kThinMaxCout * kThinMaxCin];
    thin_load_w(a, ws);
    const int ng = a.cin8 >> 3;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < a.npix; p += (long long)gridDim.x * blockDim.x) {
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(a.dz + p * a.dz_cs);
        float d[kThinMaxCout];
#pragma unroll
        for (int o = 0; o < kThinMaxCout; ++o) d[o] = (float)dv[o];
#pragma unroll
        for (int g = 0; g < kThinMaxCin / 8; ++g) {
            if (g < ng) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (a.res) {
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.res + p * a.res_cs + g * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)rv[e];
                }
#pragma unroll
                for (int o = 0; o < kThinMaxCout; ++o) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + o * kThinMaxCin + g * 8);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws + o * kThinMaxCin + g * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = fmaf(d[o], w0[e], v[e]); v[4 + e] = fmaf(d[o], w1[e], v[4 + e]); }
                }
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x8*>(a.out + p * a.out_cs + g * 8) = ov;
            }
        }
    }
}

// dW[o][c] = sum_p dz[p][o] * x[p][c],  db[o] = sum_p dz[p][o]: per-thread partials in registers over a strided pixel range, wave
// fold by xor-shuffles (fixed order), the four waves of a workgroup meet in LDS, one partial row per workgroup
__global__ __launch_bounds__(256) void thin1x1_wgrad_bf16_kernel(const ThinArgs a) {
    __shared__ float red[4][kThinMaxCout * kThinMaxCin + kThinMaxCout];
    float acc[kThinMaxCout][kThinMaxCin];
    float db[kThinMaxCout];
#pragma unroll
    for (int o = 0; o < kThinMaxCout; ++o) {
        db[o] = 0.f;
#pragma unroll
        for (int c = 0; c < kThinMaxCin; ++c) acc[o][c] = 0.f;
    }
    const int ng = a.cin8 >> 3;
    // two pixels per iteration: the ten loads of both are issued before the first is consumed (one pixel per iteration ran the
    // kernel on one memory latency per 80 bytes: 0.13 ms for 236 MB)
    auto fold = [&](const bf16x8 dv, const bf16x8* xv) {
        float d[kThinMaxCout];
#pragma unroll
        for (int o = 0; o < kThinMaxCout; ++o) { d[o] = (float)dv[o]; db[o] += d[o]; }
#pragma unroll
        for (int g = 0; g < kThinMaxCin / 8; ++g) {
            if (g < ng) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xf = (float)xv[g][e];
#pragma unroll
                    for (int o = 0; o < kThinMaxCout; ++o) acc[o][g * 8 + e] = fmaf(d[o], xf, acc[o][g * 8 + e]);
                }
            }
        }
    };
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; p + stride < a.npix; p += 2 * stride) {
        const bf16x8 dv0 = *reinterpret_cast<const bf16x8*>(a.dz + p * a.dz_cs);
        const bf16x8 dv1 = *reinterpret_cast<const bf16x8*>(a.dz + (p + stride) * a.dz_cs);
        bf16x8 x0[kThinMaxCin / 8], x1[kThinMaxCin / 8];
#pragma unroll
        for (int g = 0; g < kThinMaxCin / 8; ++g)
            if (g < ng) {
                x0[g] = *reinterpret_cast<const bf16x8*>(a.x + p * a.x_cs + g * 8);
                x1[g] = *reinterpret_cast<const bf16x8*>(a.x + (p + stride) * a.x_cs + g * 8);
            }
        fold(dv0, x0);
        fold(dv1, x1);
    }
    for (; p < a.npix; p += stride) {
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(a.dz + p * a.dz_cs);
        bf16x8 xv[kThinMaxCin / 8];
#pragma unroll
        for (int g = 0; g < kThinMaxCin / 8; ++g)
            if (g < ng) xv[g] = *reinterpret_cast<const bf16x8*>(a.x + p * a.x_cs + g * 8);
        fold(dv, xv);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < kThinMaxCout; ++o) {
#pragma unroll
        for (int c = 0; c < kThinMaxCin; ++c) {
            float v = acc[o][c];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
            if (lane == 0) red[wave][o * kThinMaxCin + c] = v;
        }
        float v = db[o];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave][kThinMaxCout * kThinMaxCin + o] = v;
    }
    __syncthreads();
    constexpr int NV = kThinMaxCout * kThinMaxCin + kThinMaxCout;
    if (threadIdx.x < NV)
        a.partial[(long long)blockIdx.x * NV + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// seven row lanes per value walk the partial rows (fixed order per lane), LDS, then one fixed-order sum of the seven
constexpr int kThinFinalLanes = 7;
__global__ __launch_bounds__(1024) void thin1x1_wgrad_final_kernel(const float* __restrict__ partial, int nblocks, int cin, int cout,
                                                                   float* __restrict__ dw, float* __restrict__ dbias) {
    constexpr int NV = kThinMaxCout * kThinMaxCin + kThinMaxCout;
    static_assert(NV * kThinFinalLanes <= 1024, "thin1x1_wgrad_final_kernel: one workgroup");
    __shared__ double red[kThinFinalLanes][NV];
    const int i = threadIdx.x % NV, rl = threadIdx.x / NV;
    if (rl < kThinFinalLanes) {
        double s[4] = {0, 0, 0, 0};
        int b = rl;
        for (; b + 3 * kThinFinalLanes < nblocks; b += 4 * kThinFinalLanes) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += (double)partial[(long long)(b + u * kThinFinalLanes) * NV + i];
        }
        for (; b < nblocks; b += kThinFinalLanes) s[0] += (double)partial[(long long)b * NV + i];
        red[rl][i] = (s[0] + s[1]) + (s[2] + s[3]);
    }
    __syncthreads();
    if (threadIdx.x >= NV) return;
    double t = 0;
#pragma unroll
    for (int l = 0; l < kThinFinalLanes; ++l) t += red[l][i];
    const float v = (float)t;
    if (i < kThinMaxCout * kThinMaxCin) {
        const int o = i / kThinMaxCin, c = i - o * kThinMaxCin;
        if (o < cout && c < cin) dw[o * cin + c] = v;
    } else if (dbias) {
        const int o = i - kThinMaxCout * kThinMaxCin;
        if (o < cout) dbias[o] = v;
    }
}

float* conv_workspace(hipStream_t stream, size_t bytes);   // conv_igemm.hip: grow-only per-stream scratch

static int thin_check(long long npix, int cin, int cout, const void* p, int cs, int need, const char* what) {
    W2L_REQUIRE(npix >= 1 && cin >= 1 && cin <= kThinMaxCin && cout >= 1 && cout <= kThinMaxCout,
                "%s: the thin 1x1 path serves cin <= %d, cout <= %d (got %d -> %d)", what, kThinMaxCin, kThinMaxCout, cin, cout);
    W2L_REQUIRE(p && cs >= need && (cs & 7) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0,
                "%s: tensor must be 16-byte aligned with a channel stride that is a multiple of 8 and >= %d (cs=%d)", what, need, cs);
    return W2L_OK;
}

static int thin_grid(long long npix) {
    long long g = (npix + 255) / 256;
    if (g > 2048) g = 2048;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace w2l

using namespace w2l;

extern "C" {

int w2l_thin1x1_forward_bf16(void* stream, long long npix, int cin, int cout, const void* x, int x_cs, const float* w,
                             const float* bias, int act, void* y, int y_cs) {
    const int cin8 = (cin + 7) & ~7;
    if (thin_check(npix, cin, cout, x, x_cs, cin8, "thin1x1_forward_bf16 x") != W2L_OK ||
        thin_check(npix, cin, cout, y, y_cs, 8, "thin1x1_forward_bf16 y") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(w && act >= W2L_ACT_NONE && act <= W2L_ACT_LEAKY, "thin1x1_forward_bf16: bad argument");
    ThinArgs a = {};
    a.x = static_cast<const __bf16*>(x); a.out = static_cast<__bf16*>(y); a.w = w; a.bias = bias; a.npix = npix;
    a.cin = cin; a.cin8 = cin8; a.cout = cout; a.x_cs = x_cs; a.out_cs = y_cs; a.act = act;
    hipLaunchKernelGGL(thin1x1_forward_bf16_kernel, dim3(thin_grid(npix)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_thin1x1_dgrad_bf16(void* stream, long long npix, int cin, int cout, const void* dz, int dz_cs, const float* w,
                           const void* res, int res_cs, void* dx, int dx_cs) {
    const int cin8 = (cin + 7) & ~7;
    if (thin_check(npix, cin, cout, dz, dz_cs, 8, "thin1x1_dgrad_bf16 dz") != W2L_OK ||
        thin_check(npix, cin, cout, dx, dx_cs, cin8, "thin1x1_dgrad_bf16 dx") != W2L_OK ||
        (res != nullptr && thin_check(npix, cin, cout, res, res_cs, cin8, "thin1x1_dgrad_bf16 res") != W2L_OK))
        return W2L_ERR_ARG;
    W2L_REQUIRE(w, "thin1x1_dgrad_bf16: NULL weight");
    ThinArgs a = {};
    a.dz = static_cast<const __bf16*>(dz); a.res = static_cast<const __bf16*>(res); a.out = static_cast<__bf16*>(dx); a.w = w;
    a.npix = npix; a.cin = cin; a.cin8 = cin8; a.cout = cout; a.dz_cs = dz_cs; a.res_cs = res_cs; a.out_cs = dx_cs;
    hipLaunchKernelGGL(thin1x1_dgrad_bf16_kernel, dim3(thin_grid(npix)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_thin1x1_wgrad_bf16(void* stream, long long npix, int cin, int cout, const void* x, int x_cs, const void* dz, int dz_cs,
                           float* dweight, float* dbias) {
    const int cin8 = (cin + 7) & ~7;
    if (thin_check(npix, cin, cout, x, x_cs, cin8, "thin1x1_wgrad_bf16 x") != W2L_OK ||
        thin_check(npix, cin, cout, dz, dz_cs, 8, "thin1x1_wgrad_bf16 dz") != W2L_OK)
        return W2L_ERR_ARG;
    W2L_REQUIRE(dweight, "thin1x1_wgrad_bf16: NULL output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    long long nb = (npix + 256 * 8 - 1) / (256 * 8);      // >= 8 pixels per thread: the register partials are worth their fold
    if (nb > 512) nb = 512;                               // two workgroups per CU; every partial row is one more row of the final walk
    if (nb < 1) nb = 1;
    constexpr int NV = kThinMaxCout * kThinMaxCin + kThinMaxCout;
    ThinArgs a = {};
    a.x = static_cast<const __bf16*>(x); a.dz = static_cast<const __bf16*>(dz); a.npix = npix;
    a.cin = cin; a.cin8 = cin8; a.cout = cout; a.x_cs = x_cs; a.dz_cs = dz_cs;
    a.partial = conv_workspace(s, (size_t)nb * NV * sizeof(float));
    if (!a.partial) return W2L_ERR_NOMEM;
    hipLaunchKernelGGL(thin1x1_wgrad_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(thin1x1_wgrad_final_kernel, dim3(1), dim3(1024), 0, s, a.partial, (int)nb, cin, cout, dweight, dbias);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // extern "C"

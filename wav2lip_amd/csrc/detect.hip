// S3FD face-detector glue kernels (reference face_detection/detection/sfd/): everything between the 3x3 convolutions, which
// are the SAME fused conv launches as the generator's (conv_igemm.hip / conv_wino.hip with bias + ReLU, no BatchNorm).
//   w2l_s3fd_pack       detect.py:57-58 + api.py:62  uint8 BGR frames -> RGB order, minus (104,117,123), fp32 NHWC4
//   w2l_maxpool2x2      net_s3fd.py:75,79,85,91,97   F.max_pool2d(h, 2, 2)
//   w2l_l2norm_scale    net_s3fd.py:6-19             x / (sqrt(sum_c x^2) + 1e-10) * weight[c]
//   w2l_s3fd_decode     net_s3fd.py:123-126 (max-out background), detect.py:66-84 (softmax, priors, decode)
//   w2l_s3fd_nms        sfd_detector.py:39-45 gate + bbox.py:44-64 greedy NMS, bit-exact keep list
// All HBM-bound, NHWC, float4 where the channel count allows.
#include "w2l_common.h"

namespace w2l {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void s3fd_pack_kernel(long long npix, const uint8_t* __restrict__ x, float* __restrict__ y, int y_cs,
                                 float m0, float m1, float m2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const uint8_t* p = x + i * 3;
        // images[..., ::-1] (api.py:62) then float64 subtraction of the per-channel mean, cast to float32 (detect.py:57-63)
        const float c0 = (float)((double)p[2] - (double)m0);
        const float c1 = (float)((double)p[1] - (double)m1);
        const float c2 = (float)((double)p[0] - (double)m2);
        float* o = y + i * y_cs;
        if (y_cs >= 4 && (y_cs & 3) == 0) *reinterpret_cast<f32x4*>(o) = f32x4{c0, c1, c2, 0.f};
        else { o[0] = c0; o[1] = c1; o[2] = c2; }
    }
}

// one thread per (output pixel, float4 channel group)
__global__ void maxpool2x2_kernel(int N, int H, int W, int C4, const float* __restrict__ x, int x_cs, float* __restrict__ y,
                                  int y_cs) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long pix = i / C4;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        const float* p = x + (((long long)n * H + 2 * oy) * W + 2 * ox) * x_cs + c4 * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + x_cs);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + (long long)W * x_cs);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p + (long long)W * x_cs + x_cs);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
        *reinterpret_cast<f32x4*>(y + (((long long)n * Ho + oy) * Wo + ox) * y_cs + c4 * 4) = o;
    }
}

// one wave per pixel: norm over channels, then scale by weight[c]
__global__ void l2norm_scale_kernel(long long rows, int C, const float* __restrict__ x, int x_cs,
                                    const float* __restrict__ w, float* __restrict__ y, int y_cs) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + row * x_cs;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float norm = sqrtf(s) + 1e-10f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] / norm * ww[e];
        *reinterpret_cast<f32x4*>(y + row * y_cs + c) = o;
    }
}

// one thread per feature-map position: (x1, y1, x2, y2, score)
__global__ void s3fd_decode_kernel(int B, int FH, int FW, int stride, const float* __restrict__ cls, int cls_cs, int ncls,
                                   const float* __restrict__ reg, int reg_cs, float* __restrict__ out) {
    const long long total = (long long)B * FH * FW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int wx = (int)(i % FW);
        const int hy = (int)((i / FW) % FH);
        const float* c = cls + i * cls_cs;
        float bg, fg;
        if (ncls == 4) { bg = fmaxf(fmaxf(c[0], c[1]), c[2]); fg = c[3]; }   // max-out background label
        else { bg = c[0]; fg = c[1]; }
        // F.softmax over the two classes
        const float mx = fmaxf(bg, fg);
        const float eb = expf(bg - mx), ef = expf(fg - mx);
        const float score = ef / (eb + ef);
        const float* l = reg + i * reg_cs;
        const float axc = (float)stride / 2.f + (float)wx * (float)stride;
        const float ayc = (float)stride / 2.f + (float)hy * (float)stride;
        const float pw = (float)(stride * 4);
        // decode (bbox.py:91-108), variances (0.1, 0.2)
        float cx = axc + l[0] * 0.1f * pw;
        float cy = ayc + l[1] * 0.1f * pw;
        const float bw = pw * expf(l[2] * 0.2f);
        const float bh = pw * expf(l[3] * 0.2f);
        cx -= bw / 2.f;
        cy -= bh / 2.f;
        float* o = out + i * 5;
        o[0] = cx; o[1] = cy; o[2] = bw + cx; o[3] = bh + cy; o[4] = score;
    }
}


// ---- greedy non-maximum suppression of one image's candidate boxes (bbox.py:44-64; sfd_detector.py:39-45 gates the
// candidates at score > 0.05 first).  One 1024-thread workgroup per image:
//   1. candidates = table rows with score > gate, as 64-bit keys (score bits << 32 | row): positive floats order like their bit
//      patterns, so a descending key order is "score descending, later row first among equal scores" - the order a reversed
//      stable ascending sort gives (the reference reverses numpy's unstable argsort: equal scores have no defined order there)
//   2. rank sort: rank[c] = #{j : key[j] > key[c]} (all pairs, n^2 / 1024 compares per thread: a few hundred candidates per
//      frame, 10^4 at worst), order[rank] = row
//   3. the reference's loop: the best remaining box is kept; every remaining box whose overlap with it is NOT <= thresh is
//      removed (a NaN overlap - empty union - removes, as `np.where(ovr <= thresh)` does).  "Remaining" is a bit per
//      candidate in LDS; one parallel pass + one barrier per KEPT box.
// The overlap is the reference's float32 expression, operation by operation, no contraction: w * h / (area_i + area_j - w * h)
// with area = (x2 - x1 + 1) * (y2 - y1 + 1), so the keep list is bit-exact.
constexpr int kNmsThreads = 1024;
constexpr int kNmsMaxCand = 1 << 18;              // remaining-bits in LDS: 32 KB; bounds the boxes that PASS THE GATE, not the table

// sort key of a score: order-preserving for every float (sign bit set -> all bits flipped, else the sign bit set), so that negative
// scores and -0.0 rank below the positive ones as bbox.py:50 `scores.argsort()[::-1]` ranks them; ties by the higher row index
__device__ __forceinline__ unsigned nms_score_key(float sc) {
    const unsigned u = __float_as_uint(sc);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float nms_area(const float* b) {
    return __fmul_rn(__fadd_rn(__fsub_rn(b[2], b[0]), 1.f), __fadd_rn(__fsub_rn(b[3], b[1]), 1.f));
}

__global__ __launch_bounds__(kNmsThreads) void s3fd_nms_kernel(int P, const float* __restrict__ table, float gate, float thresh,
                                                               unsigned long long* __restrict__ keys, int* __restrict__ order,
                                                               int* __restrict__ keep, int* __restrict__ counts) {
    __shared__ unsigned s_alive[kNmsMaxCand / 32];
    __shared__ int s_n, s_kept;
    const int b = blockIdx.x, t = threadIdx.x;
    const float* tb = table + (long long)b * P * 5;
    unsigned long long* kb = keys + (long long)b * P;
    int* ob = order + (long long)b * P;
    int* kp = keep + (long long)b * P;
    if (t == 0) { s_n = 0; s_kept = 0; }
    __syncthreads();
    for (int i = t; i < P; i += kNmsThreads) {
        const float sc = tb[i * 5 + 4];
        if (sc > gate) kb[atomicAdd(&s_n, 1)] = ((unsigned long long)nms_score_key(sc) << 32) | (unsigned)i;
    }
    __syncthreads();
    const int n = s_n;
    if (n > kNmsMaxCand) {          // more survivors of the gate than the alive bitmap holds: reported, the caller's host pass runs
        if (t == 0) counts[b] = -1;
        return;
    }
    for (int c = t; c < n; c += kNmsThreads) {
        const unsigned long long k = kb[c];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += kb[j] > k ? 1 : 0;
        ob[rank] = (int)(k & 0xffffffffu);
    }
    for (int w = t; w < (n + 31) / 32; w += kNmsThreads) s_alive[w] = 0xffffffffu;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (!((s_alive[i >> 5] >> (i & 31)) & 1u)) continue;      // uniform: written before the last barrier
        const float* bi = tb + ob[i] * 5;
        const float x1 = bi[0], y1 = bi[1], x2 = bi[2], y2 = bi[3];
        const float ai = nms_area(bi);
        if (t == 0) kp[s_kept++] = ob[i];
        for (int j = i + 1 + t; j < n; j += kNmsThreads) {
            if (!((s_alive[j >> 5] >> (j & 31)) & 1u)) continue;
            const float* bj = tb + ob[j] * 5;
            const float xx1 = fmaxf(x1, bj[0]), yy1 = fmaxf(y1, bj[1]), xx2 = fminf(x2, bj[2]), yy2 = fminf(y2, bj[3]);
            const float w = fmaxf(0.f, __fadd_rn(__fsub_rn(xx2, xx1), 1.f)), h = fmaxf(0.f, __fadd_rn(__fsub_rn(yy2, yy1), 1.f));
            const float inter = __fmul_rn(w, h);
            const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, nms_area(bj)), inter));
            if (!(ovr <= thresh)) atomicAnd(&s_alive[j >> 5], ~(1u << (j & 31)));
        }
        __syncthreads();
    }
    if (t == 0) counts[b] = s_kept;
}

static inline int grid1d(long long work, int block, int cap = 16384) {
    long long g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace w2l

using namespace w2l;

extern "C" {

int w2l_s3fd_pack(void* stream, long long npix, const uint8_t* bgr, float* y, int y_cs) {
    W2L_REQUIRE(bgr && y && npix >= 1 && y_cs >= 3, "bad s3fd_pack arguments");
    W2L_REQUIRE((y_cs & 3) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) == 0, "s3fd_pack: y must be 16-byte aligned");
    hipLaunchKernelGGL(s3fd_pack_kernel, dim3(grid1d(npix, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), npix, bgr, y,
                       y_cs, 104.f, 117.f, 123.f);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_maxpool2x2(void* stream, int N, int H, int W, int C, const float* x, int x_cs, float* y, int y_cs) {
    W2L_REQUIRE(x && y && N >= 1 && H >= 2 && W >= 2 && C >= 4 && (C & 3) == 0, "bad maxpool2x2 arguments (C %% 4 == 0, H, W >= 2)");
    W2L_REQUIRE((x_cs & 3) == 0 && (y_cs & 3) == 0 && x_cs >= C && y_cs >= C &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                "maxpool2x2: 16-byte aligned tensors with channel strides that are multiples of 4");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid1d(total, 256, 65536)), dim3(256), 0, static_cast<hipStream_t>(stream), N, H,
                       W, C / 4, x, x_cs, y, y_cs);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_l2norm_scale(void* stream, long long rows, int C, const float* x, int x_cs, const float* weight, float* y, int y_cs) {
    W2L_REQUIRE(x && y && weight && rows >= 1 && C >= 4 && (C & 3) == 0, "bad l2norm_scale arguments");
    W2L_REQUIRE((x_cs & 3) == 0 && (y_cs & 3) == 0 && x_cs >= C && y_cs >= C &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(weight)) & 15) == 0,
                "l2norm_scale: 16-byte aligned tensors with channel strides that are multiples of 4");
    const long long blocks = (rows + 3) / 4;
    W2L_REQUIRE(blocks < (1ll << 31), "too many rows");
    hipLaunchKernelGGL(l2norm_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rows, C, x,
                       x_cs, weight, y, y_cs);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_s3fd_decode(void* stream, int B, int FH, int FW, int stride, const float* cls, int cls_cs, int ncls, const float* reg,
                    int reg_cs, float* out) {
    W2L_REQUIRE(cls && reg && out && B >= 1 && FH >= 1 && FW >= 1 && stride >= 1, "bad s3fd_decode arguments");
    W2L_REQUIRE((ncls == 2 || ncls == 4) && cls_cs >= ncls && reg_cs >= 4, "s3fd_decode: ncls must be 2 or 4");
    const long long total = (long long)B * FH * FW;
    hipLaunchKernelGGL(s3fd_decode_kernel, dim3(grid1d(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), B, FH, FW,
                       stride, cls, cls_cs, ncls, reg, reg_cs, out);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_s3fd_nms(void* stream, int B, int P, const float* table, float gate, float thresh, int* keep, int* counts,
                 void* scratch, long long scratch_bytes) {
    W2L_REQUIRE(table && keep && counts && scratch && B >= 1 && P >= 1, "bad s3fd_nms arguments");
    W2L_REQUIRE((long long)B * P * 5 < (1ll << 31), "s3fd_nms: table too large");
    W2L_REQUIRE(scratch_bytes >= (long long)B * P * 12 && (reinterpret_cast<uintptr_t>(scratch) & 7) == 0,
                "s3fd_nms: scratch must hold 12 bytes per box (8-byte aligned), got %lld for %d x %d", scratch_bytes, B, P);
    unsigned long long* keys = static_cast<unsigned long long*>(scratch);
    int* order = reinterpret_cast<int*>(keys + (long long)B * P);
    hipLaunchKernelGGL(s3fd_nms_kernel, dim3(B), dim3(kNmsThreads), 0, static_cast<hipStream_t>(stream), P, table, gate, thresh,
                       keys, order, keep, counts);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // extern "C"

// The two cv2.resize calls either side of the generator and the paste-back, on uint8 frames resident in HBM
// (reference inference.py:121-126 face crop -> 96x96; :270-271 generated 96x96 -> box size, pasted into the frame).
// cv::resize(INTER_LINEAR) semantics for CV_8UC3 (OpenCV 4.1.0 modules/imgproc/src/resize.cpp): copy when the sizes match,
// the 2x2 INTER_AREA fast path for an exact 2x down-scale, otherwise the 11-bit fixed-point bilinear path (horizontal pass
// in int32, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2).  One thread per destination pixel (3 channels);
// byte traffic only — HBM-bound, no LDS needed (each source pixel is touched by <= 4 neighbouring threads: L1/L2 hits).
#include "w2l_common.h"

namespace w2l {

struct AxisTap {
    int s0, s1;     // source indices
    int a0, a1;     // 11-bit coefficients
};

// horizontal convention (coefficient zeroed at the clamped edges); vertical = the same without the edge zeroing, rows clipped
__device__ __forceinline__ AxisTap axis_tap(int d, int ssize, int dsize, bool horizontal) {
    const double inv = __ddiv_rn((double)dsize, (double)ssize);
    const double scale = __ddiv_rn(1.0, inv);
    float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    AxisTap t;
    if (horizontal) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
        t.s0 = s;
        t.s1 = min(s + 1, ssize - 1);
    } else {
        t.s0 = min(max(s, 0), ssize - 1);
        t.s1 = min(max(s + 1, 0), ssize - 1);
    }
    t.a1 = (int)rintf(__fmul_rn(f, 2048.f));
    t.a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    return t;
}

// src: pointer to pixel (0,0) of the source region, row stride in bytes; Hs x Ws -> pixel (dx,dy) of an h x w result
__device__ __forceinline__ void resize_px(const uint8_t* __restrict__ src, long long row_stride, int Hs, int Ws, int dx,
                                          int dy, int w, int h, uint8_t out[3]) {
    if (Hs == h && Ws == w) {
        const uint8_t* p = src + dy * row_stride + dx * 3;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
        return;
    }
    if (Hs == 2 * h && Ws == 2 * w) {
        const uint8_t* p0 = src + (2 * dy) * row_stride + (2 * dx) * 3;
        const uint8_t* p1 = p0 + row_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = (uint8_t)(((int)p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2);
        return;
    }
    const AxisTap tx = axis_tap(dx, Ws, w, true);
    const AxisTap ty = axis_tap(dy, Hs, h, false);
    const uint8_t* r0 = src + ty.s0 * row_stride;
    const uint8_t* r1 = src + ty.s1 * row_stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[tx.s0 * 3 + c] * tx.a0 + (int)r0[tx.s1 * 3 + c] * tx.a1;
        const int h1 = (int)r1[tx.s0 * 3 + c] * tx.a0 + (int)r1[tx.s1 * 3 + c] * tx.a1;
        int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out[c] = (uint8_t)min(max(v, 0), 255);
    }
}

__global__ void crop_resize_kernel(int B, const uint8_t* __restrict__ frames, int H, int W,
                                   const int32_t* __restrict__ frame_idx, const int32_t* __restrict__ boxes, int S,
                                   uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int4 box = *reinterpret_cast<const int4*>(boxes + 4 * b);   // y1, y2, x1, x2
    const int fi = frame_idx ? frame_idx[b] : b;
    const uint8_t* src = frames + ((long long)fi * H + box.x) * W * 3 + (long long)box.z * 3;
    const int Hs = box.y - box.x, Ws = box.w - box.z;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S * S; i += gridDim.x * blockDim.x) {
        const int dy = i / S, dx = i - dy * S;
        uint8_t px[3];
        resize_px(src, (long long)W * 3, Hs, Ws, dx, dy, S, S, px);
        uint8_t* o = out + ((long long)b * S * S + i) * 3;
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
    }
}

__global__ void resize_paste_kernel(int B, const uint8_t* __restrict__ pred, int S, const int32_t* __restrict__ boxes,
                                    const int32_t* __restrict__ frame_idx, uint8_t* __restrict__ frames, int H, int W) {
    const int b = blockIdx.y;
    const int4 box = *reinterpret_cast<const int4*>(boxes + 4 * b);
    const int fo = frame_idx ? frame_idx[b] : b;
    const int h = box.y - box.x, w = box.w - box.z;
    const uint8_t* src = pred + (long long)b * S * S * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
        const int dy = i / w, dx = i - dy * w;
        uint8_t px[3];
        resize_px(src, (long long)S * 3, S, S, dx, dy, w, h, px);
        uint8_t* o = frames + (((long long)fo * H + box.x + dy) * W + box.z + dx) * 3;
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
    }
}

// whole-frame resize (inference.py:203: `cv2.resize(frame, (w // resize_factor, h // resize_factor))`)
__global__ void resize_frames_kernel(int B, const uint8_t* __restrict__ src, int Hs, int Ws, uint8_t* __restrict__ dst,
                                     int Hd, int Wd) {
    const int b = blockIdx.y;
    const uint8_t* s = src + (long long)b * Hs * Ws * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hd * Wd; i += gridDim.x * blockDim.x) {
        const int dy = i / Wd, dx = i - dy * Wd;
        uint8_t px[3];
        resize_px(s, (long long)Ws * 3, Hs, Ws, dx, dy, Wd, Hd, px);
        uint8_t* o = dst + ((long long)b * Hd * Wd + i) * 3;
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
    }
}

}  // namespace w2l

using namespace w2l;

extern "C" {

int w2l_crop_resize_u8(void* stream, int B, const uint8_t* frames, int H, int W, const int32_t* frame_idx,
                       const int32_t* boxes, int S, uint8_t* out) {
    W2L_REQUIRE(frames && boxes && out && B >= 1 && H >= 1 && W >= 1 && S >= 1, "bad crop_resize arguments");
    W2L_REQUIRE(B <= 65535 && (reinterpret_cast<uintptr_t>(boxes) & 15) == 0, "crop_resize: B <= 65535 and 16-byte aligned boxes");
    int gx = ceil_div(S * S, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(crop_resize_kernel, dim3(gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), B, frames, H, W,
                       frame_idx, boxes, S, out);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_resize_u8(void* stream, int B, const uint8_t* src, int Hs, int Ws, uint8_t* dst, int Hd, int Wd) {
    W2L_REQUIRE(src && dst && B >= 1 && Hs >= 1 && Ws >= 1 && Hd >= 1 && Wd >= 1, "bad resize arguments");
    W2L_REQUIRE(B <= 65535 && (long long)Hd * Wd < (1ll << 31) && (long long)Hs * Ws < (1ll << 31), "resize: frame too large");
    int gx = ceil_div(Hd * Wd, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(resize_frames_kernel, dim3(gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), B, src, Hs, Ws, dst,
                       Hd, Wd);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_resize_paste_u8(void* stream, int B, const uint8_t* pred, int S, const int32_t* boxes, const int32_t* frame_idx,
                        uint8_t* frames, int H, int W, int max_box_pixels) {
    W2L_REQUIRE(pred && boxes && frames && B >= 1 && H >= 1 && W >= 1 && S >= 1 && max_box_pixels >= 1,
                "bad resize_paste arguments");
    W2L_REQUIRE(B <= 65535 && (reinterpret_cast<uintptr_t>(boxes) & 15) == 0, "resize_paste: B <= 65535 and 16-byte aligned boxes");
    int gx = ceil_div(max_box_pixels, 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(resize_paste_kernel, dim3(gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), B, pred, S, boxes,
                       frame_idx, frames, H, W);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // extern "C"

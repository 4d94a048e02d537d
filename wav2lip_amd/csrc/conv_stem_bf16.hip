// The 7x7 STEM convolutions of the bf16-storage training path - a handful of input channels at full resolution: the generator's
// first layer (6 -> 16 at 96x96, models/wav2lip.py:13), SyncNet's (15 -> 32 at 48x96, models/syncnet.py:12), the discriminator's
// (3 -> 32 at 48x96, models/wav2lip.py:131) - forward only (their inputs are images: no data gradient).
//
// Why.  On the implicit GEMM the A operand of such a layer is gathered 16 bytes per (pixel, tap): 49 taps x 16 B = 784 B from L2 into
// LDS per pixel for a 16 B pixel - 2.3 GB per launch of the generator's first layer (320 frames) whose input is 47 MB; the layer runs
// at 82 TFLOP/s, 0.34 ms, against a byte floor of 0.03 ms.  Same remedy as conv_box_bf16.hip: one 8-wave workgroup per
// CU, persistent over 16x16-pixel tiles, the WHOLE weight set resident in LDS (49 x cin_p x 32 couts: 26 / 51 KB, fetched once) and
// the tile's 22x22-pixel input box (tile + halo of 3: 8 / 16 KB) fetched once per tile by LDS-DMA while the previous tile is
// computed; a tap is a row offset into the box.  Weights are the MFMA's A operand, pixels its B operand: a lane ends up with 16 couts
// of one pixel, the epilogue is register-only (v_permlane32_swap makes 16-byte rows), stores leave after the tile barrier.
// K order = (tap, channel): with 16 channels per pixel a 32x32x16 MFMA takes one tap (the two half waves read the two 16-byte halves of
// the box pixel); with 8 channels it takes TWO taps (half wave h reads tap 2s + h), 25 steps, the 50th tap being zero weights.
// BatchNorm statistics are not taken here (such a launch reports "no partials" and the stand-alone reduction over z follows: one
// pass over a 16- or 32-channel tensor).
#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kStemOob = 0x80000000u;
constexpr int kStemT = 16;                       // tile edge
constexpr int kStemE = kStemT + 6;               // box edge (halo 3)
constexpr int kStemPix = kStemE * kStemE;        // 484

struct StemArgs {
    const void* x;
    void* y;
    const void* w;         // bf16 [cout_p][kp], K = (tap, c), c < cin_p
    const float* scale;
    const float* shift;
    const int* taps;       // 49 x (dy & 0xffff) | (dx << 16), |dy|, |dx| <= 3
    int N, H, W, x_cs, y_cs, cout, cout_p, kp, act;
    int tiles_x, tiles_y, ntiles;
};

typedef __attribute__((address_space(3))) void* stem_lds_t;

// CP = channels per pixel in the tensor (8 or 16)
template <int CP>
__global__ __launch_bounds__(512, 1) void conv_stem_bf16_kernel(const StemArgs a) {
    constexpr int RB = CP * 2;                               // box pixel row (bytes)
    constexpr int NSTEP = CP == 16 ? 49 : 25;                // K-steps of 16
    constexpr int WROW = NSTEP * 32 + 16;                    // weight row (bytes), + one pad slot: 1584 / 816 = 12 mod 32 dwords
    constexpr int WITEMS = 32 * (WROW / 16);                 // 16-byte items of the 32 resident rows
    constexpr int WDMA = (WITEMS + 63) / 64;
    constexpr int BDMA = (kStemPix * RB + 1023) / 1024;      // one-KB requests per box: 8 / 16
    constexpr int WVALID = 49 * CP * 2;                      // bytes of a weight row that exist (the rest: zero)
    __shared__ __attribute__((aligned(16))) char Wl[WDMA * 1024];
    __shared__ __attribute__((aligned(16))) char Box0[BDMA * 1024];
    __shared__ __attribute__((aligned(16))) char Box1[BDMA * 1024];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5;
    const int n = lane & 31;

    const long long npix = (long long)a.N * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)(((npix - 1) * a.x_cs + CP) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.cout_p * a.kp * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + 32) * 2), 0x00020000);

    // tap offsets (dy * kStemE + dx) in scalar registers; with 8 channels a K-step pairs taps 2s and 2s + 1 (the 50th: offset 0, zero weights)
    int tapoff[50];
#pragma unroll
    for (int i = 0; i < 49; ++i) {
        const int tv = __builtin_amdgcn_readfirstlane(a.taps[i]);
        tapoff[i] = ((int)(short)(tv & 0xffff)) * kStemE + (tv >> 16);
    }
    tapoff[49] = 0;

    // ---- the weight set, once: item q = row * (WROW / 16) + c of the padded LDS rows <- global row * kp * 2 + c * 16 (zero beyond)
    for (int i = wave; i < WDMA; i += 8) {
        const int q = i * 64 + lane;
        const int row = q / (WROW / 16), c = q - row * (WROW / 16);
        const bool ok = q < WITEMS && row < a.cout_p && c * 16 < WVALID;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (stem_lds_t)(Wl + i * 1024), 16, (int)(ok ? (unsigned)(row * a.kp * 2 + c * 16) : kStemOob), 0, 0, 0);
    }

    // ---- this thread's box items: request i = wave + 8 j covers 16-byte items 64 i + lane; item -> box pixel bp, 16-byte slot of its row
    constexpr int SL = RB / 16;                  // slots per box pixel: 1 / 2
    constexpr int NB = (BDMA + 7) / 8;
    int b_it[NB];                                // by | bx << 8 | slot << 16, or -1
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i = wave + 8 * j;
        const int item = i * 64 + lane;
        const int bp = item / SL, slot = item - bp * SL;
        const int by = bp / kStemE, bx = bp - by * kStemE;
        b_it[j] = (i < BDMA && bp < kStemPix) ? (by | (bx << 8) | (slot << 16)) : -1;
    }
    auto tile_coords = [&](int tile, int& img, int& ty0, int& tx0) {
        const int per = a.tiles_x * a.tiles_y;
        img = tile / per;
        const int r = tile - img * per;
        const int ty = r / a.tiles_x;
        ty0 = ty * kStemT;
        tx0 = (r - ty * a.tiles_x) * kStemT;
    };
    auto box_dma = [&](int tile, char* box) {
        int img, ty0, tx0;
        tile_coords(tile, img, ty0, tx0);
        const bool tile_ok = tile < a.ntiles;
        const int base = ((img * a.H + ty0 - 3) * a.W + tx0 - 3) * a.x_cs * 2;      // box pixel (0, 0); only used with in-range (by, bx)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = wave + 8 * j;
            if (i < BDMA) {
                const int by = b_it[j] & 0xff, bx = (b_it[j] >> 8) & 0xff, sl = (b_it[j] >> 16) & 1;
                const bool ok = tile_ok & (b_it[j] >= 0) & ((unsigned)(ty0 - 3 + by) < (unsigned)a.H) & ((unsigned)(tx0 - 3 + bx) < (unsigned)a.W);
                const unsigned rel = (unsigned)((by * a.W + bx) * a.x_cs * 2 + sl * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (stem_lds_t)(box + i * 1024), 16, (int)(ok ? (unsigned)base + rel : kStemOob), 0, 0, 0);
            }
        }
    };

    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    const bool is_sigmoid = a.act == W2L_ACT_SIGMOID;
    const int py = 2 * wave + (n >> 4), px = n & 15;
    const int pbase = (py + 3) * kStemE + (px + 3);
    const int wrow0 = n * WROW + h * 16;
    const int cout8 = (a.cout + 7) & ~7;

    auto compute = [&](int tile, const char* box) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int img, ty0, tx0;
        tile_coords(tile, img, ty0, tx0);
        const bool pix_ok = (ty0 + py < a.H) & (tx0 + px < a.W);
        const unsigned opix = (unsigned)((img * a.H + ty0 + py) * a.W + tx0 + px);
        // NSTEP K-steps as one software pipeline: the two fragments of step s + 2 are requested before the MFMA of step s
        constexpr int FD = 2;
        bf16x8 fx[FD + 1], fw[FD + 1];
        auto frag = [&](int step, int set) {
            // 16 channels: tap = step, half wave h reads the 16-byte half h of the box pixel;  8 channels: tap = 2 step + h
            const int p = pbase + (CP == 16 ? tapoff[step] : (h ? tapoff[2 * step + 1] : tapoff[2 * step]));
            fx[set] = *reinterpret_cast<const bf16x8*>(box + p * RB + (CP == 16 ? h * 16 : 0));
            fw[set] = *reinterpret_cast<const bf16x8*>(Wl + wrow0 + step * 32);
        };
#pragma unroll
        for (int i = 0; i < FD; ++i) frag(i, i);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + FD < NSTEP) frag(step + FD, (step + FD) % (FD + 1));
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[step % (FD + 1)], fx[step % (FD + 1)], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: register 4 g + e of lane (n, h) = cout 8 g + 4 h + e of pixel n; the swap leaves lane (n, 0) with couts
        // 16 k .. 16 k + 7 and lane (n, 1) with 16 k + 8 .. 16 k + 15 (inline asm with both operands read-write: conv_box_bf16.hip)
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = acc[r];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(vv[8 * k + e]), "+v"(vv[8 * k + 4 + e]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next box landed (this wave's requests)
        __syncthreads();                                       // ... every wave's; nobody still reads this tile's box
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c0 = 16 * k + 8 * h;                     // this lane's 8 consecutive couts
            if (c0 < cout8) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool cv = c0 + j < a.cout;
                    float v = vv[8 * k + j] * (cv && a.scale ? a.scale[c0 + j] : 1.f) + (cv && a.shift ? a.shift[c0 + j] : 0.f);
                    if (is_sigmoid) v = 1.0f / (1.0f + expf(-v));
                    else v = act_leaky(v, neg_slope);
                    o[j] = (__bf16)(cv ? v : 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry,
                                                       (int)(pix_ok ? (opix * (unsigned)a.y_cs + (unsigned)c0) * 2u : kStemOob), 0, 0);
            }
        }
    };

    int tile = blockIdx.x;
    box_dma(tile, Box0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    while (tile < a.ntiles) {
        box_dma(tile + gridDim.x, Box1);
        compute(tile, Box0);
        tile += gridDim.x;
        if (tile >= a.ntiles) break;
        box_dma(tile + gridDim.x, Box0);
        compute(tile, Box1);
        tile += gridDim.x;
    }
}

// ---- host side (called from conv_bf16.hip's launcher).  A shape-only rule: 7x7 / stride 1 / pad 3, 8 or 16 channels per pixel, at most
// 32 couts, a LARGE launch (>= 1024 tiles: the weight set is fetched once per workgroup) whose extents fill their 16x16 tiles to 85 %
bool stem_ok(int transposed, int kh, int kw, int sh, int sw, int ph, int pw, int cin_p, int cout, int N, int H, int W) {
    if (transposed || kh != 7 || kw != 7 || sh != 1 || sw != 1 || ph != 3 || pw != 3 || (cin_p != 8 && cin_p != 16) || cout > 32) return false;
    const long long ty = (H + kStemT - 1) / kStemT, tx = (W + kStemT - 1) / kStemT;
    return (long long)N * ty * tx >= 1024 && (long long)H * W * 100 >= 85ll * ty * tx * kStemT * kStemT;
}

int stem_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* w, int cout_p, int kp, const float* scale,
                const float* shift, const int* taps, int N, int H, int W, int cin_p, int cout, int act) {
    StemArgs a;
    a.x = x; a.y = y; a.w = w; a.scale = scale; a.shift = shift; a.taps = taps;
    a.N = N; a.H = H; a.W = W; a.x_cs = x_cs; a.y_cs = y_cs; a.cout = cout; a.cout_p = cout_p; a.kp = kp; a.act = act;
    a.tiles_x = (W + kStemT - 1) / kStemT; a.tiles_y = (H + kStemT - 1) / kStemT;
    const long long tiles = (long long)N * a.tiles_x * a.tiles_y;
    W2L_REQUIRE(tiles < (1ll << 30), "grid too large");
    a.ntiles = (int)tiles;
    // 8 channels: 43 KB of LDS and 88 registers per workgroup - two workgroups per CU hide each other's per-tile barrier and DMA wait
    const long long wgs = cin_p == 8 ? 512 : 256;
    const dim3 grid((unsigned)(tiles < wgs ? tiles : wgs)), block(512);
    if (cin_p == 16) hipLaunchKernelGGL((conv_stem_bf16_kernel<16>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_stem_bf16_kernel<8>), grid, block, 0, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

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
// The same kernel serves the 3x3 / stride-1 layers with FEW channels on one side at full resolution (template KS = 3): the output
// block 80 -> 32 at 96x96 (models/wav2lip.py:83; 45 K-steps, 46 KB of weights, 176-byte padded box rows), its data gradient 32 -> 80
// (three cout tiles) and the 32 -> 32 residual blocks at 48x48 (models/wav2lip.py:15-17), forward and data gradient - all of them
// L2-bound on the implicit GEMM for the same reason.  Box rows that are a multiple of 64 bytes or 160 bytes get one 16-byte pad slot
// (bank-conflict-free fragment reads without a swizzle).
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

struct StemArgs {
    const void* x;
    void* y;
    const void* res;       // optional residual rows (may alias y: accumulate)
    const void* w;         // bf16 [cout_p][kp], K = (tap, c), c < cin_p
    const float* scale;
    const float* shift;
    const int* taps;       // 49 x (dy & 0xffff) | (dx << 16), |dy|, |dx| <= 3
    int N, H, W, x_cs, y_cs, res_cs, cout, cout_p, kp, act;
    int tiles_x, tiles_y, ntiles;
};

typedef __attribute__((address_space(3))) void* stem_lds_t;

// CP = channels per pixel in the tensor (8, 16, 32, 80), KS = kernel side (7 or 3), MT = 32-cout tiles, RES = residual rows added
template <int CP, int KS, int MT, bool RES>
__global__ __launch_bounds__(512, 1) void conv_stem_bf16_kernel(const StemArgs a) {
    constexpr int NTAP = KS * KS;
    constexpr int HALO = KS / 2;
    constexpr int kStemE = kStemT + 2 * HALO;                // box edge
    constexpr int kStemPix = kStemE * kStemE;
    constexpr int RBV = CP * 2;                              // bytes of a box pixel that exist
    constexpr int RB = CP <= 16 ? RBV : RBV + 16;            // box pixel row (bytes): + one pad slot from 32 channels up
    constexpr int NSTEP = CP == 8 ? (NTAP + 1) / 2 : NTAP * (CP / 16);   // K-steps of 16
    constexpr int WROW = NSTEP * 32 + 16;                    // weight row (bytes), + one pad slot: 1584 / 816 = 12 mod 32 dwords
    constexpr int WITEMS = 32 * MT * (WROW / 16);            // 16-byte items of the resident rows
    constexpr int WDMA = (WITEMS + 63) / 64;
    constexpr int BDMA = (kStemPix * RB + 1023) / 1024;      // one-KB requests per box: 8 / 16
    constexpr int WVALID = NTAP * CP * 2;                    // bytes of a weight row that exist (the rest: zero)
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
    const int cout8 = (a.cout + 7) & ~7;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + cout8) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(RES && a.res ? a.res : a.y), 0, RES && a.res ? (int)(((npix - 1) * a.res_cs + cout8) * 2) : 0, 0x00020000);

    // tap offsets (dy * kStemE + dx) in scalar registers; with 8 channels a K-step pairs taps 2s and 2s + 1 (the 50th: offset 0, zero weights)
    int tapoff[NTAP + 1];
#pragma unroll
    for (int i = 0; i < NTAP; ++i) {
        const int tv = __builtin_amdgcn_readfirstlane(a.taps[i]);
        tapoff[i] = ((int)(short)(tv & 0xffff)) * kStemE + (tv >> 16);
    }
    tapoff[NTAP] = 0;

    // ---- the weight set, once: item q = row * (WROW / 16) + c of the padded LDS rows <- global row * kp * 2 + c * 16 (zero beyond)
    for (int i = wave; i < WDMA; i += 8) {
        const int q = i * 64 + lane;
        const int row = q / (WROW / 16), c = q - row * (WROW / 16);
        const bool ok = q < WITEMS && row < a.cout_p && c * 16 < WVALID;      // rows beyond cout_p (pad cout tiles): zero
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (stem_lds_t)(Wl + i * 1024), 16, (int)(ok ? (unsigned)(row * a.kp * 2 + c * 16) : kStemOob), 0, 0, 0);
    }

    // ---- this thread's box items: request i = wave + 8 j covers 16-byte items 64 i + lane; item -> box pixel bp, 16-byte slot of its row
    constexpr int SL = RB / 16;                  // slots per box pixel (the pad slot, if any, is the last)
    constexpr int NB = (BDMA + 7) / 8;
    int b_it[NB];                                // by | bx << 8 | slot << 16, or -1
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i = wave + 8 * j;
        const int item = i * 64 + lane;
        const int bp = item / SL, slot = item - bp * SL;
        const int by = bp / kStemE, bx = bp - by * kStemE;
        b_it[j] = (i < BDMA && bp < kStemPix && slot * 16 < RBV) ? (by | (bx << 8) | (slot << 16)) : -1;
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
        const int base = ((img * a.H + ty0 - HALO) * a.W + tx0 - HALO) * a.x_cs * 2;      // box pixel (0, 0); only used with in-range (by, bx)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = wave + 8 * j;
            if (i < BDMA) {
                const int by = b_it[j] & 0xff, bx = (b_it[j] >> 8) & 0xff, sl = (b_it[j] >> 16) & 15;
                const bool ok = tile_ok & (b_it[j] >= 0) & ((unsigned)(ty0 - HALO + by) < (unsigned)a.H) & ((unsigned)(tx0 - HALO + bx) < (unsigned)a.W);
                const unsigned rel = (unsigned)((by * a.W + bx) * a.x_cs * 2 + sl * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (stem_lds_t)(box + i * 1024), 16, (int)(ok ? (unsigned)base + rel : kStemOob), 0, 0, 0);
            }
        }
    };

    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    const bool is_sigmoid = a.act == W2L_ACT_SIGMOID;
    const int py = 2 * wave + (n >> 4), px = n & 15;
    const int pbase = (py + HALO) * kStemE + (px + HALO);
    const int wrow0 = n * WROW + h * 16;

    auto compute = [&](int tile, const char* box) {
        f32x16 acc[MT];
#pragma unroll
        for (int t2 = 0; t2 < MT; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
        int img, ty0, tx0;
        tile_coords(tile, img, ty0, tx0);
        const bool pix_ok = (ty0 + py < a.H) & (tx0 + px < a.W);
        const unsigned opix = (unsigned)((img * a.H + ty0 + py) * a.W + tx0 + px);
        u32x4 rv[MT][2];                                       // residual rows, requested before the matrix work
#pragma unroll
        for (int t2 = 0; t2 < MT; ++t2)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rv[t2][k] = u32x4{0u, 0u, 0u, 0u};
                const int c0 = 32 * t2 + 16 * k + 8 * h;
                if (RES && a.res)
                    rv[t2][k] = __builtin_amdgcn_raw_buffer_load_b128(
                        rr, (int)(pix_ok && c0 < cout8 ? (opix * (unsigned)a.res_cs + (unsigned)c0) * 2u : kStemOob), 0, 0);
            }
        // NSTEP K-steps as one software pipeline: the fragments of step s + 2 are requested before the MFMAs of step s
        constexpr int FD = 2;
        bf16x8 fx[FD + 1], fw[FD + 1][MT];
        auto frag = [&](int step, int set) {
            // >= 16 channels: tap = step / (CP / 16), the half wave h reads the 16-byte chunk 2 sub + h of the box pixel;
            // 8 channels: tap = 2 step + h, the whole 16-byte pixel
            constexpr int TS = CP == 8 ? 1 : CP / 16;
            const int tap = CP == 8 ? 0 : step / TS, sub = CP == 8 ? 0 : step - tap * TS;
            const int p = pbase + (CP == 8 ? (h ? tapoff[2 * step + 1] : tapoff[2 * step]) : tapoff[tap]);
            fx[set] = *reinterpret_cast<const bf16x8*>(box + p * RB + (CP == 8 ? 0 : sub * 32 + h * 16));
#pragma unroll
            for (int t2 = 0; t2 < MT; ++t2) fw[set][t2] = *reinterpret_cast<const bf16x8*>(Wl + wrow0 + t2 * 32 * WROW + step * 32);
        };
#pragma unroll
        for (int i = 0; i < FD; ++i) frag(i, i);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + FD < NSTEP) frag(step + FD, (step + FD) % (FD + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t2 = 0; t2 < MT; ++t2)
                acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[step % (FD + 1)][t2], fx[step % (FD + 1)], acc[t2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: register 4 g + e of lane (n, h) = cout 32 t2 + 8 g + 4 h + e of pixel n; the swap leaves lane (n, 0) with couts
        // 16 k .. 16 k + 7 and lane (n, 1) with 16 k + 8 .. 16 k + 15 of the tile (inline asm with both operands read-write: conv_box_bf16.hip)
        float vv[MT][16];
#pragma unroll
        for (int t2 = 0; t2 < MT; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[t2][r] = acc[t2][r];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(vv[t2][8 * k + e]), "+v"(vv[t2][8 * k + 4 + e]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next box landed (this wave's requests), residual rows here
        __syncthreads();                                       // ... every wave's; nobody still reads this tile's box
#pragma unroll
        for (int t2 = 0; t2 < MT; ++t2)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c0 = 32 * t2 + 16 * k + 8 * h;       // this lane's 8 consecutive couts
                if (c0 < cout8) {
                    const bf16x8 rb = __builtin_bit_cast(bf16x8, rv[t2][k]);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool cv = c0 + j < a.cout;
                        float v = vv[t2][8 * k + j] * (cv && a.scale ? a.scale[c0 + j] : 1.f) + (cv && a.shift ? a.shift[c0 + j] : 0.f) +
                                  (float)rb[j];
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

// ---- host side (called from conv_bf16.hip's launcher).  Shape-only rules: the layer families above, a LARGE launch (the weight set is
// fetched once per workgroup) whose extents fill their 16x16 tiles to 85 %.  `transposed` stride-1 layers arrive here as their
// equivalent forward geometry (same tap table convention: input offsets per tap).
static int stem_family(int kh, int kw, int sh, int sw, int ph, int pw, int cin_p, int cout) {
    if (sh != 1 || sw != 1 || kh != kw || ph != kh / 2 || pw != kw / 2) return 0;
    if (kh == 7 && (cin_p == 8 || cin_p == 16) && cout <= 32) return 1;     // stems
    if (kh == 3 && cin_p == 80 && cout <= 32) return 2;                     // output block
    if (kh == 3 && cin_p == 32 && cout <= 32) return 3;                     // 32 -> 32 residual blocks
    if (kh == 3 && cin_p == 32 && cout > 64 && cout <= 96) return 4;        // data gradient of the output block
    return 0;
}

bool stem_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cin_p, int cout, int N, int H, int W, bool has_res) {
    const int fam = stem_family(kh, kw, sh, sw, ph, pw, cin_p, cout);
    if (!fam) return false;
    // families 1 and 2 (7x7 stems, the 80 -> 32 output block) are instantiated without the residual read (no layer of the path
    // has one there): a launch that does carry a residual - or a data gradient accumulating into gx - stays on the implicit GEMM
    if (has_res && fam <= 2) return false;
    const long long ty = (H + kStemT - 1) / kStemT, tx = (W + kStemT - 1) / kStemT;
    return (long long)N * ty * tx >= (fam == 1 ? 1024 : 2048) && (long long)H * W * 100 >= 85ll * ty * tx * kStemT * kStemT;
}

int stem_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* res, int res_cs, const void* w, int cout_p,
                int kp, const float* scale, const float* shift, const int* taps, int N, int H, int W, int kh, int cin_p, int cout, int act) {
    StemArgs a;
    a.x = x; a.y = y; a.res = res; a.w = w; a.scale = scale; a.shift = shift; a.taps = taps;
    a.N = N; a.H = H; a.W = W; a.x_cs = x_cs; a.y_cs = y_cs; a.res_cs = res_cs; a.cout = cout; a.cout_p = cout_p; a.kp = kp; a.act = act;
    a.tiles_x = (W + kStemT - 1) / kStemT; a.tiles_y = (H + kStemT - 1) / kStemT;
    const long long tiles = (long long)N * a.tiles_x * a.tiles_y;
    W2L_REQUIRE(tiles < (1ll << 30), "grid too large");
    a.ntiles = (int)tiles;
    const int fam = stem_family(kh, kh, 1, 1, kh / 2, kh / 2, cin_p, cout);
    // two workgroups per CU where LDS and registers allow (they hide each other's per-tile barrier and DMA wait)
    const long long wgs = (fam == 1 && cin_p == 8) || fam == 3 ? 512 : 256;
    const dim3 grid((unsigned)(tiles < wgs ? tiles : wgs)), block(512);
    if (fam == 1 && cin_p == 16) hipLaunchKernelGGL((conv_stem_bf16_kernel<16, 7, 1, false>), grid, block, 0, stream, a);
    else if (fam == 1) hipLaunchKernelGGL((conv_stem_bf16_kernel<8, 7, 1, false>), grid, block, 0, stream, a);
    else if (fam == 2) hipLaunchKernelGGL((conv_stem_bf16_kernel<80, 3, 1, false>), grid, block, 0, stream, a);
    else if (fam == 3 && res) hipLaunchKernelGGL((conv_stem_bf16_kernel<32, 3, 1, true>), grid, block, 0, stream, a);
    else if (fam == 3) hipLaunchKernelGGL((conv_stem_bf16_kernel<32, 3, 1, false>), grid, block, 0, stream, a);
    else if (fam == 4 && res) hipLaunchKernelGGL((conv_stem_bf16_kernel<32, 3, 3, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_stem_bf16_kernel<32, 3, 3, false>), grid, block, 0, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

// 3x3 / stride 1 / 64 -> 64 channel convolutions of the bf16-storage training path with the INPUT BOX and the WHOLE WEIGHT SET
// resident in LDS: the residual blocks of the generator's last decoder stage at 96x96 (models/wav2lip.py:79-81 through
// models/conv.py:5-19), forward and data gradient, inside wav2lip_train.py:220-231.
//
// Why.  conv_bf16.hip fetches the A operand once per TAP: a 128-pixel tile of a 64-channel 3x3 layer moves 9 x 16 KB of input and
// 9 x 8 KB of weights from L2 into LDS for 2 us of matrix work - 3.4 GB + 1.7 GB per launch at 320 frames of 96x96 against 0.38 GB
// of input in HBM; those layers run at 0.45 of their byte floor.  Here a workgroup (8 waves, one per CU, persistent
// over 16x16-pixel tiles) keeps all 9 x 64 x 64 weights (74 KB, fetched once per launch) and two 18x18-pixel input boxes (41 KB
// each: the tile + its halo, fetched ONCE per tile by LDS-DMA while the previous tile is computed) in LDS; a tap is a row offset
// into the box.  L2 -> LDS traffic per 256 pixels: 41 KB instead of 432 KB; no barrier inside a tile.
//
// Roles are swapped against conv_bf16.hip: the WEIGHTS are the MFMA's A operand (rows = couts) and the pixels its B operand
// (columns), so that a lane ends up with 16 couts of ONE pixel per accumulator and the epilogue needs no LDS transpose - there is
// no LDS left for one.  Wave w owns pixel rows 2w, 2w+1 of the tile (32 pixels) x all 64 couts: 2 accumulators, per K-substep two
// weight fragments + one pixel fragment for two 32x32x16 MFMAs.  Box rows are 128 bytes (64 channels): the 16-byte slot s of box
// pixel p holds K chunk s ^ (p & 7) (applied to the DMA's SOURCE address; the reader recomputes it for the shifted pixel), weight
// rows are padded to 1168 bytes: both fragment reads are bank-conflict free.
// Epilogue per lane and accumulator: four groups of 4 consecutive couts of its pixel -> scale / shift (bias), residual, activation,
// optional BatchNorm statistics of the ROUNDED outputs (per-lane partials over ALL tiles of the workgroup, folded over the 32
// pixel lanes once at the end: the per-wave partial rows bn_stats_from_partials already consumes), 8-byte bf16x4 stores.
// Not here: split-K, phases, other channel counts (the launcher falls back to conv_bf16.hip) and BatchNorm-backward sums (such a
// launch runs here and reports "not fused": 64 more partial sums + the block's z / y rows next to everything else spilled 112
// registers, and on these memory-bound layers the stand-alone reduction costs less than the slower conv it would ride on).
#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kBoxOob = 0x80000000u;
constexpr int kBoxT = 16;                        // tile edge (pixels)
constexpr int kBoxE = kBoxT + 2;                 // box edge
constexpr int kBoxPix = kBoxE * kBoxE;           // 324
constexpr int kBoxDma = (kBoxPix * 8 + 63) / 64; // 41 one-KB requests per box
constexpr int kBoxBytes = kBoxDma * 1024;        // 41 984 (the last request's tail is zero-filled)
constexpr int kWRow = 9 * 128 + 16;              // weight row: 576 bf16 + one pad slot
constexpr int kWItems = 64 * (kWRow / 16);       // 4 672 sixteen-byte items = 73 requests exactly
constexpr int kWBytes = 64 * kWRow;

struct BoxArgs {
    const void* x;
    void* y;
    const void* res;
    const void* w;         // bf16 [64][576], K = (tap, c)
    const float* scale;    // [cout] or NULL
    const float* shift;    // [cout] or NULL
    const int* taps;       // 9 x (dy & 0xffff) | (dx << 16), |dy|, |dx| <= 1
    float* stats;          // NULL or [gridDim.x * 8][2][64]
    int N, H, W, x_cs, y_cs, res_cs, cout, act;
    int tiles_x, tiles_y, ntiles;
    // BWD variant (a data-gradient launch that completes the dy of a batch-statistics block, w2l_convb_forward_bnbwd): that block's
    // pre-BatchNorm output z, its output y (or NULL: ReLU block without residual, sign of z * bscale + bshift), its statistics
    const void* bz;
    const void* by;
    const float* bmean;
    const float* brstd;
    const float* bscale;
    const float* bshift;
    int bz_cs, by_cs, bstore_g;
    float bneg;
};

typedef __attribute__((address_space(3))) void* box_lds_t;



// BWD: `stats` receives the BatchNorm-backward column sums (sum g, sum g * zhat; g = dy * act'(block output)) of the block whose
// dy this launch writes instead of forward statistics, and with bstore_g the launch stores g.  The residual rows and the
// block's z / y rows are requested AFTER the matrix work of a tile (48 more registers across it spilled): STATS and RES are
// then compile-time true / run-time respectively.
template <bool STATS, bool RES, bool BWD = false>
__global__ __launch_bounds__(512, 1) void conv_box64_bf16_kernel(const BoxArgs a) {
    __shared__ __attribute__((aligned(16))) char Wl[kWBytes];
    __shared__ __attribute__((aligned(16))) char Box0[kBoxBytes];
    __shared__ __attribute__((aligned(16))) char Box1[kBoxBytes];

    constexpr bool want_stats = STATS;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5;
    const int n = lane & 31;

    const long long npix = (long long)a.N * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)(((npix - 1) * a.x_cs + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 64 * 576 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + 64) * 2) : 0, 0x00020000);
    const bool have_by = BWD && a.by != nullptr;
    const __amdgpu_buffer_rsrc_t rbz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(BWD ? a.bz : a.y), 0, BWD ? (int)(((npix - 1) * a.bz_cs + 64) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rby = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(have_by ? a.by : a.y), 0, have_by ? (int)(((npix - 1) * a.by_cs + 64) * 2) : 0, 0x00020000);

    // (dy * kBoxE + dx) of every tap in scalar registers (an LDS table read inside the fragment pipeline would make every step wait
    // for all reads in flight: LDS returns in order)
    int tapoff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int tv = __builtin_amdgcn_readfirstlane(a.taps[i]);
        tapoff[i] = ((int)(short)(tv & 0xffff)) * kBoxE + (tv >> 16);
    }
    // ---- the weight set, once: item q = row * 73 + c of the padded LDS rows <- global (row * 1152 + c * 16) bytes, pad slot zero
    for (int i = wave; i < kWItems / 64; i += 8) {
        const int q = i * 64 + lane;
        const int row = q / 73, c = q - row * 73;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (box_lds_t)(Wl + i * 1024), 16, (int)(c < 72 ? (unsigned)(row * 1152 + c * 16) : kBoxOob), 0, 0, 0);
    }

    // ---- this thread's box items (tile-independent): request i = wave + 8 j covers items 64 i + lane; item -> box pixel bp = item >> 3,
    // slot = item & 7 holding K chunk slot ^ (bp & 7)
    constexpr int NB = (kBoxDma + 7) / 8;     // 6 requests per wave at most
    int b_it[NB];                             // by | bx << 8 | K chunk << 16, or -1: no item
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i = wave + 8 * j;
        const int item = i * 64 + lane;
        const int bp = item >> 3, slot = item & 7;
        const int by = bp / kBoxE, bx = bp - by * kBoxE;
        b_it[j] = (i < kBoxDma && bp < kBoxPix) ? (by | (bx << 8) | ((slot ^ (bp & 7)) << 16)) : -1;
    }
    auto tile_coords = [&](int tile, int& img, int& ty0, int& tx0) {
        const int per = a.tiles_x * a.tiles_y;
        img = tile / per;
        const int r = tile - img * per;
        const int ty = r / a.tiles_x;
        ty0 = ty * kBoxT;
        tx0 = (r - ty * a.tiles_x) * kBoxT;
    };
    auto box_dma = [&](int tile, char* box) {
        int img, ty0, tx0;
        tile_coords(tile, img, ty0, tx0);
        const bool tile_ok = tile < a.ntiles;
        // byte offset of box pixel (0, 0) = image pixel (ty0 - 1, tx0 - 1); may be "negative": only used together with in-range (by, bx)
        const int base = ((img * a.H + ty0 - 1) * a.W + tx0 - 1) * a.x_cs * 2;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = wave + 8 * j;
            if (i < kBoxDma) {
                const int by = b_it[j] & 0xff, bx = (b_it[j] >> 8) & 0xff, kc = (b_it[j] >> 16) & 7;
                const bool ok = tile_ok & (b_it[j] >= 0) & ((unsigned)(ty0 - 1 + by) < (unsigned)a.H) & ((unsigned)(tx0 - 1 + bx) < (unsigned)a.W);
                const unsigned rel = (unsigned)((by * a.W + bx) * a.x_cs * 2 + kc * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (box_lds_t)(box + i * 1024), 16, (int)(ok ? (unsigned)base + rel : kBoxOob), 0, 0, 0);
            }
        }
    };

    // ---- per-lane epilogue: accumulator t2 (cout tile), register group g -> couts 32 t2 + 8 g + 4 h + {0..3}; scale / shift are
    // read per group inside the epilogue (L1 hits): 64 registers of them next to the accumulators spilled
    float st0[2][2][8], st1[2][2][8];           // [accumulator][k][j]: cout 32 t2 + 16 k + 8 h + j
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) { st0[t2][k][j] = 0.f; st1[t2][k][j] = 0.f; }

    constexpr bool has_res = RES && !BWD;        // BWD: the residual is a run-time switch, its rows are requested late
    const bool late_res = BWD && a.res != nullptr;
    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    const bool is_sigmoid = a.act == W2L_ACT_SIGMOID;

    // this lane's pixel inside a tile: rows 2 wave, 2 wave + 1
    const int py = 2 * wave + (n >> 4), px = n & 15;
    const int pbase = (py + 1) * kBoxE + (px + 1);
    const int wrow0 = n * kWRow + h * 16;

    auto compute = [&](int tile, const char* box) {
        f32x16 acc[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
        // the residual rows of this tile are requested BEFORE the matrix work (16 registers across it): behind it, every tile paid
        // their latency in front of the barrier (ablation, profiles/r04/u_*)
        int img, ty0, tx0;
        tile_coords(tile, img, ty0, tx0);
        const bool pix_ok = (ty0 + py < a.H) & (tx0 + px < a.W);          // ragged last tile row / column
        const unsigned opix = (unsigned)((img * a.H + ty0 + py) * a.W + tx0 + px);
        u32x4 rv[2][2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int k = 0; k < 2; ++k) rv[t2][k] = u32x4{0u, 0u, 0u, 0u};
        if (has_res) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    rv[t2][k] = __builtin_amdgcn_raw_buffer_load_b128(
                        rr, (int)(pix_ok ? (opix * (unsigned)a.res_cs + (unsigned)(32 * t2 + 16 * k + 8 * h)) * 2u : kBoxOob), 0, 0);
        }
        // BWD: the residual, z and y rows (twelve 16-byte rows per lane) are requested in front of the matrix work like the residual
        // rows above (requested behind it, every tile waited a memory round trip for them: the launch took twice as long)
        u32x4 zr[2][2], yr[2][2];
        auto late_rows = [&](int t2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned c0 = (unsigned)(32 * t2 + 16 * k + 8 * h);
                if (late_res)
                    rv[t2][k] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(pix_ok ? (opix * (unsigned)a.res_cs + c0) * 2u : kBoxOob), 0, 0);
                zr[t2][k] = __builtin_amdgcn_raw_buffer_load_b128(rbz, (int)(pix_ok ? (opix * (unsigned)a.bz_cs + c0) * 2u : kBoxOob), 0, 0);
                yr[t2][k] = u32x4{0u, 0u, 0u, 0u};
                if (have_by)
                    yr[t2][k] = __builtin_amdgcn_raw_buffer_load_b128(rby, (int)(pix_ok ? (opix * (unsigned)a.by_cs + c0) * 2u : kBoxOob), 0, 0);
            }
        };
        if (BWD) { late_rows(0); late_rows(1); }
        // 36 (tap, K-substep) steps as ONE software pipeline: the three fragments of step s + FD are requested before the two MFMAs
        // of step s are issued (FD + 1 register sets; the scheduling fences keep the requests where they are written - left alone
        // the compiler sinks every read to its use and each step waits a full LDS latency, 10 us per tile instead of 3)
        constexpr int FD = 2;               // fragments requested FD steps ahead, FD + 1 register sets (3 and 4: no faster, profiles/r04/w_*)
        bf16x8 fx[FD + 1], fw0[FD + 1], fw1[FD + 1];
        int pb = pbase;
        // with statistics (64 more live registers) the 36 fragment addresses are recomputed per tile (3 VALU each) instead of being
        // hoisted out of the tile loop into 36 registers - which spilled; without, hoisted is 7 % faster (profiles/r04/v_*, w_*)
        if (STATS) asm volatile("" : "+v"(pb));
        auto frag = [&](int step, int set) {
            const int tap = step >> 2, ks = step & 3;
            const int p = pb + tapoff[tap];
            const char* brow = box + p * 128;
            const char* wtap = Wl + wrow0 + tap * 128 + ks * 32;
            fx[set] = *reinterpret_cast<const bf16x8*>(brow + (((2 * ks + h) ^ (p & 7)) * 16));
            fw0[set] = *reinterpret_cast<const bf16x8*>(wtap);
            fw1[set] = *reinterpret_cast<const bf16x8*>(wtap + 32 * kWRow);
        };
#pragma unroll
        for (int i = 0; i < FD; ++i) frag(i, i);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            if (step + FD < 36) frag(step + FD, (step + FD) % (FD + 1));
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[step % (FD + 1)], fx[step % (FD + 1)], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[step % (FD + 1)], fx[step % (FD + 1)], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue.  Accumulator register 4 g + e of lane (n, h) = cout 32 t2 + 8 g + 4 h + e of pixel n.  v_permlane32_swap
        // between the register groups g = 2k (upper half wave) and g = 2k + 1 (lower half wave) leaves lane (n, 0) with couts
        // 16 k .. 16 k + 7 and lane (n, 1) with 16 k + 8 .. 16 k + 15 of its accumulator: 16-byte rows, two lanes = 32 contiguous bytes.
        // The residual rows are requested here, the wait for them (and for the next box's DMA) and the tile's barrier come next, the
        // stores leave AFTER the barrier and nobody waits for their acknowledgement before the following tile's end.
        // (inline asm with BOTH operands read-write: through __builtin_amdgcn_permlane32_swap the compiler kept only the first
        //  result of each swap here - the second operand's register was reused at once - and half of every 8-cout row came out wrong;
        //  tools/microbench/permlane32_swap.hip probes the lane map)
        float vv[2][16];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[t2][r] = acc[t2][r];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    // lower lanes keep group 2k and receive the upper lanes' group 2k; upper lanes receive the lower lanes' group 2k + 1
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(vv[t2][8 * k + e]), "+v"(vv[t2][8 * k + 4 + e]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next box landed (this wave's requests), residual rows here
        __syncthreads();                                       // ... every wave's; nobody still reads this tile's box
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bf16x8 rb = __builtin_bit_cast(bf16x8, rv[t2][k]);
                const int c0 = 32 * t2 + 16 * k + 8 * h;          // this lane's 8 consecutive couts
                f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
                if (a.scale) { sc0 = *reinterpret_cast<const f32x4*>(a.scale + c0); sc1 = *reinterpret_cast<const f32x4*>(a.scale + c0 + 4); }
                if (a.shift) { sh0 = *reinterpret_cast<const f32x4*>(a.shift + c0); sh1 = *reinterpret_cast<const f32x4*>(a.shift + c0 + 4); }
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // after the swap: values j = 0..3 sit in register group 2k, j = 4..7 in group 2k + 1 (both halves of the wave)
                    const float av = vv[t2][8 * k + j];
                    float v = av * (j < 4 ? sc0[j & 3] : sc1[j & 3]) + (j < 4 ? sh0[j & 3] : sh1[j & 3]) + (float)rb[j];
                    if (is_sigmoid) v = 1.0f / (1.0f + expf(-v));
                    else v = act_leaky(v, neg_slope);
                    o[j] = (__bf16)((c0 + j < a.cout) ? v : 0.f);
                    if (want_stats && !BWD) {
                        const float vr = pix_ok ? (float)o[j] : 0.f;
                        st0[t2][k][j] += vr;
                        st1[t2][k][j] += vr * vr;
                    }
                }
                if (BWD) {
                    // the block's vectors hold round8(cout) = 64 entries; read per group (L1 hits), as scale / shift above
                    const f32x4 mu0 = *reinterpret_cast<const f32x4*>(a.bmean + c0), mu1 = *reinterpret_cast<const f32x4*>(a.bmean + c0 + 4);
                    const f32x4 rs0 = *reinterpret_cast<const f32x4*>(a.brstd + c0), rs1 = *reinterpret_cast<const f32x4*>(a.brstd + c0 + 4);
                    f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0, bh0 = bs0, bh1 = bs0;
                    if (!have_by) {
                        bs0 = *reinterpret_cast<const f32x4*>(a.bscale + c0); bs1 = *reinterpret_cast<const f32x4*>(a.bscale + c0 + 4);
                        bh0 = *reinterpret_cast<const f32x4*>(a.bshift + c0); bh1 = *reinterpret_cast<const f32x4*>(a.bshift + c0 + 4);
                    }
                    const bf16x8 zb = __builtin_bit_cast(bf16x8, zr[t2][k]), yb = __builtin_bit_cast(bf16x8, yr[t2][k]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float vr = (float)o[j];                       // the dy that is stored
                        const float zf = (float)zb[j];
                        const float yy = have_by ? (float)yb[j] : zf * (j < 4 ? bs0[j & 3] : bs1[j & 3]) + (j < 4 ? bh0[j & 3] : bh1[j & 3]);
                        const float g = pix_ok ? vr * (yy > 0.f ? 1.f : a.bneg) : 0.f;
                        st0[t2][k][j] += g;
                        st1[t2][k][j] += g * ((zf - (j < 4 ? mu0[j & 3] : mu1[j & 3])) * (j < 4 ? rs0[j & 3] : rs1[j & 3]));
                        if (a.bstore_g) o[j] = yy > 0.f ? o[j] : (__bf16)0.f;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry,
                                                       (int)(pix_ok ? (opix * (unsigned)a.y_cs + (unsigned)c0) * 2u : kBoxOob), 0, 0);
            }
        }
    };

    // ---- persistent loop over tiles, two per iteration so that the box buffers are compile-time names (the compiler then tracks the
    // DMA destinations per array and does not put vmcnt(0) in front of every LDS read while the next box is in flight)
    int tile = blockIdx.x;
    box_dma(tile, Box0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    while (tile < a.ntiles) {
        box_dma(tile + gridDim.x, Box1);
        compute(tile, Box0);              // (ends with the wait for the next box + the tile barrier, then its stores)
        tile += gridDim.x;
        if (tile >= a.ntiles) break;
        box_dma(tile + gridDim.x, Box0);
        compute(tile, Box1);
        tile += gridDim.x;
    }

    if (STATS) {
        // fold the 32 pixel lanes of each half wave (xor 1..16), then lanes 0 and 32 hold the wave's sums of their 32 couts
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int m = 1; m < 32; m <<= 1) {
                        st0[t2][k][j] += __shfl_xor(st0[t2][k][j], m);
                        st1[t2][k][j] += __shfl_xor(st1[t2][k][j], m);
                    }
        if (n == 0) {
            float* dst = a.stats + (long long)(blockIdx.x * 8 + wave) * 2 * 64;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c = 32 * t2 + 16 * k + 8 * h;
                    *reinterpret_cast<f32x4*>(dst + c) = f32x4{st0[t2][k][0], st0[t2][k][1], st0[t2][k][2], st0[t2][k][3]};
                    *reinterpret_cast<f32x4*>(dst + c + 4) = f32x4{st0[t2][k][4], st0[t2][k][5], st0[t2][k][6], st0[t2][k][7]};
                    *reinterpret_cast<f32x4*>(dst + 64 + c) = f32x4{st1[t2][k][0], st1[t2][k][1], st1[t2][k][2], st1[t2][k][3]};
                    *reinterpret_cast<f32x4*>(dst + 64 + c + 4) = f32x4{st1[t2][k][4], st1[t2][k][5], st1[t2][k][6], st1[t2][k][7]};
                }
        }
    }
}

// ---- host side (called from conv_bf16.hip's launcher)
struct BoxBwd {             // the block whose dy a BWD launch completes
    const void* z;
    const void* y;
    int z_cs, y_cs, store_g;
    float neg;
    const float* mean;
    const float* rstd;
    const float* scale;
    const float* shift;
};

// A shape-only rule (bit-reproducible): the layer must be LARGE - every workgroup fetches the 74 KB weight set once, which 8 tiles
// amortise and 3 do not (64 @24x24 x 320 frames and 64 @46x47 x 64 pairs were measured slower here than on the implicit GEMM,
// profiles/r04/z_*) - and its extents must fill their 16x16 tiles to 85 % (24x24 fills 56 %)
bool box64_ok(int nphase, int ntaps, int cin_p, int cout, int cout_p, int N, int H, int W, int Ho, int Wo, int sy, int sx) {
    // the epilogue stores all 64 channels of a pixel and reads scale / shift / residual [0, 64) unguarded: only layers whose
    // 8-padded cout IS 64 (cout 57..64; the pad channels are zero by the layout contract) - a 33..56-cout layer has cout_p == 64
    // too and would write over the neighbouring channels of a narrower or concat buffer
    if (!(nphase == 1 && ntaps == 9 && cin_p == 64 && cout_p == 64 && round_up(cout, 8) == 64 && Ho == H && Wo == W && sy == 1 && sx == 1))
        return false;
    const long long ty = (H + kBoxT - 1) / kBoxT, tx = (W + kBoxT - 1) / kBoxT;
    return (long long)N * ty * tx >= 2048 && (long long)H * W * 100 >= 85ll * ty * tx * kBoxT * kBoxT;
}

int box64_grid(int N, int H, int W) {
    const long long tiles = (long long)N * ((H + kBoxT - 1) / kBoxT) * ((W + kBoxT - 1) / kBoxT);
    return (int)(tiles < 256 ? tiles : 256);
}

int box64_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* res, int res_cs, const void* w,
                 const float* scale, const float* shift, const int* taps, float* stats, int N, int H, int W, int cout, int act,
                 const BoxBwd* bwd) {
    BoxArgs a;
    a.bz = nullptr; a.by = nullptr; a.bmean = nullptr; a.brstd = nullptr; a.bscale = nullptr; a.bshift = nullptr;
    a.bz_cs = 0; a.by_cs = 0; a.bstore_g = 0; a.bneg = 1.f;
    if (bwd) {
        a.bz = bwd->z; a.by = bwd->y; a.bz_cs = bwd->z_cs; a.by_cs = bwd->y_cs; a.bmean = bwd->mean; a.brstd = bwd->rstd;
        a.bscale = bwd->scale; a.bshift = bwd->shift; a.bstore_g = bwd->store_g; a.bneg = bwd->neg;
    }
    a.x = x; a.y = y; a.res = res; a.w = w; a.scale = scale; a.shift = shift; a.taps = taps; a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.x_cs = x_cs; a.y_cs = y_cs; a.res_cs = res_cs; a.cout = cout; a.act = act;
    a.tiles_x = (W + kBoxT - 1) / kBoxT; a.tiles_y = (H + kBoxT - 1) / kBoxT;
    const long long tiles = (long long)N * a.tiles_x * a.tiles_y;
    W2L_REQUIRE(tiles < (1ll << 30), "grid too large");
    a.ntiles = (int)tiles;
    // statistics and residual are template switches: the statistics' 64 per-lane partial sums and the residual's 16 registers
    // across the matrix work do not fit next to each other without spilling, and no layer of the path asks for both
    const dim3 grid((unsigned)box64_grid(N, H, W)), block(512);
    if (bwd) hipLaunchKernelGGL((conv_box64_bf16_kernel<true, true, true>), grid, block, 0, stream, a);
    else if (stats && res) hipLaunchKernelGGL((conv_box64_bf16_kernel<true, true>), grid, block, 0, stream, a);
    else if (stats) hipLaunchKernelGGL((conv_box64_bf16_kernel<true, false>), grid, block, 0, stream, a);
    else if (res) hipLaunchKernelGGL((conv_box64_bf16_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_box64_bf16_kernel<false, false>), grid, block, 0, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

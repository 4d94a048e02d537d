// Winograd F(2x2, 3x3) convolution for gfx950 with an fp32 RESULT from the bf16 matrix cores ("split-operand" arithmetic, the
// same as conv_igemm_bf16_kernel<.., 3>): every transformed input value and every transformed weight is the exact sum of three
// bf16 pieces, a K-chunk of 16 channels is the six piece products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 (smallest first,
// fp32 accumulation), i.e. 6/16 of the matrix-pipe time of the eight fp32 MFMAs it replaces.  For the 3x3 / stride 1 / pad 1
// layers (models/wav2lip.py:61-81 via models/conv.py:5-19):
//   y = act( A^T [ sum_c (G g G^T)[xi] * (B^T d B)[xi] ] A * scale + shift (+ res) ),   xi = (i, j) in 4 x 4
// F(2x2) and not F(4x4) (conv_wino4.hip) because of what the bf16 pipe's speed leaves as the bound: a 36-position workgroup holds
// 295 KB of accumulators, which caps it at 32 tiles x 64 couts, and at 32 tiles per weight fetch the pre-split weights need
// 64 B/clk per CU from L2 at the matrix rate - all the vector memory path has; the 16-position workgroup below covers 64 tiles x
// 64 couts (half the weight bytes per MFMA), and its whole weight stream for a K-step fits the registers of the waves that
// consume it, two slots ahead.
//
// Workgroup = 8 waves = 64 tiles (2x2 outputs each) x 64 couts x 16 positions, one per CU, persistent over work items.
//   wave (g, wn, il):  row i = 2 g + il of the 4 x 4 position grid (4 positions), cout half wn, ALL 64 tiles (two 32-tile
//   MFMA row blocks share every weight fragment): 8 accumulators = 128 registers.
// The two row halves g = 0 / 1 run half a K-step apart ("slots", one workgroup barrier each):
//   slot 2s     g0: B^T d B rows 0-1 of chunk s -> V (three bf16 planes, LDS)      g1: MFMAs of chunk s-1
//   slot 2s+1   g0: MFMAs of chunk s (48 per wave)                                 g1: rows 2-3 of chunk s -> V
// so on every SIMD one wave feeds the matrix pipe while the other does the vector work (transform + 3-piece split, ~120 VALU per
// task); a row half of V is written and read by its own four waves in alternate slots and needs ONE buffer (96 KB for both).
// The raw input block of a chunk ((2bh+2) x (2bw+2) pixels per image for a bh x bw x ni tile block, 16 channels) goes through
// registers into one of two 30 KB buffers, laid out in (channel half, x parity) planes so that the transform's ds_read_b128
// groups are conflict-free (the host picks row pitch / image stride per block shape): every wave requests four 16-byte slots at
// the START of its MFMA slot - the only phase with registers to spare - and stores them at its end; g1's half of a chunk lands
// two slots before the chunk's first reader, g0's half one slot before.  (LDS-DMA was built and measured first: a request with 64
// scattered 16-byte pieces holds the issuing wave for ~0.15 us, 1.3 us per chunk, wherever it is placed - profiles/r05.)
// Weights: U = G g G^T in fp64, rounded to fp32, split into three bf16 planes, stored as MFMA B fragments
//   u[(((nb * nkc + kc) * 16 + pos) * 3 + plane) * 512 + lane * 8 + e] = piece_plane( U_pos[nb*32 + (lane&31)][kc*16 + 8*(lane>>5) + e] )
// a wave's 12 fragments of chunk s+1 are requested right after the same registers fed chunk s (a full K-step of latency cover).
// Epilogue: every wave applies A^T along j in registers, the four row waves meet in an LDS staging tile (two rounds of 32 tiles)
// and the float4 output pass applies A^T along i, scale / shift / residual / activation.
#include <type_traits>

#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef W2S_DBG
#define W2S_DBG 0      // timing ablations (variant builds only, tools/build_variant.sh): 1 no MFMA, 2 no transform, 4 no raw DMA, 8 no weight loads, 16 no epilogue, 32 raw loads out of range (issue only)
#endif

#ifdef W2S_TRACE
// phase timestamps (variant builds only; tools/wino2s_trace.py): [workgroup][wave 0 / wave 4][stamp < 48] s_memtime values of the
// SECOND work item of a workgroup (steady state), s_memrealtime (100 MHz) at stamps 0 and 47 to calibrate the shader clock
__device__ unsigned long long w2s_trace_buf[256 * 2 * 48];
__device__ unsigned long long w2s_trace_rt[256 * 2 * 2];
#define W2S_STAMP(k)                                                                                                 \
    do {                                                                                                             \
        if (trace_on && (k) < 48) {                                                                                  \
            w2s_trace_buf[(blockIdx.x * 2 + g) * 48 + (k)] = __builtin_readcyclecounter();                           \
            if ((k) == 0) w2s_trace_rt[(blockIdx.x * 2 + g) * 2] = __builtin_amdgcn_s_memrealtime();                  \
        }                                                                                                            \
    } while (0)
#define W2S_STAMP_END()                                                                                              \
    do {                                                                                                             \
        if (trace_on) w2s_trace_rt[(blockIdx.x * 2 + g) * 2 + 1] = __builtin_amdgcn_s_memrealtime();                  \
    } while (0)
#else
#define W2S_STAMP(k)
#define W2S_STAMP_END()
#endif

constexpr unsigned kSOob = 0x80000000u;
constexpr int kS_BT = 64;                      // 2x2 output tiles per workgroup
constexpr int kS_BC = 64;                      // couts per workgroup
constexpr int kS_KS = 16;                      // channels per K-step (the bf16 MFMA's K)
constexpr int kS_VPOS = 2 * kS_BT * 16;        // bytes of one (plane, position): [channel half][tile][8 bf16]
constexpr int kS_VPLANE = 16 * kS_VPOS;
constexpr int kS_VBYTES = 3 * kS_VPLANE;       // 98 304
constexpr int kS_CELLS = 240;                  // raw plane: cells (= 2 channel quads = 32 B) per (channel half, x parity) plane
constexpr int kS_RAWSLOTS = 4 * kS_CELLS * 2;  // 16-byte slots per raw buffer (1 920)
constexpr int kS_RAWBYTES = kS_RAWSLOTS * 16;  // 30 720
constexpr int kS_NRAW = (kS_RAWSLOTS + 511) / 512;   // raw-block slots per thread and chunk (4)
constexpr int kS_LDY = kS_BC + 4;              // staging row stride (floats)
static_assert(4 * 32 * 2 * kS_LDY * 4 <= kS_VBYTES, "one round of the four row-partial staging tiles must fit in V");
static_assert(kS_VBYTES + 2 * kS_RAWBYTES + 2 * kS_BT * 4 <= 160 * 1024, "LDS budget");

struct Wino2sKArgs {
    const float* x;
    float* y;
    const float* res;
    const __bf16* u;     // pre-split transformed weights in fragment order (wino2s_pack)
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;
    int cout, y_cs, res_cs;
    int TH, TW;          // 2x2 output tiles per image
    int bh, bw, ni;      // tile block of a workgroup
    int nby, nbx, ngi;
    int RH, RW;          // raw region per image (2bh+2, 2bw+2)
    int pitch, istride;  // raw planes: cells per region row (>= bw+1) and per image (>= RH*pitch)
    float inv_pitch, inv_istride;
    int nkc;             // cin / 16
    int tiles_n;         // cout / 64
    long long total;
    int act;
};

// x = h + m + l exactly (each piece RNE to bf16 of what the pieces before it left), two values per call
__device__ __forceinline__ unsigned s_pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void s_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = s_pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = s_pack_bf16x2(r0, r1);
    l = s_pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

// Workgroup barrier of the slot loop: this wave's LDS traffic done, then s_barrier - WITHOUT the release fence of __syncthreads(),
// which makes the compiler complete every LDS-DMA request in flight (s_waitcnt vmcnt) at every barrier: the raw-block requests
// are issued two slots before their first reader and must stay in flight across the barrier in between.  The one barrier that
// publishes them is preceded by an explicit s_waitcnt vmcnt(0) of the waves that issued them.
__device__ __forceinline__ void slot_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 1) void conv_wino2s_kernel(const Wino2sKArgs a) {
    __shared__ __attribute__((aligned(16))) char Vs[kS_VBYTES];      // [plane 3][pos 16][channel half 2][tile 64][8 bf16]
    __shared__ __attribute__((aligned(16))) char Raw0[kS_RAWBYTES];  // chunks 0, 2, 4, ...
    __shared__ __attribute__((aligned(16))) char Raw1[kS_RAWBYTES];  // chunks 1, 3, 5, ...
    __shared__ int s_opix[kS_BT];    // output pixel (2ty, 2tx) of a tile or -1
    __shared__ int s_oflag[kS_BT];   // bit0: column 2tx+1 exists, bit1: row 2ty+1 exists

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);

    const int t0 = threadIdx.x;
    const int lane0 = t0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6);
    const int g = wave >> 2;            // row half: rows 2g, 2g+1; slot parity
    const int wn = (wave >> 1) & 1;     // cout half (MFMA phase) / tile half (transform phase)
    const int row = 2 * g + (wave & 1); // position row i of this wave (both phases)
    const int bhw = a.bh * a.bw;

    // ---- transform task of this wave: row `row` of B^T d B for tile (wn * 32 + (lane >> 1)), channel quad q = lane & 1 of each
    // channel half:  row i of B^T d:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3   ==  d[ra] + sg * d[rb]
    const int ra = (row == 0) ? 0 : (row == 2 ? 2 : 1);
    const int rb = (row == 0) ? 2 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
    const float sg = (row == 1) ? 1.0f : -1.0f;
    // Registers that live across the whole kernel are scarce (128 accumulators + 48 for the weight fragments in flight): only
    // tf_a (two divisions) and the packed request descriptors stay resident; every other per-lane address is re-derived from an
    // opaque copy of the lane id at the start of the phase that uses it (the compiler would otherwise hoist it out of the loops)
    int tf_a;                            // byte offset (plane 0, column 0) of this thread's first raw row
    {
        const int tl = wn * 32 + (lane0 >> 1);
        const int il = tl / bhw, r = tl - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int ilc = il < a.ni ? il : 0;      // unused tile slots read image 0's region: finite, never stored
        const int cell0 = ilc * a.istride + 2 * tyl * a.pitch + txl;
        tf_a = (cell0 + ra * a.pitch) * 32 + (lane0 & 1) * 16;
    }
    const int tf_ba = (rb - ra) * a.pitch * 32;      // second raw row relative to the first (wave-uniform)
    // ---- A fragments: lane reads channel half (lane >> 5) of tile mb * 32 + (lane & 31)
    const unsigned bl_row = (unsigned)(4 * row) * 3072u;

    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    const int nsteps = a.nkc;
#ifdef W2S_TRACE
    int trace_item = -1;
#endif
    for (unsigned jw = blockIdx.x >> 3; jw < per; jw += gw) {
        const unsigned bid = xcd * per + jw;
        if (bid >= total) break;
#ifdef W2S_TRACE
        ++trace_item;
#ifdef W2S_TRACE_OFF
        const bool trace_on = false;
#else
        const bool trace_on = trace_item == 1 && (threadIdx.x & 255) == 0;     // lane 0 of waves 0 (g = 0) and 4 (g = 1)
#endif
        int ts = 0;
#endif
        W2S_STAMP(0);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));      // per-item values are re-derived from an opaque copy: nothing per-lane stays live across items
        const int lane = t & 63;
        const int tile_n = (int)(bid % (unsigned)a.tiles_n);
        unsigned mbk = bid / (unsigned)a.tiles_n;
        const int bx_i = (int)(mbk % (unsigned)a.nbx);
        mbk /= (unsigned)a.nbx;
        const int by_i = (int)(mbk % (unsigned)a.nby);
        const int gi = (int)(mbk / (unsigned)a.nby);
        const int n0 = tile_n * kS_BC;

        if (t < kS_BT) {                 // tile table of the epilogue
            const int il = t / bhw, r = t - il * bhw;
            const int tyl = r / a.bw, txl = r - tyl * a.bw;
            const int n = gi * a.ni + il, ty = by_i * a.bh + tyl, tx = bx_i * a.bw + txl;
            int o = -1, f = 0;
            if (il < a.ni && n < a.N && ty < a.TH && tx < a.TW) {
                o = (n * a.H + 2 * ty) * a.W + 2 * tx;
                f = ((2 * tx + 1 < a.W) ? 1 : 0) | ((2 * ty + 1 < a.H) ? 2 : 0);
            }
            s_opix[t] = o;
            s_oflag[t] = f;
        }

        // ---- raw block: slot e of a chunk's buffer, e = (P * kS_CELLS + cell) * 2 + qq, plane P = (channel half, x parity),
        // cell = il * istride + ry * pitch + (rx >> 1).  Thread -> slots e = half * 1024 + 256 k + 64 (wave & 3) + lane, k = 0..3, with
        // half = 1 - g (g1 carries slots [0, 1024), g0 the rest).  The byte offsets of a thread's four slots do not depend on the chunk
        // (the chunk is the scalar offset of the load): decoded once per item; a pad slot reads out of range = zero.
        unsigned goff[kS_NRAW];
#pragma unroll
        for (int k = 0; k < kS_NRAW; ++k) {
            const int e = (1 - g) * 1024 + 256 * k + 64 * (wave & 3) + lane;
            unsigned off = kSOob;
            if (e < kS_RAWSLOTS) {
                // exact small-integer divisions through reciprocals (half-integer numerators, e < 1920, cell < 240)
                const int P = (int)(((float)e + 0.5f) * (1.0f / (2 * kS_CELLS)));
                const int rem = e - P * (2 * kS_CELLS);
                const int cell = rem >> 1, qq = rem & 1;
                const int il = (int)(((float)cell + 0.5f) * a.inv_istride);
                const int r2 = cell - il * a.istride;
                const int ry = (int)(((float)r2 + 0.5f) * a.inv_pitch);
                const int rxx = 2 * (r2 - ry * a.pitch) + (P & 1);
                const int n = gi * a.ni + il;
                const int iy = 2 * by_i * a.bh - 1 + ry, ix = 2 * bx_i * a.bw - 1 + rxx;
                if (il < a.ni && ry < a.RH && rxx < a.RW && n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(((P >> 1) * 2 + qq) * 4)) * 4u;
            }
            goff[k] = off;
        }
        f32x4 rawreg[kS_NRAW];
        auto raw_gload = [&](int chunk) {            // chunks past the end read zero
#if !(W2S_DBG & 4)
            const unsigned soff = (unsigned)(chunk * kS_KS * 4);
#pragma unroll
            for (int k = 0; k < kS_NRAW; ++k)
                rawreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rx, (int)((chunk < nsteps && !(W2S_DBG & 32)) ? goff[k] : kSOob), (int)soff, 0));
#endif
        };
        auto raw_store = [&](char* raw) {
#if !(W2S_DBG & 4)
            char* dst = raw + ((1 - g) * 1024 + 64 * (wave & 3) + lane) * 16;
#pragma unroll
            for (int k = 0; k < kS_NRAW; ++k)
                if ((1 - g) * 1024 + 256 * k + 64 * (wave & 3) < kS_RAWSLOTS)      // wave-uniform
                    *reinterpret_cast<f32x4*>(dst + 256 * k * 16) = rawreg[k];
#endif
        };

        // B^T d B of one chunk (this wave's row, 32 tiles, both channel halves) -> V
        auto transform = [&](const char* raw) {
#if W2S_DBG & 2
            return;
#endif
            int ln = threadIdx.x & 63;
            asm volatile("" : "+v"(ln));
            // V write address of (plane 0, position (row, 0), channel half 0): + plane * kS_VPLANE + j * kS_VPOS + kh * 1024
            char* const vwr = Vs + (4 * row) * kS_VPOS + (wn * 32 + (ln >> 1)) * 16 + (ln & 1) * 8;
            const int tf_b = tf_a + tf_ba;
            // The wave's time in this phase is its LDS round trips (a read group answers in ~200 cycles beside the partner's fragment
            // reads), not its ~200 vector instructions: all eight reads of a channel half are requested at once, and the next half's
            // while this half's four positions are split and stored (the first version waited four times per slot for four reads
            // each: 1.65 us per slot against 1.15 for the partner's 48 MFMAs, profiles/r05)
            f32x4 va[4], vb[4];
            auto rd = [&](int kh) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int po = ((2 * kh + (c & 1)) * kS_CELLS + (c >> 1)) * 32;      // plane of the column, next cell for c = 2, 3
                    va[c] = *reinterpret_cast<const f32x4*>(raw + tf_a + po);
                    vb[c] = *reinterpret_cast<const f32x4*>(raw + tf_b + po);
                }
            };
            rd(0);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                f32x4 da[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) da[c][e] = fmaf(sg, vb[c][e], va[c][e]);
                __builtin_amdgcn_sched_barrier(0);
                if (kh == 0) rd(1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 v;
                    switch (j) {
                        case 0: v = da[0] - da[2]; break;
                        case 1: v = da[1] + da[2]; break;
                        case 2: v = da[2] - da[1]; break;
                        default: v = da[1] - da[3]; break;
                    }
                    unsigned h0, m0, l0, h1, m1, l1;
                    s_split3_pair(v[0], v[1], h0, m0, l0);
                    s_split3_pair(v[2], v[3], h1, m1, l1);
                    char* dst = vwr + j * kS_VPOS + kh * 1024;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(dst + kS_VPLANE) = u32x2{m0, m1};
                    *reinterpret_cast<u32x2*>(dst + 2 * kS_VPLANE) = u32x2{l0, l1};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        // ---- B operand: this wave's 12 fragments per chunk (4 positions of its row x 3 planes)
        const int nb = (n0 >> 5) + wn;
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<__bf16*>(a.u + (long long)nb * a.nkc * (16 * 3 * 512)), 0, a.nkc * (16 * 3 * 1024), 0x00020000);
        auto bload = [&](int kc, int j, int plane, int ln) {       // past-the-end chunks read zero (never used)
            const unsigned soff = (unsigned)kc * 49152u + bl_row + (unsigned)(j * 3 + plane) * 1024u;
            return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ru, ln * 16, (int)soff, 0));
        };
        bf16x8 bq[4][3];
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[j][p] = bload(0, j, p, ln);
        }

        f32x16 acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mb][r] = 0.f;

        // the six piece products of a K-chunk, smallest first: (a2 b0) (a1 b1) (a0 b2) | (a1 b0) (a0 b1) | (a0 b0), each on both tile
        // blocks.  ONE set of A fragments: a piece's registers take the next position's piece as soon as its last product is issued
        // (a2 after product 1, a1 after 4, a0 after 6: every reload has at least four MFMAs = ~130 cycles to land)
        auto mfma_slot = [&](int step, int rchunk, char* rbuf) {
            raw_gload(rchunk);               // this wave's four slots of chunk `rchunk`: in flight under the slot's matrix work
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const char* const ard = Vs + (4 * row) * kS_VPOS + (ln >> 5) * 1024 + (ln & 31) * 16;
            auto aload = [&](int j, int plane, int mb) {
                return *reinterpret_cast<const bf16x8*>(ard + plane * kS_VPLANE + j * kS_VPOS + mb * 512);
            };
            bf16x8 af[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) af[p][mb] = aload(0, p, mb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                constexpr int kPa[6] = {2, 1, 0, 1, 0, 0};
                constexpr int kPb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
#if W2S_DBG & 1
                        asm volatile("" : "+v"(acc[j][mb]) : "v"(af[kPa[u]][mb]), "v"(bq[j][kPb[u]]));
#else
                        acc[j][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kPa[u]][mb], bq[j][kPb[u]], acc[j][mb], 0, 0, 0);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 3) {
                        const int dead = u == 0 ? 2 : (u == 3 ? 1 : (u == 5 ? 0 : -1));
                        if (dead >= 0) {
#pragma unroll
                            for (int mb = 0; mb < 2; ++mb) af[dead][mb] = aload(j + 1, dead, mb);
                        }
                    }
                }
                // these registers fed chunk `step`: ask for chunk step + 1 now, a full K-step ahead of its use
#if !(W2S_DBG & 8)
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[j][p] = bload(step + 1, j, p, ln);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            raw_store(rbuf);
        };

        // ---- prologue: chunks 0 and 1 of the input block (every wave its own slots)
        {
            // both chunks requested before either is awaited (one memory latency per item, not two)
            f32x4 first[kS_NRAW];
            raw_gload(0);
#pragma unroll
            for (int k = 0; k < kS_NRAW; ++k) first[k] = rawreg[k];
            raw_gload(1);
            f32x4 second[kS_NRAW];
#pragma unroll
            for (int k = 0; k < kS_NRAW; ++k) { second[k] = rawreg[k]; rawreg[k] = first[k]; }
            raw_store(Raw0);
#pragma unroll
            for (int k = 0; k < kS_NRAW; ++k) rawreg[k] = second[k];
            raw_store(Raw1);
        }
        __syncthreads();                 // raw chunks 0 / 1, tile table
        W2S_STAMP(1);
#ifdef W2S_TRACE
        ts = 2;
#endif

        // ---- slots.  Both row halves run the SAME straight-line loop body (transform, barrier, MFMAs, barrier - no branch around
        // the accumulators, which a compiler turns into 128-register copies at the merge); g = 1 enters it one barrier late and g = 0
        // leaves it one barrier late, so g1's transform of chunk s runs beside g0's MFMAs of chunk s and g1's MFMAs of chunk s
        // beside g0's transform of chunk s + 1.  An odd chunk count runs one more chunk of zeros (raw_gload() and the weight
        // descriptor both return zeros past the end).
        // Raw buffer of chunk X (Raw[X & 1]): read by g0 in its transform slot (time 2X) and by g1 in the next (2X + 1); free for
        // chunk X + 2 from time 2X + 2.  g1's MFMA slot of chunk s is time 2s + 2: it carries its half of chunk s + 2 (stored at the
        // end of that slot, two slots before the first reader); g0's MFMA slot of chunk s is time 2s + 1: it carries its half of
        // chunk s + 1 (stored at the end of time 2s + 1, one slot before the reader).  (Chunk 1 is complete from the prologue: g0's
        // first MFMA slot stores its half of it once more.)
        if (g == 1) slot_barrier();
#ifdef W2S_TRACE
#define W2S_T() do { W2S_STAMP(ts); ++ts; } while (0)
#else
#define W2S_T()
#endif
        for (int step = 0; step < nsteps; step += 2) {
            W2S_T();                     // per chunk: +0 -, +1 transform done, +2 barrier, +3 MFMAs issued + raw stored, +4 -, +5 barrier
            transform(Raw0);
            W2S_T();
            slot_barrier();
            W2S_T();
            mfma_slot(step, step + 1 + g, g ? Raw0 : Raw1);
            W2S_T();
            W2S_T();
            slot_barrier();
            W2S_T();
            W2S_T();
            transform(Raw1);
            W2S_T();
            slot_barrier();
            W2S_T();
            mfma_slot(step + 1, step + 2 + g, g ? Raw1 : Raw0);
            W2S_T();
            W2S_T();
            slot_barrier();
            W2S_T();
        }
        if (g == 0) slot_barrier();
        __syncthreads();                 // (a fenced barrier before the staging tile overwrites V)
        W2S_STAMP(46);

        // ---- epilogue.  acc[j][mb][r] = M[row][j] for cout n0 + wn*32 + (lane & 31), tile mb*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)
        //   inner[b] = sum_j M[row][j] * AT[b][j]:   b = 0: m0 + m1 + m2     b = 1: m1 - m2 - m3
        //   out(a, b) = sum_i AT[a][i] * inner_i[b]:  a = 0: rows 0 + 1 + 2   a = 1: rows 1 - 2 - 3
        // Two rounds (mb = 0, 1): every wave stages inner[0 / 1] of its 32 tiles x 32 couts at S[row][tile][b][LDY]; the float4 pass
        // combines the four rows, applies scale / shift / residual / activation and stores.
        float* Ss = reinterpret_cast<float*>(Vs);
        constexpr int kRowPart = 32 * 2 * kS_LDY;        // floats per row partial of a round
        const long long npix = (long long)a.N * a.H * a.W;
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + a.cout) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0, 0x00020000);
        constexpr int CG = kS_BC / 4;                    // float4 column groups per pixel
        constexpr int NIT = 32 * 4 * CG / 512;           // 4 float4 per thread and round
        const int c4 = t % CG;
        const int ch = n0 + c4 * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
        const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
        auto out_pix = [&](int round, int i) {           // output pixel of this thread's i-th float4 of the round, or -1
            const int id = i * 512 + t;
            const int pxl = id / CG;                     // tile32 * 4 + a * 2 + b
            const int tile = 32 * round + (pxl >> 2), oa = (pxl >> 1) & 1, ob = pxl & 1;
            const int opix = s_opix[tile];
            const int fl = s_oflag[tile];
            const bool ok = (opix >= 0) & ((oa == 0) | ((fl & 2) != 0)) & ((ob == 0) | ((fl & 1) != 0));
            return ok ? opix + oa * a.W + ob : -1;
        };
        f32x4 rv[NIT];
        auto res_load = [&](int round) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int pix = out_pix(round, i);
                rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rres, (int)(pix >= 0 ? ((unsigned)pix * (unsigned)a.res_cs + (unsigned)ch) * 4u : kSOob), 0, 0));
            }
        };
#if W2S_DBG & 16
        if (a.N > 0) { __syncthreads(); continue; }
#endif
        res_load(0);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            {
                float* srow = Ss + row * kRowPart + wn * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m0 = acc[0][round][r], m1 = acc[1][round][r], m2 = acc[2][round][r], m3 = acc[3][round][r];
                    const int tl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    srow[(tl * 2 + 0) * kS_LDY] = (m0 + m1) + m2;
                    srow[(tl * 2 + 1) * kS_LDY] = (m1 - m2) - m3;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int id = i * 512 + t;
                const int pxl = id / CG;
                const int oa = (pxl >> 1) & 1;
                const int pix = out_pix(round, i);
                // staging row of (tile32, b): rows i0, i0+1, i0+2 with i0 = oa
                const float* src = Ss + ((pxl >> 2) * 2 + (pxl & 1)) * kS_LDY + c4 * 4 + oa * kRowPart;
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(src);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(src + kRowPart);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(src + 2 * kRowPart);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = oa == 0 ? (p0[e] + p1[e]) + p2[e] : (p0[e] - p1[e]) - p2[e];
                    const float xv = fmaf(m, sc[e], sh[e]) + rv[i][e];
                    v[e] = fmaf(neg_slope, fminf(xv, 0.f), fmaxf(xv, 0.f));
                }
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, v), ry, (int)(pix >= 0 ? ((unsigned)pix * (unsigned)a.y_cs + (unsigned)ch) * 4u : kSOob), 0, 0);
            }
            if (round == 0) res_load(1);
            __syncthreads();             // staging tile (and, after round 1, V / the tile table) free for the next writer
        }
        W2S_STAMP(47);
        W2S_STAMP_END();
    }   // persistent loop
}

// ---- weight planes: the fp32 transformed weights U = G g G^T the fp32 F(2x2) kernels use (wino_pack, conv_wino.hip: fp64 transform
// rounded once; u32[((nb * nks + kc8) * 16 + pos) * 256 + (h * 32 + n) * 4 + e] = U_pos[nb*32 + n][kc8*8 + 4h + e]) split into three
// bf16 pieces and re-ordered into the bf16 MFMA's B fragments (K = 16 channels, 8 per lane)
struct Wino2sPackArgs {
    const float* u32;
    __bf16* u;        // [cout/32][cin/16][16 pos][3 planes][64 lanes][8]
    int cin, cout;
};

__global__ void wino2s_pack_kernel(const Wino2sPackArgs a) {
    const long long total = (long long)a.cout * a.cin * 16;
    const int nkc = a.cin / 16, nks = a.cin / 8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        // idx = (((nb * nkc + kc) * 16 + pos) * 64 + lane) * 8 + e: consecutive threads write consecutive bf16 of one fragment
        const int e = (int)(idx & 7), ln = (int)((idx >> 3) & 63), pos = (int)((idx >> 9) & 15);
        const int blk = (int)(idx >> 13);        // (nb, kc)
        const int kc = blk % nkc, nb = blk / nkc;
        const int ci = kc * 16 + 8 * (ln >> 5) + e;
        const float v = a.u32[((long long)(nb * nks + (ci >> 3)) * 16 + pos) * 256 + (((ci >> 2) & 1) * 32 + (ln & 31)) * 4 + (ci & 3)];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        __bf16* d = a.u + ((long long)blk * 16 + pos) * (3 * 512) + ln * 8 + e;
        d[0] = h;
        d[512] = m;
        d[1024] = (__bf16)(r1 - (float)m);
    }
}

// ---- host side --------------------------------------------------------------------------------
struct W2sBlock { int bh, bw, ni; };
// candidate tile blocks: bh*bw*ni <= 64 tiles, ni * (2bh+2) * (bw+1) <= kS_CELLS plane cells
static const W2sBlock kW2sBlocks[] = {{8, 8, 1}, {4, 8, 2}, {8, 4, 2}, {4, 4, 4}, {2, 8, 4}, {8, 2, 4}, {4, 8, 1}, {8, 4, 1},
                                      {2, 4, 8}, {4, 2, 8}, {4, 4, 2}, {3, 3, 7}, {2, 2, 12}, {2, 2, 8}, {3, 3, 3}, {1, 4, 12},
                                      {4, 1, 8}, {1, 2, 20}, {2, 1, 13}, {1, 1, 30}, {6, 6, 1}, {3, 6, 3}, {6, 3, 3}};

static bool wino2s_block_fits(const W2sBlock& b) {
    return b.bh * b.bw * b.ni <= kS_BT && b.ni * (2 * b.bh + 2) * (b.bw + 1) <= kS_CELLS;
}

// Raw-plane geometry of a block: the row pitch and image stride (cells of 32 bytes) under which the 16 lanes of every ds_read_b128
// lane group of a transform read (lane = (tile, quad); the hardware's groups are {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32) land on as many distinct 16-byte bank slots (slot index mod 16) as possible.  Exposed for tools/lds_conflicts.py.
void wino2s_plane_geom(int bh, int bw, int ni, int* pitch, int* istride) {
    static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int RH = 2 * bh + 2, bhw = bh * bw;
    int best = 1 << 30;
    *pitch = bw + 1;
    *istride = RH * (bw + 1);
    for (int p = bw + 1; p <= bw + 8; ++p)
        for (int is = RH * p; is <= RH * p + 15; ++is) {
            if (ni * is > kS_CELLS) break;
            int cost = (p - bw - 1) + (is - RH * p);
            for (int half = 0; half < 2; ++half)          // the two tile halves (transform waves wn = 0 / 1)
                for (int gq = 0; gq < 4; ++gq) {
                    int cnt[16] = {0};
                    for (int k = 0; k < 16; ++k) {
                        const int lane = kGroup[gq & 1][k] + 32 * (gq >> 1);
                        const int tl = half * 32 + (lane >> 1), q = lane & 1;
                        const int il = tl / bhw, r = tl % bhw;
                        const int ilc = il < ni ? il : 0;
                        ++cnt[((ilc * is + 2 * (r / bw) * p + r % bw) * 2 + q) & 15];
                    }
                    for (int k = 0; k < 16; ++k) cost += cnt[k] > 1 ? (cnt[k] - 1) * 64 * cnt[k] : 0;
                }
            if (cost < best) { best = cost; *pitch = p; *istride = is; }
        }
}

static W2sBlock wino2s_pick_block(int N, int TH, int TW) {
    W2sBlock best = {1, 1, 1};
    double best_cost = 1e300;
    for (const W2sBlock& b : kW2sBlocks) {
        if (!wino2s_block_fits(b)) continue;
        const double items = (double)ceil_div(TH, b.bh) * ceil_div(TW, b.bw) * ceil_div(N, b.ni);
        const double halo = (double)(2 * b.bh + 2) * (2 * b.bw + 2) / (4.0 * b.bh * b.bw);
        const double cost = items * (1.0 + 0.05 * halo);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}

bool wino2s_ok(int cin, int cout) { return cin % kS_KS == 0 && cout % kS_BC == 0; }

long long wino2s_u_elems(int cin, int cout) { return (long long)cout * cin * 16 * 3; }

int wino2s_pack(const float* u32, __bf16* u, int cin, int cout, hipStream_t stream) {
    Wino2sPackArgs pa;
    pa.u32 = u32; pa.u = u; pa.cin = cin; pa.cout = cout;
    long long blocks = ((long long)cin * cout * 16 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(wino2s_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int wino2s_launch(const WinoKArgs& w, const __bf16* u, hipStream_t stream, long long* flops_out) {
    Wino2sKArgs a;
    a.x = w.x; a.y = w.y; a.res = w.res; a.u = u; a.scale = w.scale; a.shift = w.shift;
    a.N = w.N; a.H = w.H; a.W = w.W; a.cin = w.cin; a.x_cs = w.x_cs;
    a.cout = w.cout; a.y_cs = w.y_cs; a.res_cs = w.res_cs; a.act = w.act;
    a.TH = (a.H + 1) / 2;
    a.TW = (a.W + 1) / 2;
    const W2sBlock b = wino2s_pick_block(a.N, a.TH, a.TW);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(a.TH, b.bh);
    a.nbx = ceil_div(a.TW, b.bw);
    a.ngi = ceil_div(a.N, b.ni);
    a.RH = 2 * b.bh + 2;
    a.RW = 2 * b.bw + 2;
    wino2s_plane_geom(b.bh, b.bw, b.ni, &a.pitch, &a.istride);
    a.inv_pitch = 1.0f / (float)a.pitch;
    a.inv_istride = 1.0f / (float)a.istride;
    a.nkc = a.cin / kS_KS;
    a.tiles_n = a.cout / kS_BC;
    a.total = (long long)a.ngi * a.nby * a.nbx * a.tiles_n;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE((long long)a.N * a.H * a.W < (1ll << 31), "tensor too large");
    if (flops_out) {   // dry run: 16 position-GEMMs of [items*64] x [64] x cin, six bf16 piece products per product
        *flops_out = 6ll * 2ll * 16 * a.total * kS_BT * kS_BC * a.cin;
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 256) grid = 256;        // persistent, one 512-thread workgroup per CU
    hipLaunchKernelGGL(conv_wino2s_kernel, dim3((unsigned)grid), dim3(512), 0, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

#ifdef W2S_TRACE
extern "C" int w2l_dbg_w2s_trace(unsigned long long* out_dev) {      // out_dev: device memory, 256 * 2 * 50 uint64
    if (hipMemcpyFromSymbol(out_dev, HIP_SYMBOL(w2l::w2s_trace_buf), sizeof(w2l::w2s_trace_buf), 0, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out_dev + 256 * 2 * 48, HIP_SYMBOL(w2l::w2s_trace_rt), sizeof(w2l::w2s_trace_rt), 0, hipMemcpyDeviceToDevice) != hipSuccess) return 2;
    return (int)hipDeviceSynchronize();
}
#endif

// EXPERIMENT (not part of libw2l_hip.so; never measured - written after the round's GPU budget ended, see EXPERIMENTS.md):
// a drop-in replacement unit for wav2lip_amd/csrc/conv_wino4.hip with the SAME interface (wino4_ok / wino4_u_floats /
// wino4_pack / wino4_init_attrs / wino4_launch) and the same arithmetic per product, but a different split of the work
// inside the workgroup.  Build and select it with
//     bash tools/build_variant.sh w4w ../../tools/experiments/conv_wino4w.hip "" conv_wino4
//     W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_w4w.so python -m pytest tests/test_conv_gpu.py -m gpu -q -k wino4
//
// Why.  conv_wino4.hip gives every wave a 3 x 3 block of the 36 Winograd positions for 32 tiles x 32 couts (9 accumulators
// of 32x32x2 MFMAs).  The inverse transform A^T M A then needs all 36 positions of a (tile, cout) pair, which live in FOUR
// waves: the partial results meet in LDS in four rounds (write 4x the output, barrier, read, barrier), and the phase
// trace (tools/wino4_trace.py) shows that this epilogue + the prologue are 36 % of a 64-channel work item with
// nothing overlapping them.
// Here every wave owns ALL 36 positions of a 16-cout x 16-tile block on `v_mfma_f32_16x16x4_f32` (36 accumulators of 4
// registers = the same 144), with the weights as the A operand so that a lane ends up with 4 consecutive couts of ONE
// tile for every position: the whole inverse transform is per-lane packed arithmetic on registers, the output leaves as
// float4 stores, and the epilogue touches LDS only to look up the tile's pixel.  No staging tile, no epilogue barriers.
// Because the epilogue leaves V and the raw buffers alone, the K-steps of successive work items of a workgroup run as ONE
// software pipeline: the last two steps of an item request the first two input blocks of the next one, its last step
// transforms the next item's first block, its last weight requests are the next item's first fragments.  Only the first
// item of a workgroup pays a prologue (two exposed HBM latencies per item in conv_wino4.hip).
// -DW4W_HALF=1 builds the same kernel as a 256-thread workgroup over 16 tiles x 64 couts (one wave per cout quarter, three
// transform waves with two rows each, 61 KB of LDS): TWO independent workgroups per CU, one wave of each per SIMD, so that what
// one workgroup cannot overlap (epilogue, index math, barrier waits) the other's MFMAs cover.  Price: every weight fragment is
// fetched per 16 tiles instead of per 32 (twice the L2 -> CU weight traffic), and 16-tile blocks cover the 6x6- and 3x3-tile
// images of the 24x24 / 12x12 layers at 75 % - it is meant for the 96x96 and 48x48 layers, the ones at 0.40-0.46 today.
// Price: 16x16x4 MFMAs read twice the operands per FLOP (18 ds_read_b128 + 18 global dwordx4 per wave and K-step instead
// of 9 + 9) - 32 B/clk/CU of LDS reads, far below the 248 B/clk measured for ds_read_b128 (tools/microbench/lds_rate.hip).
//
// Layouts (K-step = 8 channels; MFMA g = 0, 1 of a position takes channel 2*kq + g from the lane with k index kq = lane >> 4):
//   V (LDS, per buffer)  [pp = position pair 18][th = tile half 2][slot 64][4]: slot sigma(kq, row) = kq*16 + ((row + 8*(kq>>1)) & 15)
//                        holds {V[2pp][tile th*16+row][2kq], V[2pp][..][2kq+1], V[2pp+1][..][2kq], V[2pp+1][..][2kq+1]}:
//                        one ds_read_b128 per lane feeds the 4 MFMAs of a position pair, conflict-free for the readers (linear
//                        up to the rotation) and for the 6 ds_write_b128 per lane of the transform waves (the rotation by 8
//                        puts the two channel quads of a 16-lane store group on disjoint bank halves).
//   U (global)           [cb = cout / 16][kc][pp 18][lane 64][4] with lane = kq*16 + cout % 16 and the same 4 floats: one
//                        coalesced 1 KB load per wave and position pair.
#include <type_traits>

#include "w2l_common.h"
#include "w2l_pk.h"

namespace w2l {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2 pk2_mul4(f32x2 a) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, 4.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ f32x4 pk_mul4(f32x4 a) { return cat(pk2_mul4(a.lo), pk2_mul4(a.hi)); }
__device__ __forceinline__ f32x2 pk2_fma_v(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x4 pk_fma_v(f32x4 a, f32x4 b, f32x4 c) { return cat(pk2_fma_v(a.lo, b.lo, c.lo), pk2_fma_v(a.hi, b.hi, c.hi)); }

// One row of A^T applied to six values m0..m5 (rows of A^T: (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1));
// products by 2, 4, 8 are exact.
template <int ROW>
__device__ __forceinline__ f32x4 w4w_at(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5) {
    if (ROW == 0) return pk_add(pk_add(m0, pk_add(m1, m2)), pk_add(m3, m4));
    if (ROW == 1) {
        const f32x4 d = pk_sub(m3, m4);
        return pk_add(pk_sub(m1, m2), pk_add(d, d));
    }
    if (ROW == 2) return pk_add(pk_add(m1, m2), pk_mul4(pk_add(m3, m4)));
    const f32x4 d = pk_sub(m3, m4);
    return pk_add(pk_add(pk_sub(m1, m2), pk_mul4(pk_add(d, d))), m5);
}

#ifdef W4_TRACE
__device__ unsigned long long w4_trace_buf[256 * 16 * 8];
__device__ unsigned long long w4_trace_rt[256 * 16 * 2];
#define W4_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && trace_item < 16 && blockIdx.x < 256) {                                          \
            w4_trace_buf[(blockIdx.x * 16 + trace_item) * 8 + (k)] = __builtin_readcyclecounter();              \
            if ((k) == 0 || (k) == 7)                                                                            \
                w4_trace_rt[(blockIdx.x * 16 + trace_item) * 2 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                        \
    } while (0)
#else
#define W4_STAMP(k)
#endif

#ifndef W4W_RING
#define W4W_RING 3      // weight fragments in flight per wave (must divide 18)
#endif
#ifndef W4W_HALF
#define W4W_HALF 0      // 1: work item = 16 tiles x 64 couts on a 4-wave workgroup, TWO independent workgroups per CU (one wave of each
#endif                  // per SIMD): what one workgroup cannot overlap - epilogue, index math, barrier waits - the other's MFMAs cover

constexpr unsigned kW4Oob = 0x80000000u;
constexpr bool kHalf = W4W_HALF != 0;
constexpr int kW4Threads = kHalf ? 256 : 512;
constexpr int kW4BT = kHalf ? 16 : 32;   // 4x4 output tiles per workgroup
constexpr int kW4BC = 64;          // couts per workgroup
constexpr int kW4KS = 8;           // channels per K-step
constexpr int kW4VPP = (kW4BT / 16) * 64 * 4;    // floats per position pair (tile halves x 64 slots x 4)
constexpr int kW4VBUF = 18 * kW4VPP;             // 36 KB (18 KB) per buffer
constexpr int kW4NRAW = 3;
constexpr int kW4RAW4 = kW4Threads * kW4NRAW;    // float4 slots per raw buffer
// raw planes exactly as in conv_wino4.hip: entry (16 bytes = one channel quad of one pixel) = q * QS + (x & 3) * PS + cell
constexpr int kW4PS = kHalf ? 90 : 186;       // 16-tile blocks whose planes fit: (4,4,1), (2,8,1), (2,3,2), (3,2,2), (2,2,3), ...
constexpr int kW4QS = 4 * kW4PS;
static_assert(2 * kW4QS <= kW4RAW4 && kW4PS % 8 == 2 && kW4QS % 16 == 8, "raw plane geometry");
constexpr int kW4LdsFloats = 2 * kW4VBUF + 2 * kW4RAW4 * 4;
constexpr int kW4LdsBytes = kW4LdsFloats * 4 + 2 * 2 * kW4BT * 4;      // + two (pixel, flags) tile tables, by item parity
static_assert(kW4LdsBytes <= 160 * 1024, "LDS budget");
static_assert(18 % W4W_RING == 0, "the weight ring must divide the 18 position pairs of a K-step");

struct Wino4KArgs {
    const float* x;
    float* y;
    const float* res;
    const float* u;
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;
    int cout, y_cs, res_cs;
    int TH, TW;
    int bh, bw, ni;
    int nby, nbx, ngi;
    int RH, RW, R4;
    int pitch, istride;
    float inv_rw, inv_rh;
    int nks;
    int tiles_n;
    long long total;
    int act;
};

__global__ __launch_bounds__(kW4Threads, 2) void conv_wino4w_f32_kernel(const Wino4KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);                  // [2][18][2][64][4]
    float* Rs = Vs + 2 * kW4VBUF;                                // [2][RAW4] float4 slots
    int* s_tab = reinterpret_cast<int*>(Rs + 2 * kW4RAW4 * 4);   // [2][{pixel, flags}][BT]

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u), 0, (int)((long long)a.cout * a.cin * 36 * 4), 0x00020000);

    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int th = kHalf ? 0 : (wave & 1);       // tile half of the MFMA work
    const int cq = kHalf ? wave : (wave >> 1);   // cout quarter of the MFMA work
    const int bhw = a.bh * a.bw;
    const int nsteps = a.cin / kW4KS;

    // ---- raw block slots of this thread: slot e = t + 512*k -> channel quad q = (e >> 3) & 1 of pixel (e >> 4) * 8 + (e & 7) of the
    // block's input region.  The LDS entry of a slot is recomputed where it is stored (a dozen VALU instructions per slot and
    // K-step) instead of living in three registers through the K loop: that is what pays for a deeper weight ring.
    auto raw_slot = [&](int k) {          // byte offset of the slot's entry inside a raw buffer, -1: no pixel
        const int e = t + kW4Threads * k;
        const int q = (e >> 3) & 1, pix = (e >> 4) * 8 + (e & 7);
        int st = -1;
        if (pix < a.R4) {
            const int p2 = (int)(((float)pix + 0.5f) * a.inv_rw);
            const int rxx = pix - p2 * a.RW;
            const int il = (int)(((float)p2 + 0.5f) * a.inv_rh);
            const int ry = p2 - il * a.RH;
            st = (q * kW4QS + (rxx & 3) * kW4PS + il * a.istride + ry * a.pitch + (rxx >> 2)) * 16;
        }
        return st;
    };
    // global byte offsets of the three slots for the work item at (gi, by_i, bx_i); kW4Oob where the slot has no pixel
    auto item_goff = [&](int gi, int by_i, int bx_i, bool valid, unsigned (&goff)[kW4NRAW]) {
#pragma unroll
        for (int k = 0; k < kW4NRAW; ++k) {
            const int e = t + kW4Threads * k;
            const int q = (e >> 3) & 1, pix = (e >> 4) * 8 + (e & 7);
            unsigned off = kW4Oob;
            if (valid && pix < a.R4) {
                // exact small-integer division through the reciprocal (half-integer numerators, pix < 768)
                const int p2 = (int)(((float)pix + 0.5f) * a.inv_rw);
                const int rxx = pix - p2 * a.RW;
                const int il = (int)(((float)p2 + 0.5f) * a.inv_rh);
                const int ry = p2 - il * a.RH;
                const int n = gi * a.ni + il;
                const int iy = 4 * by_i * a.bh - 1 + ry, ix = 4 * bx_i * a.bw - 1 + rxx;
                if (n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(q * 4)) * 4u;
            }
            goff[k] = off;
        }
    };
    auto item_coords = [&](unsigned bid, int& tile_n, int& bx_i, int& by_i, int& gi) {
        tile_n = (int)(bid % (unsigned)a.tiles_n);
        unsigned mb = bid / (unsigned)a.tiles_n;
        bx_i = (int)(mb % (unsigned)a.nbx);
        mb /= (unsigned)a.nbx;
        by_i = (int)(mb % (unsigned)a.nby);
        gi = (int)(mb / (unsigned)a.nby);
    };

    f32x4 rawreg[kW4NRAW];
    auto raw_gload = [&](const unsigned (&goff)[kW4NRAW], int step) {
        const unsigned soff = (unsigned)(step * kW4KS * 4);
#pragma unroll
        for (int k = 0; k < kW4NRAW; ++k)
            rawreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)goff[k], (int)soff, 0));
    };
    auto raw_store = [&](int buf) {
        char* dst = reinterpret_cast<char*>(Rs) + buf * (kW4RAW4 * 16);
#pragma unroll
        for (int k = 0; k < kW4NRAW; ++k) {
            const int st = raw_slot(k);
            if (st >= 0) *reinterpret_cast<f32x4*>(dst + st) = rawreg[k];
        }
    };

    // ---- input transform: row `trow` of B^T d B for (tile tl, channel quad q = lane & 1), as conv_wino4.hip.  512 threads: waves
    // 0..5 take one row each over 32 tiles; 256 threads: waves 0..2 take two rows each (one per 32-lane half) over 16 tiles
    const bool tf_wave = kHalf ? wave < 3 : wave < 6;
    const int trow = kHalf ? 2 * wave + (lane >> 5) : wave;
    const int q = lane & 1;
    int ra, rb, rc, rd;
    float ca, cb, cc;
    switch (trow) {
        case 0: ra = 0; ca = 0.f; rb = 0; cb = 4.f; rc = 2; cc = -5.f; rd = 4; break;
        case 1: ra = 1; ca = -4.f; rb = 2; cb = -4.f; rc = 3; cc = 1.f; rd = 4; break;
        case 2: ra = 1; ca = 4.f; rb = 2; cb = -4.f; rc = 3; cc = -1.f; rd = 4; break;
        case 3: ra = 1; ca = -2.f; rb = 2; cb = -1.f; rc = 3; cc = 2.f; rd = 4; break;
        case 4: ra = 1; ca = 2.f; rb = 2; cb = -1.f; rc = 3; cc = -2.f; rd = 4; break;
        default: ra = 1; ca = 0.f; rb = 1; cb = 4.f; rc = 3; cc = -5.f; rd = 5; break;
    }
    const int tl = kHalf ? ((lane & 31) >> 1) : (lane >> 1);
    int tf_base;
    {
        const int il = tl / bhw, r = tl - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int ilc = il < a.ni ? il : 0;      // unused tile slots read image 0's region: finite, never stored
        tf_base = (q * kW4QS + ilc * a.istride + 4 * tyl * a.pitch + txl) * 16;
    }
    const int rp = a.pitch * 16;
    const int o_a = tf_base + ra * rp, o_b = tf_base + rb * rp, o_c = tf_base + rc * rp, o_d = tf_base + rd * rp;
    // V slot of this lane's channels 4q, 4q+1 (kq = 2q; channels 4q+2, 4q+3 sit 16 slots further): positions 6*wave + j
    float* const vwr = Vs + ((trow < 6 ? trow : 0) * 3) * kW4VPP + (tl >> 4) * 256 + ((2 * q) * 16 + (((tl & 15) + 8 * q) & 15)) * 4;
    f32x4 rr[6];
    auto tf_rows = [&](int buf, int c) {
        const char* src = reinterpret_cast<const char*>(Rs) + buf * (kW4RAW4 * 16);
        const int co = ((c & 3) * kW4PS + (c >> 2)) * 16;
        const f32x4 va = *reinterpret_cast<const f32x4*>(src + o_a + co), vb = *reinterpret_cast<const f32x4*>(src + o_b + co);
        const f32x4 vc = *reinterpret_cast<const f32x4*>(src + o_c + co), vd = *reinterpret_cast<const f32x4*>(src + o_d + co);
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[c][e] = fmaf(ca, va[e], fmaf(cb, vb[e], fmaf(cc, vc[e], vd[e])));
    };
    // (B^T d) B along the columns for channels 4q + 2*half, +1 of the quad: 6 positions -> 3 V stores (two slots per K-step, so that
    // only half of the column transform's values are live next to the six row registers)
    auto tf_cols_store = [&](int buf, auto HALF) {
        constexpr int h = decltype(HALF)::value;
        float* dst = vwr + buf * kW4VBUF + 64 * h;
        f32x2 v0, v1, v2, v3, v4, v5;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = 2 * h + e2;
            const float p = fmaf(-4.f, rr[2][e], rr[4][e]);
            const float qq = fmaf(4.f, rr[1][e], -rr[3][e]);
            const float p2 = rr[4][e] - rr[2][e];
            const float q2 = 2.f * (rr[3][e] - rr[1][e]);
            v0[e2] = fmaf(4.f, rr[0][e], fmaf(-5.f, rr[2][e], rr[4][e]));
            v1[e2] = p - qq;
            v2[e2] = p + qq;
            v3[e2] = p2 + q2;
            v4[e2] = p2 - q2;
            v5[e2] = fmaf(4.f, rr[1][e], fmaf(-5.f, rr[3][e], rr[5][e]));
        }
        *reinterpret_cast<f32x4*>(dst + 0 * kW4VPP) = cat(v0, v1);
        *reinterpret_cast<f32x4*>(dst + 1 * kW4VPP) = cat(v2, v3);
        *reinterpret_cast<f32x4*>(dst + 2 * kW4VPP) = cat(v4, v5);
    };

    // ---- weight fragments: one dwordx4 per position pair of the (16-cout block, K-step) at byte offset ublk * 18432
    const unsigned bl_lane = (unsigned)(lane * 16);
    auto bload = [&](unsigned ublk, int pp) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, (int)bl_lane, (int)(ublk * 18432u + (unsigned)pp * 1024u), 0));
    };
    constexpr int RING = W4W_RING;
    f32x4 bq[RING];
    // V fragment of this lane: slot sigma(kq = lane >> 4, row = lane & 15) of its tile half
    const float* const Abase = Vs + th * 256 + ((lane & 48) | (((lane & 15) + ((lane >> 5) << 3)) & 15)) * 4;

    // The K-steps of successive work items form ONE pipeline: the last two steps of an item request the first two raw blocks of
    // the next item, its last step transforms the next item's first block and its last weight loads are the next item's first
    // fragments - the epilogue in between touches neither V nor the raw buffers.  Only the first item of a workgroup has a prologue.
    int item_parity = 0;      // which tile table this item uses
    int gpar = 0;             // parity of the pipeline's step counter at the item's first K-step
    bool first = true;
#ifdef W4_TRACE
    int trace_item = -1;
#endif
    for (unsigned jw = blockIdx.x >> 3; jw < per; jw += gw) {
    const unsigned bid = xcd * per + jw;
    if (bid >= total) break;
#ifdef W4_TRACE
    ++trace_item;
#endif
    W4_STAMP(0);
    int tile_n, bx_i, by_i, gi;
    item_coords(bid, tile_n, bx_i, by_i, gi);
    const int n0 = tile_n * kW4BC;
    const unsigned ublk0 = (unsigned)(((n0 >> 4) + cq) * nsteps);
    // the next item of this workgroup (its loads start inside this item's K loop); none: loads go out of range
    const unsigned bid_n = xcd * per + jw + gw;
    const bool has_next = (jw + gw < per) & (bid_n < total);
    int tile_nn, bx_n, by_n, gi_n;
    item_coords(has_next ? bid_n : bid, tile_nn, bx_n, by_n, gi_n);
    const unsigned ublk_n = (unsigned)(((tile_nn * kW4BC >> 4) + cq) * nsteps);
    // the epilogue of the previous item reads its table without a barrier behind it: this item writes the other one
    int* const s_opix = s_tab + item_parity * 2 * kW4BT;
    int* const s_oflag = s_opix + kW4BT;
    item_parity ^= 1;

    if (t < kW4BT) {
        const int il = t / bhw, r = t - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int n = gi * a.ni + il, ty = by_i * a.bh + tyl, tx = bx_i * a.bw + txl;
        int o = -1, f = 0;
        if (il < a.ni && n < a.N && ty < a.TH && tx < a.TW) {
            o = (n * a.H + 4 * ty) * a.W + 4 * tx;
#pragma unroll
            for (int k = 0; k < 4; ++k) f |= ((4 * ty + k < a.H) ? (1 << k) : 0) | ((4 * tx + k < a.W) ? (16 << k) : 0);
        }
        s_opix[t] = o;
        s_oflag[t] = f;
    }
    unsigned goff[kW4NRAW];
    item_goff(gi, by_i, bx_i, true, goff);

    f32x4 acc[36];
#pragma unroll
    for (int s = 0; s < 36; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (first) {      // ---- prologue of the workgroup's first item
        first = false;
        raw_gload(goff, 0);
#pragma unroll
        for (int i = 0; i < RING; ++i) bq[i] = bload(ublk0, i);
        raw_store(gpar);
        raw_gload(goff, nsteps > 1 ? 1 : 0);
        __syncthreads();                 // raw block 0, tile table
        W4_STAMP(1);
        if (tf_wave) {
#pragma unroll
            for (int c = 0; c < 6; ++c) tf_rows(gpar, c);
            tf_cols_store(gpar, std::integral_constant<int, 0>{});
            tf_cols_store(gpar, std::integral_constant<int, 1>{});
        }
        raw_store(gpar ^ 1);
        __syncthreads();                 // V of step 0, raw block 1
    } else {
        W4_STAMP(1);                     // (trace builds) later items have no prologue: stamps 0-2 bracket their index math
    }
    W4_STAMP(2);

    for (int step = 0; step < nsteps; ++step) {
        const int buf = (gpar + step) & 1;
        const float* Ab = Abase + buf * kW4VBUF;
        const bool tail = step + 2 >= nsteps;            // the raw block requested in this step belongs to the next item
        const unsigned ub_next = (step + 1 < nsteps) ? ublk0 + (unsigned)(step + 1) : ublk_n;
        f32x4 vf = *reinterpret_cast<const f32x4*>(Ab);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const f32x4 vc = vf;
            if (s < 17) vf = *reinterpret_cast<const f32x4*>(Ab + (s + 1) * kW4VPP);
            const f32x4 uc = bq[s % RING];
            bq[s % RING] = (s < 18 - RING) ? bload(ublk0 + (unsigned)step, s + RING) : bload(ub_next, s + RING - 18);
            // the rest of the K-step between the MFMA groups (even slots): raw block two steps ahead requested, the row transform
            // of the next step (waves 0-5, one column per slot), the column transform + 6 V stores, the requested raw block -> LDS
            if (s == 0) {
                if (!tail) {
                    raw_gload(goff, step + 2);
                } else {
                    unsigned gn[kW4NRAW];
                    item_goff(gi_n, by_n, bx_n, has_next, gn);
                    raw_gload(gn, step + 2 - nsteps);
                }
            } else if (s >= 2 && s <= 12 && (s & 1) == 0) {
                if (tf_wave) tf_rows(buf ^ 1, (s >> 1) - 1);
            } else if (s == 14) {
                if (tf_wave) tf_cols_store(buf ^ 1, std::integral_constant<int, 0>{});
            } else if (s == 15) {
                if (tf_wave) tf_cols_store(buf ^ 1, std::integral_constant<int, 1>{});
            } else if (s == 16) {
                raw_store(buf);
            }
            __builtin_amdgcn_sched_barrier(0);
            // weights are the A operand (rows = couts), V the B operand (columns = tiles); the two MFMAs of one position
            // are two issues apart (16x16x4: 32-cycle issue, 40-cycle dependent latency)
            acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[0], vc[0], acc[2 * s], 0, 0, 0);
            acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[2], vc[2], acc[2 * s + 1], 0, 0, 0);
            acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[1], vc[1], acc[2 * s], 0, 0, 0);
            acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[3], vc[3], acc[2 * s + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    gpar = (gpar + nsteps) & 1;

    W4_STAMP(3);
    // ---- epilogue, per lane: acc[6i + j][r] = M[i][j] of tile th*16 + (lane & 15), cout n0 + cq*16 + 4*(lane >> 4) + r.
    // Y = A^T M A row by row (every row of the 4x4 output tile needs all 36 positions; the shared sums are recomputed per row
    // instead of keeping 24 intermediate float4 next to the accumulators), then scale / shift / residual / activation and one
    // float4 store per pixel.  No LDS traffic besides the tile lookup, no barrier.
    {
        const long long npix = (long long)a.N * a.H * a.W;
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + a.cout) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0, 0x00020000);
        const int tile = th * 16 + (lane & 15);
        const int opix = s_opix[tile];
        const int fl = s_oflag[tile];
        const int ch = n0 + cq * 16 + 4 * (lane >> 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
        const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
        const bool relu = a.act == W2L_ACT_RELU;
        auto out_row = [&](auto OA) {
            constexpr int oa = decltype(OA)::value;
            const bool rok = (opix >= 0) & (((fl >> oa) & 1) != 0);
            unsigned yo[4], ro[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const bool ok = rok & (((fl >> (4 + ob)) & 1) != 0);
                const unsigned pix = (unsigned)(opix + oa * a.W + ob);
                yo[ob] = ok ? (pix * (unsigned)a.y_cs + (unsigned)ch) * 4u : kW4Oob;
                ro[ob] = ok ? (pix * (unsigned)a.res_cs + (unsigned)ch) * 4u : kW4Oob;
            }
            f32x4 rv[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
                rv[ob] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)ro[ob], 0, 0));
            f32x4 T[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) T[j] = w4w_at<oa>(acc[j], acc[6 + j], acc[12 + j], acc[18 + j], acc[24 + j], acc[30 + j]);
            f32x4 Y[4];
            Y[0] = w4w_at<0>(T[0], T[1], T[2], T[3], T[4], T[5]);
            Y[1] = w4w_at<1>(T[0], T[1], T[2], T[3], T[4], T[5]);
            Y[2] = w4w_at<2>(T[0], T[1], T[2], T[3], T[4], T[5]);
            Y[3] = w4w_at<3>(T[0], T[1], T[2], T[3], T[4], T[5]);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const f32x4 xv = pk_add(pk_fma_v(Y[ob], sc, sh), rv[ob]);
                f32x4 v;
                if (relu) {          // wave-uniform: one v_max per element instead of min / max / fma
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(xv[e], 0.f);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(neg_slope, fminf(xv[e], 0.f), fmaxf(xv[e], 0.f));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (int)yo[ob], 0, 0);
            }
        };
        out_row(std::integral_constant<int, 0>{});
        W4_STAMP(4);
        out_row(std::integral_constant<int, 1>{});
        W4_STAMP(5);
        out_row(std::integral_constant<int, 2>{});
        W4_STAMP(6);
        out_row(std::integral_constant<int, 3>{});
    }
    W4_STAMP(7);
    }   // persistent loop
}

// ---- weight transform: U = G g G^T (6x6) in fp64, rounded once, in the fragment order of the kernel above
struct Wino4PackArgs {
    const float* w;   // [cout][cin][3][3], or (transposed) [cin][cout][3][3] read as the flipped kernel with swapped roles
    float* u;         // [cout/16][cin/8][18][64][4]
    int cin, cout;
    int transposed;
};

// one thread per (cout, cin) pair: j = ((cb * nks + kc) * 64 + lane) * 2 + g with lane = kq * 16 + cout % 16, cin = kc*8 + 2*kq + g
__global__ void wino4w_pack_kernel(const Wino4PackArgs a) {
    const int total = a.cout * a.cin;
    const int nks = a.cin / 8;
    constexpr double G[6][3] = {{0.25, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int g1 = j & 1, ln = (j >> 1) & 63;
        const int blk = j >> 7;                  // (cb, kc)
        const int kc = blk % nks, cbk = blk / nks;
        const int co = cbk * 16 + (ln & 15);
        const int ci = kc * 8 + 2 * (ln >> 4) + g1;
        const float* g = a.transposed ? a.w + ((long long)ci * a.cout + co) * 9 : a.w + ((long long)co * a.cin + ci) * 9;
        double gd[3][3];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) gd[aa][bb] = (double)(a.transposed ? g[(2 - aa) * 3 + (2 - bb)] : g[aa * 3 + bb]);
        float* dst = a.u + (long long)blk * 18 * 256 + ln * 4 + g1;
#pragma unroll
        for (int pi = 0; pi < 6; ++pi)
#pragma unroll
            for (int pj = 0; pj < 6; ++pj) {
                double sum = 0.0;
#pragma unroll
                for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) sum += G[pi][aa] * gd[aa][bb] * G[pj][bb];
                const int pos = pi * 6 + pj;
                dst[(pos >> 1) * 256 + (pos & 1) * 2] = (float)sum;
            }
    }
}

struct W4Block { int bh, bw, ni; };
static const W4Block kW4Blocks[] = {{4, 8, 1}, {8, 4, 1}, {4, 4, 2}, {2, 8, 2}, {8, 2, 1}, {2, 4, 3}, {4, 2, 3}, {3, 3, 3},
                                    {2, 2, 6}, {2, 3, 4}, {3, 2, 4}, {1, 4, 6}, {4, 1, 5}, {1, 2, 10}, {2, 1, 9}, {1, 1, 15},
                                    // 16-tile blocks (W4W_HALF); wino4_block_fits drops what exceeds the tile / plane budget
                                    {4, 4, 1}, {2, 8, 1}, {2, 4, 2}, {4, 2, 2}, {2, 2, 4}, {2, 3, 2}, {3, 2, 2}, {1, 4, 4}, {4, 1, 3},
                                    {1, 2, 8}, {2, 1, 8}, {1, 1, 16}, {3, 3, 1}, {2, 2, 3}};

static bool wino4_block_fits(const W4Block& b) {
    const int RH = 4 * b.bh + 2, RW = 4 * b.bw + 2;
    return b.bh * b.bw * b.ni <= kW4BT && b.ni * RH * RW * 2 <= kW4RAW4 && b.ni * RH * (b.bw + 1) <= kW4PS;
}

// raw-plane geometry of a block (as conv_wino4.hip): pitch / image stride under which the 16 lanes of every ds_read_b128 lane
// group of a transform read land on as many distinct 16-byte bank slots as possible
static void wino4_plane_geom(const W4Block& b, int* pitch, int* istride) {
    static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int RH = 4 * b.bh + 2, bhw = b.bh * b.bw;
    int best = 1 << 30;
    *pitch = b.bw + 1;
    *istride = RH * (b.bw + 1);
    for (int p = b.bw + 1; p <= b.bw + 4; ++p)
        for (int is = RH * p; is <= RH * p + 15; ++is) {
            if (b.ni * is > kW4PS) break;
            int cost = (p - b.bw - 1) + (is - RH * p);
            for (int g = 0; g < 4; ++g) {
                int cnt[16] = {0};
                for (int k = 0; k < 16; ++k) {
                    const int lane = kGroup[g & 1][k] + 32 * (g >> 1);
                    const int tl = kHalf ? ((lane & 31) >> 1) : (lane >> 1), q = lane & 1;
                    const int il = tl / bhw, r = tl % bhw;
                    const int ilc = il < b.ni ? il : 0;
                    ++cnt[(q * kW4QS + ilc * is + 4 * (r / b.bw) * p + r % b.bw) & 15];
                }
                for (int k = 0; k < 16; ++k) cost += cnt[k] > 1 ? (cnt[k] - 1) * 64 * cnt[k] : 0;
            }
            if (cost < best) { best = cost; *pitch = p; *istride = is; }
        }
}

static W4Block wino4_pick_block(int N, int TH, int TW) {
    W4Block best = {1, 1, 1};
    double best_cost = 1e300;
    for (const W4Block& b : kW4Blocks) {
        if (!wino4_block_fits(b)) continue;
        const double items = (double)ceil_div(TH, b.bh) * ceil_div(TW, b.bw) * ceil_div(N, b.ni);
        const double halo = (double)(4 * b.bh + 2) * (4 * b.bw + 2) / (16.0 * b.bh * b.bw);
        const double cost = items * (1.0 + 0.05 * halo);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}

bool wino4_ok(int cin, int cout) { return cin % kW4KS == 0 && cin >= 2 * kW4KS && cout % kW4BC == 0; }   // the item pipeline looks two K-steps ahead

long long wino4_u_floats(int cin, int cout) { return (long long)cout * cin * 36; }

int wino4_pack(const float* w, float* u, int cin, int cout, int transposed, hipStream_t stream) {
    Wino4PackArgs pa;
    pa.w = w; pa.u = u; pa.cin = cin; pa.cout = cout; pa.transposed = transposed;
    long long blocks = ((long long)cin * cout + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino4w_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int wino4_init_attrs() {
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4w_f32_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kW4LdsBytes));
    done = true;
    return W2L_OK;
}

int wino4_launch(const WinoKArgs& w, const float* u4, hipStream_t stream, long long* flops_out) {
    Wino4KArgs a;
    a.x = w.x; a.y = w.y; a.res = w.res; a.u = u4; a.scale = w.scale; a.shift = w.shift;
    a.N = w.N; a.H = w.H; a.W = w.W; a.cin = w.cin; a.x_cs = w.x_cs;
    a.cout = w.cout; a.y_cs = w.y_cs; a.res_cs = w.res_cs; a.act = w.act;
    a.TH = (a.H + 3) / 4;
    a.TW = (a.W + 3) / 4;
    const W4Block b = wino4_pick_block(a.N, a.TH, a.TW);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(a.TH, b.bh);
    a.nbx = ceil_div(a.TW, b.bw);
    a.ngi = ceil_div(a.N, b.ni);
    a.RH = 4 * b.bh + 2;
    a.RW = 4 * b.bw + 2;
    a.R4 = b.ni * a.RH * a.RW;
    wino4_plane_geom(b, &a.pitch, &a.istride);
    a.inv_rw = 1.0f / (float)a.RW;
    a.inv_rh = 1.0f / (float)a.RH;
    a.nks = a.cin / 8;
    a.tiles_n = a.cout / kW4BC;
    a.total = (long long)a.ngi * a.nby * a.nbx * a.tiles_n;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE((long long)a.N * a.H * a.W < (1ll << 31), "tensor too large");
    if (flops_out) {
        *flops_out = 2ll * 36 * a.total * kW4BT * kW4BC * a.cin;
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > (kHalf ? 512 : 256)) grid = kHalf ? 512 : 256;      // persistent: one 512-thread or two 256-thread workgroups per CU
    hipLaunchKernelGGL(conv_wino4w_f32_kernel, dim3((unsigned)grid), dim3(kW4Threads), kW4LdsBytes, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

#ifdef W4_TRACE
extern "C" int w2l_dbg_w4_trace(unsigned long long* out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(w2l::w4_trace_buf), sizeof(w2l::w4_trace_buf)) != hipSuccess) return 1;
    return (int)hipMemcpyFromSymbol(out + 256 * 16 * 8, HIP_SYMBOL(w2l::w4_trace_rt), sizeof(w2l::w4_trace_rt));
}
#endif

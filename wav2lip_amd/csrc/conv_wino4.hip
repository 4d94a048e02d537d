// Winograd F(4x4, 3x3) convolution for gfx950 on the fp32 matrix cores: 36 multiplies per 4x4 output tile and (cin, cout)
// pair instead of 144 (direct) or 64 (F(2x2,3x3), conv_wino2.hip) - 1.78x less matrix work than the F(2x2) kernels for the
// 3x3 / stride 1 / pad 1 layers of the generator (models/wav2lip.py:61-81 via models/conv.py:5-19).
//
//   y = act( A^T [ sum_c (G g G^T)[xi] * (B^T d B)[xi] ] A * scale + shift (+ res) ),   xi = (i, j) in 6 x 6
//   B^T = | 4  0 -5  0  1  0 |   G = | 1/4    0     0  |   A^T = | 1  1  1  1  1  0 |
//         | 0 -4 -4  1  1  0 |       |-1/6  -1/6  -1/6 |         | 0  1 -1  2 -2  0 |
//         | 0  4 -4 -1  1  0 |       |-1/6   1/6  -1/6 |         | 0  1  1  4  4  0 |
//         | 0 -2 -1  2  1  0 |       | 1/24  1/12  1/6 |         | 0  1 -1  8 -8  1 |
//         | 0  2 -1 -2  1  0 |       | 1/24 -1/12  1/6 |
//         | 0  4  0 -5  0  1 |       |  0     0     1  |
// (interpolation points 0, +-1, +-2, inf).  All products and sums in fp32; the weight side is transformed in fp64 and rounded
// once.  Accuracy, measured on the whole generator before the kernel was written (tools/wino_f4x4_accuracy.py,
// profiles/r02/wino_f4x4_accuracy.txt): 3e-6 per layer relative to the layer's scale, pixel L-inf 7.3e-7 against fp64 (direct
// fp32: 3.1e-7), against a parity budget of 1e-3.
//
// Structure = conv_wino2.hip with the position grid cut 2 x 2 instead of 1 x 2.  Workgroup = 8 waves (two per SIMD) = 32 tiles
// x 64 couts x 36 positions, one per CU; wave (wn, I, J) owns 32 tiles x 32 couts x the 3 x 3 position block i in 3I..3I+2,
// j in 3J..3J+2 (9 accumulators = 144 registers).  A^T M A is bilinear in the blocks: every wave forms  A^T[:, I] M[I, J] A[J, :]
// (a 4 x 4 partial result per tile and cout) in registers; the four partials meet in an LDS staging tile, in four rounds over
// groups of 8 tiles (a quarter of the accumulators dies per round: room for the next round's residual, requested one round
// ahead), and the float4 output pass adds them.  Per K-step (8 channels): the input block of the workgroup ((4bh+2) x (4bw+2)
// pixels per image for bh x bw tiles) is loaded once into LDS (<= 3 float4 per thread) in (channel quad, x & 3) planes; waves
// 0-5 each compute one row of B^T d B for every (tile, channel quad) - 24 conflict-free LDS reads, 124 VALU, 6 LDS writes per
// thread, one raw column per MFMA slot; every wave reads 9 V fragments and 9 weight fragments for its 36 MFMAs.
// What was measured and dropped on the way (profiles/r02/k2-k5): a two-workgroup / 32-cout / 4-channel design,
// epilogue prefetches that spill next to the accumulators, a carried-state item loop that spills raw-block offsets in the K loop.
#include "w2l_common.h"
#include "w2l_pk.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2 pk2_mul4(f32x2 a) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, 4.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}

// One 3-column block of A^T applied along one axis of a wave's 3 x 3 position block, two tiles per instruction:
//   block 0 (positions 0-2): rows (1 1 1), (0 1 -1), (0 1 1), (0 1 -1)        -> o0 = m0 + s, o1 = o3 = d, o2 = s     (s, d = m1 +- m2)
//   block 1 (positions 3-5): rows (1 1 0), (2 -2 0), (4 4 0), (8 -8 1)        -> o0 = s, o1 = 2d, o2 = 4s, o3 = 8d + m2 (s, d = m0 +- m1)
// Products by 2, 4, 8 are exact, so every output is rounded where the textbook order rounds it.
template <int BLK>
__device__ __forceinline__ void w4_at3(f32x2 m0, f32x2 m1, f32x2 m2, f32x2& o0, f32x2& o1, f32x2& o2, f32x2& o3) {
    if (BLK == 0) {
        const f32x2 sm = pk2_add(m1, m2);
        o1 = pk2_sub(m1, m2);
        o0 = pk2_add(m0, sm);
        o2 = sm;
        o3 = o1;
    } else {
        const f32x2 sm = pk2_add(m0, m1);
        const f32x2 df = pk2_sub(m0, m1);
        o0 = sm;
        o1 = pk2_add(df, df);
        o2 = pk2_mul4(sm);
        o3 = pk2_add(pk2_mul4(o1), m2);
    }
}

// Partial inverse transform of one wave, accumulator registers r, r+1 (two tiles of one cout):  P[a][b] = sum_il sum_jl
// AT[a][3PI+il] M[il][jl] AT[b][3PJ+jl], written to the staging tile at y[((tile*4 + a)*4 + b) * LDY], tiles r and r+1 being
// `tstride` floats apart.  Block 0 has two equal output rows (1 and 3): 18-35 packed instructions per pair instead of the 168
// scalar ones of the generic wave-uniform-coefficient form.
template <int PI, int PJ, int LDY>
__device__ __forceinline__ void w4_partial_store(const f32x16 (&acc)[9], const int r, float* y, const int tstride) {
    f32x2 R[4][3];
#pragma unroll
    for (int jl = 0; jl < 3; ++jl) {
        const f32x2 m0 = {acc[0 + jl][r], acc[0 + jl][r + 1]};
        const f32x2 m1 = {acc[3 + jl][r], acc[3 + jl][r + 1]};
        const f32x2 m2 = {acc[6 + jl][r], acc[6 + jl][r + 1]};
        w4_at3<PI>(m0, m1, m2, R[0][jl], R[1][jl], R[2][jl], R[3][jl]);
    }
    f32x2 P1[4];                        // row 1, kept for row 3 of block 0
#pragma unroll
    for (int oa = 0; oa < 4; ++oa) {
        f32x2 P[4];
        if (PI == 0 && oa == 3) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) P[ob] = P1[ob];
        } else {
            w4_at3<PJ>(R[oa][0], R[oa][1], R[oa][2], P[0], P[1], P[2], P[3]);
        }
        if (oa == 1) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) P1[ob] = P[ob];
        }
        float* yo = y + (oa * 4) * LDY;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            yo[ob * LDY] = P[ob].x;
            yo[ob * LDY + tstride] = P[ob].y;
        }
    }
}

#ifndef W4_DBG
#define W4_DBG 0      // timing ablations of the K loop (variant builds only, tools/build_variant.sh): 1 no MFMA, 2 no transform, 4 no raw loads, 8 no weight loads
#endif
#ifdef W4_TRACE
// phase timestamps (variant builds only; tools/wino4_trace.py): [workgroup][item < 16][stamp < 8] s_memtime values of wave 0, and
// s_memrealtime (100 MHz) at stamps 0 and 7 to calibrate the shader clock
__device__ unsigned long long w4_trace_buf[256 * 16 * 8];
__device__ unsigned long long w4_trace_rt[256 * 16 * 2];
#define W4_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && trace_item < 16) {                                                               \
            w4_trace_buf[(blockIdx.x * 16 + trace_item) * 8 + (k)] = __builtin_readcyclecounter();              \
            if ((k) == 0 || (k) == 7)                                                                            \
                w4_trace_rt[(blockIdx.x * 16 + trace_item) * 2 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                        \
    } while (0)
#else
#define W4_STAMP(k)
#endif

constexpr unsigned kW4Oob = 0x80000000u;
constexpr int kW4BT = 32;          // 4x4 output tiles per workgroup
constexpr int kW4BC = 64;          // couts per workgroup
constexpr int kW4KS = 8;           // channels per K-step
constexpr int kW4LDK = kW4KS + 4;  // V row stride (floats)
constexpr int kW4VPOS = kW4BT * kW4LDK;
constexpr int kW4VBUF = 36 * kW4VPOS;
constexpr int kW4NRAW = 3;
constexpr int kW4RAW4 = 512 * kW4NRAW;           // float4 slots per raw buffer
// The raw block lives in LDS "planar": entry (16 bytes = one channel quad of one pixel) = q * QS + (x & 3) * PS + cell, with
// cell = image * istride + y * pitch + (x >> 2).  A transform read (all lanes the same (row, column) of their own tile) then
// touches consecutive cells of one plane instead of entries 4 pixels apart (4-way bank conflicts in the first version), and
// the host picks pitch / istride so that the 16 lanes of a ds_read_b128 group land on 16 distinct 16-byte bank slots.
constexpr int kW4PS = 186;                       // = 2 mod 8: 8 consecutive pixels of a store group hit 8 distinct slots
constexpr int kW4QS = 4 * kW4PS;                 // = 8 mod 16: the two channel quads of a read group use disjoint halves
static_assert(2 * kW4QS <= kW4RAW4 && kW4PS % 8 == 2 && kW4QS % 16 == 8, "raw plane geometry");
constexpr int kW4LDY = kW4BC + 4;
constexpr int kW4LdsFloats = 2 * kW4VBUF + 2 * kW4RAW4 * 4;
constexpr int kW4LdsBytes = kW4LdsFloats * 4 + 2 * kW4BT * 4;
static_assert(4 * kW4BT * 4 * kW4LDY <= kW4LdsFloats, "one round of the four partial staging tiles must fit");
static_assert(kW4LdsBytes <= 160 * 1024, "LDS budget");

struct Wino4KArgs {
    const float* x;
    float* y;
    const float* res;
    const float* u;      // transformed weights, wino4_pack below
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;
    int cout, y_cs, res_cs;
    int TH, TW;          // 4x4 output tiles per image
    int bh, bw, ni;      // tile block of a workgroup
    int nby, nbx, ngi;
    int RH, RW, R4;      // raw region per image (4bh+2, 4bw+2) and pixels per K-step ni*RH*RW (two float4 each)
    int pitch, istride;  // raw planes: cells per region row (>= bw+1) and per image (>= RH*pitch)
    float inv_rw, inv_rh;
    int nks;             // cin / 8
    int tiles_n;         // cout / 64
    long long total;
    int act;
};

__global__ __launch_bounds__(512, 2) void conv_wino4_f32_kernel(const Wino4KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);                  // [2][36][32][LDK]
    float* Rs = Vs + 2 * kW4VBUF;                                // [2][RAW4] float4 slots, linear in the load index
    int* s_opix = reinterpret_cast<int*>(Rs + 2 * kW4RAW4 * 4);  // [32] output pixel (4ty, 4tx) of a tile or -1
    int* s_oflag = s_opix + kW4BT;                               // [32] valid rows (bits 0-3) and columns (bits 4-7) of the tile

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);

    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
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
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave & 1;            // cout half
    const int pb = wave >> 1;           // position block
    const int PI = pb >> 1, PJ = pb & 1;
    const int tile_n = (int)(bid % (unsigned)a.tiles_n);
    unsigned mb = bid / (unsigned)a.tiles_n;
    const int bx_i = (int)(mb % (unsigned)a.nbx);
    mb /= (unsigned)a.nbx;
    const int by_i = (int)(mb % (unsigned)a.nby);
    const int gi = (int)(mb / (unsigned)a.nby);
    const int n0 = tile_n * kW4BC;
    const int bhw = a.bh * a.bw;

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

    // ---- raw block loads: slot e = t + 512*k -> channel quad q = (e >> 3) & 1 of pixel (e >> 4) * 8 + (e & 7) of the block's
    // input region (8 consecutive lanes = 8 consecutive pixels of one quad: conflict-free 16-byte LDS stores into the planes)
    unsigned goff[kW4NRAW];
    int rst[kW4NRAW];                   // byte offset of the slot's entry inside a raw buffer, -1: no pixel
#pragma unroll
    for (int k = 0; k < kW4NRAW; ++k) {
        const int e = t + 512 * k;
        const int q = (e >> 3) & 1, pix = (e >> 4) * 8 + (e & 7);
        unsigned off = kW4Oob;
        int st = -1;
        if (pix < a.R4) {
            // exact small-integer division through the reciprocal (half-integer numerators, pix < 768)
            const int p2 = (int)(((float)pix + 0.5f) * a.inv_rw);
            const int rxx = pix - p2 * a.RW;
            const int il = (int)(((float)p2 + 0.5f) * a.inv_rh);
            const int ry = p2 - il * a.RH;
            st = (q * kW4QS + (rxx & 3) * kW4PS + il * a.istride + ry * a.pitch + (rxx >> 2)) * 16;
            const int n = gi * a.ni + il;
            const int iy = 4 * by_i * a.bh - 1 + ry, ix = 4 * bx_i * a.bw - 1 + rxx;
            if (n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(q * 4)) * 4u;
        }
        goff[k] = off;
        rst[k] = st;
    }
    f32x4 rawreg[kW4NRAW];
    auto raw_gload = [&](int step) {
        const unsigned soff = (unsigned)(step * kW4KS * 4);
#pragma unroll
        for (int k = 0; k < kW4NRAW; ++k)
            rawreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)goff[k], (int)soff, 0));
    };
    auto raw_store = [&](int buf) {
        char* dst = reinterpret_cast<char*>(Rs) + buf * (kW4RAW4 * 16);
#pragma unroll
        for (int k = 0; k < kW4NRAW; ++k)
            if (rst[k] >= 0) *reinterpret_cast<f32x4*>(dst + rst[k]) = rawreg[k];
    };

    // ---- transform item of waves 0..5: row i = wave of B^T d B for (tile = lane>>1, channel quad q = lane&1):
    //   row i of B^T d  =  ca*d[ra] + cb*d[rb] + cc*d[rc] + d[rd]           (wave-uniform rows and coefficients)
    const bool tf_wave = wave < 6;
    const int q = lane & 1;
    int ra, rb, rc, rd;
    float ca, cb, cc;
    switch (wave) {
        case 0: ra = 0; ca = 0.f; rb = 0; cb = 4.f; rc = 2; cc = -5.f; rd = 4; break;
        case 1: ra = 1; ca = -4.f; rb = 2; cb = -4.f; rc = 3; cc = 1.f; rd = 4; break;
        case 2: ra = 1; ca = 4.f; rb = 2; cb = -4.f; rc = 3; cc = -1.f; rd = 4; break;
        case 3: ra = 1; ca = -2.f; rb = 2; cb = -1.f; rc = 3; cc = 2.f; rd = 4; break;
        case 4: ra = 1; ca = 2.f; rb = 2; cb = -1.f; rc = 3; cc = -2.f; rd = 4; break;
        default: ra = 1; ca = 0.f; rb = 1; cb = 4.f; rc = 3; cc = -5.f; rd = 5; break;
    }
    int tf_base;
    {
        const int tl = lane >> 1;
        const int il = tl / bhw, r = tl - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int ilc = il < a.ni ? il : 0;      // unused tile slots read image 0's region: finite, never stored
        tf_base = (q * kW4QS + ilc * a.istride + 4 * tyl * a.pitch + txl) * 16;      // bytes, plane (q, x & 3 = 0)
    }
    const int rp = a.pitch * 16;
    const int o_a = tf_base + ra * rp, o_b = tf_base + rb * rp, o_c = tf_base + rc * rp, o_d = tf_base + rd * rp;
    float* const vwr = Vs + ((wave < 6 ? wave : 0) * 6) * kW4VPOS + (lane >> 1) * kW4LDK + q * 4;
    f32x4 rr[6];
    auto tf_rows = [&](int buf, int c0) {         // rr[c0] = row i of B^T d, column c0
        const char* src = reinterpret_cast<const char*>(Rs) + buf * (kW4RAW4 * 16);
#pragma unroll
        for (int c = c0; c < c0 + 1; ++c) {
            const int co = ((c & 3) * kW4PS + (c >> 2)) * 16;      // plane of the column, next cell for columns 4, 5
            const f32x4 va = *reinterpret_cast<const f32x4*>(src + o_a + co), vb = *reinterpret_cast<const f32x4*>(src + o_b + co);
            const f32x4 vc = *reinterpret_cast<const f32x4*>(src + o_c + co), vd = *reinterpret_cast<const f32x4*>(src + o_d + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) rr[c][e] = fmaf(ca, va[e], fmaf(cb, vb[e], fmaf(cc, vc[e], vd[e])));
        }
    };
    auto tf_cols_store = [&](int buf) {           // (B^T d) B: the same matrix along the columns; 6 positions -> V
        float* dst = vwr + buf * kW4VBUF;
        f32x4 v0, v1, v2, v3, v4, v5;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = fmaf(-4.f, rr[2][e], rr[4][e]);
            const float qq = fmaf(4.f, rr[1][e], -rr[3][e]);
            const float p2 = rr[4][e] - rr[2][e];
            const float q2 = 2.f * (rr[3][e] - rr[1][e]);
            v0[e] = fmaf(4.f, rr[0][e], fmaf(-5.f, rr[2][e], rr[4][e]));
            v1[e] = p - qq;
            v2[e] = p + qq;
            v3[e] = p2 + q2;
            v4[e] = p2 - q2;
            v5[e] = fmaf(4.f, rr[1][e], fmaf(-5.f, rr[3][e], rr[5][e]));
        }
        *reinterpret_cast<f32x4*>(dst + 0 * kW4VPOS) = v0;
        *reinterpret_cast<f32x4*>(dst + 1 * kW4VPOS) = v1;
        *reinterpret_cast<f32x4*>(dst + 2 * kW4VPOS) = v2;
        *reinterpret_cast<f32x4*>(dst + 3 * kW4VPOS) = v3;
        *reinterpret_cast<f32x4*>(dst + 4 * kW4VPOS) = v4;
        *reinterpret_cast<f32x4*>(dst + 5 * kW4VPOS) = v5;
    };

    // ---- B operand: u[((nb * nks + kc) * 36 + pos) * 256 + (h*32 + n)*4 + e] = U_pos[nb*32 + n][kc*8 + 4h + e], pos = 6i + j;
    // this wave reads pos(s) = 6*(3*PI + s/3) + 3*PJ + s%3, s = 0..8
    const int nb = (n0 >> 5) + wn;
    const int F = a.nks * 36;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u + (long long)nb * F * 256), 0, F * 1024, 0x00020000);
    const unsigned bl_lane = (unsigned)(lane * 16);
    const int pos0 = 18 * PI + 3 * PJ;             // position of (il, jl) = (0, 0)
    const unsigned bl_pb = (unsigned)pos0 * 1024u;
    auto bload = [&](int kc, int s) {              // s compile-time at every call site
        const unsigned soff = (unsigned)kc * 36864u + bl_pb + (unsigned)(6 * (s / 3) + (s % 3)) * 1024u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, (int)bl_lane, (int)soff, 0));
    };
    constexpr int RING = 3;
    f32x4 bq[RING];

    f32x16 acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // ---- prologue
    const int nsteps = a.cin / kW4KS;
    raw_gload(0);
#pragma unroll
    for (int i = 0; i < RING; ++i) bq[i] = bload(0, i);
    raw_store(0);
    raw_gload(1);
    __syncthreads();                 // raw[0], tile table
    W4_STAMP(1);
    if (tf_wave) {
#pragma unroll
        for (int c = 0; c < 6; ++c) tf_rows(0, c);
        tf_cols_store(0);
    }
    raw_store(1);
    __syncthreads();                 // V[0], raw[1]
    W4_STAMP(2);

    const float* Abase = Vs + pos0 * kW4VPOS + (lane & 31) * kW4LDK + (lane >> 5) * 4;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const float* Ab = Abase + buf * kW4VBUF;
        f32x4 af = *reinterpret_cast<const f32x4*>(Ab);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const f32x4 ac = af;
            if (s < 8) af = *reinterpret_cast<const f32x4*>(Ab + (6 * ((s + 1) / 3) + ((s + 1) % 3)) * kW4VPOS);
            const f32x4 bc = bq[s % RING];
#if !(W4_DBG & 8)
            bq[s % RING] = (s < 9 - RING) ? bload(step, s + RING) : bload(step + 1, s + RING - 9);
#endif
            // the rest of the K-step between the MFMA groups: slot 0 requests the raw block of step+2; slots 1-6 the row
            // transform of step+1 (waves 0-5, one column each); slot 7 the column transform + 6 V stores; slot 8 raw(step+2) -> LDS
            if (s == 0) {
                if (!(W4_DBG & 4)) raw_gload(step + 2);
            } else if (s >= 1 && s <= 6) {
                if (!(W4_DBG & 2) && tf_wave) tf_rows(buf ^ 1, s - 1);
            } else if (s == 7) {
                if (!(W4_DBG & 2) && tf_wave) tf_cols_store(buf ^ 1);
            } else if (s == 8) {
                if (!(W4_DBG & 4)) raw_store(buf);
            }
            __builtin_amdgcn_sched_barrier(0);
#if W4_DBG & 1
            asm volatile("" : "+v"(acc[s]) : "v"(ac), "v"(bc));
#else
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[e], bc[e], acc[s], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    W4_STAMP(3);
    // ---- epilogue.  acc[3*il + jl][r] = M[3PI + il][3PJ + jl] for cout lane&31, tile (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Partial result of this wave:  P[a][b] = sum_il sum_jl  AT[a][3PI+il] * M[il][jl] * AT[b][3PJ+jl]  (the A^T entries are
    // 0, +-1, +-2, +-4, +-8: every product is exact, so the partials differ from the textbook order only in the order of sums)
    // Four rounds, one per group of 8 tiles (accumulator registers 4k..4k+3): every wave stores the 4 x 4 partial outputs of its 8
    // tiles into staging tile [pb][tile][a][b][LDY]; the float4 pass sums the four partials, applies scale / shift / residual /
    // activation and stores.  Rounds run over TILES, not output rows: a quarter of the accumulators dies with every round, which
    // is what leaves registers for the residual of the next round (requested one round ahead, so that a wait in this epilogue
    // never meets a request just issued) without spilling next to the 144 accumulators.
    float* Ys = Vs;
    const long long npix = (long long)a.N * a.H * a.W;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + a.cout) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0, 0x00020000);
    constexpr int CG = kW4BC / 4;
    const int c4 = t % CG;
    const int ch = n0 + c4 * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    constexpr int kPart = 8 * 16 * kW4LDY;         // floats per partial staging tile (8 tiles x 16 pixels)
    constexpr int NIT = 8 * 16 * CG / 512;         // 4 float4 per thread and round
    auto out_pix = [&](int round, int i) {         // output pixel of this thread's i-th float4 of the round, or -1
        const int id = i * 512 + t;
        const int pxl = id / CG;                   // tile8 * 16 + a * 4 + b
        const int tile = 8 * round + (pxl >> 4), oa = (pxl >> 2) & 3, ob = pxl & 3;
        const int opix = s_opix[tile];
        const int fl = s_oflag[tile];
        const bool ok = (opix >= 0) & (((fl >> oa) & 1) != 0) & (((fl >> (4 + ob)) & 1) != 0);
        return ok ? opix + oa * a.W + ob : -1;
    };
    f32x4 rv[NIT];
    auto res_load = [&](int round) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pix = out_pix(round, i);
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rres, (int)(pix >= 0 ? ((unsigned)pix * (unsigned)a.res_cs + (unsigned)ch) * 4u : kW4Oob), 0, 0));
        }
    };
    res_load(0);
#pragma unroll
    for (int round = 0; round < 4; ++round) {       // tiles 8*round .. 8*round + 7
        {
            // tiles (r & 3) + 4 * (lane >> 5) of this round's group, r = 4 * round + {0, 1}, {2, 3}
            float* yrow = Ys + pb * kPart + wn * 32 + (lane & 31) + (4 * (lane >> 5)) * 16 * kW4LDY;
            constexpr int TS = 16 * kW4LDY;
            if (pb == 0) {
                w4_partial_store<0, 0, kW4LDY>(acc, 4 * round, yrow, TS);
                w4_partial_store<0, 0, kW4LDY>(acc, 4 * round + 2, yrow + 2 * TS, TS);
            } else if (pb == 1) {
                w4_partial_store<0, 1, kW4LDY>(acc, 4 * round, yrow, TS);
                w4_partial_store<0, 1, kW4LDY>(acc, 4 * round + 2, yrow + 2 * TS, TS);
            } else if (pb == 2) {
                w4_partial_store<1, 0, kW4LDY>(acc, 4 * round, yrow, TS);
                w4_partial_store<1, 0, kW4LDY>(acc, 4 * round + 2, yrow + 2 * TS, TS);
            } else {
                w4_partial_store<1, 1, kW4LDY>(acc, 4 * round, yrow, TS);
                w4_partial_store<1, 1, kW4LDY>(acc, 4 * round + 2, yrow + 2 * TS, TS);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 512 + t;
            const int pix = out_pix(round, i);
            const bool ok = pix >= 0;
            const f32x4 rvv = rv[i];
            const float* src = Ys + (id / CG) * kW4LDY + c4 * 4;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(src);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(src + kPart);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(src + 2 * kPart);
            const f32x4 p3 = *reinterpret_cast<const f32x4*>(src + 3 * kPart);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xv = fmaf((p0[e] + p1[e]) + (p2[e] + p3[e]), sc[e], sh[e]) + rvv[e];
                v[e] = fmaf(neg_slope, fminf(xv, 0.f), fmaxf(xv, 0.f));
            }
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(u32x4, v), ry, (int)(ok ? ((unsigned)pix * (unsigned)a.y_cs + (unsigned)ch) * 4u : kW4Oob), 0, 0);
        }
        if (round < 3) res_load(round + 1);
        __syncthreads();
        W4_STAMP(4 + round);
    }
    }   // persistent loop
}

// ---- weight transform: U = G g G^T (6x6) in fp64, rounded once, in MFMA B-fragment order
struct Wino4PackArgs {
    const float* w;   // [cout][cin][3][3], or (transposed) [cin][cout][3][3] read as the flipped kernel with swapped roles
    float* u;         // [cout/32][cin/8][36][2][32][4]
    int cin, cout;
    int transposed;
};

// one thread per (cout, cin) pair: the 9 taps are read once and the 36 positions leave as 36 coalesced stores (a weight update of a
// training step re-runs this for every 3x3 layer: the first version - one thread per output element, 64-bit div / mod by 36 and a
// runtime-indexed G in scratch - cost 3 ms per cfg4 step)
__global__ void wino4_pack_kernel(const Wino4PackArgs a) {
    const int total = a.cout * a.cin;
    const int nks = a.cin / 8;
    constexpr double G[6][3] = {{0.25, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int e = j & 3, n = (j >> 2) & 31, h = (j >> 7) & 1;
        const int blk = j >> 8;                  // (nb, kc)
        const int kc = blk % nks, nbk = blk / nks;
        const int co = nbk * 32 + n;
        const int ci = kc * 8 + 4 * h + e;
        const float* g = a.transposed ? a.w + ((long long)ci * a.cout + co) * 9 : a.w + ((long long)co * a.cin + ci) * 9;
        double gd[3][3];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) gd[aa][bb] = (double)(a.transposed ? g[(2 - aa) * 3 + (2 - bb)] : g[aa * 3 + bb]);
        float* dst = a.u + (long long)blk * 36 * 256 + (j & 255);
#pragma unroll
        for (int pi = 0; pi < 6; ++pi)
#pragma unroll
            for (int pj = 0; pj < 6; ++pj) {
                double sum = 0.0;
#pragma unroll
                for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) sum += G[pi][aa] * gd[aa][bb] * G[pj][bb];
                dst[(pi * 6 + pj) * 256] = (float)sum;
            }
    }
}

struct W4Block { int bh, bw, ni; };
// candidate tile blocks: bh*bw*ni <= 32 tiles, ni*(4bh+2)*(4bw+2) <= 768 pixels, ni*(4bh+2)*(bw+1) <= PS plane cells
static const W4Block kW4Blocks[] = {{4, 8, 1}, {8, 4, 1}, {4, 4, 2}, {2, 8, 2}, {8, 2, 1}, {2, 4, 3}, {4, 2, 3}, {3, 3, 3},
                                    {2, 2, 6}, {2, 3, 4}, {3, 2, 4}, {1, 4, 6}, {4, 1, 5}, {1, 2, 10}, {2, 1, 9}, {1, 1, 15}};

static bool wino4_block_fits(const W4Block& b) {
    const int RH = 4 * b.bh + 2, RW = 4 * b.bw + 2;
    return b.bh * b.bw * b.ni <= kW4BT && b.ni * RH * RW * 2 <= kW4RAW4 && b.ni * RH * (b.bw + 1) <= kW4PS;
}

// raw-plane geometry of a block: the row pitch and image stride (cells) under which the 16 lanes of every ds_read_b128 lane group
// of a transform read (lane = (tile, quad); the hardware's groups are {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32)
// land on as many distinct 16-byte bank slots (entry mod 16) as possible
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
                    const int tl = lane >> 1, q = lane & 1;
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

bool wino4_ok(int cin, int cout) { return cin % kW4KS == 0 && cout % kW4BC == 0; }

long long wino4_u_floats(int cin, int cout) { return (long long)cout * cin * 36; }

int wino4_pack(const float* w, float* u, int cin, int cout, int transposed, hipStream_t stream) {
    Wino4PackArgs pa;
    pa.w = w; pa.u = u; pa.cin = cin; pa.cout = cout; pa.transposed = transposed;
    long long blocks = ((long long)cin * cout + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int wino4_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4_f32_kernel),
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
    if (flops_out) {   // dry run: 36 position-GEMMs of [items*32] x [64] x cin
        *flops_out = 2ll * 36 * a.total * kW4BT * kW4BC * a.cin;
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 256) grid = 256;        // persistent, one 512-thread workgroup per CU
    hipLaunchKernelGGL(conv_wino4_f32_kernel, dim3((unsigned)grid), dim3(512), kW4LdsBytes, stream, a);
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

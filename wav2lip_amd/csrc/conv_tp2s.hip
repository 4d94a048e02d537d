// Stride-2 transposed 3x3 convolution (nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1) + BN + ReLU, models/wav2lip.py:63-81 via
// models/conv.py:33-44) with all four output phases in ONE workgroup (conv_tp2.hip's structure) AND split operands (conv_igemm.hip's
// arithmetic): every fp32 operand enters the bf16 matrix cores as the exact sum of three bf16 pieces, a K-chunk of 16
// channels of one (tap, phase) product is the six piece products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 (smallest first, fp32
// accumulate) - an fp32 result with the fp32 kernels' error.
//
// Why: the split implicit GEMM runs the four phases as separate workgroups; each re-gathers AND re-splits its own A tiles (9 tile
// loads + 9 splits per input block where 4 shifted views of ONE staged block serve all 9 (tap, phase) products) and the one- and
// two-tap phases are short K loops.  Here a workgroup owns a block of input pixels (bh x bw in each of ni images, <= 128 rows) x 64
// couts x 4 phases; per K-step of 16 channels the input block (+1 halo row / column) is loaded once, split once (104 vector
// instructions per thread per 108 MFMAs of its wave) and stored as three bf16 planes [plane][k-half][pixel][8]; every wave reads its
// shifted A fragments from there (a shift is a pixel offset); the pre-split weights come from L2 in fragment order, one 1 KB fragment
// per (tap, plane, 32 couts, 16 channels), and feed two 32-row blocks each.  Wave (wm, wn) = 64 rows x 32 couts x 4 phases = 8
// accumulators (128 registers); two workgroups per CU.
#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kTsOob = 0x80000000u;
constexpr int kTsBM = 128;         // input pixels (GEMM rows) per workgroup
constexpr int kTsBC = 64;          // couts per workgroup
constexpr int kTsKS = 16;          // channels per K-step (the bf16 MFMA's K)
constexpr int kTsRP = 256;         // raw pixels per buffer
constexpr int kTsLDY = kTsBC + 4;
constexpr int kTsKhBytes = kTsRP * 16;                    // one k-half of one plane: [pixel][8 bf16]
constexpr int kTsPlaneBytes = 2 * kTsKhBytes;
constexpr int kTsBufBytes = 3 * kTsPlaneBytes;            // 24 KB
constexpr int kTsStageBytes = kTsBM * 2 * kTsLDY * 4;     // one epilogue round = 128 rows x 2 phases x 64 couts (fp32)
constexpr int kTsMainBytes = kTsStageBytes > 2 * kTsBufBytes ? kTsStageBytes : 2 * kTsBufBytes;
constexpr int kTsLdsBytes = kTsMainBytes + kTsBM * 4;
static_assert(2 * kTsLdsBytes <= 160 * 1024, "two workgroups per CU");

struct Tp2sKArgs {
    const float* x;
    float* y;
    const __bf16* u;     // packed pre-split weights, tp2s_pack below
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;      // input; output is N x 2H x 2W
    int cout, y_cs;
    int bh, bw, ni;      // pixel block of a workgroup: bh x bw input pixels in each of ni images (bh*bw*ni <= 128)
    int nby, nbx, ngi;
    int RH, RW, RP;      // raw region per image (bh+1, bw+1) and pixels per K-step ni*RH*RW (<= 256)
    int nkc;             // cin / 16
    int tiles_n;         // cout / 64
    long long total;     // work items: blocks * cout-tiles * ksplit
    int act;
    // split-K (layers with few pixel blocks, 1024 -> 512 at 3x3: 72 items for 256 CUs): K-steps are cut into ksplit ranges of
    // steps_per_split, work item = (range, block, cout-tile); each writes its raw partial sums to ws[range][pixel][cout] and
    // splitk_reduce_kernel (conv_igemm.hip) adds them up in a fixed order and applies scale / shift / activation
    int ksplit, steps_per_split;
    long long items;     // blocks * cout-tiles
    float* ws;
};

// taps in conv_tp2.hip's numbering: tap t has phase tp_phase(t) and input shift d = 2*dy + dx = tp_shift(t); this kernel walks them
// grouped by shift (one set of A fragments per shift): 0 2 4 8 | 1 7 | 3 6 | 5
__device__ __forceinline__ constexpr int ts_phase(int t) { return t == 0 ? 0 : (t < 3 ? 1 : (t < 5 ? 2 : 3)); }
__device__ __forceinline__ constexpr int ts_shift(int t) {
    return (t == 1 || t == 7) ? 1 : ((t == 3 || t == 6) ? 2 : (t == 5 ? 3 : 0));
}
__device__ __forceinline__ constexpr int ts_seq(int i) {
    return i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 4 : i == 3 ? 8 : i == 4 ? 1 : i == 5 ? 7 : i == 6 ? 3 : i == 7 ? 6 : 5;
}

__device__ __forceinline__ unsigned ts_pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (x0, x1) -> the three bf16 pieces of each, packed pairwise: x = h + m + l exactly (RNE at every step)
__device__ __forceinline__ void ts_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = ts_pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = ts_pack_bf16x2(r0, r1);
    l = ts_pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

__global__ __launch_bounds__(256, 2) void conv_tp2s_kernel(const Tp2sKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_opix = reinterpret_cast<int*>(smem + kTsMainBytes);        // [128] output pixel (2qy, 2qx) of a row or -1

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);
    const int Wo = 2 * a.W;
    const int bhw = a.bh * a.bw;

    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    for (unsigned jw = blockIdx.x >> 3; jw < per; jw += gw) {
    const unsigned bid0 = xcd * per + jw;
    if (bid0 >= total) break;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int kz = (int)(bid0 / (unsigned)a.items);
    const unsigned bid = bid0 - (unsigned)kz * (unsigned)a.items;
    const int k0 = kz * a.steps_per_split;
    const int k1 = k0 + a.steps_per_split < a.nkc ? k0 + a.steps_per_split : a.nkc;
    const int tile_n = (int)(bid % (unsigned)a.tiles_n);
    unsigned mb = bid / (unsigned)a.tiles_n;
    const int bx_i = (int)(mb % (unsigned)a.nbx);
    mb /= (unsigned)a.nbx;
    const int by_i = (int)(mb % (unsigned)a.nby);
    const int gi = (int)(mb / (unsigned)a.nby);
    const int n0 = tile_n * kTsBC;

    if (t < kTsBM) {                 // row table of the epilogue
        const int il = t / bhw, r = t - il * bhw;
        const int qyl = r / a.bw, qxl = r - qyl * a.bw;
        const int n = gi * a.ni + il, qy = by_i * a.bh + qyl, qx = bx_i * a.bw + qxl;
        s_opix[t] = (il < a.ni && n < a.N && qy < a.H && qx < a.W) ? (n * 2 * a.H + 2 * qy) * Wo + 2 * qx : -1;
    }

    // ---- raw block loads: slot e = t + 256*k -> (pixel p = e>>1 of the block's input region, k-half kh = e&1: 8 channels = 32 bytes)
    unsigned goff[2];
    int lds_off[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = t + 256 * k;
        const int kh = e & 1, p = e >> 1;
        unsigned off = kTsOob;
        if (p < a.RP) {
            const int rxx = p % a.RW, p2 = p / a.RW;
            const int ry = p2 % a.RH, il = p2 / a.RH;
            const int n = gi * a.ni + il;
            const int iy = by_i * a.bh + ry, ix = bx_i * a.bw + rxx;
            if (n < a.N && iy < a.H && ix < a.W)
                off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(kh * 8)) * 4u;
        }
        goff[k] = off;
        lds_off[k] = kh * kTsKhBytes + p * 16;
    }
    // slots 256 .. 511 hold pixels 128 .. 255: a block's raw region is ~150 pixels (9 x 17, 2 x 9 x 9, 3 x 7 x 7), so the second slot of
    // waves 1 (mostly), 2 and 3 is past it - a wave-uniform test skips its loads, split and stores
    const bool slot1 = (256 + wave * 64) < 2 * a.RP;
    f32x4 rawreg[2][2];
    auto raw_gload = [&](int step) {
#if defined(TS_DBG) && (TS_DBG & 4)
        return;
#endif
        const unsigned soff = (unsigned)(step * kTsKS * 4);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !slot1) break;
            rawreg[k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)goff[k], (int)soff, 0));
            rawreg[k][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(goff[k] + 16u), (int)soff, 0));
        }
    };
    auto raw_store = [&](int buf) {      // split once, three 16-byte stores per slot
#if defined(TS_DBG) && (TS_DBG & 2)
        return;
#endif
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !slot1) break;
            unsigned h[4], m[4], l[4];
            ts_split3_pair(rawreg[k][0][0], rawreg[k][0][1], h[0], m[0], l[0]);
            ts_split3_pair(rawreg[k][0][2], rawreg[k][0][3], h[1], m[1], l[1]);
            ts_split3_pair(rawreg[k][1][0], rawreg[k][1][1], h[2], m[2], l[2]);
            ts_split3_pair(rawreg[k][1][2], rawreg[k][1][3], h[3], m[3], l[3]);
            char* d = smem + buf * kTsBufBytes + lds_off[k];
            *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(d + kTsPlaneBytes) = u32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<u32x4*>(d + 2 * kTsPlaneBytes) = u32x4{l[0], l[1], l[2], l[3]};
        }
    };

    // ---- A fragments: row m = wm*64 + b*32 + (lane&31) -> raw pixel of (qy, qx); shift d adds dy*RW + dx pixels; lane>>5 = k-half
    int abase[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = wm * 64 + b * 32 + (lane & 31);
        const int il = m / bhw, r = m - il * bhw;
        const int qyl = r / a.bw, qxl = r - qyl * a.bw;
        const int p = il < a.ni ? (il * a.RH + qyl) * a.RW + qxl : 0;      // unused row slots read pixel 0: finite, never stored
        abase[b] = p * 16 + (lane >> 5) * kTsKhBytes;
    }
    const int shb1 = 16, shb2 = a.RW * 16, shb3 = (a.RW + 1) * 16;

    // ---- B operand: u[(((nb * nkc + kc) * 9 + tap) * 3 + plane) * 512 + lane * 8 + e]
    //               = piece `plane` of w[kc*16 + 8*(lane>>5) + e][nb*32 + (lane&31)][ky(tap)][kx(tap)]
    const int nb = (n0 >> 5) + wn;
    const int F = a.nkc * 27;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.u + (long long)nb * F * 512), 0, F * 1024, 0x00020000);
    const unsigned bl_lane = (unsigned)(lane * 16);
    auto bload = [&](int kc, int tap, int plane) {       // past-the-end chunks read zero (never used)
#if defined(TS_DBG) && (TS_DBG & 8)
        return __builtin_bit_cast(bf16x8, u32x4{(unsigned)kc, (unsigned)tap, (unsigned)plane, 0u});
#endif
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ru, (int)bl_lane, (int)((unsigned)((kc * 9 + tap) * 3 + plane) * 1024u), 0));
    };
    constexpr int RING = 3;
    bf16x8 bq[RING][3];

    f32x16 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][p][r] = 0.f;

    // ---- prologue: raw(k0) -> LDS[0]; raw(k0 + 1) in registers
    raw_gload(k0);
#pragma unroll
    for (int i = 0; i < RING; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[i][p] = bload(k0, ts_seq(i), p);
    raw_store(0);
    raw_gload(k0 + 1);
    __syncthreads();

    // the six piece products of a K-chunk, smallest first: (a2 b0) (a1 b1) (a0 b2) (a1 b0) (a0 b1) (a0 b0)
    constexpr int kPa[6] = {2, 1, 0, 1, 0, 0};
    constexpr int kPb[6] = {0, 1, 2, 0, 1, 0};

    for (int step = k0; step < k1; ++step) {
        const int buf = (step - k0) & 1;
        // raw(step+1) -> LDS[buf^1] (last read during step-1, a barrier ago), then request raw(step+2)
#ifndef TS_LATE_STORE
        raw_store(buf ^ 1);
        raw_gload(step + 2);
#endif
        const char* Ab = smem + buf * kTsBufBytes;
        bf16x8 af[2][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int tap = ts_seq(i);
            const int ph = ts_phase(tap), sd = ts_shift(tap);
            if (i == 0 || i == 4 || i == 6 || i == 8) {           // first tap of a shift group: this shift's A fragments
                const int sh = sd == 0 ? 0 : (sd == 1 ? shb1 : (sd == 2 ? shb2 : shb3));
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        af[b][p] = *reinterpret_cast<const bf16x8*>(Ab + p * kTsPlaneBytes + abase[b] + sh);
            }
            bf16x8 bc[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bc[p] = bq[i % RING][p];
            // ring of 3: sequence positions 3..8 of this chunk, then 0..2 of the next
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bq[i % RING][p] = (i < 6) ? bload(step, ts_seq(i + 3), p) : bload(step + 1, ts_seq(i - 6), p);
            __builtin_amdgcn_sched_barrier(0);
#ifdef TS_PRIO
            __builtin_amdgcn_s_setprio(2);
#endif
#if defined(TS_DBG) && (TS_DBG & 1)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p) acc[b][ph][p] += (float)af[b][p][0] * (float)bc[p][0];
#else
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[b][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[b][kPa[u]], bc[kPb[u]], acc[b][ph], 0, 0, 0);
#endif
#ifdef TS_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifdef TS_LATE_STORE
            if (i == TS_LATE_STORE) {
                raw_store(buf ^ 1);
                raw_gload(step + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        __syncthreads();
    }

    // ---- epilogue, two rounds of two phases: accumulators -> LDS staging [128 rows][2 phases][LDY] -> float4 rows of y.
    // acc[b][p][r]: row wm*64 + b*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), cout wn*32 + (lane&31), phase p = 2*py + px
    float* Ys = reinterpret_cast<float*>(smem);
    const long long npix = (long long)a.N * 2 * a.H * Wo;
    const bool partial = a.ksplit > 1;          // raw sums into this range's slab of the workspace, pixel stride cout
    const int ycs = partial ? a.cout : a.y_cs;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        partial ? a.ws + (long long)kz * npix * a.cout : a.y, 0, (int)(((npix - 1) * ycs + a.cout) * 4), 0x00020000);
    constexpr int CG = kTsBC / 4;
    const int c4 = t % CG;
    const int ch = n0 + c4 * 4;
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 sc = partial ? one : *reinterpret_cast<const f32x4*>(a.scale + ch);
    const f32x4 sh = partial ? zero : *reinterpret_cast<const f32x4*>(a.shift + ch);
    const float neg_slope = partial ? 1.f : (a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f));
#pragma unroll
    for (int round = 0; round < 2; ++round) {          // round = py
        {
            float* yrow = Ys + wn * 32 + (lane & 31);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = wm * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        yrow[(m * 2 + pp) * kTsLDY] = acc[b][2 * round + pp][r];
                    }
        }
        __syncthreads();
        constexpr int NIT = kTsBM * 2 * CG / 256;      // 16 float4 per thread
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 256 + t;
            const int rowp = id / CG;                  // m*2 + px
            const int m = rowp >> 1, px = rowp & 1;
            const int opix = s_opix[m];
            const f32x4 c = *reinterpret_cast<const f32x4*>(Ys + rowp * kTsLDY + c4 * 4);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xv = fmaf(c[e], sc[e], sh[e]);
                v[e] = fmaf(neg_slope, fminf(xv, 0.f), fmaxf(xv, 0.f));
            }
            const int pix = opix + round * Wo + px;
#if defined(TS_DBG) && (TS_DBG & 16)
            if (v[0] == 1.2345f)
#endif
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(u32x4, v), ry,
                (int)(opix >= 0 ? ((unsigned)pix * (unsigned)ycs + (unsigned)ch) * 4u : kTsOob), 0, 0);
        }
        __syncthreads();
    }
    }   // persistent loop
}

// ---- weight packing: conv_tp2's fp32 fragment-ordered weights (tp2_pack: u32[((nb*nks + kc8)*9 + tap)*256 + (h*32 + n)*4 + e4] =
// w[kc8*8 + 4h + e4][nb*32 + n][tap]) -> three bf16 pieces per value in this kernel's fragment order
struct Tp2sPackArgs {
    const float* u32;
    __bf16* u;
    int cin, cout;
};

__global__ void tp2s_pack_kernel(const Tp2sPackArgs a) {
    const long long total = (long long)a.cout * a.cin * 9;
    const int nkc = a.cin / 16, nks = a.cin / 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int ln = (int)((i >> 3) & 63);
        const long long rest = i >> 9;
        const int tap = (int)(rest % 9);
        const long long r2 = rest / 9;
        const int kc = (int)(r2 % nkc);
        const int nbk = (int)(r2 / nkc);
        const int kc8 = kc * 2 + (ln >> 5), h = e >> 2, e4 = e & 3, n = ln & 31;
        const float v = a.u32[(((long long)nbk * nks + kc8) * 9 + tap) * 256 + (h * 32 + n) * 4 + e4];
        const __bf16 hp = (__bf16)v;
        const float r1 = v - (float)hp;
        const __bf16 mp = (__bf16)r1;
        __bf16* d = a.u + (((long long)nbk * nkc + kc) * 9 + tap) * (3 * 512) + ln * 8 + e;
        d[0] = hp;
        d[512] = mp;
        d[1024] = (__bf16)(r1 - (float)mp);
    }
}

struct TsBlock { int bh, bw, ni; };
static const TsBlock kTsBlocks[] = {{8, 16, 1}, {16, 8, 1}, {8, 8, 2}, {4, 16, 2}, {4, 8, 4}, {4, 6, 5}, {6, 4, 5}, {2, 12, 5},
                                    {4, 12, 2}, {12, 4, 2}, {6, 6, 3}, {3, 12, 3}, {6, 12, 1}, {12, 6, 1}, {4, 4, 8}, {3, 3, 14},
                                    {2, 4, 16}, {2, 2, 28}, {1, 4, 25}, {1, 1, 64}};

static TsBlock tp2s_pick_block(int N, int H, int W) {
    TsBlock best = {1, 1, 1};
    double best_cost = 1e300;
    for (const TsBlock& b : kTsBlocks) {
        if (b.ni * (b.bh + 1) * (b.bw + 1) > kTsRP || b.bh * b.bw * b.ni > kTsBM) continue;
        const double items = (double)ceil_div(H, b.bh) * ceil_div(W, b.bw) * ceil_div(N, b.ni);
        const double halo = (double)(b.bh + 1) * (b.bw + 1) / ((double)b.bh * b.bw);
        const double cost = items * (1.0 + 0.03 * halo);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}

bool tp2s_ok(const w2l_conv_geom& g) { return tp2_ok(g) && g.cin % kTsKS == 0; }

long long tp2s_u_elems(int cin, int cout) { return (long long)cout * cin * 9 * 3; }

int tp2s_pack(const float* tp2_u32, __bf16* u, int cin, int cout, hipStream_t stream) {
    Tp2sPackArgs pa;
    pa.u32 = tp2_u32; pa.u = u; pa.cin = cin; pa.cout = cout;
    long long blocks = ((long long)cin * cout * 9 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(tp2s_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int tp2s_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tp2s_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kTsLdsBytes));
    done = true;
    return W2L_OK;
}

// ksplit > 1: `ws` holds ksplit * N * 4HW * cout floats of partial sums, the caller runs the reduce launch; *ksplit_out = the number of
// non-empty K ranges actually used
int tp2s_launch(const float* x, int x_cs, float* y, int y_cs, const __bf16* u, const float* scale, const float* shift, int N, int H,
                int W, int cin, int cout, int act, int ksplit, float* ws, int* ksplit_out, hipStream_t stream, long long* flops_out) {
    Tp2sKArgs a;
    a.x = x; a.y = y; a.u = u; a.scale = scale; a.shift = shift;
    a.N = N; a.H = H; a.W = W; a.cin = cin; a.x_cs = x_cs; a.cout = cout; a.y_cs = y_cs; a.act = act;
    const TsBlock b = tp2s_pick_block(N, H, W);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(H, b.bh);
    a.nbx = ceil_div(W, b.bw);
    a.ngi = ceil_div(N, b.ni);
    a.RH = b.bh + 1;
    a.RW = b.bw + 1;
    a.RP = b.ni * a.RH * a.RW;
    a.nkc = cin / kTsKS;
    a.tiles_n = cout / kTsBC;
    a.items = (long long)a.ngi * a.nby * a.nbx * a.tiles_n;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > a.nkc) ksplit = a.nkc;
    a.steps_per_split = ceil_div(a.nkc, ksplit);
    a.ksplit = ceil_div(a.nkc, a.steps_per_split);   // drop empty trailing ranges
    a.ws = ws;
    a.total = a.items * a.ksplit;
    if (ksplit_out) *ksplit_out = a.ksplit;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE((long long)N * 4 * H * W < (1ll << 31), "tensor too large");
    if (flops_out) {   // dry run: 9 (tap, phase) GEMMs of [items*128] x [64] x cin, six bf16 piece products per product
        *flops_out = 6ll * 2 * 9 * a.items * kTsBM * kTsBC * cin;
        return W2L_OK;
    }
    W2L_REQUIRE(a.ksplit == 1 || ws != nullptr, "tp2s: split-K without a workspace");
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(conv_tp2s_kernel, dim3((unsigned)grid), dim3(256), kTsLdsBytes, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

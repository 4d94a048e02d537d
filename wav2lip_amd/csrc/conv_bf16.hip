// bf16-STORAGE convolution / transposed convolution for the training configurations (BASELINE configs[3]/[4]: bf16):
//     y = act( conv(x, w) * scale + shift (+ res) ),   x, res, y: NHWC bf16 in HBM;  w: bf16 (packed from the fp32 master
// weights every optimiser step);  accumulation, scale / shift / residual / activation in fp32;  one rounding on the way out.
// Replaces, in bf16 mode, the torch ops behind models/conv.py:5-44 (forward) and their data gradients (the dgrad of a conv is a
// transposed conv over the same weight tensor and vice versa - only the packer's reading of the tensor changes).
//
// Implicit GEMM on v_mfma_f32_32x32x16_bf16: M = N*Hq*Wq "q" positions, N_gemm = cout, K = taps * cin_p (cin_p = cin rounded up
// to 8 so that a 16-byte chunk of K never straddles a tap).  What differs from the fp32 kernel (conv_igemm.hip):
//   * operands are bf16 in HBM: half the fetched bytes, no conversion instructions;
//   * K-step of 64 (4 MFMAs per 32x32 tile between barriers instead of the round-2 bf16 kernel's 2);
//   * both operand tiles travel HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write
//     (ds_write_b128 moves 79 B/clk/CU - at bf16 MFMA rates the VGPR->LDS path was the bound); an out-of-range offset makes
//     the DMA write zeros (probed: tools/microbench/probe_lds.hip, P2), which is how padding taps, ragged rows and ragged
//     couts are realised, as in the fp32 kernels;
//   * the LDS image of a DMA is lane-linear (destination = wave base + lane*16), so rows are unpadded 128-byte rows and the
//     bank-conflict-free layout is an XOR swizzle applied to the SOURCE address: the 16-byte slot p of tile row r holds K
//     chunk p ^ ((r >> 1) & 7); the 16 rows of a ds_read_b128 lane group then cover all 16 slots of the 256-byte bank window;
//   * two LDS buffers, the next K-step's DMAs are issued before this step's MFMAs and drained (vmcnt(0)) at the step's one
//     barrier; two workgroups per CU (64 KB LDS each at 128x128) overlap one's drain with the other's MFMAs;
//   * epilogue per WAVE through a private fp32 LDS tile (no workgroup barrier): rows leave as 16-byte bf16x8 stores.
#include <mutex>
#include <new>
#include <vector>

#include <stdlib.h>

#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef W2L_CONVB_DBG
#define W2L_CONVB_DBG 0   // timing ablations of the K loop (variant builds only): 1 no MFMA, 2 no K-loop DMA, 4 no fragment reads, 8 no step barrier
#endif
constexpr int kBKH = 64;                  // K elements per step = one 128-byte LDS row
constexpr unsigned kOobH = 0x80000000u;   // byte offset beyond any bound buffer (extents are checked < 2^31 on the host)

struct ConvBArgs {
    const void* x;
    void* y;
    const void* res;
    const void* w;        // bf16 [phase][cout_p][kp]
    const float* scale;   // [cout] or NULL (= 1)
    const float* shift;   // [cout] or NULL (= 0)
    const int* taps;      // (dy & 0xffff) | (dx << 16) per tap-table entry
    int N, H, W, cin_p, x_cs;
    int Ho, Wo, cout, cout_p, y_cs, res_cs;
    int Hq, Wq, sy, sx, omy, omx;
    int act;
    int M, tiles_m, tiles_n;
    int ksplit, steps_per_split;
    float* ws;            // split-K partial sums [ksplit][N*Ho*Wo][cout_p] fp32
    float* stats;         // NULL, or per-(phase, M tile, wave row) column partials [npart][2][cout_p]: sum and sum of squares of
                          // the bf16-ROUNDED outputs (the stored z; no activation on such launches) - BatchNorm statistics in the
                          // conv epilogue.  fp32 per-lane partials of <= a few dozen rows, every later level in fp64: the one-pass
                          // E[x^2]-E[x]^2 form loses ~|mean|^2/var * 1e-7 relative on the variance (tests: |mean| = 30 std)
    // BatchNorm-BACKWARD sums in the epilogue of a data-gradient launch (w2l_convb_forward_bnbwd): this launch's output is the dy of
    // a batch-statistics BatchNorm block whose pre-BatchNorm output is bz; with bz != NULL the stats partials are
    //   [..][0][c] = sum g,  [..][1][c] = sum g * zhat,   g = y_stored * act'(block output),  zhat = (bz - bmean) * brstd
    // over the bf16-ROUNDED outputs (what the elementwise pass re-reads).  by == NULL: a ReLU block without residual, whose output
    // sign is recomputed as bz * bscale + bshift > 0 (the forward's own expression); else by is the block output.
    const void* bz;
    const void* by;
    const float* bmean;
    const float* brstd;
    const float* bscale;
    const float* bshift;
    int bz_cs, by_cs;
    float bneg;           // act'(.) on the non-positive side: 0 ReLU, 0.01 LeakyReLU, 1 none
    int bstore_g;         // ReLU block: store g = dy * act'(.) (the masked gradient) instead of dy
    int bmask_only;       // no statistics: the output is dy * act'(by) of an activation block WITHOUT BatchNorm (w2l_convb_forward_actbwd)
    ConvPhase ph[kMaxPhases];   // kp / w_off in ELEMENTS
};

__device__ __forceinline__ float actb(int act, float v) {
    switch (act) {
        case W2L_ACT_RELU: return act_leaky(v, 0.f);          // the same expression as the un-split epilogue (w2l_common.h)
        case W2L_ACT_LEAKY: return act_leaky(v, 0.01f);
        case W2L_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

template <int BM, int BN>
constexpr int convb_lds_bytes() { return 2 * (BM + BN) * 128 + BM * 4 + 128 * 4; }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) void conv_bf16s_kernel(const ConvBArgs a) {
    constexpr int NW = WM * WN;                      // waves per workgroup: 4 (two workgroups per CU) or 8 (one: the 256x256 tile)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1 && TN <= 2, "wave tile: at least 32x32, at most 64 columns");
    constexpr int PR = 8 * NW;    // rows of one DMA pass: every wave 8 rows
    constexpr int PA = BM / PR;   // DMA passes over the A tile
    constexpr int PB = BN / PR;
    static_assert(BM % PR == 0 && BN % PR == 0, "tile rows per DMA pass");
    constexpr int STAGE = (BM + BN) * 128;          // bytes of one (A, B) buffer pair
    constexpr int LDCW = TN * 32 + 4;               // floats per row of a wave's private epilogue tile
    static_assert(NW * 32 * LDCW * 4 <= 2 * STAGE, "epilogue tiles must fit in the staging buffers");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_orow = reinterpret_cast<int*>(smem + 2 * STAGE);   // [BM] output pixel index or -1
    int* s_taps = s_orow + BM;                                 // [64][2]: (dy,dx), byte offset of the tap

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;

    const ConvPhase ph = a.ph[blockIdx.y];
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = bid % a.tiles_n;
    const int tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int HWq = a.Hq * a.Wq;
    const int kfirst = blockIdx.z * a.steps_per_split;

    if (t < 64) {
        const int tv = (t < ph.ntaps) ? a.taps[ph.tap_off + t] : 0;
        s_taps[2 * t] = tv;
        s_taps[2 * t + 1] = (((int)(short)(tv & 0xffff)) * a.W + (tv >> 16)) * a.x_cs * 2;
    }
    for (int r = t; r < BM; r += NW * 64) {
        const int m = m0 + r;
        int o = -1;
        if (m < a.M) {
            const int n = m / HWq;
            const int rem = m - n * HWq;
            const int qy = rem / a.Wq;
            const int qx = rem - qy * a.Wq;
            const int oy = qy * a.omy + ph.po_y;
            const int ox = qx * a.omx + ph.po_x;
            if (oy < a.Ho && ox < a.Wo) o = (n * a.Ho + oy) * a.Wo + ox;
        }
        s_orow[r] = o;
    }

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin_p) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.w) + ph.w_off * 2), 0, (int)((long long)a.cout_p * ph.kp * 2), 0x00020000);

    // ---- DMA coordinates of this lane: tile row 32*pass + 8*wave + (lane >> 3), 16-byte slot lane & 7 of that row, which
    // holds K chunk (lane & 7) ^ ((row >> 1) & 7); (row >> 1) & 7 = (4*wave + (lane >> 4)) & 7 for every pass
    const int rsub = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    int a_iy0[PA], a_ix0[PA];
    unsigned a_base[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + PR * p + 8 * wave + rsub;
        if (m < a.M) {
            const int n = m / HWq;
            const int rem = m - n * HWq;
            const int qy = rem / a.Wq;
            const int qx = rem - qy * a.Wq;
            a_iy0[p] = qy * a.sy;
            a_ix0[p] = qx * a.sx;
            a_base[p] = (unsigned)((n * a.H + a_iy0[p]) * a.W + a_ix0[p]) * (unsigned)a.x_cs * 2u;
        } else {
            a_iy0[p] = -0x4000;
            a_ix0[p] = -0x4000;
            a_base[p] = 0;
        }
    }
    unsigned b_off[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int gn = n0 + PR * p + 8 * wave + rsub;
        b_off[p] = gn < a.cout_p ? ((unsigned)gn * (unsigned)ph.kp + (unsigned)(kfirst * kBKH + kc * 8)) * 2u : kOobH;
    }
    const int nsteps = min(a.steps_per_split, ph.kp / kBKH - kfirst);

    __syncthreads();   // s_taps / s_orow visible

    const int dq = kBKH / a.cin_p, dc = kBKH % a.cin_p;
    int g_tap = (kfirst * kBKH + kc * 8) / a.cin_p;
    int g_c = (kfirst * kBKH + kc * 8) % a.cin_p;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma = [&](int step, int buf) {
        const bool tap_ok = g_tap < ph.ntaps;
        const int2 tv = *reinterpret_cast<const int2*>(s_taps + 2 * (tap_ok ? g_tap : 0));
        const int dy = (int)(short)(tv.x & 0xffff);
        const int dx = tv.x >> 16;
        const unsigned delta = (unsigned)(tv.y + g_c * 2);
        char* Ab = smem + buf * STAGE + (8 * wave) * 128;
        char* Bb = smem + buf * STAGE + BM * 128 + (8 * wave) * 128;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = tap_ok & ((unsigned)(a_iy0[p] + dy) < (unsigned)a.H) & ((unsigned)(a_ix0[p] + dx) < (unsigned)a.W);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(Ab + p * PR * 128), 16, (int)(ok ? a_base[p] + delta : kOobH), 0, 0, 0);
        }
        const bool step_ok = step < nsteps;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Bb + p * PR * 128), 16, (int)(step_ok ? b_off[p] : kOobH), 0, 0, 0);
            b_off[p] += (b_off[p] == kOobH) ? 0u : kBKH * 2u;
        }
        g_c += dc;
        g_tap += dq;
        if (g_c >= a.cin_p) { g_c -= a.cin_p; ++g_tap; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // a wave that owns ONE 32x32 tile would chain its four MFMAs of a K-step through one accumulator: alternate two
    constexpr bool kDual = (TM * TN == 1);
    f32x16 acc_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;

    // fragment addresses: lane reads row (lane & 31) of a 32-row tile, K chunk 2*ksub + (lane >> 5), stored at slot chunk ^ swz
    const int frow = lane & 31;
    const int fswz = (lane >> 1) & 7;
    const int fhi = lane >> 5;
    const int a_row_off = (wm * TM * 32 + frow) * 128;
    const int b_row_off = BM * 128 + (wn * TN * 32 + frow) * 128;

    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8 af[2][TM], bfr[2][TN];
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (!(W2L_CONVB_DBG & 2) && step + 1 < nsteps) dma(step + 1, buf ^ 1);
        const char* Sb = smem + buf * STAGE;
        // two fragment sets: the reads of K-substep ks+1 are requested BEFORE the MFMAs of substep ks are issued (the scheduling
        // fences keep them there: left alone, the compiler folds the two sets back into one and every substep waits a full LDS
        // latency in front of its MFMAs)
        auto frags = [&](int ks, int set) {
            const int slot = ((2 * ks + fhi) ^ fswz) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const bf16x8*>(Sb + a_row_off + i * 32 * 128 + slot);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[set][j] = *reinterpret_cast<const bf16x8*>(Sb + b_row_off + j * 32 * 128 + slot);
        };
        if (!(W2L_CONVB_DBG & 4) || step == 0) frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3 && (!(W2L_CONVB_DBG & 4) || step == 0)) frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#if W2L_CONVB_DBG & 1
                    asm volatile("" : "+v"(acc[i][j]) : "v"(af[ks & 1][i]), "v"(bfr[ks & 1][j]));
#else
                    if (kDual && (ks & 1))
                        acc_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc_odd, 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next step's tiles have landed (this wave's DMAs)
        if (!(W2L_CONVB_DBG & 8)) __syncthreads();         // ... every wave's, and nobody still reads this step's buffer
    }
    if (kDual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
    }

    // ---- epilogue, per wave: accumulators -> private fp32 LDS tile [32][LDCW] -> rows of 8 channels
    float* Cw = reinterpret_cast<float*>(smem) + wave * (32 * LDCW);
    constexpr int CGW = TN * 4;          // 8-channel groups per row of the wave tile
    constexpr int RPPW = 64 / CGW;       // rows per pass
    const int cg = lane % CGW;
    const int rl = lane / CGW;
    const int ch = n0 + wn * TN * 32 + cg * 8;       // first of this lane's 8 output channels
    const int cout8 = (a.cout + 7) & ~7;
    const bool ch_ok = ch < cout8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool v = ch + e < a.cout;
        sc[e] = v ? (a.scale ? a.scale[ch + e] : 1.f) : 0.f;
        sh[e] = v ? (a.shift ? a.shift[ch + e] : 0.f) : 0.f;
    }
    float st0[8], st1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st0[e] = 0.f; st1[e] = 0.f; }
    const long long npix = (long long)a.N * a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + cout8) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + cout8) * 2) : 0, 0x00020000);
    // Straight-line per tile: the residual rows of ALL passes are requested before the first is consumed, the activation is an
    // expression (slope on the negative side; the sigmoid of the output block is a uniform branch after it), split-K, residual
    // and statistics are wave-uniform switches around whole loops.  (The first version asked for one residual row per pass and
    // waited for it at once, behind per-element activation branches: 8-16 exposed memory latencies per workgroup - 28 % of the
    // kernel on a 36-step layer, 40 % on a 9-step one; ablation in EXPERIMENTS.md.)
    constexpr int NPS = 32 / RPPW;
    const bool split = a.ksplit > 1;
    const bool has_res = a.res != nullptr;
    const bool want_stats = a.stats != nullptr;
    const bool bwd_sums = want_stats && a.bz != nullptr;      // BatchNorm-backward sums instead of forward statistics
    const bool have_by = a.by != nullptr;
    const bool mask_only = a.bmask_only != 0;     // (then bwd_sums is false: a.stats == NULL)
    float bmu[8], brs[8], bsc[8], bsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bmu[e] = 0.f; brs[e] = 0.f; bsc[e] = 0.f; bsh[e] = 0.f; }
    if (bwd_sums && ch_ok) {      // the per-channel vectors hold round8(cout) entries (header contract), not cout_p
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bmu[e] = a.bmean[ch + e];
            brs[e] = a.brstd[ch + e];
            if (!have_by) { bsc[e] = a.bscale[ch + e]; bsh[e] = a.bshift[ch + e]; }
        }
    }
    const __amdgpu_buffer_rsrc_t rbz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(bwd_sums ? a.bz : a.y), 0, bwd_sums ? (int)(((npix - 1) * a.bz_cs + cout8) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rby = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>((bwd_sums || mask_only) && have_by ? a.by : a.y), 0,
        (bwd_sums || mask_only) && have_by ? (int)(((npix - 1) * a.by_cs + cout8) * 2) : 0, 0x00020000);
    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    const bool is_sigmoid = a.act == W2L_ACT_SIGMOID;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDCW + j * 32 + (lane & 31)] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        int opix[NPS];
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) opix[ps] = s_orow[(wm * TM + i) * 32 + ps * RPPW + rl];
        if (split) {
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = ps * RPPW + rl;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cw + row * LDCW + cg * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cw + row * LDCW + cg * 8 + 4);
                if (ch_ok && opix[ps] >= 0 && ch < a.cout_p) {
                    float* wsz = a.ws + ((long long)blockIdx.z * npix + opix[ps]) * a.cout_p + ch;
                    *reinterpret_cast<f32x4*>(wsz) = c0;
                    *reinterpret_cast<f32x4*>(wsz + 4) = c1;
                }
            }
        } else {
            u32x4 rv[NPS];
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) rv[ps] = u32x4{0u, 0u, 0u, 0u};
            if (has_res) {
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    const bool ok = ch_ok & (opix[ps] >= 0);
                    rv[ps] = __builtin_amdgcn_raw_buffer_load_b128(
                        rr, (int)(ok ? ((unsigned)opix[ps] * (unsigned)a.res_cs + (unsigned)ch) * 2u : kOobH), 0, 0);
                }
            }
            u32x4 zr[NPS], yr[NPS];
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) { zr[ps] = u32x4{0u, 0u, 0u, 0u}; yr[ps] = u32x4{0u, 0u, 0u, 0u}; }
            if (mask_only) {
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    const bool ok = ch_ok & (opix[ps] >= 0);
                    yr[ps] = __builtin_amdgcn_raw_buffer_load_b128(
                        rby, (int)(ok ? ((unsigned)opix[ps] * (unsigned)a.by_cs + (unsigned)ch) * 2u : kOobH), 0, 0);
                }
            }
            if (bwd_sums) {
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    const bool ok = ch_ok & (opix[ps] >= 0);
                    zr[ps] = __builtin_amdgcn_raw_buffer_load_b128(
                        rbz, (int)(ok ? ((unsigned)opix[ps] * (unsigned)a.bz_cs + (unsigned)ch) * 2u : kOobH), 0, 0);
                    if (have_by)
                        yr[ps] = __builtin_amdgcn_raw_buffer_load_b128(
                            rby, (int)(ok ? ((unsigned)opix[ps] * (unsigned)a.by_cs + (unsigned)ch) * 2u : kOobH), 0, 0);
                }
            }
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = ps * RPPW + rl;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cw + row * LDCW + cg * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cw + row * LDCW + cg * 8 + 4);
                const bool ok = ch_ok & (opix[ps] >= 0);
                const bf16x8 rb = __builtin_bit_cast(bf16x8, rv[ps]);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? c0[e] : c1[e - 4]) * sc[e] + sh[e] + (float)rb[e];
                if (mask_only) {
                    // dz of the block in front: the ROUNDED dy times act'(its output), rounded again by the store - bit for bit what
                    // w2l_act_bwd_bf16 computes from the stored dy
                    const bf16x8 yb = __builtin_bit_cast(bf16x8, yr[ps]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] * ((float)yb[e] > 0.f ? 1.f : a.bneg);
                }
                if (bwd_sums) {
                    const float m = ok ? 1.f : 0.f;
                    const bf16x8 zb = __builtin_bit_cast(bf16x8, zr[ps]), yb = __builtin_bit_cast(bf16x8, yr[ps]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float vr = (float)(__bf16)v[e];                    // the dy that is stored
                        const float zf = (float)zb[e];
                        const float yy = have_by ? (float)yb[e] : zf * bsc[e] + bsh[e];
                        const float g = vr * (yy > 0.f ? 1.f : a.bneg) * m;
                        st0[e] += g;
                        st1[e] += g * ((zf - bmu[e]) * brs[e]);
                        // ReLU: what is stored IS g (the rounded dy or zero): the block's BatchNorm-backward pass then reads neither
                        // its output (the mask) again nor writes g for the residual path
                        if (a.bstore_g) v[e] = yy > 0.f ? v[e] : 0.f;
                    }
                } else if (want_stats) {
                    const float m = ok ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {   // of the ROUNDED value: the z that is stored is the z affine_act / bn_train_bwd
                        const float vr = (float)(__bf16)v[e];       // normalise and the split-K route reduces (one definition of
                        const float vm = vr * m;                    // mean / var whatever the launch shape)
                        st0[e] += vm;
                        st1[e] += vm * vr;
                    }
                }
                bf16x8 o;
                if (is_sigmoid) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)((ch + e < a.cout) ? 1.0f / (1.0f + expf(-v[e])) : 0.f);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        o[e] = (__bf16)((ch + e < a.cout) ? act_leaky(v[e], neg_slope) : 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry,
                                                       (int)(ok ? ((unsigned)opix[ps] * (unsigned)a.y_cs + (unsigned)ch) * 2u : kOobH), 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (a.stats) {
        // the lanes rl = 0 .. RPPW-1 that share a column group meet by xor-shuffles over the row-lane bits (lane = rl * CGW + cg);
        // lane rl == 0 then holds this wave's column sums over its TM * 32 rows: one partial row per (phase, M tile, wave row)
#pragma unroll
        for (int m = CGW; m < 64; m <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                st0[e] += __shfl_xor(st0[e], m);
                st1[e] += __shfl_xor(st1[e], m);
            }
        if (rl == 0 && ch < a.cout_p) {
            float* dst = a.stats + ((long long)((blockIdx.y * a.tiles_m + tile_m) * WM + wm)) * 2 * a.cout_p + ch;
            *reinterpret_cast<f32x4*>(dst) = f32x4{st0[0], st0[1], st0[2], st0[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{st0[4], st0[5], st0[6], st0[7]};
            *reinterpret_cast<f32x4*>(dst + a.cout_p) = f32x4{st1[0], st1[1], st1[2], st1[3]};
            *reinterpret_cast<f32x4*>(dst + a.cout_p + 4) = f32x4{st1[4], st1[5], st1[6], st1[7]};
        }
    }
}

// ---- split-K reduce: y = act( sum_z ws[z] * scale + shift (+ res) ), one thread per (output pixel, 8-channel group)
struct ReduceBArgs {
    const float* ws;
    void* y;
    const void* res;
    const float* scale;
    const float* shift;
    long long npix;
    int ksplit, cout, cout_p, y_cs, res_cs, act;
};

__global__ void splitk_reduce_bf16_kernel(const ReduceBArgs a) {
    const int groups = (a.cout + 7) >> 3;
    const long long total = a.npix * groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / groups;
        const int c = (int)(i - pix * groups) * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int z = 0; z < a.ksplit; ++z) {
            const float* p = a.ws + ((long long)z * a.npix + pix) * a.cout_p + c;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(p), p1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += p0[e]; v[4 + e] += p1[e]; }
        }
        bf16x8 rb;
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e] = (__bf16)0.f;
        if (a.res) rb = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(a.res) + pix * a.res_cs + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = c + e < a.cout;
            const float s = ok ? (a.scale ? a.scale[c + e] : 1.f) : 0.f;
            const float h = ok ? (a.shift ? a.shift[c + e] : 0.f) : 0.f;
            o[e] = (__bf16)(ok ? actb(a.act, v[e] * s + h + (float)rb[e]) : 0.f);
        }
        *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(a.y) + pix * a.y_cs + c) = o;
    }
}

// ---- weight packing: fp32 torch layout -> bf16 per-phase [cout_p][kp] slabs, K = (tap, c) with c fastest
struct PackBArgs {
    const float* w;   // OIHW (conv) or IOHW (transposed)
    __bf16* out;
    const int* tapk;  // (ky & 0xffff) | (kx << 16) per tap-table entry
    int transposed, cin, cout, kh, kw, cin_p, cout_p;
    int nphase;
    ConvPhase ph[kMaxPhases];
};

// A workgroup packs TILES of the weight tensor: 64 input channels x `rows` slab rows (8 for kernels up to 3x3, 2 above) x all
// kh*kw taps.  The tile's source values are `rows` (conv: OIHW, one run per row) or 64 (transposed / data-gradient view: IOHW, one
// run per channel) CONTIGUOUS runs of floats: they are loaded coalesced into LDS as [channel][row][tap], then every (row, tap of a
// phase, 8-channel group) leaves as ONE 16-byte store, c fastest - the slab order the kernels read.  Index arithmetic is
// multiply-shift division by small run-time constants (exact for the ranges used: x < 2^13, divisor <= 98).
// (First version: one thread per output element, two 64-bit divisions and a kh*kw-strided 4-byte gather each, 2-byte stores:
//  0.22 - 0.26 ms per launch for 36 M parameters, twice per wav2lip_train step and four times per hq step.)  Same values, same bytes.
constexpr int kPackCT = 64;                         // channels per tile
constexpr int kPackLdsFloats = kPackCT * (2 * 49 + 1);

__device__ __forceinline__ unsigned pack_magic(unsigned d) { return ((1u << 20) + d - 1) / d; }
__device__ __forceinline__ unsigned pack_div(unsigned x, unsigned magic) { return (x * magic) >> 20; }   // x * d < 2^20

__device__ __forceinline__ void pack_tiles_bf16(const PackBArgs& a, int first_tile, int tile_stride, float* T) {
    const int khw = a.kh * a.kw;
    const int rows = khw <= 9 ? 8 : 2;
    const int crow = rows * khw + 1;                // LDS floats per channel (+1: the 8-channel groups of a store land on different banks)
    const int ctiles = (a.cin_p + kPackCT - 1) / kPackCT;
    const int ntiles = (a.cout_p / rows) * ctiles;  // cout_p is a multiple of 32
    const unsigned m_khw = pack_magic((unsigned)khw), m_rk = pack_magic((unsigned)(rows * khw));
    const int t = threadIdx.x;
    for (int tile = first_tile; tile < ntiles; tile += tile_stride) {
        const int nt = tile / ctiles, ct = tile - nt * ctiles;
        const int n0 = nt * rows, c0 = ct * kPackCT;
        __syncthreads();                            // the previous tile's stores have read T
        if (!a.transposed) {
            // row n: the run w[n][c0 .. c0+63][*][*]
            const int run = kPackCT * khw;
            for (int r = 0; r < rows; ++r) {
                const int n = n0 + r;
                const long long base = ((long long)n * a.cin + c0) * khw;
                for (int e = t; e < run; e += blockDim.x) {
                    const int c = (int)pack_div((unsigned)e, m_khw);
                    const int tap = e - c * khw;
                    T[c * crow + r * khw + tap] = (n < a.cout && c0 + c < a.cin) ? a.w[base + e] : 0.f;
                }
            }
        } else {
            // channel c: the run w[c][n0 .. n0+rows-1][*][*]
            const int run = rows * khw;
            for (int i = t; i < kPackCT * run; i += blockDim.x) {
                const int c = (int)pack_div((unsigned)i, m_rk);
                const int e = i - c * run;
                const int r = (int)pack_div((unsigned)e, m_khw);
                T[c * crow + e] = (c0 + c < a.cin && n0 + r < a.cout) ? a.w[((long long)(c0 + c) * a.cout + n0) * khw + e] : 0.f;
            }
        }
        __syncthreads();
        for (int p = 0; p < a.nphase; ++p) {
            const ConvPhase& ph = a.ph[p];
            const unsigned m_nt = pack_magic((unsigned)(ph.ntaps > 0 ? ph.ntaps : 1));
            const int items = rows * ph.ntaps * 8;
            for (int it = t; it < items; it += blockDim.x) {
                const int g = it & 7, rt = it >> 3;
                const int r = (int)pack_div((unsigned)rt, m_nt);
                const int tt = rt - r * ph.ntaps;
                const int c = c0 + g * 8;
                if (c >= a.cin_p) continue;
                const int tk = a.tapk[ph.tap_off + tt];
                const int tl = (tk & 0xffff) * a.kw + (tk >> 16);
                const float* src = T + (g * 8) * crow + r * khw + tl;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)src[j * crow];
                *reinterpret_cast<bf16x8*>(a.out + ph.w_off + (long long)(n0 + r) * ph.kp + (long long)tt * a.cin_p + c) = o;
            }
            // K entries beyond the last tap (K is padded to the K-step): zeros, written by the first channel tile of the row block
            const int tail0 = ph.ntaps * a.cin_p, tailg = (ph.kp - tail0) >> 3;
            if (ct == 0 && tailg > 0) {
                const unsigned m_tg = pack_magic((unsigned)tailg);
                bf16x8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
                for (int it = t; it < rows * tailg; it += blockDim.x) {
                    const int r = (int)pack_div((unsigned)it, m_tg);
                    *reinterpret_cast<bf16x8*>(a.out + ph.w_off + (long long)(n0 + r) * ph.kp + tail0 + (it - r * tailg) * 8) = z;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const PackBArgs a) {
    __shared__ float T[kPackLdsFloats];
    pack_tiles_bf16(a, blockIdx.x, gridDim.x, T);
}

// Every layer of a train graph in ONE launch (w2l_convb_update_many): an optimiser step invalidates the bf16 slabs of all ~120
// layers (forward + data-gradient variants), and 240 launches of 5-10 us each were 1.6-2 ms of a 28 ms wav2lip_train step.
// Workgroup b serves table entry blk[b].x as its blk[b].z-th of blk[b].w workgroups (tiles z, z + w, ...); same arithmetic, same
// bytes as the per-layer kernel.
__global__ __launch_bounds__(256) void pack_weights_bf16_many_kernel(const PackBArgs* __restrict__ tab, const int4* __restrict__ blk) {
    __shared__ float T[kPackLdsFloats];
    const int4 b = blk[blockIdx.x];
    pack_tiles_bf16(tab[b.x], b.z, b.w, T);
}

// ---- host side ------------------------------------------------------------------------------
struct BTile {
    int bm, bn, wm;
    void (*kernel)(const ConvBArgs);
    int lds, threads;
};
#define W2L_BTILE(BM, BN, WM, WN) { BM, BN, WM, conv_bf16s_kernel<BM, BN, WM, WN>, convb_lds_bytes<BM, BN>(), WM * WN * 64 }
static const BTile kBTiles[] = {
    W2L_BTILE(128, 128, 2, 2),   // 0
    W2L_BTILE(128, 64, 2, 2),    // 1
    W2L_BTILE(64, 128, 2, 2),    // 2
    W2L_BTILE(64, 64, 2, 2),     // 3
    W2L_BTILE(128, 32, 4, 1),    // 4
    W2L_BTILE(256, 256, 2, 4),   // 5: eight waves with 128x64 wave tiles (6 fragment reads per 8 MFMAs), one workgroup per CU
};
constexpr int kNumBTiles = sizeof(kBTiles) / sizeof(kBTiles[0]);

struct BVariant {
    int nphase = 0;
    int sy = 1, sx = 1, omy = 1, omx = 1;
    bool q_is_out = true;
    ConvPhase ph[kMaxPhases];
    int* taps_dev = nullptr;    // [0, ntab): (dy, dx); [ntab, 2 ntab): (ky, kx) for the packer
    std::vector<int> taps_host; // the (dy, dx) half on the host: the shape rules of the special-case kernels read it
    int ntab = 0;
    __bf16* w_dev = nullptr;
    long long w_elems = 0;
    bool built = false;
};

float* conv_workspace(hipStream_t stream, size_t bytes);   // conv_igemm.hip: grow-only split-K scratch, one per stream
// conv_box_bf16.hip: 3x3 / stride 1 / 64 -> 64 channels with the input box and the weight set resident in LDS
// conv_stem_bf16.hip: 7x7 / stride 1 stems with 8 or 16 channels per pixel (input box + weight set resident in LDS)
bool stem_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cin_p, int cout, int N, int H, int W, bool has_res);
int stem_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* res, int res_cs, const void* w, int cout_p,
                int kp, const float* scale, const float* shift, const int* taps, int N, int H, int W, int kh, int cin_p, int cout, int act);
bool box64_ok(int nphase, int ntaps, int cin_p, int cout, int cout_p, int N, int H, int W, int Ho, int Wo, int sy, int sx);
int box64_grid(int N, int H, int W);
// conv_tp2b_bf16.hip: 3x3 / stride 2 transposed layers (and the data gradients of 3x3 / stride 2 convs) with all four output phases
// in one workgroup
bool tp2b_ok(int transposed, int kh, int kw, int sh, int sw, int ph, int pw, int nphase, const ConvPhase* phs, const int* taps_host, int cin_p,
             int cout_p, int N, int H, int W, int Ho, int Wo);
int tp2b_npart(int cout_p, int N, int H, int W);
int tp2b_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* res, int res_cs, const void* w, long long w_elems,
                const float* scale, const float* shift, float* stats, const ConvPhase* phs, const int* taps_host, int N, int H, int W,
                int cin_p, int cout, int cout_p, int act);
struct BoxBwd {             // conv_box_bf16.hip: the block whose dy a BWD launch completes
    const void* z;
    const void* y;
    int z_cs, y_cs, store_g;
    float neg;
    const float* mean;
    const float* rstd;
    const float* scale;
    const float* shift;
};
int box64_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* res, int res_cs, const void* w,
                 const float* scale, const float* shift, const int* taps, float* stats, int N, int H, int W, int cout, int act,
                 const BoxBwd* bwd);

}  // namespace w2l

struct w2l_convb {
    w2l_conv_geom g;
    int cin_p, cout_p;
    w2l::BVariant generic;
    w2l::BVariant unit_in;   // transposed, stride 1, 1x1 input: one single-tap phase per output position
    int tile_override = -1;
};

namespace w2l {

// tiles of pack_tiles_bf16 for a layer: (cout_p / rows) x ceil(cin_p / 64), rows = 8 for kernels up to 3x3 else 2
static int pack_tiles(const w2l_convb* c) {
    const int rows = c->g.kh * c->g.kw <= 9 ? 8 : 2;
    return (c->cout_p / rows) * ((c->cin_p + 63) / 64);
}

static PackBArgs pack_args(const w2l_convb* c, const BVariant& v, const float* weight, long long* maxtot_out) {
    PackBArgs pa;
    pa.w = weight; pa.out = v.w_dev; pa.tapk = v.taps_dev + v.ntab;
    pa.transposed = c->g.transposed; pa.cin = c->g.cin; pa.cout = c->g.cout; pa.kh = c->g.kh; pa.kw = c->g.kw;
    pa.cin_p = c->cin_p; pa.cout_p = c->cout_p; pa.nphase = v.nphase;
    long long maxtot = 1;
    for (int i = 0; i < v.nphase; ++i) {
        pa.ph[i] = v.ph[i];
        const long long tot = (long long)c->cout_p * v.ph[i].kp;
        if (tot > maxtot) maxtot = tot;
    }
    for (int i = v.nphase; i < kMaxPhases; ++i) pa.ph[i] = ConvPhase();
    if (maxtot_out) *maxtot_out = maxtot;
    return pa;
}

static int packb(const w2l_convb* c, const BVariant& v, const float* weight, hipStream_t stream) {
    const PackBArgs pa = pack_args(c, v, weight, nullptr);
    int blocks = pack_tiles(c);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

static int buildb(w2l_convb* c, BVariant& v, bool unit_input) {
    const w2l_conv_geom& g = c->g;
    std::vector<int> tapd, tapk;
    long long woff = 0;
    v.nphase = 0;
    auto add_phase = [&](int poy, int pox) -> ConvPhase& {
        ConvPhase& p = v.ph[v.nphase++];
        p.ntaps = 0; p.po_y = poy; p.po_x = pox; p.tap_off = (int)tapd.size(); p.pad_ = 0;
        return p;
    };
    auto add_tap = [&](ConvPhase& p, int dy, int dx, int ky, int kx) {
        tapd.push_back((int)(((unsigned)dy & 0xffffu) | ((unsigned)dx << 16)));
        tapk.push_back((int)(((unsigned)ky & 0xffffu) | ((unsigned)kx << 16)));
        ++p.ntaps;
    };
    auto close_phase = [&](ConvPhase& p) {
        p.kp = round_up(p.ntaps * c->cin_p, kBKH);
        p.w_off = woff;
        woff += (long long)c->cout_p * p.kp;
    };
    if (!g.transposed) {
        v.sy = g.sh; v.sx = g.sw; v.omy = 1; v.omx = 1; v.q_is_out = true;
        ConvPhase& p = add_phase(0, 0);
        for (int ky = 0; ky < g.kh; ++ky)
            for (int kx = 0; kx < g.kw; ++kx) add_tap(p, ky - g.ph, kx - g.pw, ky, kx);
        close_phase(p);
    } else if (unit_input) {
        v.sy = 1; v.sx = 1; v.omy = 1; v.omx = 1; v.q_is_out = false;
        const int Ho = g.kh - 2 * g.ph + g.oph, Wo = g.kw - 2 * g.pw + g.opw;
        for (int ky = 0; ky < g.kh; ++ky)
            for (int kx = 0; kx < g.kw; ++kx) {
                const int oy = ky - g.ph, ox = kx - g.pw;
                if (oy < 0 || ox < 0 || oy >= Ho || ox >= Wo) continue;
                if (v.nphase >= kMaxPhases) { set_error("bf16 convT unit-input: too many phases"); return W2L_ERR_ARG; }
                ConvPhase& p = add_phase(oy, ox);
                add_tap(p, 0, 0, ky, kx);
                close_phase(p);
            }
    } else {
        // out oy = iy*s - p + ky.  Phase py = oy mod s: taps with ky == (py+p) mod s, iy = q + (py+p-ky)/s
        v.sy = 1; v.sx = 1; v.omy = g.sh; v.omx = g.sw; v.q_is_out = false;
        if (g.sh * g.sw > kMaxPhases) { set_error("bf16 convT stride %dx%d unsupported", g.sh, g.sw); return W2L_ERR_ARG; }
        for (int py = 0; py < g.sh; ++py)
            for (int px = 0; px < g.sw; ++px) {
                ConvPhase& p = add_phase(py, px);
                for (int ky = 0; ky < g.kh; ++ky) {
                    if ((py + g.ph - ky) % g.sh != 0) continue;
                    for (int kx = 0; kx < g.kw; ++kx) {
                        if ((px + g.pw - kx) % g.sw != 0) continue;
                        add_tap(p, (py + g.ph - ky) / g.sh, (px + g.pw - kx) / g.sw, ky, kx);
                    }
                }
                close_phase(p);
            }
    }
    for (int i = 0; i < v.nphase; ++i)
        if (v.ph[i].ntaps > 64) { set_error("bf16 conv: too many taps"); return W2L_ERR_ARG; }
    const int ntab = (int)tapd.size();
    v.ntab = ntab;
    v.taps_host = tapd;
    v.w_elems = woff;
    W2L_HIP_CHECK(hipMalloc(&v.taps_dev, sizeof(int) * 2 * (ntab > 0 ? ntab : 1)));
    W2L_HIP_CHECK(hipMalloc(&v.w_dev, sizeof(__bf16) * (woff > 0 ? woff : 1)));
    if (ntab > 0) {
        W2L_HIP_CHECK(hipMemcpy(v.taps_dev, tapd.data(), sizeof(int) * ntab, hipMemcpyHostToDevice));
        W2L_HIP_CHECK(hipMemcpy(v.taps_dev + ntab, tapk.data(), sizeof(int) * ntab, hipMemcpyHostToDevice));
    }
    v.built = true;
    return W2L_OK;
}

static void freeb(BVariant& v) {
    if (v.taps_dev) (void)hipFree(v.taps_dev);
    if (v.w_dev) (void)hipFree(v.w_dev);
    v.taps_dev = nullptr; v.w_dev = nullptr; v.built = false;
}

static int convb_init_attrs() {
    static std::mutex m;
    static bool done = false;
    std::lock_guard<std::mutex> lock(m);
    if (done) return W2L_OK;
    for (int i = 0; i < kNumBTiles; ++i)
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kBTiles[i].kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kBTiles[i].lds));
    done = true;
    return W2L_OK;
}

static int max_stepsb(const BVariant& v) {
    int m = 0;
    for (int i = 0; i < v.nphase; ++i) m = v.ph[i].kp / kBKH > m ? v.ph[i].kp / kBKH : m;
    return m;
}

// launch configuration = a function of the shape only (bit-reproducible).  Rules read off a same-box sweep of every tile on the
// generator's training shapes (profiles/r03/d_bf16_conv_sweep.txt): N-tile = the layer's couts up to 128 (a 32-cout layer on a
// 128-wide tile multiplies 75 % padding), M-tile 128 unless that leaves the chip under-filled, split-K for the deep
// small-spatial layers whose grid cannot reach one workgroup per CU otherwise
static void pickb(const w2l_convb* c, const BVariant& v, int M, int* tile, int* ksplit) {
    const int bn = c->cout_p <= 32 ? 32 : (c->cout_p <= 64 ? 64 : 128);
    int bm = 128;
    const long long blocks128 = (long long)ceil_div(M, 128) * ceil_div(c->cout_p, bn) * v.nphase;
    if (bn != 32 && blocks128 < 384) bm = 64;
    int ti = 0;
    for (int i = 0; i < kNumBTiles; ++i)
        if (kBTiles[i].bm == bm && kBTiles[i].bn == bn) ti = i;
    // the eight-wave 256x256 tile (128x64 wave tiles: 6 fragment reads per 8 MFMAs instead of 8): only where its N tiles are full
    // (cout a multiple of 256) and >= 512 of them fill the chip twice - 256 -> 256 at 24x24 x 320 frames 0.227 -> 0.209 ms, 384-
    // and 512-channel layers lose (half-empty N tile / too few M tiles): profiles/r06/i_bf16_sweep_all_tiles.log.  W2L_CONVB_T256=0: off
    static const bool t256_on = [] { const char* e = getenv("W2L_CONVB_T256"); return e ? atoi(e) != 0 : true; }();
    if (t256_on && bm == 128 && c->cout_p % 256 == 0 && (long long)ceil_div(M, 256) * (c->cout_p / 256) * v.nphase >= 512) ti = 5;
    const long long blocks = (long long)ceil_div(M, kBTiles[ti].bm) * ceil_div(c->cout_p, kBTiles[ti].bn) * v.nphase;
    const int steps = max_stepsb(v);
    int ks = 1;
    while (ks < 16 && blocks * ks * 2 <= 512 && steps / (ks * 2) >= 4) ks *= 2;
    *tile = ti;
    *ksplit = ks;
}

}  // namespace w2l

using namespace w2l;

extern "C" {

int w2l_convb_create(const w2l_conv_geom* g, const float* weight, void* stream, w2l_convb_t** out) {
    W2L_REQUIRE(g && weight && out, "NULL argument");
    W2L_REQUIRE(g->cin >= 1 && g->cout >= 1 && g->kh >= 1 && g->kw >= 1 && g->kh * g->kw <= kMaxTaps, "bad geometry");
    W2L_REQUIRE(g->sh >= 1 && g->sw >= 1 && g->ph >= 0 && g->pw >= 0, "bad stride/pad");
    W2L_REQUIRE(g->act >= W2L_ACT_NONE && g->act <= W2L_ACT_LEAKY, "bad act %d", g->act);
    W2L_REQUIRE(g->transposed || (g->oph == 0 && g->opw == 0), "output_padding on a plain conv");
    if (convb_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    w2l_convb* c = new (std::nothrow) w2l_convb();
    if (!c) { set_error("out of host memory"); return W2L_ERR_NOMEM; }
    c->g = *g;
    c->cin_p = round_up(g->cin, 8);
    c->cout_p = round_up(g->cout, 32);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = buildb(c, c->generic, false);
    if (rc == W2L_OK && g->transposed && g->sh == 1 && g->sw == 1 && g->kh * g->kw <= kMaxPhases) rc = buildb(c, c->unit_in, true);
    if (rc == W2L_OK) rc = w2l_convb_update(c, weight, stream);
    if (rc == W2L_OK && hipStreamSynchronize(s) != hipSuccess) { set_error("sync after bf16 weight packing failed"); rc = W2L_ERR_HIP; }
    if (rc != W2L_OK) { w2l_convb_destroy(c); return rc; }
    *out = c;
    return W2L_OK;
}

int w2l_convb_update(w2l_convb_t* c, const float* weight, void* stream) {
    W2L_REQUIRE(c && weight, "NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = packb(c, c->generic, weight, s);
    if (rc == W2L_OK && c->unit_in.built) rc = packb(c, c->unit_in, weight, s);
    return rc;
}

// ---- w2l_convb_update_many: the (entry, workgroup) tables of a set of layers live on the device and are re-used for as long as
// the same (handle, weight pointer) list comes back - which is every optimiser step of a training run
namespace {
struct PackGroup {
    unsigned long long key = 0;
    int n = 0;
    std::vector<const void*> ids;   // handle, weight, handle, weight, ...
    w2l::PackBArgs* tab = nullptr;
    int4* blk = nullptr;
    int nblocks = 0;
};
std::mutex g_packgroup_mutex;
std::vector<PackGroup> g_packgroups;
constexpr size_t kMaxPackGroups = 16;
}  // namespace

int w2l_convb_update_many(int n, w2l_convb_t* const* handles, const float* const* weights, void* stream) {
    W2L_REQUIRE(n >= 1 && n <= 4096 && handles && weights, "convb_update_many: 1..4096 layers");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<const void*> ids(2 * (size_t)n);
    unsigned long long key = 1469598103934665603ull;
    for (int i = 0; i < n; ++i) {
        W2L_REQUIRE(handles[i] && weights[i], "convb_update_many: NULL handle or weight at %d", i);
        ids[2 * i] = handles[i];
        ids[2 * i + 1] = weights[i];
        key = (key ^ (unsigned long long)reinterpret_cast<uintptr_t>(handles[i])) * 1099511628211ull;
        key = (key ^ (unsigned long long)reinterpret_cast<uintptr_t>(weights[i])) * 1099511628211ull;
    }
    std::lock_guard<std::mutex> lock(g_packgroup_mutex);
    PackGroup* grp = nullptr;
    for (PackGroup& g : g_packgroups)
        if (g.key == key && g.n == n && g.ids == ids) { grp = &g; break; }
    if (!grp) {
        std::vector<PackBArgs> tab;
        std::vector<int4> blk;
        for (int i = 0; i < n; ++i) {
            const w2l_convb* c = handles[i];
            const BVariant* vars[2] = {&c->generic, c->unit_in.built ? &c->unit_in : nullptr};
            for (const BVariant* v : vars) {
                if (!v) continue;
                const int e = (int)tab.size();
                tab.push_back(pack_args(c, *v, weights[i], nullptr));
                int nb = (pack_tiles(c) + 1) / 2;                 // two tiles per workgroup, at most 512 workgroups per layer
                if (nb > 512) nb = 512;
                if (nb < 1) nb = 1;
                for (int j = 0; j < nb; ++j) blk.push_back(make_int4(e, 0, j, nb));
            }
        }
        if (g_packgroups.size() >= kMaxPackGroups) {              // evict the oldest table (its launches may still be queued)
            W2L_HIP_CHECK(hipDeviceSynchronize());
            (void)hipFree(g_packgroups.front().tab);
            (void)hipFree(g_packgroups.front().blk);
            g_packgroups.erase(g_packgroups.begin());
        }
        PackGroup g;
        g.key = key; g.n = n; g.ids = ids; g.nblocks = (int)blk.size();
        if (hipMalloc(&g.tab, tab.size() * sizeof(PackBArgs)) != hipSuccess || hipMalloc(&g.blk, blk.size() * sizeof(int4)) != hipSuccess) {
            if (g.tab) (void)hipFree(g.tab);
            set_error("convb_update_many: hipMalloc of the pack tables failed");
            return W2L_ERR_NOMEM;
        }
        W2L_HIP_CHECK(hipMemcpy(g.tab, tab.data(), tab.size() * sizeof(PackBArgs), hipMemcpyHostToDevice));
        W2L_HIP_CHECK(hipMemcpy(g.blk, blk.data(), blk.size() * sizeof(int4), hipMemcpyHostToDevice));
        g_packgroups.push_back(std::move(g));
        grp = &g_packgroups.back();
    }
    hipLaunchKernelGGL(pack_weights_bf16_many_kernel, dim3((unsigned)grp->nblocks), dim3(256), 0, s, grp->tab, grp->blk);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// a destroyed handle must not be found by a cached table again (its address may be re-used by a new layer)
static void packgroups_forget(const w2l_convb* c) {
    std::lock_guard<std::mutex> lock(g_packgroup_mutex);
    for (size_t i = 0; i < g_packgroups.size();) {
        bool hit = false;
        for (size_t k = 0; k < g_packgroups[i].ids.size(); k += 2) hit |= g_packgroups[i].ids[k] == c;
        if (hit) {
            (void)hipDeviceSynchronize();
            (void)hipFree(g_packgroups[i].tab);
            (void)hipFree(g_packgroups[i].blk);
            g_packgroups.erase(g_packgroups.begin() + (long)i);
        } else {
            ++i;
        }
    }
}

int w2l_convb_destroy(w2l_convb_t* c) {
    if (c) packgroups_forget(c);
    if (!c) return W2L_OK;
    freeb(c->generic);
    freeb(c->unit_in);
    delete c;
    return W2L_OK;
}

int w2l_convb_set_tile(w2l_convb_t* c, int tile) {
    W2L_REQUIRE(c && tile >= -1 && tile < kNumBTiles, "bad tile id %d", tile);
    c->tile_override = tile;
    return W2L_OK;
}
int w2l_convb_num_tiles(void) { return kNumBTiles; }

}  // extern "C"

// stats_out != NULL: BatchNorm statistics wanted.  If this launch can carry them in its epilogue (no split-K) *stats_out receives
// the partial buffer [*npart_out][2][cout_p] (stream scratch), else NULL and the caller runs the column reduction over y.
struct BnBwdOperands {      // the BatchNorm block whose dy this launch produces (w2l_convb_forward_bnbwd)
    const void* z;
    const void* y;          // or NULL: ReLU block without residual, mask from z * scale + shift
    int z_cs, y_cs, act;
    const float* mean;
    const float* rstd;
    const float* scale;
    const float* shift;
    int store_g;            // ReLU block: the launch stores the masked gradient (W2L_BNBWD_STORE_MASKED)
};

static int convb_forward_impl(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                              const void* res, int res_cs, const float* scale, const float* shift, int ksplit_force,
                              float** stats_out, int* npart_out, const BnBwdOperands* bb = nullptr) {
    W2L_REQUIRE(c && x && y, "NULL argument");
    W2L_REQUIRE(N >= 1 && H >= 1 && W >= 1, "bad shape N=%d H=%d W=%d", N, H, W);
    const int cout8 = round_up(c->g.cout, 8);
    W2L_REQUIRE(x_cs >= c->cin_p && (x_cs & 7) == 0, "x_cs=%d must be a multiple of 8 and >= %d", x_cs, c->cin_p);
    W2L_REQUIRE(y_cs >= cout8 && (y_cs & 7) == 0, "y_cs=%d must be a multiple of 8 and >= %d", y_cs, cout8);
    W2L_REQUIRE(res == nullptr || (res_cs >= cout8 && (res_cs & 7) == 0), "res_cs=%d must be a multiple of 8 and >= %d", res_cs, cout8);
    W2L_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) == 0,
                "x / y / res must be 16-byte aligned");
    const w2l_conv_geom& g = c->g;
    int Ho, Wo;
    if (w2l_conv_out_hw(&g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(Ho >= 1 && Wo >= 1, "empty output %dx%d", Ho, Wo);
    const long long lim = 1ll << 31;   // buffer descriptors use 32-bit byte offsets with 0x80000000 as "out of range"
    W2L_REQUIRE(((long long)N * H * W * x_cs) * 2 < lim && ((long long)N * Ho * Wo * y_cs) * 2 < lim &&
                    (res == nullptr || ((long long)N * Ho * Wo * res_cs) * 2 < lim),
                "activation buffer larger than 2 GiB: split the batch");
    const bool unit = g.transposed && g.sh == 1 && g.sw == 1 && H == 1 && W == 1 && c->unit_in.built;
    const BVariant& v = unit ? c->unit_in : c->generic;
    ConvBArgs a;
    a.x = x; a.y = y; a.res = res; a.w = v.w_dev; a.scale = scale; a.shift = shift; a.taps = v.taps_dev;
    a.N = N; a.H = H; a.W = W; a.cin_p = c->cin_p; a.x_cs = x_cs;
    a.Ho = Ho; a.Wo = Wo; a.cout = g.cout; a.cout_p = c->cout_p; a.y_cs = y_cs; a.res_cs = res_cs;
    if (unit) { a.Hq = 1; a.Wq = 1; }
    else if (v.q_is_out) { a.Hq = Ho; a.Wq = Wo; }
    else { a.Hq = ceil_div(Ho, v.omy); a.Wq = ceil_div(Wo, v.omx); }
    a.sy = v.sy; a.sx = v.sx; a.omy = v.omy; a.omx = v.omx;
    a.act = g.act;
    const long long M = (long long)N * a.Hq * a.Wq;
    W2L_REQUIRE(M < lim, "tensor too large");
    a.M = (int)M;
    for (int i = 0; i < v.nphase; ++i) a.ph[i] = v.ph[i];
    // W2L_CONVB_BOX=0 (read once): every layer on the implicit GEMM below - A/B switch of the LDS-resident-box kernel
    static const int box_level = [] { const char* e = getenv("W2L_CONVB_BOX"); return e ? atoi(e) : 1; }();   // 2: without the 3x3 small-channel families (A/B)
    const bool box_on = box_level != 0;
    if (box_on && !unit && (v.q_is_out || (v.omy == 1 && v.omx == 1)) && c->tile_override < 0 && ksplit_force < 1 && v.nphase == 1 &&
        v.ph[0].ntaps == g.kh * g.kw && Ho == H && Wo == W && v.sy == 1 && v.sx == 1 && (g.kh == 7 || box_level != 2) &&
        stem_ok(g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, c->cin_p, g.cout, N, H, W, res != nullptr)) {
        if (stats_out) *stats_out = nullptr;      // no partials / sums: the stand-alone reductions follow (few channels: cheap passes)
        if (flops_counting()) flops_add(2ll * N * H * W * c->cout_p * v.ph[0].kp, 5);
        return stem_launch(static_cast<hipStream_t>(stream), x, x_cs, y, y_cs, res, res_cs, v.w_dev, c->cout_p, v.ph[0].kp, scale, shift,
                           v.taps_dev, N, H, W, g.kh, c->cin_p, g.cout, g.act);
    }
    if (box_on && !unit && (v.q_is_out || (v.omy == 1 && v.omx == 1)) && c->tile_override < 0 && ksplit_force < 1 && g.kh == 3 && g.kw == 3 && g.ph == 1 && g.pw == 1 &&
        v.ph[0].kp == 576 && box64_ok(v.nphase, v.ph[0].ntaps, c->cin_p, g.cout, c->cout_p, N, H, W, Ho, Wo, v.sy, v.sx)) {
        // forward statistics, or (bb) the BatchNorm-backward sums of the block whose dy this launch completes: per-wave partials
        static const bool box_bwd = [] { const char* e = getenv("W2L_BOX_BWD_SUMS"); return e ? atoi(e) != 0 : true; }();   // A/B switch
        hipStream_t s = static_cast<hipStream_t>(stream);
        float* stats = nullptr;
        BoxBwd bw;
        const bool bwd = bb != nullptr && bb->z != nullptr && box_bwd && stats_out != nullptr;
        if (stats_out) {
            *stats_out = nullptr;
            if (!bb || bwd) {
                const int npart = box64_grid(N, H, W) * 8;
                stats = conv_workspace(s, (size_t)npart * 2 * c->cout_p * sizeof(float));
                if (!stats) return W2L_ERR_NOMEM;
                *stats_out = stats;
                *npart_out = npart;
            }
        }
        if (bwd) {
            bw.z = bb->z; bw.y = bb->y; bw.z_cs = bb->z_cs; bw.y_cs = bb->y_cs; bw.store_g = bb->store_g;
            bw.neg = bb->act == W2L_ACT_RELU ? 0.f : (bb->act == W2L_ACT_LEAKY ? 0.01f : 1.f);
            bw.mean = bb->mean; bw.rstd = bb->rstd; bw.scale = bb->scale; bw.shift = bb->shift;
        }
        if (flops_counting()) flops_add(2ll * N * H * W * 64 * 576, 5);
        return box64_launch(s, x, x_cs, y, y_cs, res, res_cs, v.w_dev, scale, shift, v.taps_dev, stats, N, H, W, g.cout, g.act,
                            bwd ? &bw : nullptr);
    }
    // W2L_CONVB_TP2B=0 (read once): the stride-2 transposed layers stay on the four-phase implicit GEMM below (A/B switch)
    static const bool tp2b_on = [] { const char* e = getenv("W2L_CONVB_TP2B"); return e ? atoi(e) != 0 : true; }();
    if (tp2b_on && !unit && c->tile_override < 0 && ksplit_force < 1 &&
        tp2b_ok(g.transposed, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, v.nphase, v.ph, v.taps_host.data(), c->cin_p, c->cout_p, N, H, W, Ho, Wo)) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        float* stats = nullptr;
        if (stats_out) {
            *stats_out = nullptr;       // (a data-gradient launch with `bb` reports "not fused": the stand-alone reduction runs)
            if (!bb) {
                const int npart = tp2b_npart(c->cout_p, N, H, W);
                stats = conv_workspace(s, (size_t)npart * 2 * c->cout_p * sizeof(float));
                if (!stats) return W2L_ERR_NOMEM;
                *stats_out = stats;
                *npart_out = npart;
            }
        }
        if (flops_counting()) {
            long long kp = 0;
            for (int i = 0; i < v.nphase; ++i) kp += (long long)v.ph[i].ntaps * c->cin_p;
            flops_add(2ll * N * H * W * c->cout_p * kp, 5);
        }
        return tp2b_launch(s, x, x_cs, y, y_cs, res, res_cs, v.w_dev, v.w_elems, scale, shift, stats, v.ph, v.taps_host.data(), N, H, W,
                           c->cin_p, g.cout, c->cout_p, g.act);
    }
    int ti, ks;
    pickb(c, v, a.M, &ti, &ks);
    if (c->tile_override >= 0) { ti = c->tile_override; ks = 1; }
    if (ksplit_force >= 1) ks = ksplit_force;
    const int steps = max_stepsb(v);
    if (ks > steps) ks = steps;
    if (ks < 1) ks = 1;
    a.steps_per_split = ceil_div(steps, ks);
    a.ksplit = ceil_div(steps, a.steps_per_split);
    a.ws = nullptr;
    a.stats = nullptr;
    a.bz = nullptr; a.by = nullptr; a.bmean = nullptr; a.brstd = nullptr; a.bscale = nullptr; a.bshift = nullptr;
    a.bz_cs = 0; a.by_cs = 0; a.bneg = 1.f; a.bstore_g = 0; a.bmask_only = 0;
    const BTile& tc = kBTiles[ti];
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long npix = (long long)N * Ho * Wo;
    if (stats_out && !(bb && bb->z == nullptr)) {
        *stats_out = nullptr;
        if (a.ksplit == 1) {
            const int npart = v.nphase * ceil_div(a.M, tc.bm) * tc.wm;
            a.stats = conv_workspace(s, (size_t)npart * 2 * c->cout_p * sizeof(float));
            if (!a.stats) return W2L_ERR_NOMEM;
            *stats_out = a.stats;
            *npart_out = npart;
            if (bb) {
                a.bz = bb->z; a.by = bb->y; a.bz_cs = bb->z_cs; a.by_cs = bb->y_cs;
                a.bmean = bb->mean; a.brstd = bb->rstd; a.bscale = bb->scale; a.bshift = bb->shift;
                a.bneg = bb->act == W2L_ACT_RELU ? 0.f : (bb->act == W2L_ACT_LEAKY ? 0.01f : 1.f);
                a.bstore_g = bb->store_g;
            }
        }
    }
    if (bb && bb->z == nullptr && stats_out) {      // mask-only request (w2l_convb_forward_actbwd): served by un-split launches
        *stats_out = nullptr;
        if (a.ksplit == 1) {
            a.by = bb->y; a.by_cs = bb->y_cs; a.bmask_only = 1;
            a.bneg = bb->act == W2L_ACT_RELU ? 0.f : (bb->act == W2L_ACT_LEAKY ? 0.01f : 1.f);
            *npart_out = -1;                         // "masked": the output IS the masked gradient
            if (bb->store_g) {
                // ... and its per-wave column sums (the forward-statistics accumulators over the stored values: sum dz is the
                // block's bias gradient) are wanted too
                const int npart = v.nphase * ceil_div(a.M, tc.bm) * tc.wm;
                a.stats = conv_workspace(s, (size_t)npart * 2 * c->cout_p * sizeof(float));
                if (!a.stats) return W2L_ERR_NOMEM;
                *stats_out = a.stats;
                *npart_out = -npart - 1;             // <= -2: masked, with npart partial rows
            }
        }
    }
    if (a.ksplit > 1) {
        a.ws = conv_workspace(s, (size_t)a.ksplit * npix * c->cout_p * sizeof(float));
        if (!a.ws) return W2L_ERR_NOMEM;
        // Every (split, output pixel, channel < round8(cout)) entry the reduce kernel reads is WRITTEN by the workgroup that owns
        // the pixel's tile row in that split - a split whose K range holds only padding writes zeros - provided every phase has
        // taps (a phase without taps launches no K-step and its pixels would stay unwritten: kernel smaller than the stride).
        // Only then are the partials cleared first.  (Rounds 3-5 cleared them always: 46 memsets per wav2lip_train step.)
        // W2L_SPLITK_MEMSET=1 restores that (A/B).
        static const bool always = [] { const char* e = getenv("W2L_SPLITK_MEMSET"); return e ? atoi(e) != 0 : false; }();
        bool need = always;
        for (int i = 0; i < v.nphase; ++i) need |= v.ph[i].ntaps <= 0;
        need |= !v.q_is_out && !unit && (Ho % v.omy || Wo % v.omx);      // ragged transposed extents: as conv_igemm.hip keeps it
        if (need) W2L_HIP_CHECK(hipMemsetAsync(a.ws, 0, (size_t)a.ksplit * npix * c->cout_p * sizeof(float), s));
    }
    a.tiles_m = ceil_div(a.M, tc.bm);
    a.tiles_n = ceil_div(c->cout_p, tc.bn);
    const long long nblk = (long long)a.tiles_m * a.tiles_n;
    W2L_REQUIRE(nblk < lim, "grid too large");
    if (flops_counting()) {
        long long kp = 0;
        for (int i = 0; i < v.nphase; ++i) kp += v.ph[i].kp;
        flops_add(2ll * a.tiles_m * tc.bm * a.tiles_n * tc.bn * kp, 5);
    }
    hipLaunchKernelGGL(tc.kernel, dim3((unsigned)nblk, v.nphase, a.ksplit), dim3(tc.threads), tc.lds, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    if (a.ksplit > 1) {
        ReduceBArgs r;
        r.ws = a.ws; r.y = y; r.res = res; r.scale = scale; r.shift = shift;
        r.npix = npix; r.ksplit = a.ksplit; r.cout = g.cout; r.cout_p = c->cout_p; r.y_cs = y_cs; r.res_cs = res_cs; r.act = g.act;
        long long gsz = (npix * ((g.cout + 7) / 8) + 255) / 256;
        if (gsz > 4096) gsz = 4096;
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)gsz), dim3(256), 0, s, r);
        W2L_HIP_CHECK(hipGetLastError());
    }
    return W2L_OK;
}

namespace w2l {
int bn_stats_from_partials(hipStream_t s, const float* part, int npart, int cout_p, long long rows, int C, int Cvalid,
                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* rstd, float* scale, float* shift);   // train_bf16.hip
int bn_bwd_sums_from_partials(hipStream_t s, const float* part, int npart, int cout_p, int C, int Cvalid, float* dgamma,
                              float* dbeta);                                        // train_bf16.hip
}

extern "C" {

int w2l_convb_forward(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                      const void* res, int res_cs, const float* scale, const float* shift, int ksplit_force) {
    return convb_forward_impl(c, stream, N, H, W, x, x_cs, y, y_cs, res, res_cs, scale, shift, ksplit_force, nullptr, nullptr);
}

int w2l_convb_forward_bn(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* z, int z_cs,
                         const float* bias, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* rstd, float* scale, float* shift) {
    W2L_REQUIRE(c && mean && rstd && scale && shift, "NULL argument");
    W2L_REQUIRE(c->g.act == W2L_ACT_NONE, "convb_forward_bn: the layer in front of a batch-statistics BatchNorm has no activation");
    float* part = nullptr;
    int npart = 0;
    int rc = convb_forward_impl(c, stream, N, H, W, x, x_cs, z, z_cs, nullptr, 0, nullptr, bias, 0, &part, &npart);
    if (rc != W2L_OK) return rc;
    int Ho, Wo;
    if (w2l_conv_out_hw(&c->g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    const long long rows = (long long)N * Ho * Wo;
    const int C8 = round_up(c->g.cout, 8);
    if (part)
        return bn_stats_from_partials(static_cast<hipStream_t>(stream), part, npart, c->cout_p, rows, C8, c->g.cout, gamma, beta, eps,
                                      momentum, running_mean, running_var, mean, rstd, scale, shift);
    return w2l_bn_train_stats_bf16(stream, rows, C8, c->g.cout, z, z_cs, gamma, beta, eps, momentum, running_mean, running_var, mean,
                                   rstd, scale, shift);
}

int w2l_convb_forward_bnbwd(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                            const void* res, int res_cs, const void* bz, int bz_cs, const void* by, int by_cs, int bact,
                            const float* mean, const float* rstd, const float* bscale, const float* bshift, float* dgamma,
                            float* dbeta, int* fused_out) {
    W2L_REQUIRE(c && bz && mean && rstd && dgamma && dbeta && fused_out, "NULL argument");
    W2L_REQUIRE(c->g.act == W2L_ACT_NONE, "convb_forward_bnbwd: a data-gradient launch has no activation");
    const int store_g = (bact & W2L_BNBWD_STORE_MASKED) != 0;
    bact &= ~W2L_BNBWD_STORE_MASKED;
    W2L_REQUIRE(!store_g || bact == W2L_ACT_RELU, "convb_forward_bnbwd: the masked gradient can be stored for a ReLU block only");
    W2L_REQUIRE(bact == W2L_ACT_NONE || bact == W2L_ACT_RELU || bact == W2L_ACT_LEAKY, "convb_forward_bnbwd: block activation %d", bact);
    W2L_REQUIRE(by != nullptr || (bact == W2L_ACT_RELU && bscale && bshift),
                "convb_forward_bnbwd: the block output may be omitted only for a ReLU block without residual, with scale / shift given");
    const int C8 = round_up(c->g.cout, 8);
    W2L_REQUIRE(bz_cs >= C8 && (bz_cs & 7) == 0 && (by == nullptr || (by_cs >= C8 && (by_cs & 7) == 0)) &&
                    ((reinterpret_cast<uintptr_t>(bz) | reinterpret_cast<uintptr_t>(by)) & 15) == 0,
                "convb_forward_bnbwd: z / y must be 16-byte aligned with channel strides that are multiples of 8 and >= %d", C8);
    int Ho, Wo;
    if (w2l_conv_out_hw(&c->g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(((long long)N * Ho * Wo * bz_cs) * 2 < (1ll << 31) && (by == nullptr || ((long long)N * Ho * Wo * by_cs) * 2 < (1ll << 31)),
                "activation buffer larger than 2 GiB: split the batch");
    BnBwdOperands bb = {bz, by, bz_cs, by_cs, bact, mean, rstd, bscale, bshift, store_g};
    float* part = nullptr;
    int npart = 0;
    *fused_out = 0;
    int rc = convb_forward_impl(c, stream, N, H, W, x, x_cs, y, y_cs, res, res_cs, nullptr, nullptr, 0, &part, &npart, &bb);
    if (rc != W2L_OK || !part) return rc;        // split-K launch: no sums, the caller runs the stand-alone reduction
    *fused_out = 1;
    return bn_bwd_sums_from_partials(static_cast<hipStream_t>(stream), part, npart, c->cout_p, C8, c->g.cout, dgamma, dbeta);
}

/* header: w2l_convb_forward_actbwd */
int w2l_convb_forward_actbwd(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                             const void* res, int res_cs, const void* by, int by_cs, int bact, float* dbias, int* fused_out) {
    W2L_REQUIRE(c && by && fused_out, "NULL argument");
    W2L_REQUIRE(c->g.act == W2L_ACT_NONE, "convb_forward_actbwd: a data-gradient launch has no activation");
    W2L_REQUIRE(bact == W2L_ACT_RELU || bact == W2L_ACT_LEAKY, "convb_forward_actbwd: block activation %d (ReLU / LeakyReLU)", bact);
    const int C8 = round_up(c->g.cout, 8);
    W2L_REQUIRE(by_cs >= C8 && (by_cs & 7) == 0 && (reinterpret_cast<uintptr_t>(by) & 15) == 0,
                "convb_forward_actbwd: the block output must be 16-byte aligned with a channel stride that is a multiple of 8 and >= %d", C8);
    int Ho, Wo;
    if (w2l_conv_out_hw(&c->g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(((long long)N * Ho * Wo * by_cs) * 2 < (1ll << 31), "activation buffer larger than 2 GiB: split the batch");
    BnBwdOperands bb = {nullptr, by, 0, by_cs, bact, nullptr, nullptr, nullptr, nullptr, dbias != nullptr ? 1 : 0};
    float* part = nullptr;
    int npart = 0;
    *fused_out = 0;
    const int rc = convb_forward_impl(c, stream, N, H, W, x, x_cs, y, y_cs, res, res_cs, nullptr, nullptr, 0, &part, &npart, &bb);
    if (rc != W2L_OK || npart >= 0) return rc;
    *fused_out = 1;
    if (dbias && part && npart <= -2)         // column sums of the stored dz: fixed-order two-level finish, pad entries zero
        return bn_bwd_sums_from_partials(static_cast<hipStream_t>(stream), part, -npart - 1, c->cout_p, C8, c->g.cout, nullptr, dbias);
    return W2L_OK;
}

}  // extern "C"

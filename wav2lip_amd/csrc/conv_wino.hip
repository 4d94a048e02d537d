// Winograd F(2x2, 3x3) convolution for gfx950 on the fp32 matrix cores: the 3x3 / stride 1 / pad 1 layers of the
// generator (the residual blocks of models/wav2lip.py:61-81 and models/conv.py:5-19 — 73 % of its multiply-adds).
//
//   y = act( A^T [ sum_c (G g G^T)[xi] * (B^T d B)[xi] ] A * scale + shift (+ res) )
//
// 2.25x fewer multiplies than the direct form at fp32 accuracy (all products and sums in fp32; the transforms use only
// +,- on the data side, the weight side is transformed once in fp64 on the host side of the ABI).
//
// GEMM view: for each of the 16 transform positions xi an independent GEMM  M_xi[tile][cout] += V_xi[tile][cin] *
// U_xi[cout][cin]  with tile = (n, ty, tx) over 2x2 output tiles.  One workgroup (4 waves, one per SIMD, up to 512
// registers each) owns BT tiles x BC couts; every wave owns 32 tiles x 32 couts for ALL 16 positions = 16 accumulators
// of 32x32 (256 registers), so that the inverse transform A^T M A is a per-lane register operation in the epilogue.
// Per K-step (KS input channels):
//   * every thread gathers a 3x4 pixel patch (one float4 of channels each) of one tile straight from the NHWC input
//     (out-of-image taps read zero through out-of-range buffer offsets), applies its half of B^T d B in registers and
//     writes 8 positions of V to LDS ([buf][xi][tile][KS+4], conflict-free b128 rows); loads run one K-step ahead;
//   * the A operand (V) is read from LDS one position ahead, the B operand (U) comes straight from global memory/L2 in
//     MFMA fragment order (pre-packed so that a wave reads 1 KiB contiguous per position) four positions ahead.
#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kWOob = 0x80000000u;

__device__ __forceinline__ f32x4 wbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ float wbuf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
__device__ __forceinline__ void wbuf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}

template <int WM, int WN, int KS>
constexpr int wino_lds_bytes() {
    return 2 * 16 * (32 * WM) * (KS + 4) * 4 + 2 * (32 * WM) * 4;
}

template <int WM, int WN, int KS>
__global__ __launch_bounds__(256, 1) void conv_wino_f32_kernel(const WinoKArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(KS == 8 || KS == 16, "K-step of 8 or 16 channels");
    constexpr int BT = 32 * WM;        // tiles per workgroup
    constexpr int BC = 32 * WN;        // couts per workgroup
    constexpr int LDK = KS + 4;        // V row stride (floats)
    constexpr int QN = KS / 4;         // float4 channel groups per K-step
    constexpr int NSUB = KS / 8;       // 8-channel MFMA sub-steps per K-step
    static_assert(BT * QN * 2 == 256, "one transform item per thread");
    constexpr int VPOS = BT * LDK;     // floats per position slab
    constexpr int VBUF = 16 * VPOS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);             // [2][16][BT][LDK]
    int* s_opix = reinterpret_cast<int*>(Vs + 2 * VBUF);    // [BT] output pixel of (2ty, 2tx) or -1
    int* s_oflag = s_opix + BT;                              // [BT] bit0: column 2tx+1 exists, bit1: row 2ty+1 exists

    const int THW = a.TH * a.TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);

    // PERSISTENT workgroups: the grid is one workgroup per CU (a wave needs all 512 registers, so nothing else is resident
    // anyway) and every workgroup walks several (tile_m, tile_n) work items.  A fresh workgroup per item cost ~20 us of
    // dispatch + drain on top of ~11-25 us of work (fit over the generator's layers: t = 5.5 ns x WG-steps + 85 ns x WGs).
    // XCD x (hardware ids x, x+8, ...) sweeps the contiguous range [x*per, (x+1)*per): neighbouring tiles share one L2.
    const unsigned total = (unsigned)a.tiles_m * (unsigned)a.tiles_n;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    for (unsigned jw = blockIdx.x >> 3; jw < per; jw += gw) {
    const unsigned bid = xcd * per + jw;
    if (bid >= total) break;
    // per-thread coordinates are re-derived in every work item from an opaque copy of the thread id: hoisted out of the
    // loop they would stay live across the epilogue (256 accumulators + 64 residuals + addresses) and spill
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // provably wave-uniform: descriptors built from it stay in SGPRs
    const int wm = wave / WN;
    const int wn = wave % WN;
    // ---- transform item of this thread: tile tl, channel quad q, half (rows {0,1} or {2,3} of B^T d).
    // Rows of the 4x4 input tile are loaded in the order (A, B, C) = (d0, d2, d1) for half 0 and (d2, d1, d3) for half 1,
    // so that both halves run the same code:  row 2*half of B^T d = A - B,  row 2*half+1 = B + sgn*C  (sgn = +1 / -1).
    const int half = wave >> 1;                // BT*QN = 128 items = 2 waves per half
    const float sgn = half ? -1.0f : 1.0f;
    const int tl = (t % (BT * QN)) / QN;
    const int q = t % QN;
    // an XCD's contiguous item range is walked cout-tile fastest (the cout-tiles of one M-tile share its input block in L2) unless
    // the layer's weights outweigh its input (512 @6x6: then an XCD keeps ONE cout-tile's weights and streams the M-tiles past them)
    const int tile_n = a.m_fastest ? bid / a.tiles_m : bid % a.tiles_n;
    const int tile_m = a.m_fastest ? bid % a.tiles_m : bid / a.tiles_n;
    const int m0 = tile_m * BT;
    const int n0 = tile_n * BC;

    if (t < BT) {
        const int m = m0 + t;
        int o = -1, f = 0;
        if (m < a.M) {
            const int n = m / THW;
            const int rem = m - n * THW;
            const int ty = rem / a.TW;
            const int tx = rem - ty * a.TW;
            o = (n * a.H + 2 * ty) * a.W + 2 * tx;
            f = ((2 * tx + 1 < a.W) ? 1 : 0) | ((2 * ty + 1 < a.H) ? 2 : 0);
        }
        s_opix[t] = o;
        s_oflag[t] = f;
    }

    unsigned goff[3][4];                       // byte offsets of the 3x4 patch
    {
        const int m = m0 + tl;
        const bool mv = m < a.M;
        const int n = mv ? m / THW : 0;
        const int rem = m - n * THW;
        const int ty = rem / a.TW;
        const int tx = rem - ty * a.TW;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int rowsel = half ? (r == 0 ? 2 : (r == 1 ? 1 : 3)) : (r == 0 ? 0 : (r == 1 ? 2 : 1));
            const int iy = 2 * ty - 1 + rowsel;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = 2 * tx - 1 + c;
                const bool ok = mv & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
                goff[r][c] = ok ? ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(q * 4)) * 4u : kWOob;
            }
        }
    }
    f32x4 raw[3][4];
    f32x4 ra[4], rb[4];
    auto gload_col = [&](int step, int c) {
        const unsigned soff = (unsigned)(step * KS * 4);
#pragma unroll
        for (int r = 0; r < 3; ++r) raw[r][c] = wbuf_load4(rx, goff[r][c], soff);
    };
    auto row_tf = [&](int c) {       // column c of the two B^T d rows of this thread
        ra[c] = raw[0][c] - raw[1][c];
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[c][e] = fmaf(sgn, raw[2][c][e], raw[1][c][e]);
    };
    float* const vwr = Vs + (half * 8) * VPOS + tl * LDK + q * 4;
    auto col_tf_store = [&](int buf, int j) {   // position (2*half + j/4, j%4) of B^T d B -> LDS
        const f32x4* rr_ = (j < 4) ? ra : rb;
        f32x4 v;
        switch (j & 3) {
            case 0: v = rr_[0] - rr_[2]; break;
            case 1: v = rr_[1] + rr_[2]; break;
            case 2: v = rr_[2] - rr_[1]; break;
            default: v = rr_[1] - rr_[3]; break;
        }
        *reinterpret_cast<f32x4*>(vwr + buf * VBUF + j * VPOS) = v;
    };

    // ---- B operand: u[((nb * nks + kc) * 16 + pos) * 256 + (h*32 + n)*4 + e] = U_pos[nb*32 + n][kc*8 + 4h + e]
    const int nb = (n0 >> 5) + wn;
    const int F = a.nks * 16;                  // (8-channel chunk, position) pairs
    const bool wave_live = nb * 32 < a.cout;   // a wave past the last cout block only keeps the barriers company
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u + (long long)nb * F * 256), 0, wave_live ? F * 1024 : 0, 0x00020000);
    constexpr int RING = 8;                    // B fragments in flight (positions ahead)
    f32x4 bq[RING];
    auto bload = [&](int f) { return wbuf_load4(ru, (unsigned)(lane * 16), (unsigned)f * 1024u); };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int nsteps = a.cin / KS;
#pragma unroll
    for (int c = 0; c < 4; ++c) gload_col(0, c);
#pragma unroll
    for (int i = 0; i < RING; ++i) bq[i] = bload(i);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        row_tf(c);
        gload_col(1, c);                        // past-the-end steps read zero (cin bound of the descriptor)
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) col_tf_store(0, j);
    __syncthreads();

    const float* Abase = Vs + (wm * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    int f = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const float* Ab = Abase + buf * VBUF;
        f32x4 af = *reinterpret_cast<const f32x4*>(Ab);
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int slot = u * 16 + p;                   // 16*NSUB slots of 4 MFMAs per K-step
                const f32x4 ac = af;
                if (!(u == NSUB - 1 && p == 15)) {
                    const int pn = (p + 1) & 15, un = (p == 15) ? u + 1 : u;
                    af = *reinterpret_cast<const f32x4*>(Ab + pn * VPOS + un * 8);
                }
                const f32x4 bc = bq[slot % RING];
                bq[slot % RING] = bload(f + RING);             // past-the-end loads read zero (never used)
                ++f;
                // the input transform of the NEXT K-step rides in the MFMA shadow, one small piece per slot:
                // slots 0-3: row transform of column c, then the patch column is re-requested for step+2;
                // slots 4-11: one transformed position each -> LDS buffer buf^1
                if (slot < 4) {
                    row_tf(slot);
                    gload_col(step + 2, slot);
                } else if (slot < 12) {
                    col_tf_store(buf ^ 1, slot - 4);
                }
                // The 4 MFMAs of a slot accumulate into the SAME 32x32 tile: an instruction issued between two of them costs
                // ~43 cycles (the dependent-accumulator cliff, MI355X_MICROARCH.md), the same instruction issued between
                // MFMAs on DIFFERENT accumulators ~6.  So everything else of the slot is fenced in front of the group (it
                // runs in the shadow of the previous slot's last MFMA) and the group issues back to back.
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[e], bc[e], acc[p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  Lane holds cout (lane&31), tile rows (r&3) + 8*(r>>2) + 4*(lane>>5): the inverse transform A^T M A is
    // per-lane register work.  The scaled result is then staged through LDS (the V buffers are dead after the last
    // barrier) so that the residual loads and the output stores are whole float4 rows of the NHWC tensors: one dword per
    // lane per store is store-ISSUE-bound (64 stores per lane cost ~7 us per workgroup, measured against a no-epilogue
    // build: 0.588 -> 0.442 ms on the 64-channel 96x96 layer); 16 float4 stores per thread move the same bytes.
    constexpr int LDY = BC + 4;
    float* Ys = Vs;                              // [BT][4 pixels][LDY]
    static_assert(BT * 4 * LDY <= 2 * VBUF, "output staging tile must fit in the V buffers");
    if (wave_live) {
        const int co = nb * 32 + (lane & 31);
        const bool co_ok = co < a.cout;
        const float sc = co_ok ? a.scale[co] : 0.f;
        const float sh = co_ok ? a.shift[co] : 0.f;
        float* yrow = Ys + wn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // t[i'][j] = (A^T M)[i'][j]
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
                t1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
            }
            float o[4];
            o[0] = t0[0] + t0[1] + t0[2];
            o[1] = t0[1] - t0[2] - t0[3];
            o[2] = t1[0] + t1[1] + t1[2];
            o[3] = t1[1] - t1[2] - t1[3];
            const int tlr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int k = 0; k < 4; ++k) yrow[(tlr * 4 + k) * LDY] = o[k] * sc + sh;
        }
    }
    __syncthreads();
    {
        constexpr int CG = BC / 4;               // float4 column groups per pixel
        constexpr int NIT = BT * 4 * CG / 256;   // items per thread
        static_assert(NIT * 256 == BT * 4 * CG, "whole passes");
        const long long npix = (long long)a.N * a.H * a.W;
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + a.cout) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0,
            0x00020000);
        const int c4 = t % CG;
        const int ch = n0 + c4 * 4;
        const bool ch_ok = ch < a.cout;
        const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
        int pixv[NIT];
        f32x4 rv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 256 + t;
            const int px = (id / CG) & 3;
            const int tile = id / (CG * 4);
            const int opix = s_opix[tile];
            const int fl = s_oflag[tile];
            const bool ok = ch_ok & (opix >= 0) & (((px & 1) == 0) | ((fl & 1) != 0)) & (((px & 2) == 0) | ((fl & 2) != 0));
            const int pix = ok ? opix + (px & 1) + (px >> 1) * a.W : -1;
            pixv[i] = pix;
            u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
                rr, (int)(pix >= 0 ? ((unsigned)pix * (unsigned)a.res_cs + (unsigned)ch) * 4u : kWOob), 0, 0);
            rv[i] = __builtin_bit_cast(f32x4, raw);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 256 + t;
            const f32x4 c = *reinterpret_cast<const f32x4*>(Ys + (id / CG) * LDY + c4 * 4);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // none / ReLU / LeakyReLU(0.01) without a branch per element: max(x,0) + slope * min(x,0) is exact for all three
                // (one of the two terms is zero); sigmoid layers never come here (wino_allowed, conv_igemm.hip)
                const float x = c[e] + rv[i][e];
                v[e] = fmaf(neg_slope, fminf(x, 0.f), fmaxf(x, 0.f));
            }
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(u32x4, v), ry,
                (int)(pixv[i] >= 0 ? ((unsigned)pixv[i] * (unsigned)a.y_cs + (unsigned)ch) * 4u : kWOob), 0, 0);
        }
    }
    __syncthreads();   // s_opix / s_oflag are rewritten by the next work item
    }   // persistent loop
}

// ---- weight transform: U = G g G^T in fp64, rounded once to fp32, written in MFMA B-fragment order
struct WinoPackArgs {
    const float* w;   // [cout][cin][3][3], or (transposed) the nn.ConvTranspose2d layout [cin][cout][3][3]
    float* u;         // [cout_p/32][cin/8][16][2][32][4]
    int cin, cout, cout_p;
    int transposed;   // 1: a stride-1 pad-1 transposed conv = the conv with g'[co][ci][ky][kx] = w[ci][co][2-ky][2-kx]
};

// one thread per (cout, cin) pair: the 9 taps are read once, the 16 positions leave as 16 coalesced stores
__global__ void wino_pack_kernel(const WinoPackArgs a) {
    const int total = a.cout_p * a.cin;
    const int nks = a.cin / 8;
    constexpr double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int e = j & 3, n = (j >> 2) & 31, h = (j >> 7) & 1;
        const int blk = j >> 8;                  // (nb, kc)
        const int kc = blk % nks, nbk = blk / nks;
        const int co = nbk * 32 + n;
        const int ci = kc * 8 + 4 * h + e;
        float* dst = a.u + (long long)blk * 16 * 256 + (j & 255);
        if (co >= a.cout) {
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) dst[pos * 256] = 0.f;
            continue;
        }
        const float* g = a.transposed ? a.w + ((long long)ci * a.cout + co) * 9 : a.w + ((long long)co * a.cin + ci) * 9;
        double gd[3][3];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) gd[aa][bb] = (double)(a.transposed ? g[(2 - aa) * 3 + (2 - bb)] : g[aa * 3 + bb]);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                double sum = 0.0;
#pragma unroll
                for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) sum += G[pi][aa] * gd[aa][bb] * G[pj][bb];
                dst[(pi * 4 + pj) * 256] = (float)sum;
            }
    }
}

struct WinoCfg {
    int bt, bc, ks;
    void (*kernel)(const WinoKArgs);
    int lds;
};

static const WinoCfg kWino[] = {
    {64, 64, 8, conv_wino_f32_kernel<2, 2, 8>, wino_lds_bytes<2, 2, 8>()},
    {32, 128, 16, conv_wino_f32_kernel<1, 4, 16>, wino_lds_bytes<1, 4, 16>()},
};
constexpr int kNumWino = sizeof(kWino) / sizeof(kWino[0]);

int wino_num_cfgs() { return kNumWino; }

int wino_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    for (int i = 0; i < kNumWino; ++i)
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kWino[i].kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kWino[i].lds));
    done = true;
    return W2L_OK;
}

bool wino_cfg_ok(int cfg, int cin, int cout) {
    if (cfg < 0 || cfg >= kNumWino) return false;
    return cin % kWino[cfg].ks == 0 && cout % kWino[cfg].bc == 0;
}

// u_out: device buffer of wino_u_floats(cin, cout) floats
long long wino_u_floats(int cin, int cout) { return (long long)round_up(cout, 32) * cin * 16; }

int wino_pack(const float* w, float* u, int cin, int cout, int transposed, hipStream_t stream) {
    WinoPackArgs pa;
    pa.w = w; pa.u = u; pa.cin = cin; pa.cout = cout; pa.cout_p = round_up(cout, 32); pa.transposed = transposed;
    long long blocks = ((long long)pa.cout_p * cin + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int wino_launch(int cfg, WinoKArgs a, hipStream_t stream, long long* flops_out) {
    const WinoCfg& wc = kWino[cfg];
    a.TH = (a.H + 1) / 2;
    a.TW = (a.W + 1) / 2;
    const long long M = (long long)a.N * a.TH * a.TW;
    W2L_REQUIRE(M < (1ll << 31), "tensor too large");
    a.M = (int)M;
    a.nks = a.cin / 8;
    a.tiles_n = ceil_div(a.cout, wc.bc);
    a.tiles_m = ceil_div(a.M, wc.bt);
    const long long nblk = (long long)a.tiles_m * a.tiles_n;
    W2L_REQUIRE(nblk < (1ll << 31), "grid too large");
    {
        const long long xb = 4ll * a.N * a.H * a.W * a.cin, wb = 64ll * a.cin * a.cout;   // x and U = 16 cin cout floats
        a.m_fastest = fetch_cout_slowest(xb, wb, a.tiles_n, 1) < fetch_cout_fastest(xb, wb, a.tiles_n, 1) ? 1 : 0;
    }
#ifdef W2L_ORDER_ENV
    static const char* me = getenv("W2L_WINO_MFAST");
    if (me) a.m_fastest = me[0] == '1';
#endif
    if (flops_out) {   // dry run: 16 position-GEMMs of [tiles_m*bt] x [tiles_n*bc] x cin
        *flops_out = 2ll * 16 * nblk * wc.bt * wc.bc * a.cin;
        return W2L_OK;
    }
    // persistent: at most one workgroup per CU (256), a multiple of 8 so that every XCD gets the same number
    long long grid = (nblk + 7) / 8 * 8;
#ifndef W2L_WINO_NONPERSISTENT
    if (grid > 256) grid = 256;
#endif
    hipLaunchKernelGGL(wc.kernel, dim3((unsigned)grid), dim3(256), wc.lds, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

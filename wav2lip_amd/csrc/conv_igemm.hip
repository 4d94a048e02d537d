// Fused convolution / transposed convolution for gfx950 (MI355X) as an fp32 implicit GEMM on the
// matrix cores:  y = act( conv(x, w) * scale + shift (+ res) ),  NHWC activations, exact fp32
// (v_mfma_f32_32x32x2_f32 is bitwise an fmaf chain).
//
// Replaces the torch ops behind models/conv.py:5-44 of the reference (nn.Conv2d / nn.ConvTranspose2d
// + eval BatchNorm2d + residual add + ReLU / LeakyReLU / Sigmoid).
//
// GEMM view: M = N*Hq*Wq "q" positions, N_gemm = cout, K = ntaps*cin_p.  One workgroup of 4 waves
// owns a BM x BN output tile; each wave owns (BM/WM) x (BN/WN) as TM x TN accumulators of 32x32.
// Per K-step (32 floats) the A tile (im2col gather, 16 B per lane; padding taps / ragged rows read as zero
// through out-of-range buffer offsets, no divergent control flow) and the B tile (pre-packed [cout_p][kp]
// weights, K contiguous) are register-prefetched one step ahead and written to a double-buffered LDS image
// with 36-float rows (conflict-free ds_read_b128 fragment reads and ds_write_b128 staging writes).
// The epilogue stages the accumulators through LDS so that residual loads and output stores are whole
// float4 rows of the NHWC tensors.  A conv-transpose runs as s*s output phases (blockIdx.y), each a small
// conv over the taps of its parity, so no zero-inserted input is ever multiplied.
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLDK = kBK + 4;  // LDS row stride in floats: 36 -> 16 rows map to 16 distinct 4-bank slots
constexpr unsigned kOob = 0x80000000u;  // byte offset beyond any bound buffer (extents are checked < 2^31 on the host)

template <int BM, int BN>
constexpr int conv_lds_bytes() {
    return (2 * BM * kLDK + 2 * BN * kLDK) * 4 + BM * 4 + 128 * 4;
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == W2L_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == W2L_ACT_LEAKY) return v > 0.0f ? v : 0.01f * v;
    if (ACT == W2L_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// Epilogue on the LDS-staged BM x BN accumulator tile Cs[row][BN+4]: y = act(C*scale + shift (+ res)).
// Vector form: each thread owns one float4 column group and BM*BN/1024 rows; all loads/stores are buffer ops with an
// out-of-range offset for dead rows, so there is no divergent control flow and the residual loads issue back to back.
// pair == 2 (x-paired small-cout conv): GEMM column j is channel j % cout of output pixel ox + j / cout.
// HEAD: the activated row (all cout channels live in this tile, tiles_n == 1) is contracted with a [head_c][cout] matrix
// across the CG lanes that hold it (xor shuffles inside the wave) and only head_act(. + head_b) is written.
template <int BM, int BN, int ACT, bool HEAD>
__device__ __forceinline__ void epilogue_vec(const ConvKArgs& a, const float* Cs, const int* s_orow, int n0, int t) {
    constexpr int LDC = BN + 4;
    constexpr int CG = BN / 4;     // float4 column groups per row
    constexpr int RPP = 256 / CG;  // rows per pass
    constexpr int NV = BM / RPP;
    const int c4 = t % CG;
    const int col = n0 + c4 * 4;
    const bool col_ok = col < a.ncols;
    const int pixoff = (a.pair == 2 && col >= a.cout) ? 1 : 0;
    const int ch = col - pixoff * a.cout;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
        sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
    }
    const long long npix = (long long)a.N * a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        a.y, 0, (int)(((npix - 1) * a.y_cs + (HEAD ? a.head_c : a.cout)) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0,
        0x00020000);
    int opix[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int o = s_orow[t / CG + i * RPP];
        opix[i] = o >= 0 ? o + pixoff : o;
    }
    f32x4 rv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool ok = col_ok & (opix[i] >= 0);
        rv[i] = buf_load4(rr, ok ? ((unsigned)opix[i] * (unsigned)a.res_cs + (unsigned)ch) * 4u : kOob);
    }
    float hw[4][4];
    if (HEAD) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) hw[o][e] = (o < a.head_c && col_ok) ? a.head_w[o * a.cout + ch + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int row = t / CG + i * RPP;
        const f32x4 c = *reinterpret_cast<const f32x4*>(Cs + row * LDC + c4 * 4);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fn<ACT>(c[e] * sc[e] + sh[e] + rv[i][e]);
        const bool ok = col_ok & (opix[i] >= 0);
        if (!HEAD) {
            buf_store4(ry, ok ? ((unsigned)opix[i] * (unsigned)a.y_cs + (unsigned)ch) * 4u : kOob, v);
        } else {
            float p[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                p[o] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) p[o] = fmaf(v[e], hw[o][e], p[o]);
#pragma unroll
                for (int m = 1; m < CG; m <<= 1) p[o] += __shfl_xor(p[o], m);
            }
            if (c4 == 0 && opix[i] >= 0) {
                for (int o = 0; o < a.head_c; ++o) {
                    float hv = p[o] + (a.head_b ? a.head_b[o] : 0.f);
                    if (a.head_act == W2L_ACT_SIGMOID) hv = act_fn<W2L_ACT_SIGMOID>(hv);
                    else if (a.head_act == W2L_ACT_RELU) hv = act_fn<W2L_ACT_RELU>(hv);
                    else if (a.head_act == W2L_ACT_LEAKY) hv = act_fn<W2L_ACT_LEAKY>(hv);
                    a.y[(long long)opix[i] * a.y_cs + o] = hv;
                }
            }
        }
    }
}

// Scalar form for heads whose cout / channel stride / pointer are not float4-friendly (cout 3 or 1).
template <int BM, int BN, int ACT>
__device__ __forceinline__ void epilogue_scalar(const ConvKArgs& a, const float* Cs, const int* s_orow, int n0,
                                                int t) {
    constexpr int LDC = BN + 4;
    for (int idx = t; idx < BM * BN; idx += 256) {
        const int row = idx / BN, c = idx % BN;
        const int col = n0 + c;
        int opix = s_orow[row];
        if (col >= a.ncols || opix < 0) continue;
        const int pixoff = (a.pair == 2 && col >= a.cout) ? 1 : 0;
        const int ch = col - pixoff * a.cout;
        opix += pixoff;
        float v = Cs[row * LDC + c] * a.scale[ch] + a.shift[ch];
        if (a.res) v += a.res[(long long)opix * a.res_cs + ch];
        a.y[(long long)opix * a.y_cs + ch] = act_fn<ACT>(v);
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(const ConvKArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
    constexpr int PA = BM / 32;  // A staging passes (32 rows x 8 float4 per pass)
    constexpr int PB = BN / 32;
    static_assert(BM * (BN + 4) <= 2 * (BM + BN) * kLDK, "accumulator staging tile must fit in the A/B buffers");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                // [2][BM][kLDK]
    float* Bs = As + 2 * BM * kLDK;                            // [2][BN][kLDK]
    int* s_orow = reinterpret_cast<int*>(Bs + 2 * BN * kLDK);  // [BM] output pixel index or -1
    int* s_taps = s_orow + BM;                                 // [64][2]: (dy,dx) and byte offset per tap

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;

    int phase_i, tile_m, tile_n;
    igemm_block_coords(a, phase_i, tile_m, tile_n);
    const ConvPhase ph = a.ph[phase_i];
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int HWq = a.Hq * a.Wq;
    const int kfirst = blockIdx.z * a.steps_per_split;  // split-K: this workgroup owns K-steps [kfirst, kfirst+nsteps)

    if (t < 64) {
        const int tv = (t < ph.ntaps) ? a.taps[ph.tap_off + t] : 0;
        s_taps[2 * t] = tv;                                                       // (dy, dx) for the bounds test
        s_taps[2 * t + 1] = (((int)(short)(tv & 0xffff)) * a.W + (tv >> 16)) * a.x_cs * 4;  // byte offset of the tap
    }
    for (int r = t; r < BM; r += 256) {
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

    // buffer descriptors: out-of-range offsets read as zero, which is how padding taps, ragged M rows and
    // ragged cout rows are realised without divergent control flow
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin_p) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w + ph.w_off), 0, (int)((long long)a.cout_p * ph.kp * 4), 0x00020000);

    // ---- per-thread staging coordinates: rows (t>>3)+32p, float4 column kg = t&7
    const int kg = t & 7;
    const int r0 = t >> 3;
    int a_iy0[PA], a_ix0[PA];
    unsigned a_base[PA];  // byte offset of input pixel (n, iy0, ix0), channel 0
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + r0 + 32 * p;
        if (m < a.M) {
            const int n = m / HWq;
            const int rem = m - n * HWq;
            const int qy = rem / a.Wq;
            const int qx = rem - qy * a.Wq;
            a_iy0[p] = qy * a.sy;
            a_ix0[p] = qx * a.sx;
            a_base[p] = (unsigned)((n * a.H + a_iy0[p]) * a.W + a_ix0[p]) * (unsigned)a.x_cs * 4u;
        } else {
            a_iy0[p] = -0x4000;  // forces every tap out of range
            a_ix0[p] = -0x4000;
            a_base[p] = 0;
        }
    }
    unsigned b_off[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int gn = n0 + r0 + 32 * p;
        b_off[p] = gn < a.cout_p ? ((unsigned)gn * (unsigned)ph.kp + (unsigned)(kfirst * kBK + kg * 4)) * 4u : kOob;
    }
    const int nsteps = min(a.steps_per_split, ph.kp / kBK - kfirst);  // K-steps of this split (<= 0: nothing to do)

    __syncthreads();  // s_taps / s_orow visible

    // Two register sets for the staged tiles (static indices only: runtime-indexed vector arrays would go to scratch).
    f32x4 ra[2][PA], rb[2][PB];
    // (tap, c) of this thread's float4 column, advanced by kBK per requested tile: kBK = dq*cin_p + dc, dc < cin_p
    const int dq = kBK / a.cin_p, dc = kBK % a.cin_p;
    int g_tap = (kfirst * kBK + kg * 4) / a.cin_p;
    int g_c = (kfirst * kBK + kg * 4) % a.cin_p;
    auto gload = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;
        const bool tap_ok = g_tap < ph.ntaps;
        const int2 tv = *reinterpret_cast<const int2*>(s_taps + 2 * (tap_ok ? g_tap : 0));
        const int dy = (int)(short)(tv.x & 0xffff);
        const int dx = tv.x >> 16;
        const unsigned delta = (unsigned)(tv.y + g_c * 4);
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = tap_ok & ((unsigned)(a_iy0[p] + dy) < (unsigned)a.H) &
                            ((unsigned)(a_ix0[p] + dx) < (unsigned)a.W);  // no short-circuit: no branches
            ra[S][p] = buf_load4(rx, ok ? a_base[p] + delta : kOob);
        }
        const bool step_ok = step < nsteps;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            rb[S][p] = buf_load4(rw, step_ok ? b_off[p] : kOob);
            b_off[p] += (b_off[p] == kOob) ? 0u : kBK * 4u;
        }
        g_c += dc;
        g_tap += dq;
        if (g_c >= a.cin_p) { g_c -= a.cin_p; ++g_tap; }
    };
    auto lds_store = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        float* Ab = As + buf * BM * kLDK;
        float* Bb = Bs + buf * BN * kLDK;
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<f32x4*>(Ab + (r0 + 32 * p) * kLDK + kg * 4) = ra[S][p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<f32x4*>(Bb + (r0 + 32 * p) * kLDK + kg * 4) = rb[S][p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // A wave that owns ONE 32x32 tile (64x64, 128x32, 32x128 workgroup tiles) would chain every MFMA through the same
    // accumulator, and any staging instruction issued between two MFMAs on the same accumulator costs ~43 cycles instead of
    // ~6 (dependent-accumulator cliff, MI355X_MICROARCH.md).  Such waves alternate between two accumulators (even / odd
    // k-pairs) and add them once at the end.
    constexpr bool kDual = (TM * TN == 1);
    f32x16 acc_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;

    // Software pipeline, distance 2: at step s the tile s+2 is requested from memory into register set s&1 (start of the
    // step), tile s+1 (requested one step earlier into the other set) is written to the idle LDS buffer in the middle
    // of the step, and tile s is multiplied from LDS.  A load therefore has ~1.5 steps (>= 1500 MFMA cycles) to land
    // and every staging instruction sits inside the MFMA stream (a 64-cycle MFMA leaves ~12 issue slots).  Requests
    // past the last tile are harmless: out-of-range buffer offsets read zero and that LDS buffer is never read again.
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    gload(0, Set0{});
    lds_store(0, Set0{});
    gload(1, Set1{});
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    auto do_step = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;   // free register set; the other one holds tile step+1
        using Other = std::integral_constant<int, S ^ 1>;
        const int buf = step & 1;
        const float* Ab = As + buf * BM * kLDK + (wm * TM * 32 + frag_row) * kLDK + frag_k;
        const float* Bb = Bs + buf * BN * kLDK + (wn * TN * 32 + frag_row) * kLDK + frag_k;
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * kLDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * kLDK);
#pragma unroll
        for (int kq = 0; kq < kBK / 8; ++kq) {
            const int cur = kq & 1, nxt = cur ^ 1;
            if (kq + 1 < kBK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[nxt][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * kLDK + (kq + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[nxt][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * kLDK + (kq + 1) * 8);
            }
            if (kq == 0) gload(step + 2, SET);          // tile step+2 -> free register set
            if (kq == 2) lds_store(buf ^ 1, Other{});   // tile step+1 -> idle LDS buffer
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (kDual && (e & 1))
                            acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bf[cur][j][e], acc_odd, 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bf[cur][j][e], acc[i][j],
                                                                             0, 0, 0);
                    }
#ifndef W2L_NO_PIN
            // Issue-order recipe for this group of 4*TM*TN MFMAs (masks: 0x8 MFMA, 0x2 VALU, 0x4 SALU, 0x20 VMEM read,
            // 0x100 DS read, 0x200 DS write): the next group's fragment reads go behind the first MFMA; the address
            // arithmetic + 8 buffer loads (group 0) and the 8 LDS stores (group 2) are spread one small packet per MFMA
            // so that no gap between two MFMAs holds more than ~60 cycles of other work.
            {
                constexpr int NM = 4 * TM * TN;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kq + 1 < kBK / 8) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int m = 1; m < NM; ++m) {
                    if (kq == 0) {
                        __builtin_amdgcn_sched_group_barrier(0x002, (PA >= 4 ? 128 : 72) / NM + 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);
                        if ((m * (PA + PB)) / NM != ((m - 1) * (PA + PB)) / NM)
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    if (kq == 2 && (m * (PA + PB)) / NM != ((m - 1) * (PA + PB)) / NM)
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
            // and nothing may sink behind later groups
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        __syncthreads();
    };
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        do_step(step, Set0{});
        do_step(step + 1, Set1{});
    }
    if (step < nsteps) do_step(step, Set0{});
    if (kDual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
    }

    // ---- epilogue: stage the accumulators through LDS (the A/B buffers are dead after the last barrier) so that
    // global traffic is whole float4 rows.  Lane holds column (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5) per tile.
    constexpr int LDC = BN + 4;
    float* Cs = As;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[row * LDC + (wn * TN + j) * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    if (a.ksplit > 1) {  // partial sums -> workspace [split][output pixel][cout_p]; scale/shift/res/act happen in the reduce
        constexpr int CG = BN / 4, RPP = 256 / CG, NV = BM / RPP;
        const int c4 = t % CG;
        const int col = n0 + c4 * 4;
        const long long npix = (long long)a.N * a.Ho * a.Wo;
        float* wsz = a.ws + (long long)blockIdx.z * npix * a.cout_p;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = t / CG + i * RPP;
            const int opix = s_orow[row];
            if (opix >= 0 && col < a.cout_p)
                *reinterpret_cast<f32x4*>(wsz + (long long)opix * a.cout_p + col) =
                    *reinterpret_cast<const f32x4*>(Cs + row * LDC + c4 * 4);
        }
        return;
    }
    if (a.head_w) {
        switch (a.act) {
            case W2L_ACT_RELU: epilogue_vec<BM, BN, W2L_ACT_RELU, true>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_LEAKY: epilogue_vec<BM, BN, W2L_ACT_LEAKY, true>(a, Cs, s_orow, n0, t); break;
            default: epilogue_vec<BM, BN, W2L_ACT_NONE, true>(a, Cs, s_orow, n0, t); break;
        }
    } else if (a.vec_epilogue) {
        switch (a.act) {
            case W2L_ACT_RELU: epilogue_vec<BM, BN, W2L_ACT_RELU, false>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_LEAKY: epilogue_vec<BM, BN, W2L_ACT_LEAKY, false>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_SIGMOID: epilogue_vec<BM, BN, W2L_ACT_SIGMOID, false>(a, Cs, s_orow, n0, t); break;
            default: epilogue_vec<BM, BN, W2L_ACT_NONE, false>(a, Cs, s_orow, n0, t); break;
        }
    } else {
        switch (a.act) {
            case W2L_ACT_RELU: epilogue_scalar<BM, BN, W2L_ACT_RELU>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_LEAKY: epilogue_scalar<BM, BN, W2L_ACT_LEAKY>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_SIGMOID: epilogue_scalar<BM, BN, W2L_ACT_SIGMOID>(a, Cs, s_orow, n0, t); break;
            default: epilogue_scalar<BM, BN, W2L_ACT_NONE>(a, Cs, s_orow, n0, t); break;
        }
    }
}

// ---- bf16-MFMA variant (mixed precision for the training configs, BASELINE configs 4/5): the SAME implicit GEMM over the
// SAME fp32 NHWC tensors and fp32 packed weights, but both operand tiles are rounded to bf16 (v_cvt_pk_bf16_f32, RNE) on
// their way into LDS and multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32 matrix rate).  LDS rows
// hold 32 K-elements as bf16 (64 B) + 16 B pad: the 80-B stride sends the 16 rows of a ds_read_b128 lane group to 16
// distinct 16-B slots (5*r mod 16), conflict-free.  Lane l reads 8 consecutive K-elements of row l&31 at K offset
// 8*(l>>5) (+16 for the second MFMA of the K-step).  Epilogue, split-K, phases, padding-by-OOB: as the fp32 kernel.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

//
// NP = 3 ("split" kernel, configuration ids conv_split_id(tile)): an fp32 RESULT from the bf16 matrix cores.  An fp32 value is the
// exact sum of three bf16 values (3 x 8 significant bits): x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1).  Both operand tiles go to LDS as three bf16 planes and a K-chunk of 16 is six MFMAs, the piece products
// a_i * b_j with i + j <= 2 (the three dropped ones are below 2^-24 of the product; each kept product is exact in the fp32
// accumulator), smallest first.  Six bf16 MFMAs of K = 16 cost 6/16 of the eight fp32 MFMAs of K = 2 they replace, and the
// accumulator is rounded 6 instead of 8 times per chunk: the error against an fp64 contraction is that of the fp32 kernel
// (tools/split_bf16_accuracy.py: rms 5.1e-7 against 5.2e-7 on a K = 2304 layer).  Not bitwise the fp32 kernel's fmaf chain.
constexpr int kLDKH = kBK + 8;   // bf16 elements per LDS row

template <int BM, int BN, int NP = 1>
constexpr int conv_bf16_lds_bytes() {
    constexpr int stage = 2 * NP * (BM + BN) * (NP == 3 ? kBK : kLDKH) * 2;
    constexpr int cs = BM * (BN + 4) * 4;
    return (stage > cs ? stage : cs) + BM * 4 + 128 * 4;
}

// x = h + m + l exactly (each piece RNE to bf16 of what the pieces before it left).  Pairs: one v_cvt_pk_bf16_f32 rounds two
// values, the rounded values come back as fp32 by a shift / a mask of the packed word (5.5 VALU instructions per element).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16x2(r0, r1);
    l = pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3(const f32x4 v, u32x2& h, u32x2& m, u32x2& l) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(v[0], v[1], h0, m0, l0);
    split3_pair(v[2], v[3], h1, m1, l1);
    h = u32x2{h0, h1};
    m = u32x2{m0, m1};
    l = u32x2{l0, l1};
}

template <int BM, int BN, int WM, int WN, int NP = 1>
__global__ __launch_bounds__(64 * WM * WN, NP == 1 ? 2 : 1) void conv_igemm_bf16_kernel(const ConvKArgs a) {
    // 4 waves, or 8 for the split tiles whose three planes leave room for one workgroup per CU only: with one wave per SIMD
    // nothing hides that wave's split arithmetic, LDS stores and fragment waits (profiles/r01/i_mfma_overlap_microbench.txt:
    // vector work does not overlap the same wave's MFMAs), a second wave per SIMD does
    constexpr int NT = 64 * WM * WN;
    static_assert(NT == 256 || (NT == 512 && NP == 3), "4 waves per workgroup (8 for the large split tiles)");
    static_assert(NP == 1 || NP == 3, "one bf16 plane (rounded operands) or three (split fp32 operands)");
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
    constexpr int RPS = NT / 8;        // tile rows staged per pass (8 lanes x 16 B cover the 32 floats of a row)
    constexpr int PA = BM / RPS;
    constexpr int PB = BN / RPS;
    static_assert(PA >= 1 && PB >= 1, "tile smaller than one staging pass");
    // LDS row of 32 K-elements.  NP == 1: 64 B + 16 B pad (conflict-free fragment reads, staging stores 2-way).  NP == 3: 64 B,
    // the 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3): fragment reads (lane groups of MI355X_MICROARCH.md, LDS table)
    // AND the 8- / 16-byte staging stores are conflict-free; with the pad the three planes' stores ran 2-way conflicted and the
    // LDS array was busy 41 % of the kernel (profiles/r04/u_split_igemm_pmc_256ch_24.txt)
    constexpr int LDR = NP == 3 ? kBK : kLDKH;
    constexpr int STAGE = 2 * NP * (BM + BN) * LDR * 2;
    constexpr int CSB = BM * (BN + 4) * 4;
    constexpr int REGION = STAGE > CSB ? STAGE : CSB;
    constexpr int APL = BM * LDR, BPL = BN * LDR;              // elements per plane

    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);              // [2][NP][BM][LDR]
    __bf16* Bs = As + 2 * NP * APL;                            // [2][NP][BN][LDR]
    int* s_orow = reinterpret_cast<int*>(smem + REGION);       // [BM]
    int* s_taps = s_orow + BM;                                 // [64][2]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;

    int phase_i, tile_m, tile_n;
    igemm_block_coords(a, phase_i, tile_m, tile_n);
    const ConvPhase ph = a.ph[phase_i];
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int HWq = a.Hq * a.Wq;
    const int kfirst = blockIdx.z * a.steps_per_split;

    if (t < 64) {
        const int tv = (t < ph.ntaps) ? a.taps[ph.tap_off + t] : 0;
        s_taps[2 * t] = tv;
        s_taps[2 * t + 1] = (((int)(short)(tv & 0xffff)) * a.W + (tv >> 16)) * a.x_cs * 4;
    }
    for (int r = t; r < BM; r += NT) {
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
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin_p) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = NP == 3
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.wsplit + 3 * ph.w_off), 0,
                                            (int)((long long)3 * a.cout_p * ph.kp * 2), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w + ph.w_off), 0, (int)((long long)a.cout_p * ph.kp * 4),
                                            0x00020000);

    const int kg = t & 7;
    const int r0 = t >> 3;
    int a_iy0[PA], a_ix0[PA];
    unsigned a_base[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + r0 + RPS * p;
        if (m < a.M) {
            const int n = m / HWq;
            const int rem = m - n * HWq;
            const int qy = rem / a.Wq;
            const int qx = rem - qy * a.Wq;
            a_iy0[p] = qy * a.sy;
            a_ix0[p] = qx * a.sx;
            a_base[p] = (unsigned)((n * a.H + a_iy0[p]) * a.W + a_ix0[p]) * (unsigned)a.x_cs * 4u;
        } else {
            a_iy0[p] = -0x4000;
            a_ix0[p] = -0x4000;
            a_base[p] = 0;
        }
    }
    // NP == 3: the weights are split when they are packed ([3][cout_p][kp] bf16 per phase): a thread stages 16 B (8 K-elements)
    // of one row per plane, 4 lanes per row, RB rows per pass
    constexpr int RB = NT / 4;
    constexpr int PB3 = (BN + RB - 1) / RB;
    constexpr int NB = NP == 3 ? PB3 : PB;
    const int bc = t & 3, br0 = t >> 2;
    unsigned b_off[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        if (NP == 3) {
            const int gn = n0 + br0 + RB * p;
            b_off[p] = (br0 + RB * p < BN && gn < a.cout_p)
                           ? ((unsigned)gn * (unsigned)ph.kp + (unsigned)(kfirst * kBK + bc * 8)) * 2u : kOob;
        } else {
            const int gn = n0 + r0 + RPS * p;
            b_off[p] = gn < a.cout_p ? ((unsigned)gn * (unsigned)ph.kp + (unsigned)(kfirst * kBK + kg * 4)) * 4u : kOob;
        }
    }
    const unsigned b_plane = (unsigned)a.cout_p * (unsigned)ph.kp * 2u;      // bytes between the planes of the split weights
    const int nsteps = min(a.steps_per_split, ph.kp / kBK - kfirst);

    __syncthreads();

    f32x4 ra[2][PA], rb[2][NP == 3 ? 3 * PB3 : PB];     // NP == 3: rb holds raw 16-byte pieces of the three weight planes
    const int dq = kBK / a.cin_p, dc = kBK % a.cin_p;
    int g_tap = (kfirst * kBK + kg * 4) / a.cin_p;
    int g_c = (kfirst * kBK + kg * 4) % a.cin_p;
    auto gload = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;
        const bool tap_ok = g_tap < ph.ntaps;
        const int2 tv = *reinterpret_cast<const int2*>(s_taps + 2 * (tap_ok ? g_tap : 0));
        const int dy = (int)(short)(tv.x & 0xffff);
        const int dx = tv.x >> 16;
        const unsigned delta = (unsigned)(tv.y + g_c * 4);
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = tap_ok & ((unsigned)(a_iy0[p] + dy) < (unsigned)a.H) & ((unsigned)(a_ix0[p] + dx) < (unsigned)a.W);
            ra[S][p] = buf_load4(rx, ok ? a_base[p] + delta : kOob);
        }
        const bool step_ok = step < nsteps;
        if (NP == 3) {
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const bool ok = step_ok && b_off[p] != kOob;
#pragma unroll
                for (int q = 0; q < 3; ++q) rb[S][3 * p + q] = buf_load4(rw, ok ? b_off[p] + q * b_plane : kOob);
                b_off[p] += (b_off[p] == kOob) ? 0u : kBK * 2u;
            }
        } else {
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                rb[S][p] = buf_load4(rw, step_ok ? b_off[p] : kOob);
                b_off[p] += (b_off[p] == kOob) ? 0u : kBK * 4u;
            }
        }
        g_c += dc;
        g_tap += dq;
        if (g_c >= a.cin_p) { g_c -= a.cin_p; ++g_tap; }
    };
    auto lds_store = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        __bf16* Ab = As + buf * NP * APL;
        __bf16* Bb = Bs + buf * NP * BPL;
        if (NP == 1) {
#pragma unroll
            for (int p = 0; p < PA; ++p)
                *reinterpret_cast<bf16x4*>(Ab + (r0 + RPS * p) * kLDKH + kg * 4) = __builtin_convertvector(ra[S][p], bf16x4);
#pragma unroll
            for (int p = 0; p < PB; ++p)
                *reinterpret_cast<bf16x4*>(Bb + (r0 + RPS * p) * kLDKH + kg * 4) = __builtin_convertvector(rb[S][p], bf16x4);
        } else {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                u32x2 h, m, l;
                split3(ra[S][p], h, m, l);
                const int row = r0 + RPS * p;
                __bf16* d = Ab + row * LDR + (((kg >> 1) ^ ((row >> 2) & 3)) * 8) + (kg & 1) * 4;
                *reinterpret_cast<u32x2*>(d) = h;
                *reinterpret_cast<u32x2*>(d + APL) = m;
                *reinterpret_cast<u32x2*>(d + 2 * APL) = l;
            }
#pragma unroll
            for (int p = 0; p < PB3; ++p) {
                const int row = br0 + RB * p;
                if (RB * PB3 > BN && row >= BN) continue;
                __bf16* d = Bb + row * LDR + ((bc ^ ((row >> 2) & 3)) * 8);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4*>(d + q * BPL) = rb[S][3 * p + q];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr bool kDual = (TM * TN == 1);   // one tile per wave: alternate two accumulators (see the fp32 kernel)
    f32x16 acc_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    gload(0, Set0{});
    lds_store(0, Set0{});
    gload(1, Set1{});
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;
    // NP == 3: element offset of this lane's chunk (lane >> 5) + 2 kq inside its swizzled row (tile bases are multiples of 32 rows)
    const int frag_sw[2] = {(((lane >> 5)) ^ ((frag_row >> 2) & 3)) * 8, (((lane >> 5) + 2) ^ ((frag_row >> 2) & 3)) * 8};
    auto do_step = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;
        using Other = std::integral_constant<int, S ^ 1>;
        const int buf = step & 1;
        const __bf16* Ab = As + buf * NP * APL + (wm * TM * 32 + frag_row) * LDR + (NP == 3 ? 0 : frag_k);
        const __bf16* Bb = Bs + buf * NP * BPL + (wn * TN * 32 + frag_row) * LDR + (NP == 3 ? 0 : frag_k);
        if (NP == 1) {
            bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kq][i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * kLDKH + kq * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[kq][j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * kLDKH + kq * 16);
            }
            gload(step + 2, SET);
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (kDual && kq == 1)
                            acc_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i], bfr[kq][j], acc_odd, 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i], bfr[kq][j], acc[i][j], 0, 0, 0);
                    }
                if (kq == 0) lds_store(buf ^ 1, Other{});
            }
        } else {
            // the six piece products of a K-chunk, smallest first: (a2 b0) (a1 b1) (a0 b2) | (a1 b0) (a0 b1) | (a0 b0)
            constexpr int kPa[6] = {2, 1, 0, 1, 0, 0};
            constexpr int kPb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                bf16x8 af[NP][TM], bfr[NP][TN];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[q][i] = *reinterpret_cast<const bf16x8*>(Ab + q * APL + i * 32 * LDR + frag_sw[kq]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bfr[q][j] = *reinterpret_cast<const bf16x8*>(Bb + q * BPL + j * 32 * LDR + frag_sw[kq]);
                }
                if (kq == 0) gload(step + 2, SET);
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (kDual && (u & 1))
                                acc_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kPa[u]][i], bfr[kPb[u]][j], acc_odd, 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kPa[u]][i], bfr[kPb[u]][j], acc[i][j], 0, 0, 0);
                        }
                if (kq == 0) lds_store(buf ^ 1, Other{});
            }
        }
        __syncthreads();
    };
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        do_step(step, Set0{});
        do_step(step + 1, Set1{});
    }
    if (step < nsteps) do_step(step, Set0{});
    if (kDual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
    }

    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[row * LDC + (wn * TN + j) * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    if (NT > 256 && t >= 256) return;     // the epilogue forms below are written for 256 threads
    if (a.ksplit > 1) {
        constexpr int CG = BN / 4, RPP = 256 / CG, NV = BM / RPP;
        const int c4 = t % CG;
        const int col = n0 + c4 * 4;
        const long long npix = (long long)a.N * a.Ho * a.Wo;
        float* wsz = a.ws + (long long)blockIdx.z * npix * a.cout_p;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = t / CG + i * RPP;
            const int opix = s_orow[row];
            if (opix >= 0 && col < a.cout_p)
                *reinterpret_cast<f32x4*>(wsz + (long long)opix * a.cout_p + col) =
                    *reinterpret_cast<const f32x4*>(Cs + row * LDC + c4 * 4);
        }
        return;
    }
    if (a.head_w) {
        switch (a.act) {
            case W2L_ACT_RELU: epilogue_vec<BM, BN, W2L_ACT_RELU, true>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_LEAKY: epilogue_vec<BM, BN, W2L_ACT_LEAKY, true>(a, Cs, s_orow, n0, t); break;
            default: epilogue_vec<BM, BN, W2L_ACT_NONE, true>(a, Cs, s_orow, n0, t); break;
        }
    } else if (a.vec_epilogue) {
        switch (a.act) {
            case W2L_ACT_RELU: epilogue_vec<BM, BN, W2L_ACT_RELU, false>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_LEAKY: epilogue_vec<BM, BN, W2L_ACT_LEAKY, false>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_SIGMOID: epilogue_vec<BM, BN, W2L_ACT_SIGMOID, false>(a, Cs, s_orow, n0, t); break;
            default: epilogue_vec<BM, BN, W2L_ACT_NONE, false>(a, Cs, s_orow, n0, t); break;
        }
    } else {
        switch (a.act) {
            case W2L_ACT_RELU: epilogue_scalar<BM, BN, W2L_ACT_RELU>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_LEAKY: epilogue_scalar<BM, BN, W2L_ACT_LEAKY>(a, Cs, s_orow, n0, t); break;
            case W2L_ACT_SIGMOID: epilogue_scalar<BM, BN, W2L_ACT_SIGMOID>(a, Cs, s_orow, n0, t); break;
            default: epilogue_scalar<BM, BN, W2L_ACT_NONE>(a, Cs, s_orow, n0, t); break;
        }
    }
}

// ---- split-K reduce: y = act( sum_z ws[z] * scale + shift (+ res) ), one thread per (output pixel, channel)
struct ReduceArgs {
    const float* ws;
    float* y;
    const float* res;
    const float* scale;
    const float* shift;
    long long npix;
    int ksplit, cout, cout_p, y_cs, res_cs, act;
};

__global__ void splitk_reduce_kernel(const ReduceArgs a) {
    const long long total = a.npix * a.cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / a.cout;
        const int c = (int)(i - pix * a.cout);
        float v = 0.f;
        for (int z = 0; z < a.ksplit; ++z) v += a.ws[((long long)z * a.npix + pix) * a.cout_p + c];
        v = v * a.scale[c] + a.shift[c];
        if (a.res) v += a.res[pix * a.res_cs + c];
        switch (a.act) {
            case W2L_ACT_RELU: v = act_fn<W2L_ACT_RELU>(v); break;
            case W2L_ACT_LEAKY: v = act_fn<W2L_ACT_LEAKY>(v); break;
            case W2L_ACT_SIGMOID: v = act_fn<W2L_ACT_SIGMOID>(v); break;
            default: break;
        }
        a.y[pix * a.y_cs + c] = v;
    }
}

// ---- weight packing: torch layout -> per-phase [cout_p][kp] slabs, K = (tap, c) with c fastest
struct PackArgs {
    const float* w;   // OIHW (conv) or IOHW (transposed)
    float* out;
    const int* tapk;  // (ky & 0xffff) | (kx << 16) per tap-table entry
    int transposed, cin, cout, kh, kw, cin_p, cout_p;
    int pair;  // 2: x-paired variant, row n = (n / cout) pixel offset, (n % cout) channel; tap kx is relative to the pair window
    int nphase;
    ConvPhase ph[kMaxPhases];
};

__global__ void pack_weights_kernel(const PackArgs a) {
    const ConvPhase ph = a.ph[blockIdx.y];
    const long long total = (long long)a.cout_p * ph.kp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / ph.kp);
        const int k = (int)(i - (long long)n * ph.kp);
        const int tap = k / a.cin_p;
        const int c = k - tap * a.cin_p;
        float v = 0.f;
        if (tap < ph.ntaps && c < a.cin && n < a.pair * a.cout) {
            const int tk = a.tapk[ph.tap_off + tap];
            const int off = n / a.cout, co = n - off * a.cout;
            const int ky = tk & 0xffff, kx = (tk >> 16) - off;
            if (kx >= 0 && kx < a.kw) {
                const long long src = a.transposed
                                          ? (((long long)c * a.cout + co) * a.kh + ky) * a.kw + kx
                                          : (((long long)co * a.cin + c) * a.kh + ky) * a.kw + kx;
                v = a.w[src];
            }
        }
        a.out[ph.w_off + i] = v;
    }
}

// packed fp32 slab of one phase -> its three bf16 pieces, plane-major (w = p0 + p1 + p2 exactly; see split3)
struct SplitWArgs {
    const float* w;
    __bf16* out;
    int nphase;
    long long off[kMaxPhases];     // slab offset in elements
    long long count[kMaxPhases];   // cout_p * kp
};
__global__ void split_weights_kernel(const SplitWArgs a) {
    const int ph = blockIdx.y;
    const float* src = a.w + a.off[ph];
    __bf16* dst = a.out + 3 * a.off[ph];
    const long long n = a.count[ph];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = src[i];
        const __bf16 h = (__bf16)v;
        const float r = v - (float)h;
        const __bf16 m = (__bf16)r;
        dst[i] = h;
        dst[n + i] = m;
        dst[2 * n + i] = (__bf16)(r - (float)m);
    }
}

// ---- host side ------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn;
    float eff;  // relative MFMA-pipe efficiency guess used only by the auto-picker
    void (*kernel)(const ConvKArgs);
    int lds;
    void (*kernel_bf16)(const ConvKArgs);   // same tile on the bf16 matrix cores (w2l_conv_set_precision)
    int lds_bf16;
    void (*kernel_split)(const ConvKArgs);  // same tile, fp32 operands as three bf16 pieces (configuration id conv_split_id(tile))
    int lds_split;
    int threads_split;
};

// SWM x SWN: the wave grid of the split kernel (8 waves on the tiles with at least eight 32x32 sub-tiles)
#define W2L_TILE(BM, BN, WM, WN, EFF, SWM, SWN) \
    { BM, BN, EFF, conv_igemm_f32_kernel<BM, BN, WM, WN>, conv_lds_bytes<BM, BN>(), \
      conv_igemm_bf16_kernel<BM, BN, WM, WN>, conv_bf16_lds_bytes<BM, BN>(), \
      conv_igemm_bf16_kernel<BM, BN, SWM, SWN, 3>, conv_bf16_lds_bytes<BM, BN, 3>(), 64 * SWM * SWN }

static const TileCfg kTiles[] = {
    W2L_TILE(128, 128, 2, 2, 1.00f, 2, 4),  // 0
    W2L_TILE(128, 64, 2, 2, 0.92f, 4, 2),   // 1
    W2L_TILE(64, 128, 2, 2, 0.93f, 2, 4),   // 2
    W2L_TILE(64, 64, 2, 2, 0.88f, 2, 2),    // 3
    W2L_TILE(128, 32, 4, 1, 0.78f, 4, 1),   // 4
    W2L_TILE(32, 128, 1, 4, 0.80f, 1, 4),   // 5
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

struct Variant {
    int nphase = 0;
    int sy = 1, sx = 1, omy = 1, omx = 1;
    bool q_is_out = true;  // q-grid = output grid (conv) vs ceil(out/om) (transposed, x-paired)
    int pair = 1;          // 2: x-paired small-cout conv (GEMM N = 2*cout)
    int cout_p = 0;        // padded GEMM N of this variant
    ConvPhase ph[kMaxPhases];
    int* taps_dev = nullptr;   // [0, ntab): (dy, dx) per tap-table entry; [ntab, 2*ntab): (ky, kx) for the packer
    int ntab = 0;
    float* w_dev = nullptr;
    // w_dev as three bf16 planes per phase ([3][cout_p][kp] at 3 * w_off): the split-operand kernel's B.  Allocated and filled by
    // the first split launch of the layer (on that launch's stream; like the split-K scratch, a warm-up run precedes any graph
    // capture), refreshed by every later (re)pack; layers that never run a split configuration carry nothing.
    mutable std::atomic<__bf16*> w_split{nullptr};
    mutable bool split_fresh = false;
    long long w_floats = 0;
    bool built = false;
};

}  // namespace w2l

struct w2l_conv {
    w2l_conv_geom g;
    int cin_p, cout_p;
    float* scale = nullptr;
    float* shift = nullptr;
    const float* weight_src = nullptr;  // only valid during create
    w2l::Variant generic;   // any input size
    w2l::Variant unit_in;   // transposed, stride 1, 1x1 input: one single-tap phase per output position
    w2l::Variant xpair;     // conv with cout <= 16, x-stride 1: two horizontally adjacent output pixels per GEMM row
    float* wino_u = nullptr;  // Winograd-transformed weights (3x3 s1 p1 layers), see conv_wino.hip
    float* tp2_u = nullptr;   // fragment-ordered weights of the fused-phase stride-2 transposed kernel, see conv_tp2.hip
    float* wino4_u = nullptr; // F(4x4,3x3) Winograd-transformed weights (36 positions), see conv_wino4.hip
    // F(2x2,3x3) transformed weights as three bf16 planes in fragment order (conv_wino2s.hip): built by the first launch that names
    // that configuration (lazy_weights below), refreshed by w2l_conv_update; layers that never run it carry nothing
    mutable std::atomic<__bf16*> wino2s_u{nullptr};
    // conv_tp2's weights as three bf16 planes in conv_tp2s.hip's fragment order: built the same way by the first launch on its id
    mutable std::atomic<__bf16*> tp2s_u{nullptr};
    __bf16* stem7s_u = nullptr;   // pre-split weights of the 7x7 first-layer kernel (conv_stem7s.hip), built with the layer
    __bf16* k3s_u = nullptr;      // pre-split weights of the direct 3x3 kernel for 32-cout layers (conv_k3s.hip), built with the layer
    float* head_w = nullptr;  // fused 1x1 head [head_c][cout] (device), see w2l_conv_attach_head
    float* head_b = nullptr;
    int head_c = 0, head_act = 0;
    int tile_override = -1;
    int precision = 0;        // W2L_PREC_F32 / W2L_PREC_BF16 (operands rounded to bf16 inside the kernel, fp32 accumulate)
};

namespace w2l {

enum VariantMode { kGeneric = 0, kUnitInput = 1, kXPair = 2 };

// w_dev -> w_split, asynchronous on `stream`
static int split_variant(const Variant& v, hipStream_t stream) {
    __bf16* const ws = v.w_split.load(std::memory_order_acquire);
    if (!ws) return W2L_OK;
    SplitWArgs sa;
    sa.w = v.w_dev; sa.out = ws; sa.nphase = v.nphase;
    long long maxtot = 1;
    for (int i = 0; i < v.nphase; ++i) {
        sa.off[i] = v.ph[i].w_off;
        sa.count[i] = (long long)v.cout_p * v.ph[i].kp;
        if (sa.count[i] > maxtot) maxtot = sa.count[i];
    }
    int blocks = (int)((maxtot + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_weights_kernel, dim3(blocks, v.nphase), dim3(256), 0, stream, sa);
    W2L_HIP_CHECK(hipGetLastError());
    v.split_fresh = true;
    return W2L_OK;
}

// (re)pack `weight` (torch layout) into the variant's K-major slabs; asynchronous on `stream`
static int pack_variant(const w2l_conv* c, const Variant& v, const float* weight, hipStream_t stream) {
    const w2l_conv_geom& g = c->g;
    PackArgs pa;
    pa.w = weight;
    pa.out = v.w_dev;
    pa.tapk = v.taps_dev + v.ntab;
    pa.transposed = g.transposed;
    pa.cin = g.cin; pa.cout = g.cout; pa.kh = g.kh; pa.kw = g.kw;
    pa.cin_p = c->cin_p; pa.cout_p = v.cout_p;
    pa.pair = v.pair;
    pa.nphase = v.nphase;
    for (int i = 0; i < v.nphase; ++i) pa.ph[i] = v.ph[i];
    long long maxtot = 0;
    for (int i = 0; i < v.nphase; ++i) {
        long long tot = (long long)v.cout_p * v.ph[i].kp;
        if (tot > maxtot) maxtot = tot;
    }
    int blocks = (int)((maxtot + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks, v.nphase), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    v.split_fresh = false;
    return v.w_split.load() ? split_variant(v, stream) : W2L_OK;
}

static int build_variant(w2l_conv* c, Variant& v, VariantMode mode, hipStream_t stream) {
    const bool unit_input = mode == kUnitInput;
    v.pair = mode == kXPair ? 2 : 1;
    v.cout_p = round_up(v.pair * c->g.cout, 32);
    const w2l_conv_geom& g = c->g;
    int tapd[kMaxPhases * kMaxTaps];
    int tapk[kMaxPhases * kMaxTaps];
    int ntab = 0;
    long long woff = 0;
    v.nphase = 0;
    auto add_phase = [&](int poy, int pox) -> ConvPhase& {
        ConvPhase& p = v.ph[v.nphase++];
        p.ntaps = 0;
        p.po_y = poy;
        p.po_x = pox;
        p.tap_off = ntab;
        p.pad_ = 0;
        return p;
    };
    auto add_tap = [&](ConvPhase& p, int dy, int dx, int ky, int kx) {
        tapd[ntab] = (int)(((unsigned)dy & 0xffffu) | ((unsigned)dx << 16));
        tapk[ntab] = (int)(((unsigned)ky & 0xffffu) | ((unsigned)kx << 16));
        ++ntab;
        ++p.ntaps;
    };
    auto close_phase = [&](ConvPhase& p) {
        p.kp = round_up(p.ntaps * c->cin_p, kBK);
        p.w_off = woff;
        woff += (long long)v.cout_p * p.kp;
    };
    if (mode == kXPair) {
        // outputs (oy, 2q) and (oy, 2q+1) share the kh x (kw+1) input window starting at x = 2q*sw - pw (sw == 1)
        v.sy = g.sh; v.sx = 2; v.omy = 1; v.omx = 2; v.q_is_out = false;
        ConvPhase& p = add_phase(0, 0);
        for (int ky = 0; ky < g.kh; ++ky)
            for (int kx = 0; kx <= g.kw; ++kx) add_tap(p, ky - g.ph, kx - g.pw, ky, kx);
        close_phase(p);
    } else if (!g.transposed) {
        v.sy = g.sh; v.sx = g.sw; v.omy = 1; v.omx = 1; v.q_is_out = true;
        ConvPhase& p = add_phase(0, 0);
        for (int ky = 0; ky < g.kh; ++ky)
            for (int kx = 0; kx < g.kw; ++kx) add_tap(p, ky - g.ph, kx - g.pw, ky, kx);
        close_phase(p);
    } else if (unit_input) {
        // 1x1 input, stride 1: out[ky-ph][kx-pw] = x * w[ky][kx]
        v.sy = 1; v.sx = 1; v.omy = 1; v.omx = 1; v.q_is_out = false;
        const int Ho = g.kh - 2 * g.ph + g.oph, Wo = g.kw - 2 * g.pw + g.opw;
        for (int ky = 0; ky < g.kh; ++ky)
            for (int kx = 0; kx < g.kw; ++kx) {
                const int oy = ky - g.ph, ox = kx - g.pw;
                if (oy < 0 || ox < 0 || oy >= Ho || ox >= Wo) continue;
                if (v.nphase >= kMaxPhases) { set_error("convT unit-input: too many phases"); return W2L_ERR_ARG; }
                ConvPhase& p = add_phase(oy, ox);
                add_tap(p, 0, 0, ky, kx);
                close_phase(p);
            }
    } else {
        // out oy = iy*s - p + ky.  Phase py = oy mod s: taps with ky == (py+p) mod s, iy = q + (py+p-ky)/s.
        v.sy = 1; v.sx = 1; v.omy = g.sh; v.omx = g.sw; v.q_is_out = false;
        if (g.sh * g.sw > kMaxPhases) { set_error("convT stride %dx%d unsupported", g.sh, g.sw); return W2L_ERR_ARG; }
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
        if (v.ph[i].ntaps > 64) { set_error("too many taps"); return W2L_ERR_ARG; }
    v.w_floats = woff;
    W2L_HIP_CHECK(hipMalloc(&v.taps_dev, sizeof(int) * 2 * (ntab > 0 ? ntab : 1)));
    W2L_HIP_CHECK(hipMalloc(&v.w_dev, sizeof(float) * (woff > 0 ? woff : 1)));
    // tap tables: [0,ntab) = (dy,dx), [ntab, 2ntab) = (ky,kx) for the packer
    W2L_HIP_CHECK(hipMemcpy(v.taps_dev, tapd, sizeof(int) * ntab, hipMemcpyHostToDevice));
    W2L_HIP_CHECK(hipMemcpy(v.taps_dev + ntab, tapk, sizeof(int) * ntab, hipMemcpyHostToDevice));
    v.ntab = ntab;
    if (pack_variant(c, v, c->weight_src, stream) != W2L_OK) return W2L_ERR_HIP;
    v.built = true;
    return W2L_OK;
}

static void free_variant(Variant& v) {
    if (v.taps_dev) (void)hipFree(v.taps_dev);
    if (v.w_dev) (void)hipFree(v.w_dev);
    if (v.w_split.load()) (void)hipFree(v.w_split.load());
    v.taps_dev = nullptr;
    v.w_dev = nullptr;
    v.w_split = nullptr;
    v.built = false;
}

static int geom_check(const w2l_conv_geom* g) {
    W2L_REQUIRE(g != nullptr, "geom is NULL");
    W2L_REQUIRE(g->cin >= 1 && g->cout >= 1, "bad channels %d->%d", g->cin, g->cout);
    W2L_REQUIRE(g->kh >= 1 && g->kw >= 1 && g->kh * g->kw <= kMaxTaps, "kernel %dx%d unsupported", g->kh, g->kw);
    W2L_REQUIRE(g->sh >= 1 && g->sw >= 1 && g->ph >= 0 && g->pw >= 0, "bad stride/pad");
    W2L_REQUIRE(g->act >= W2L_ACT_NONE && g->act <= W2L_ACT_LEAKY, "bad act %d", g->act);
    W2L_REQUIRE(g->transposed || (g->oph == 0 && g->opw == 0), "output_padding on a plain conv");
    return W2L_OK;
}

static int max_steps(const Variant& v) {
    int m = 0;
    for (int i = 0; i < v.nphase; ++i) m = v.ph[i].kp / kBK > m ? v.ph[i].kp / kBK : m;
    return m;
}

// Heuristic launch configuration (tile id, split-K factor) when no tuned/forced one is given: minimise
// rounds-of-256-CUs x tile area / measured tile efficiency; split K when the grid cannot fill the chip.
// whole_row: the epilogue needs every GEMM column of a row in one workgroup and no split-K (fused head)
static bool tile_allowed(const Variant& v, int tile, bool whole_row) {
    return tile >= 0 && tile < kNumTiles && (!whole_row || kTiles[tile].bn >= v.cout_p);
}

int conv_tp2_id();
int conv_wino4_id();
int conv_wino2q_id();
int conv_split_id(int tile);
int conv_wino2s_id();
int conv_tp2s_id();
int conv_stem7s_id();
int conv_k3s_id();
bool conv_family_excluded(int id);   // api.hip

// configuration ids conv_split_id(t), t < kNumTiles: implicit-GEMM tile t with the fp32 operands as three bf16 pieces (an fp32
// result from the bf16 matrix cores, see conv_igemm_bf16_kernel<.., 3>); -1 if `id` is not one of them
static int split_tile_of(int id) {
    const int t = id - conv_split_id(0);
    return (id >= 0 && t >= 0 && t < kNumTiles) ? t : -1;
}

// configuration ids kNumTiles + i select Winograd configuration i (conv_wino.hip) on eligible layers
static bool wino_allowed(const w2l_conv* c, int tile, int x_cs) {   // + wino_io_ok() on the output side
    if (!(c->wino_u != nullptr && c->precision == W2L_PREC_F32 && c->g.act != W2L_ACT_SIGMOID && tile >= kNumTiles &&
          tile < conv_tp2_id() && (x_cs & 3) == 0))
        return false;
    const int wc = tile - kNumTiles;
    if (wc < wino_num_cfgs()) return c->head_w == nullptr && wino_cfg_ok(wc, c->g.cin, c->g.cout);
    return wino2_ok(wc - wino_num_cfgs(), c->g.cin, c->g.cout, c->head_w ? c->head_c : 0);   // conv_wino2.hip (fuses a 1x1 head)
}

static void pick_config(const w2l_conv* c, const Variant& v, int M, bool whole_row, int* tile, int* ksplit) {
    int best = -1, best_ks = 1;
    double best_cost = 1e300;
    const int steps = max_steps(v);
    for (int i = 0; i < kNumTiles; ++i) {
        const TileCfg& tc = kTiles[i];
        if (!tile_allowed(v, i, whole_row)) continue;
        const long long blocks = (long long)ceil_div(M, tc.bm) * ceil_div(v.cout_p, tc.bn) * v.nphase;
        for (int ks = 1; ks <= 16; ks *= 2) {
            if (ks > 1 && (whole_row || v.pair > 1)) break;
            if (ks > 1 && (blocks * ks > 768 || steps / ks < 4)) break;
            const long long rounds = (blocks * ks + 255) / 256;
            const double per_block = (double)ceil_div(steps, ks) + 6.0;  // + prologue/epilogue in units of K-steps
            double cost = (double)rounds * per_block * tc.bm * tc.bn / tc.eff;
            if (ks > 1) cost += 1.0e5;  // the reduce launch: ~8 us = ~25 K-steps of a 128x128 block (profiles/r03/zq: priced at 2.0e6 the
                                        // heuristic never split K and a batch-2 step took longer than a batch-8 one)
            if (cost < best_cost) { best_cost = cost; best = i; best_ks = ks; }
        }
    }
    const bool ov = tile_allowed(v, c->tile_override, whole_row);
    *tile = ov ? c->tile_override : best;
    *ksplit = ov ? 1 : best_ks;
}

// grow-only device scratch for split-K partial sums (stream-ordered reuse across layers of one stream)
// Grow-only device scratch for split-K partial sums, ONE PER STREAM: reuse inside a stream is ordered by the stream itself,
// and two streams never share a buffer, so the library is re-entrant per (device, stream) as the header promises.  An
// outgrown buffer may still be in use by queued launches: it is left allocated (sizes converge after the first pass).
struct StreamWs {
    hipStream_t stream;
    float* ptr;
    size_t bytes;
};
static std::mutex g_ws_mutex;
static std::vector<StreamWs> g_ws_table;
static float* stream_workspace(hipStream_t stream, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (StreamWs& w : g_ws_table)
        if (w.stream == stream) {
            if (w.bytes >= bytes) return w.ptr;
            float* p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { set_error("hipMalloc(split-K workspace, %zu bytes) failed", bytes); return nullptr; }
            w.ptr = p;
            w.bytes = bytes;
            return p;
        }
    float* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { set_error("hipMalloc(split-K workspace, %zu bytes) failed", bytes); return nullptr; }
    g_ws_table.push_back(StreamWs{stream, p, bytes});
    return p;
}

float* conv_workspace(hipStream_t stream, size_t bytes) { return stream_workspace(stream, bytes); }

// Weight forms that only some configurations read (the split-operand kernels' bf16 planes) are built by the first launch that needs
// them: allocate, fill on that launch's stream, wait, publish.  One lock for all layers (it is taken once per layer and form), the
// pointer is published with release / read with acquire, and a stream that is being captured into a graph is refused - the build
// allocates and synchronises - with a message that says what to do (run the plan once before capturing it).
static std::mutex g_lazy_mutex;
template <class Fill>
static int lazy_weights(std::atomic<__bf16*>& slot, size_t elems, hipStream_t stream, const char* what, Fill fill) {
    if (slot.load(std::memory_order_acquire)) return W2L_OK;
    std::lock_guard<std::mutex> lock(g_lazy_mutex);
    if (slot.load(std::memory_order_acquire)) return W2L_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        set_error("%s are built by the first launch of the layer: run it once before capturing the stream", what);
        return W2L_ERR_ARG;
    }
    __bf16* p = nullptr;
    if (hipMalloc(&p, sizeof(__bf16) * (elems > 0 ? elems : 1)) != hipSuccess) {
        set_error("hipMalloc(%s) failed", what);
        return W2L_ERR_NOMEM;
    }
    if (fill(p) != W2L_OK || hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipFree(p);
        set_error("building %s failed", what);
        return W2L_ERR_HIP;
    }
    slot.store(p, std::memory_order_release);
    return W2L_OK;
}

// ---- shape-keyed launch configurations (the "tune table").  A launch whose (geometry, precision, residual, head, N, H, W)
// is in the table runs the recorded (configuration id, split-K); any other launch runs the pick_config heuristic.  Both are
// pure functions of the shape, so the summation order of a layer - and with it every bit of its output - is the same on every
// box and in every run.  The table is filled from a committed file by the host side (wav2lip_amd/tune_table.json), or by
// w2l_plan_autotune when a caller opts into stopwatch tuning.
typedef std::array<int, W2L_TUNE_KEY_INTS> TuneKey;
static std::mutex g_tune_mutex;
static std::map<TuneKey, std::pair<int, int>> g_tune;

static TuneKey tune_key(const w2l_conv* c, int N, int H, int W, bool has_res) {
    const w2l_conv_geom& g = c->g;
    return TuneKey{g.transposed, g.cin, g.cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.oph, g.opw,
                   c->precision, has_res ? 1 : 0, c->head_c, N, H, W};
}

static bool tune_lookup(const TuneKey& k, int* tile, int* ksplit) {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    auto it = g_tune.find(k);
    if (it == g_tune.end()) return false;
    *tile = it->second.first;
    *ksplit = it->second.second;
    return true;
}

void tune_store_launch(const w2l_conv* c, int N, int H, int W, bool has_res, int tile, int ksplit) {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    g_tune[tune_key(c, N, H, W, has_res)] = std::make_pair(tile, ksplit);
}

// flops_out != NULL: dry run - resolve the configuration exactly as a launch would, report the multiply-add work the matrix
// cores would EXECUTE (padded tiles, padded K, Winograd's 16 products per 2x2 tile; x2 = FLOPs) and launch nothing
extern "C" int w2l_conv_config_family(int id);   // api.hip
int conv_forward_impl(const w2l_conv* c, hipStream_t stream, int N, int H, int W, const float* x,
                      int x_cs, float* y, int y_cs, const float* res, int res_cs, int force_tile, int force_ksplit,
                      long long* flops_out, int* cfg_out) {
    W2L_REQUIRE(c && x && y, "NULL argument");
    if (!flops_out && flops_counting()) {   // w2l_flops_begin: resolve this launch once more as a dry run and book its work
        long long f = 0;
        int cfg[2];
        if (conv_forward_impl(c, stream, N, H, W, x, x_cs, y, y_cs, res, res_cs, force_tile, force_ksplit, &f, cfg) == W2L_OK)
            // 3: bf16 matrix-core work - the bf16c precision and EVERY split-operand family (5..9: ids of conv_igemm_bf16_kernel<..,3>,
            // conv_wino2s, conv_tp2s, conv_stem7s, conv_k3s), whose dry runs count six bf16 piece products per product
            flops_add(f, (c->precision == W2L_PREC_BF16 || w2l_conv_config_family(cfg[0]) >= 5) ? 3 : 0);
    }
    // a per-layer override of a family switched off by w2l_conv_exclude_families (W2L_EXACT) counts as no override: exact mode
    // is a property of the library, whichever way a launch names its configuration
    const int tile_override = (c->tile_override >= 0 && conv_family_excluded(c->tile_override)) ? -1 : c->tile_override;
    // an explicit id of a switched-off family counts as no choice at all (w2l_conv_exclude_families): the launch then resolves as an
    // unconfigured one does - the shape-keyed table first (W2L_SPLIT=0 overlays the fp32-pipe predecessors there), then the
    // heuristic.  (Dropping it AFTER the table lookup sent such launches - the borrowed per-plan lists of untuned batch sizes name
    // split ids - straight to the heuristic at split-K 1.)
    if (force_tile >= 0 && conv_family_excluded(force_tile)) { force_tile = -1; force_ksplit = 1; }
    if (force_tile < 0 && tile_override < 0) {   // no explicit choice: the shape-keyed table, else the heuristic below
        int tt, tk;
        if (tune_lookup(tune_key(c, N, H, W, res != nullptr), &tt, &tk)) { force_tile = tt; force_ksplit = tk; }
    }
    if (force_tile >= 0 && conv_family_excluded(force_tile)) { force_tile = -1; force_ksplit = 1; }   // a table entry of such a family
    W2L_REQUIRE(N >= 1 && H >= 1 && W >= 1, "bad shape N=%d H=%d W=%d", N, H, W);
    W2L_REQUIRE(x_cs >= c->cin_p && (x_cs & 3) == 0, "x_cs=%d must be a multiple of 4 and >= %d", x_cs, c->cin_p);
    W2L_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned");
    W2L_REQUIRE(c->head_w != nullptr || y_cs >= c->g.cout, "y_cs=%d < cout=%d", y_cs, c->g.cout);
    W2L_REQUIRE(res == nullptr || res_cs >= c->g.cout, "res_cs=%d < cout", res_cs);
    int Ho, Wo;
    if (w2l_conv_out_hw(&c->g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(Ho >= 1 && Wo >= 1, "empty output %dx%d", Ho, Wo);
    const bool unit = c->g.transposed && c->g.sh == 1 && c->g.sw == 1 && H == 1 && W == 1 && c->unit_in.built;
    const bool head = c->head_w != nullptr;
    W2L_REQUIRE(!head || (y_cs >= c->head_c && res == nullptr), "fused head: y_cs=%d < %d or residual given", y_cs, c->head_c);
    const bool y_vec_ok = (c->g.cout & 3) == 0 && (y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    const bool xp = c->xpair.built && !head && res == nullptr && (Wo % 2) == 0 && y_vec_ok;
    const Variant& v = unit ? c->unit_in : (xp ? c->xpair : c->generic);
    ConvKArgs a;
    a.x = x; a.y = y; a.res = res; a.w = v.w_dev; a.wsplit = v.w_split.load(std::memory_order_acquire); a.scale = c->scale; a.shift = c->shift; a.taps = v.taps_dev;
    a.N = N; a.H = H; a.W = W; a.cin_p = c->cin_p; a.x_cs = x_cs;
    a.Ho = Ho; a.Wo = Wo; a.cout = c->g.cout; a.cout_p = v.cout_p; a.y_cs = y_cs; a.res_cs = res_cs;
    a.pair = v.pair; a.ncols = v.pair * c->g.cout;
    a.head_w = c->head_w; a.head_b = c->head_b; a.head_c = c->head_c; a.head_act = c->head_act;
    if (unit) { a.Hq = 1; a.Wq = 1; }
    else if (v.q_is_out) { a.Hq = Ho; a.Wq = Wo; }
    else { a.Hq = ceil_div(Ho, v.omy); a.Wq = ceil_div(Wo, v.omx); }
    a.sy = v.sy; a.sx = v.sx; a.omy = v.omy; a.omx = v.omx;
    a.act = c->g.act;
    // float4 epilogue needs 16-byte aligned rows on y / res / scale / shift
    a.vec_epilogue = (y_vec_ok && (res == nullptr || ((res_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0)))
                         ? 1 : 0;
    const long long lim = 1ll << 31;  // buffer descriptors use 32-bit byte offsets with 0x80000000 as "out of range"
    W2L_REQUIRE(((long long)N * H * W * x_cs) * 4 < lim && ((long long)N * Ho * Wo * y_cs) * 4 < lim &&
                    (res == nullptr || ((long long)N * Ho * Wo * res_cs) * 4 < lim),
                "activation buffer larger than 2 GiB: split the batch");
    const long long M = (long long)N * a.Hq * a.Wq;
    W2L_REQUIRE(M < (1ll << 31) && (long long)N * H * W < (1ll << 31) && (long long)N * Ho * Wo < (1ll << 31), "tensor too large");
    a.M = (int)M;
    for (int i = 0; i < v.nphase; ++i) a.ph[i] = v.ph[i];
    // the direct 3x3 kernel with split operands for 32-cout layers (fuses the 1x1 head): only by explicit configuration id
    if (c->k3s_u != nullptr && c->precision == W2L_PREC_F32 && (x_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (head || ((y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)) &&
        (res == nullptr || ((res_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0)) && (!head || c->head_c <= 4) &&
        (force_tile == conv_k3s_id() || (force_tile < 0 && tile_override == conv_k3s_id()))) {
        if (cfg_out) { cfg_out[0] = conv_k3s_id(); cfg_out[1] = 1; }
        return k3s_launch(x, x_cs, y, y_cs, res, res_cs, c->k3s_u, c->scale, c->shift, head ? c->head_w : nullptr, c->head_b, c->head_c,
                          c->head_act, N, H, W, c->g.cin, c->g.act, stream, flops_out);
    }
    // the 7x7 first-layer kernel with split operands: only by explicit configuration id
    if (c->stem7s_u != nullptr && c->precision == W2L_PREC_F32 && !head && res == nullptr && x_cs >= 8 &&
        (force_tile == conv_stem7s_id() || (force_tile < 0 && tile_override == conv_stem7s_id()))) {
        if (cfg_out) { cfg_out[0] = conv_stem7s_id(); cfg_out[1] = 1; }
        return stem7s_launch(x, x_cs, y, y_cs, c->stem7s_u, c->scale, c->shift, N, H, W, c->g.act, stream, flops_out);
    }
    // fused-phase stride-2 transposed kernel with split operands: only by explicit configuration id
    if (c->tp2_u != nullptr && c->precision == W2L_PREC_F32 && !head && res == nullptr && !unit && tp2s_ok(c->g) && (y_cs & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (force_tile == conv_tp2s_id() || (force_tile < 0 && tile_override == conv_tp2s_id()))) {
        int ks = force_tile == conv_tp2s_id() ? force_ksplit : 1;      // a per-layer override (tile_override) carries no split-K
        if (ks < 1) ks = 1;
        if (ks > c->g.cin / 16) ks = c->g.cin / 16;
        const int sps = ceil_div(c->g.cin / 16, ks);
        ks = ceil_div(c->g.cin / 16, sps);
        if (cfg_out) { cfg_out[0] = conv_tp2s_id(); cfg_out[1] = ks; }
        float* ws = nullptr;
        const long long npix = (long long)N * 4 * H * W;
        if (!flops_out) {
            const int rc = lazy_weights(c->tp2s_u, (size_t)tp2s_u_elems(c->g.cin, c->g.cout), stream, "split-operand transposed weights",
                                        [&](__bf16* p) { return tp2s_pack(c->tp2_u, p, c->g.cin, c->g.cout, stream); });
            if (rc != W2L_OK) return rc;
            if (ks > 1) {
                ws = stream_workspace(stream, (size_t)ks * npix * c->g.cout * sizeof(float));
                if (!ws) return W2L_ERR_NOMEM;
            }
        }
        int used = 1;
        const int rc = tp2s_launch(x, x_cs, y, y_cs, c->tp2s_u.load(std::memory_order_acquire), c->scale, c->shift, N, H, W, c->g.cin,
                                   c->g.cout, c->g.act, ks, ws, &used, stream, flops_out);
        if (rc != W2L_OK || flops_out || used == 1) return rc;
        ReduceArgs r;
        r.ws = ws; r.y = y; r.res = nullptr; r.scale = c->scale; r.shift = c->shift;
        r.npix = npix; r.ksplit = used; r.cout = c->g.cout; r.cout_p = c->g.cout;
        r.y_cs = y_cs; r.res_cs = 0; r.act = c->g.act;
        long long g = (npix * c->g.cout + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, r);
        W2L_HIP_CHECK(hipGetLastError());
        return W2L_OK;
    }
    // fused-phase stride-2 transposed kernel: only by explicit configuration id (forced, per-layer override or tune table)
    if (c->tp2_u != nullptr && c->precision == W2L_PREC_F32 && !head && res == nullptr && !unit && (y_cs & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (force_tile == conv_tp2_id() || (force_tile < 0 && tile_override == conv_tp2_id()))) {
        if (cfg_out) { cfg_out[0] = conv_tp2_id(); cfg_out[1] = 1; }
        return tp2_launch(x, x_cs, y, y_cs, c->tp2_u, c->scale, c->shift, N, H, W, c->g.cin, c->g.cout, c->g.act, stream, flops_out);
    }
    // split-operand F(2x2,3x3) Winograd kernel: only by explicit configuration id (forced, per-layer override or tune table)
    if (c->wino_u != nullptr && c->precision == W2L_PREC_F32 && !head && c->g.act != W2L_ACT_SIGMOID && wino2s_ok(c->g.cin, c->g.cout) &&
        (x_cs & 3) == 0 && (y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (res == nullptr || ((res_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0)) &&
        (force_tile == conv_wino2s_id() || (force_tile < 0 && tile_override == conv_wino2s_id()))) {
        WinoKArgs wa;
        wa.x = x; wa.y = y; wa.res = res; wa.u = c->wino_u; wa.scale = c->scale; wa.shift = c->shift;
        wa.N = N; wa.H = H; wa.W = W; wa.cin = c->g.cin; wa.x_cs = x_cs;
        wa.cout = c->g.cout; wa.y_cs = y_cs; wa.res_cs = res_cs; wa.act = c->g.act;
        if (cfg_out) { cfg_out[0] = conv_wino2s_id(); cfg_out[1] = 1; }
        if (!flops_out) {
            const int rc = lazy_weights(c->wino2s_u, (size_t)wino2s_u_elems(c->g.cin, c->g.cout), stream, "split-operand F(2x2) weights",
                                        [&](__bf16* p) { return wino2s_pack(c->wino_u, p, c->g.cin, c->g.cout, stream); });
            if (rc != W2L_OK) return rc;
        }
        return wino2s_launch(wa, c->wino2s_u.load(std::memory_order_acquire), stream, flops_out);
    }
    // F(4x4,3x3) Winograd kernel: only by explicit configuration id (forced, per-layer override or tune table)
    if (c->wino4_u != nullptr && c->precision == W2L_PREC_F32 && !head && c->g.act != W2L_ACT_SIGMOID && (x_cs & 3) == 0 &&
        (y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (res == nullptr || ((res_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0)) &&
        (force_tile == conv_wino4_id() || (force_tile < 0 && tile_override == conv_wino4_id()))) {
        WinoKArgs wa;
        wa.x = x; wa.y = y; wa.res = res; wa.u = c->wino4_u; wa.scale = c->scale; wa.shift = c->shift;
        wa.N = N; wa.H = H; wa.W = W; wa.cin = c->g.cin; wa.x_cs = x_cs;
        wa.cout = c->g.cout; wa.y_cs = y_cs; wa.res_cs = res_cs; wa.act = c->g.act;
        if (cfg_out) { cfg_out[0] = conv_wino4_id(); cfg_out[1] = 1; }
        return wino4_launch(wa, c->wino4_u, stream, flops_out);
    }
    // quarter-split F(2x2) kernel (32-cout layers, fused head allowed): only by explicit configuration id
    if (c->wino_u != nullptr && c->precision == W2L_PREC_F32 && c->g.act != W2L_ACT_SIGMOID && (x_cs & 3) == 0 &&
        (y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (res == nullptr || ((res_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0)) &&
        wino2q_ok(c->g.cin, c->g.cout, c->head_w ? c->head_c : 0) &&
        (force_tile == conv_wino2q_id() || (force_tile < 0 && tile_override == conv_wino2q_id()))) {
        WinoKArgs wa;
        wa.x = x; wa.y = y; wa.res = res; wa.u = c->wino_u; wa.scale = c->scale; wa.shift = c->shift;
        wa.N = N; wa.H = H; wa.W = W; wa.cin = c->g.cin; wa.x_cs = x_cs;
        wa.cout = c->g.cout; wa.y_cs = y_cs; wa.res_cs = res_cs; wa.act = c->g.act;
        if (cfg_out) { cfg_out[0] = conv_wino2q_id(); cfg_out[1] = 1; }
        return wino2q_launch(wa, c->head_w, c->head_b, c->head_c, c->head_act, stream, flops_out);
    }
    {   // Winograd path: forced configuration id, or the heuristic default when the grid fills the chip
        int wt = -1;
        // the Winograd epilogue moves float4 rows: y and res must be 16-byte friendly (true for every plan buffer)
        const bool wino_io_ok = (y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                                (res == nullptr || ((res_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
        if (!wino_io_ok) wt = -2;
        else if (wino_allowed(c, force_tile, x_cs)) wt = force_tile;
        else if (force_tile < 0 && wino_allowed(c, tile_override, x_cs)) wt = tile_override;
        else if (force_tile < 0 && tile_override < 0 && wino_allowed(c, kNumTiles, x_cs) &&
                 (long long)N * ((H + 1) / 2) * ((W + 1) / 2) / 64 * (c->g.cout / 64) >= 192) wt = kNumTiles;
        if (wt >= 0) {
            WinoKArgs wa;
            wa.x = x; wa.y = y; wa.res = res; wa.u = c->wino_u; wa.scale = c->scale; wa.shift = c->shift;
            wa.N = N; wa.H = H; wa.W = W; wa.cin = c->g.cin; wa.x_cs = x_cs;
            wa.cout = c->g.cout; wa.y_cs = y_cs; wa.res_cs = res_cs; wa.act = c->g.act;
            if (cfg_out) { cfg_out[0] = wt; cfg_out[1] = 1; }
            if (wt - kNumTiles >= wino_num_cfgs())
                return wino2_launch(wt - kNumTiles - wino_num_cfgs(), wa, c->head_w, c->head_b, c->head_c, c->head_act, stream,
                                    flops_out);
            return wino_launch(wt - kNumTiles, wa, stream, flops_out);
        }
    }
    int ti, ks;
    pick_config(c, v, a.M, head, &ti, &ks);
    W2L_REQUIRE(ti >= 0, "fused head: cout=%d does not fit one tile", c->g.cout);
    if (tile_allowed(v, force_tile, head)) { ti = force_tile; ks = force_ksplit >= 1 ? force_ksplit : 1; }
    // split-operand kernel: only by explicit configuration id (forced, per-layer override or tune table), fp32 layers only
    bool split = false;
    if (c->precision == W2L_PREC_F32) {
        const int st = split_tile_of(force_tile >= 0 ? force_tile : tile_override);
        if (st >= 0 && tile_allowed(v, st, head)) {
            split = true;
            ti = st;
            ks = force_tile >= 0 ? (force_ksplit >= 1 ? force_ksplit : 1) : 1;
        }
    }
    if (head || v.pair > 1) ks = 1;
    const TileCfg& tc = kTiles[ti];
    const int steps = max_steps(v);
    if (ks > steps) ks = steps;
    if (ks < 1) ks = 1;
    a.ksplit = ks;
    a.steps_per_split = ceil_div(steps, ks);
    a.ksplit = ceil_div(steps, a.steps_per_split);   // drop empty trailing splits
    a.ws = nullptr;
    const long long npix = (long long)N * Ho * Wo;
    if (cfg_out) { cfg_out[0] = split ? conv_split_id(ti) : ti; cfg_out[1] = a.ksplit; }
    if (flops_out) {
        long long kp = 0;
        for (int i = 0; i < v.nphase; ++i) kp += v.ph[i].kp;
        // the split kernel's count is bf16 matrix-core work: six piece products per fp32 product
        *flops_out = (split ? 6ll : 1ll) * 2ll * ceil_div(a.M, tc.bm) * ceil_div(v.cout_p, tc.bn) * tc.bm * tc.bn * kp;
        return W2L_OK;
    }
    if (a.ksplit > 1) {
        a.ws = stream_workspace(stream, (size_t)a.ksplit * npix * v.cout_p * sizeof(float));
        if (!a.ws) return W2L_ERR_NOMEM;
        if (c->g.transposed && !unit && (Ho % v.omy || Wo % v.omx)) {
            // phases may not cover every output pixel of a ragged transposed conv: start the partials from zero
            W2L_HIP_CHECK(hipMemsetAsync(a.ws, 0, (size_t)a.ksplit * npix * v.cout_p * sizeof(float), stream));
        }
    }
    a.tiles_m = ceil_div(a.M, tc.bm);
    a.tiles_n = ceil_div(v.cout_p, tc.bn);
    const long long nblk = (long long)a.tiles_m * a.tiles_n;
    W2L_REQUIRE(nblk * v.nphase < (1ll << 31), "grid too large");
    {   // which operand an XCD's L2 keeps between workgroups (igemm_block_coords): the order with the fewest bytes fetched under a
        // model that reproduces the measured FETCH_SIZE ranking of every launch of the batch-128 plan (profiles/r05/l_*).  x = input
        // bytes, w = weight bytes; every phase is one pass over x unless the phases of a group of M-tiles run back to back; an XCD
        // fetches every cout-tile's weights that its range touches.  The bf16-storage training launches keep the plain order (the
        // others are measured on the inference plan only).
        long long wbytes = 0;
        for (int i = 0; i < v.nphase; ++i) wbytes += (long long)v.cout_p * v.ph[i].kp;
        wbytes *= split ? 6 : 4;
        const long long xbytes = (long long)N * H * W * c->cin_p * 4;
        a.order = kOrderPhaseMajor;
        a.order_r = 32 / a.tiles_n > 0 ? 32 / a.tiles_n : 1;   // one group's phase = one round of workgroups on an XCD's 32 CUs
        if (c->precision != W2L_PREC_BF16 && !unit) {
            const long long plain = fetch_cout_fastest(xbytes, wbytes, a.tiles_n, v.nphase);
            const long long cout_slowest = fetch_cout_slowest(xbytes, wbytes, a.tiles_n, v.nphase);
            const long long blocked = v.nphase > 1 && nblk >= 512 ? xbytes * 3 / 2 + 8 * wbytes : plain;   // >= 2 groups per XCD
            if (cout_slowest < plain && cout_slowest <= blocked) a.order = kOrderCoutSlowest;
            else if (blocked < plain) a.order = kOrderPhaseBlocked;
        }
#ifdef W2L_ORDER_ENV
        static const char* oe = getenv("W2L_IGEMM_ORDER");   // A/B builds: "0" = the plain order everywhere
        if (oe && oe[0] == '0') a.order = kOrderPhaseMajor;
        static const char* re = getenv("W2L_IGEMM_R");       // blocks per phase of a phase-blocked group
        if (re) a.order_r = atoi(re) / a.tiles_n > 0 ? atoi(re) / a.tiles_n : 1;
#endif
    }
    if (split) {
        // once per layer: filled and complete before the pointer is seen by a launch on any other stream
        const int rc = lazy_weights(v.w_split, 3 * (size_t)v.w_floats, stream, "split-operand weights", [&](__bf16* p) {
            SplitWArgs sa;
            sa.w = v.w_dev; sa.out = p; sa.nphase = v.nphase;
            long long maxtot = 1;
            for (int i = 0; i < v.nphase; ++i) {
                sa.off[i] = v.ph[i].w_off;
                sa.count[i] = (long long)v.cout_p * v.ph[i].kp;
                if (sa.count[i] > maxtot) maxtot = sa.count[i];
            }
            int blocks = (int)((maxtot + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(split_weights_kernel, dim3(blocks, v.nphase), dim3(256), 0, stream, sa);
            W2L_HIP_CHECK(hipGetLastError());
            v.split_fresh = true;
            return W2L_OK;
        });
        if (rc != W2L_OK) return rc;
        if (!v.split_fresh && split_variant(v, stream) != W2L_OK) return W2L_ERR_HIP;
        a.wsplit = v.w_split.load(std::memory_order_acquire);
    }
    if (split)
        hipLaunchKernelGGL(tc.kernel_split, dim3((unsigned)nblk, v.nphase, a.ksplit), dim3(tc.threads_split), tc.lds_split, stream, a);
    else if (c->precision == W2L_PREC_BF16)
        hipLaunchKernelGGL(tc.kernel_bf16, dim3((unsigned)nblk, v.nphase, a.ksplit), dim3(256), tc.lds_bf16, stream, a);
    else
        hipLaunchKernelGGL(tc.kernel, dim3((unsigned)nblk, v.nphase, a.ksplit), dim3(256), tc.lds, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    if (a.ksplit > 1) {
        ReduceArgs r;
        r.ws = a.ws; r.y = y; r.res = res; r.scale = c->scale; r.shift = c->shift;
        r.npix = npix; r.ksplit = a.ksplit; r.cout = c->g.cout; r.cout_p = v.cout_p;
        r.y_cs = y_cs; r.res_cs = res_cs; r.act = c->g.act;
        long long g = (npix * c->g.cout + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, r);
        W2L_HIP_CHECK(hipGetLastError());
    }
    return W2L_OK;
}

// + conv_tp2.hip, conv_wino4.hip, wino2q, then the kNumTiles split-operand ids (appended: the ids of every earlier family keep
// their values, so committed tune tables stay valid)
int conv_num_tiles() { return kNumTiles + wino_num_cfgs() + wino2_num_cfgs() + 3 + kNumTiles + 4; }
int conv_wino2s_id() { return conv_split_id(kNumTiles); }   // appended after the split ids: every earlier id keeps its meaning (committed tables)
int conv_tp2s_id() { return conv_split_id(kNumTiles) + 1; }
int conv_stem7s_id() { return conv_split_id(kNumTiles) + 2; }
int conv_k3s_id() { return conv_split_id(kNumTiles) + 3; }   // appended last
int conv_split_id(int tile) { return kNumTiles + wino_num_cfgs() + wino2_num_cfgs() + 3 + tile; }
int conv_tp2_id() { return kNumTiles + wino_num_cfgs() + wino2_num_cfgs(); }
int conv_wino4_id() { return conv_tp2_id() + 1; }
int conv_wino2q_id() { return conv_tp2_id() + 2; }
int conv_num_igemm_tiles() { return kNumTiles; }

static int init_kernel_attrs() {
    static std::mutex m;
    static bool done = false;
    std::lock_guard<std::mutex> lock(m);
    if (done) return W2L_OK;
    for (int i = 0; i < kNumTiles; ++i) {
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kTiles[i].kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kTiles[i].lds));
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kTiles[i].kernel_bf16),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kTiles[i].lds_bf16));
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kTiles[i].kernel_split),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kTiles[i].lds_split));
    }
    if (wino_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (wino2_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (tp2_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (tp2s_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (stem7s_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (k3s_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (wino2q_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (wino4_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    done = true;   // only after EVERY family's dynamic-LDS attribute is set: a failed init is retried (and reported) by the next call
    return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

extern "C" {

int w2l_conv_cin_padded(int cin) { return round_up(cin, 4); }
int w2l_conv_num_tiles(void) { return conv_num_tiles(); }

int w2l_conv_out_hw(const w2l_conv_geom* g, int H, int W, int* Ho, int* Wo) {
    if (geom_check(g) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(Ho && Wo, "NULL output");
    if (!g->transposed) {
        *Ho = (H + 2 * g->ph - g->kh) / g->sh + 1;
        *Wo = (W + 2 * g->pw - g->kw) / g->sw + 1;
        W2L_REQUIRE(H + 2 * g->ph >= g->kh && W + 2 * g->pw >= g->kw, "input %dx%d smaller than kernel", H, W);
    } else {
        *Ho = (H - 1) * g->sh - 2 * g->ph + g->kh + g->oph;
        *Wo = (W - 1) * g->sw - 2 * g->pw + g->kw + g->opw;
    }
    return W2L_OK;
}

long long w2l_conv_macs(const w2l_conv_geom* g, int N, int H, int W) {
    int Ho, Wo;
    if (w2l_conv_out_hw(g, H, W, &Ho, &Wo) != W2L_OK) return -1;
    const long long taps = (long long)g->kh * g->kw * g->cin * g->cout;
    return g->transposed ? taps * N * H * W : taps * N * Ho * Wo;
}

int w2l_conv_create(const w2l_conv_geom* g, const float* weight, const float* scale, const float* shift,
                    void* stream, w2l_conv_t** out) {
    if (geom_check(g) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(weight && scale && shift && out, "NULL argument");
    if (init_kernel_attrs() != W2L_OK) return W2L_ERR_HIP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    w2l_conv* c = new (std::nothrow) w2l_conv();
    if (!c) { set_error("out of host memory"); return W2L_ERR_NOMEM; }
    c->g = *g;
    c->cin_p = round_up(g->cin, 4);
    c->cout_p = round_up(g->cout, 32);
    c->weight_src = weight;
    int rc = W2L_OK;
    do {
        if (hipMalloc(&c->scale, sizeof(float) * g->cout) != hipSuccess ||
            hipMalloc(&c->shift, sizeof(float) * g->cout) != hipSuccess) {
            set_error("hipMalloc(scale/shift) failed");
            rc = W2L_ERR_NOMEM;
            break;
        }
        if (hipMemcpyAsync(c->scale, scale, sizeof(float) * g->cout, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(c->shift, shift, sizeof(float) * g->cout, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            set_error("copy of scale/shift failed");
            rc = W2L_ERR_HIP;
            break;
        }
        rc = build_variant(c, c->generic, kGeneric, s);
        if (rc != W2L_OK) break;
        if (g->transposed && g->sh == 1 && g->sw == 1 && g->kh * g->kw <= kMaxPhases)
            rc = build_variant(c, c->unit_in, kUnitInput, s);
        if (rc != W2L_OK) break;
        // 3x3 / stride 1 / pad 1: Winograd; the transposed form (the data gradient of such a conv) is the same conv with the
        // kernel flipped and the channel roles swapped, which only changes how the weight tensor is read by the packer
        if (g->kh == 3 && g->kw == 3 && g->sh == 1 && g->sw == 1 && g->ph == 1 && g->pw == 1 && g->oph == 0 && g->opw == 0 &&
            (wino_cfg_ok(0, g->cin, g->cout) || wino_cfg_ok(1, g->cin, g->cout) || wino2_ok(0, g->cin, g->cout, 0) ||
             wino2_ok(1, g->cin, g->cout, 0))) {
            if (hipMalloc(&c->wino_u, sizeof(float) * wino_u_floats(g->cin, g->cout)) != hipSuccess) {
                set_error("hipMalloc(winograd weights) failed");
                rc = W2L_ERR_NOMEM;
                break;
            }
            rc = wino_pack(weight, c->wino_u, g->cin, g->cout, g->transposed, s);
            if (rc != W2L_OK) break;
        }
        if (g->kh == 3 && g->kw == 3 && g->sh == 1 && g->sw == 1 && g->ph == 1 && g->pw == 1 && g->oph == 0 && g->opw == 0 &&
            c->precision == W2L_PREC_F32 && wino4_ok(g->cin, g->cout)) {
            if (hipMalloc(&c->wino4_u, sizeof(float) * wino4_u_floats(g->cin, g->cout)) != hipSuccess) {
                set_error("hipMalloc(F(4x4) winograd weights) failed");
                rc = W2L_ERR_NOMEM;
                break;
            }
            rc = wino4_pack(weight, c->wino4_u, g->cin, g->cout, g->transposed, s);
            if (rc != W2L_OK) break;
        }
        if (!g->transposed && g->sw == 1 && g->cout <= 16 && (g->cout & 3) == 0 && g->kh * (g->kw + 1) <= 64)
            rc = build_variant(c, c->xpair, kXPair, s);
        if (rc != W2L_OK) break;
        if (tp2_ok(*g)) {
            if (hipMalloc(&c->tp2_u, sizeof(float) * tp2_u_floats(g->cin, g->cout)) != hipSuccess) {
                set_error("hipMalloc(transposed-conv fragment weights) failed");
                rc = W2L_ERR_NOMEM;
                break;
            }
            rc = tp2_pack(weight, c->tp2_u, g->cin, g->cout, s);
        }
        if (rc == W2L_OK && stem7s_ok(*g)) {
            if (hipMalloc(&c->stem7s_u, sizeof(__bf16) * stem7s_u_elems()) != hipSuccess) {
                set_error("hipMalloc(first-layer split weights) failed");
                rc = W2L_ERR_NOMEM;
                break;
            }
            rc = stem7s_pack(weight, c->stem7s_u, g->cin, s);
        }
        if (rc == W2L_OK && k3s_ok(*g)) {
            if (hipMalloc(&c->k3s_u, sizeof(__bf16) * k3s_u_elems(g->cin)) != hipSuccess) {
                set_error("hipMalloc(direct 3x3 split weights) failed");
                rc = W2L_ERR_NOMEM;
                break;
            }
            rc = k3s_pack(weight, c->k3s_u, g->cin, s);
        }
    } while (0);
    // the packer reads the caller's weight tensor: finish before handing control back
    if (rc == W2L_OK && hipStreamSynchronize(s) != hipSuccess) { set_error("sync after weight packing failed"); rc = W2L_ERR_HIP; }
    c->weight_src = nullptr;
    if (rc != W2L_OK) { w2l_conv_destroy(c); return rc; }
    *out = c;
    return W2L_OK;
}

int w2l_conv_update(w2l_conv_t* c, const float* weight, const float* scale, const float* shift, void* stream) {
    W2L_REQUIRE(c, "NULL conv");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (scale) W2L_HIP_CHECK(hipMemcpyAsync(c->scale, scale, sizeof(float) * c->g.cout, hipMemcpyDeviceToDevice, s));
    if (shift) W2L_HIP_CHECK(hipMemcpyAsync(c->shift, shift, sizeof(float) * c->g.cout, hipMemcpyDeviceToDevice, s));
    if (weight) {
        Variant* vs[3] = {&c->generic, &c->unit_in, &c->xpair};
        for (Variant* v : vs)
            if (v->built && pack_variant(c, *v, weight, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->wino_u && wino_pack(weight, c->wino_u, c->g.cin, c->g.cout, c->g.transposed, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->tp2_u && tp2_pack(weight, c->tp2_u, c->g.cin, c->g.cout, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->wino4_u && wino4_pack(weight, c->wino4_u, c->g.cin, c->g.cout, c->g.transposed, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->wino2s_u.load() && wino2s_pack(c->wino_u, c->wino2s_u.load(), c->g.cin, c->g.cout, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->tp2s_u.load() && tp2s_pack(c->tp2_u, c->tp2s_u.load(), c->g.cin, c->g.cout, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->stem7s_u && stem7s_pack(weight, c->stem7s_u, c->g.cin, s) != W2L_OK) return W2L_ERR_HIP;
        if (c->k3s_u && k3s_pack(weight, c->k3s_u, c->g.cin, s) != W2L_OK) return W2L_ERR_HIP;
    }
    return W2L_OK;
}

int w2l_conv_destroy(w2l_conv_t* c) {
    if (!c) return W2L_OK;
    free_variant(c->generic);
    free_variant(c->unit_in);
    free_variant(c->xpair);
    if (c->wino_u) (void)hipFree(c->wino_u);
    if (c->tp2_u) (void)hipFree(c->tp2_u);
    if (c->wino4_u) (void)hipFree(c->wino4_u);
    if (c->wino2s_u.load()) (void)hipFree(c->wino2s_u.load());
    if (c->tp2s_u.load()) (void)hipFree(c->tp2s_u.load());
    if (c->stem7s_u) (void)hipFree(c->stem7s_u);
    if (c->k3s_u) (void)hipFree(c->k3s_u);
    if (c->head_w) (void)hipFree(c->head_w);
    if (c->head_b) (void)hipFree(c->head_b);
    if (c->scale) (void)hipFree(c->scale);
    if (c->shift) (void)hipFree(c->shift);
    delete c;
    return W2L_OK;
}

int w2l_conv_attach_head(w2l_conv_t* c, const float* head_weight, const float* head_bias, int head_c, int head_act,
                         void* stream) {
    W2L_REQUIRE(c && head_weight, "NULL argument");
    W2L_REQUIRE(head_c >= 1 && head_c <= 4, "head_c=%d: 1..4 output channels supported", head_c);
    W2L_REQUIRE(head_act >= W2L_ACT_NONE && head_act <= W2L_ACT_LEAKY, "bad head act %d", head_act);
    W2L_REQUIRE((c->g.cout & 3) == 0 && c->generic.cout_p <= 128, "fused head needs cout %% 4 == 0 and cout <= 128 (got %d)", c->g.cout);
    W2L_REQUIRE(c->head_w == nullptr, "head already attached");
    hipStream_t s = static_cast<hipStream_t>(stream);
    W2L_HIP_CHECK(hipMalloc(&c->head_w, sizeof(float) * head_c * c->g.cout));
    W2L_HIP_CHECK(hipMemcpyAsync(c->head_w, head_weight, sizeof(float) * head_c * c->g.cout, hipMemcpyDeviceToDevice, s));
    if (head_bias) {
        W2L_HIP_CHECK(hipMalloc(&c->head_b, sizeof(float) * head_c));
        W2L_HIP_CHECK(hipMemcpyAsync(c->head_b, head_bias, sizeof(float) * head_c, hipMemcpyDeviceToDevice, s));
    }
    W2L_HIP_CHECK(hipStreamSynchronize(s));
    c->head_c = head_c;
    c->head_act = head_act;
    return W2L_OK;
}

int w2l_conv_set_precision(w2l_conv_t* c, int precision) {
    W2L_REQUIRE(c, "NULL conv");
    W2L_REQUIRE(precision == W2L_PREC_F32 || precision == W2L_PREC_BF16, "bad precision %d", precision);
    c->precision = precision;
    return W2L_OK;
}

int w2l_conv_set_tile(w2l_conv_t* c, int tile_id) {
    W2L_REQUIRE(c, "NULL conv");
    W2L_REQUIRE(tile_id >= -1 && tile_id < conv_num_tiles(), "tile id %d out of range", tile_id);
    c->tile_override = tile_id;
    return W2L_OK;
}

int w2l_conv_forward(const w2l_conv_t* c, void* stream, int N, int H, int W, const float* x, int x_cs,
                     float* y, int y_cs, const float* res, int res_cs) {
    return conv_forward_impl(c, static_cast<hipStream_t>(stream), N, H, W, x, x_cs, y, y_cs, res, res_cs, -1, 1, nullptr, nullptr);
}

int w2l_tune_key_ints(void) { return W2L_TUNE_KEY_INTS; }

int w2l_tune_set(const int* key, int tile, int ksplit) {
    W2L_REQUIRE(key, "NULL key");
    W2L_REQUIRE(tile >= 0 && tile < conv_num_tiles() && ksplit >= 1 && ksplit <= 64, "bad config (%d, %d)", tile, ksplit);
    TuneKey k;
    for (int i = 0; i < W2L_TUNE_KEY_INTS; ++i) k[i] = key[i];
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    g_tune[k] = std::make_pair(tile, ksplit);
    return W2L_OK;
}

// Can configuration id `tile` run a launch with this key at all?  Pure host arithmetic on the key - the same predicates
// w2l_conv_create / conv_forward_impl apply to a live handle: an id that fails here would silently fall through to the
// heuristic at launch time, so a table entry carrying it misdescribes what runs (tools/make_tune_table.py and the CPU table
// test reject such entries).
int w2l_tune_entry_applicable(const int* key, int tile) {
    W2L_REQUIRE(key, "NULL key");
    if (tile < 0 || tile >= conv_num_tiles()) return 0;
    w2l_conv_geom g;
    g.transposed = key[0]; g.cin = key[1]; g.cout = key[2]; g.kh = key[3]; g.kw = key[4]; g.sh = key[5]; g.sw = key[6];
    g.ph = key[7]; g.pw = key[8]; g.oph = key[9]; g.opw = key[10]; g.act = W2L_ACT_NONE;
    const int prec = key[11], has_res = key[12], head_c = key[13];
    if (tile < kNumTiles) return (head_c == 0 || kTiles[tile].bn >= round_up(g.cout, 32)) ? 1 : 0;
    if (prec != W2L_PREC_F32) return 0;                       // every other family is fp32-only
    if (split_tile_of(tile) >= 0) return (head_c == 0 || kTiles[split_tile_of(tile)].bn >= round_up(g.cout, 32)) ? 1 : 0;
    if (tile == conv_tp2_id()) return (tp2_ok(g) && head_c == 0 && !has_res) ? 1 : 0;
    if (tile == conv_tp2s_id()) return (tp2s_ok(g) && head_c == 0 && !has_res) ? 1 : 0;
    if (tile == conv_stem7s_id()) return (stem7s_ok(g) && head_c == 0 && !has_res) ? 1 : 0;
    if (tile == conv_k3s_id()) return (k3s_ok(g) && head_c <= 4) ? 1 : 0;
    if (tile == conv_wino2s_id())
        return (g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 && g.oph == 0 && g.opw == 0 &&
                wino2s_ok(g.cin, g.cout) && head_c == 0) ? 1 : 0;
    const bool k3 = g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 && g.oph == 0 && g.opw == 0;
    if (!k3) return 0;
    const bool has_u = wino_cfg_ok(0, g.cin, g.cout) || wino_cfg_ok(1, g.cin, g.cout) || wino2_ok(0, g.cin, g.cout, 0) ||
                       wino2_ok(1, g.cin, g.cout, 0);        // w2l_conv_create packs the F(2x2) weights only then
    if (tile == conv_wino4_id()) return (wino4_ok(g.cin, g.cout) && head_c == 0) ? 1 : 0;
    if (tile == conv_wino2q_id()) return (has_u && wino2q_ok(g.cin, g.cout, head_c)) ? 1 : 0;
    const int wc = tile - kNumTiles;
    if (wc < wino_num_cfgs()) return (has_u && head_c == 0 && wino_cfg_ok(wc, g.cin, g.cout)) ? 1 : 0;
    return (has_u && wino2_ok(wc - wino_num_cfgs(), g.cin, g.cout, head_c)) ? 1 : 0;
}

int w2l_tune_clear(void) {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    g_tune.clear();
    return W2L_OK;
}

int w2l_tune_count(void) {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    return (int)g_tune.size();
}

int w2l_tune_export(int* out, int cap_entries) {
    W2L_REQUIRE(out || cap_entries == 0, "NULL out");
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    int n = 0;
    for (const auto& kv : g_tune) {
        if (n >= cap_entries) break;
        int* o = out + (long long)n * (W2L_TUNE_KEY_INTS + 2);
        for (int i = 0; i < W2L_TUNE_KEY_INTS; ++i) o[i] = kv.first[i];
        o[W2L_TUNE_KEY_INTS] = kv.second.first;
        o[W2L_TUNE_KEY_INTS + 1] = kv.second.second;
        ++n;
    }
    return n;
}

}  // extern "C"

// Weight gradient of a convolution / transposed convolution for gfx950 as an fp32 GEMM on the matrix cores
// (the autograd backward of nn.Conv2d / nn.ConvTranspose2d in the reference's training loops,
// wav2lip_train.py:229, color_syncnet_train.py:164, hq_wav2lip_train.py:231,256 -> models/conv.py:8,24,36).
//
// Both layer kinds reduce to ONE form.  Let P be the tensor on the coarse grid (the conv's output gradient dz, or the
// transposed conv's input x) and Q the tensor on the fine grid (the conv's input x, or the transposed conv's output
// gradient dz); a P pixel (py, px) touches Q pixels (py*s - p + ky, px*s - p + kx).  Then
//     dW[cp][cq][ky][kx] = sum over P pixels of  P[pix][cp] * Q[pix*s - p + (ky,kx)][cq]
// which is the torch weight layout in both cases ([cout][cin][kh][kw] resp. [cin][cout][kh][kw]).
//
// GEMM view: M = CP, N = (tap, cq) with cq fastest, K = pixels of the P grid.  Both operands are stored pixel-major
// in HBM (NHWC), i.e. K-major with M / N contiguous: tiles are staged into LDS as [k][m] / [k][n] rows with coalesced
// float4 loads (padding taps and ragged edges read zero through out-of-range buffer offsets), and a wave reads its
// fragments with ds_read_b64: the two floats a lane gets are rows 2r and 2r+1 of two row-interleaved 32x32 MFMA tiles,
// so one 64x64 wave tile costs one b64 read per operand per k-pair and every LDS read is conflict-free
// (one 256-B row per half-wave).  K is cut into `ksplit` pixel ranges (gridDim.z); partial sums go to a workspace
// and a second kernel adds them in a fixed order (deterministic) while scattering into the torch layout.
#include <stdlib.h>
#include <type_traits>

#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kGOob = 0x80000000u;

struct WgradKArgs {
    const float* p;   // [K][p_cs], CPp readable channels per pixel
    const float* q;   // [N][Hq][Wq][q_cs], CQp readable channels per pixel
    float* ws;        // [ksplit][Mp][Np] partial sums
    int N, Hp, Wp, CPp, p_cs;
    int Hq, Wq, CQp, q_cs;
    int sy, sx, py, px;   // q pixel = p pixel * s - pad + (ky, kx)
    int kh, kw;
    int K;            // N*Hp*Wp
    int ncols;        // kh*kw*CQp
    int Mp, Np;       // padded to whole tiles
    int tiles_n;
    int chunk;        // pixels per K split (multiple of the K-step)
    float inv_hw, inv_w;   // 1/(Hp*Wp), 1/Wp for the float-reciprocal index split (exact after one fix-up step)
};

// q = a / d, r = a % d for 0 <= a < 2^31 and a quotient below 2^22: float reciprocal estimate + one correction step
__device__ __forceinline__ void fast_divmod(int a, int d, float inv_d, int& q, int& r) {
    q = (int)((float)a * inv_d);
    r = a - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;   // branch-free single correction
    q += hi - lo;
    r += (lo - hi) * d;
}

__device__ __forceinline__ f32x4 gload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

template <int BM, int BN, int BK>
constexpr int wgrad_lds_bytes() {
    return (2 * BK * BM + 2 * BK * BN) * 4 + 2 * BK * 4 * 4;
}

// Wave tile = (32*IA) x (32*JB): IA / JB row-interleaved 32x32 MFMA tiles per operand (fragment reads of IA / JB floats).
template <int IA, int JB, int WM, int WN, int BK>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_kernel(const WgradKArgs a) {
    static_assert(WM * WN == 4 && (IA == 1 || IA == 2) && (JB == 1 || JB == 2), "4 waves; 1 or 2 interleaved tiles per operand");
    constexpr int BM = 32 * IA * WM, BN = 32 * JB * WN;
    constexpr int CGA = BM / 4, RA = 256 / CGA, PA = (BK + RA - 1) / RA;   // float4 column groups, rows per pass, passes
    constexpr int CGB = BN / 4, RB = 256 / CGB, PB = (BK + RB - 1) / RB;
    static_assert((RA >= BK || BK % RA == 0) && (RB >= BK || BK % RB == 0), "staging must tile the K-step");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);           // [2][BK][BM]
    float* Bs = As + 2 * BK * BM;                         // [2][BK][BN]
    int* s_rows = reinterpret_cast<int*>(Bs + 2 * BK * BN);  // [2][BK][4]: p pixel (or -1), q base pixel, iy0, ix0

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = bid % a.tiles_n;
    const int tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int k0 = blockIdx.z * a.chunk;
    const int k1 = min(a.K, k0 + a.chunk);
    const int nsteps = (k1 - k0 + BK - 1) / BK;
    const int HWp = a.Hp * a.Wp;

    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.p), 0, (int)((((long long)a.K - 1) * a.p_cs + a.CPp) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.q), 0, (int)((((long long)a.N * a.Hq * a.Wq - 1) * a.q_cs + a.CQp) * 4), 0x00020000);

    // row table of K-step `step` (BK P pixels): written by the first BK threads
    auto compute_rows = [&](int step) {
        if (t < BK) {
            const int pix = k0 + step * BK + t;
            int4 e = make_int4(-1, 0, -0x4000, -0x4000);
            if (pix < k1) {
                int n, rem, y, x;
                fast_divmod(pix, HWp, a.inv_hw, n, rem);
                fast_divmod(rem, a.Wp, a.inv_w, y, x);
                e.x = pix;
                e.z = y * a.sy - a.py;
                e.w = x * a.sx - a.px;
                e.y = (n * a.Hq + e.z) * a.Wq + e.w;   // may be "negative": only used when the tap is in range
            }
            *reinterpret_cast<int4*>(s_rows + ((step & 1) * BK + t) * 4) = e;
        }
    };

    // fixed staging coordinates of this thread
    const int ra0 = t / CGA, ca = (t % CGA) * 4;
    const int rb0 = t / CGB, cb = (t % CGB) * 4;
    const bool a_col_ok = (m0 + ca) < a.CPp;
    const unsigned a_col_off = (unsigned)(m0 + ca) * 4u;
    const int nb = n0 + cb;
    const bool b_col_ok = nb < a.ncols;
    const int tap = b_col_ok ? nb / a.CQp : 0;
    const int cq = nb - tap * a.CQp;
    const int ky = tap / a.kw;
    const int kx = tap - ky * a.kw;
    const int b_delta = ky * a.Wq + kx;             // q pixel offset of this thread's tap
    const unsigned b_col_off = (unsigned)cq * 4u;

    // two register sets for the staged tiles (static indices only: runtime-indexed vector arrays would go to scratch)
    f32x4 ra[2][PA], rb[2][PB];
    auto gload = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;
        const int* rows = s_rows + (step & 1) * BK * 4;
        // all row-table reads first and unconditionally (one LDS round trip), then branch-free address selects
        int pa[PA];
        int4 eb[PB];
#pragma unroll
        for (int i = 0; i < PA; ++i) pa[i] = rows[((RA >= BK) ? (ra0 < BK ? ra0 : 0) : ra0 + i * RA) * 4];
#pragma unroll
        for (int i = 0; i < PB; ++i)
            eb[i] = *reinterpret_cast<const int4*>(rows + ((RB >= BK) ? (rb0 < BK ? rb0 : 0) : rb0 + i * RB) * 4);
#pragma unroll
        for (int i = 0; i < PA; ++i) asm volatile("" : "+v"(pa[i]));
#pragma unroll
        for (int i = 0; i < PB; ++i) asm volatile("" : "+v"(eb[i].x), "+v"(eb[i].y), "+v"(eb[i].z), "+v"(eb[i].w));
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = a_col_ok & (pa[i] >= 0);
            ra[S][i] = gload4(rp, ok ? (unsigned)pa[i] * (unsigned)a.p_cs * 4u + a_col_off : kGOob);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int4 e = eb[i];
            const bool ok = b_col_ok & ((unsigned)(e.z + ky) < (unsigned)a.Hq) & ((unsigned)(e.w + kx) < (unsigned)a.Wq);
            rb[S][i] = gload4(rq, ok ? (unsigned)(e.y + b_delta) * (unsigned)a.q_cs * 4u + b_col_off : kGOob);
        }
    };
    auto lds_store = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        float* Ab = As + buf * BK * BM;
        float* Bb = Bs + buf * BK * BN;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (RA < BK || ra0 < BK) *reinterpret_cast<f32x4*>(Ab + (ra0 + i * RA) * BM + ca) = ra[S][i];
#pragma unroll
        for (int i = 0; i < PB; ++i)
            if (RB < BK || rb0 < BK) *reinterpret_cast<f32x4*>(Bb + (rb0 + i * RB) * BN + cb) = rb[S][i];
    };

    f32x16 acc[IA][JB];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Software pipeline, distance 2 (a K-step of 16 pixels is only ~2000 MFMA cycles per wave, less than a loaded
    // global-memory round trip): at step s the tile s+2 is requested into register set s&1, the row table of step s+3
    // is tabulated, tile s is multiplied from LDS buffer s&1, and tile s+1 (requested one step earlier into the other
    // set) is written to the idle LDS buffer.  Requests past the last tile read zeros (rows "-1") and are never used.
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    compute_rows(0);
    compute_rows(1);
    __syncthreads();
    gload(0, Set0{});
    gload(1, Set1{});
    __syncthreads();          // both row-table slots have been consumed
    compute_rows(2);
    lds_store(0, Set0{});
    __syncthreads();

    const int fa = IA * (lane & 31), fb = JB * (lane & 31);
    const int khalf = lane >> 5;
    auto do_step = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;            // free register set; the other one holds tile step+1
        using Other = std::integral_constant<int, S ^ 1>;
        const int buf = step & 1;
        const float* Ab = As + buf * BK * BM + wm * 32 * IA + fa;
        const float* Bb = Bs + buf * BK * BN + wn * 32 * JB + fb;
        float av[2][IA], bv[2][JB];
        auto frag = [&](int kk, int slot) {
            const int k = 2 * kk + khalf;
            if (IA == 2) { const f32x2 t2 = *reinterpret_cast<const f32x2*>(Ab + k * BM); av[slot][0] = t2[0]; av[slot][IA - 1] = t2[1]; }
            else av[slot][0] = Ab[k * BM];
            if (JB == 2) { const f32x2 t2 = *reinterpret_cast<const f32x2*>(Bb + k * BN); bv[slot][0] = t2[0]; bv[slot][JB - 1] = t2[1]; }
            else bv[slot][0] = Bb[k * BN];
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < BK / 2) frag(kk + 1, cur ^ 1);      // next fragments land behind this group's MFMAs
            if (kk == 0) {
                gload(step + 2, SET);      // row table slot (step & 1) = rows(step+2), tabulated during the previous step
                compute_rows(step + 3);    // slot (step+1) & 1: last read by gload(step+1) before the previous barrier
            }
            if (kk == BK / 4) lds_store(buf ^ 1, Other{});   // tile step+1 -> idle buffer, in the middle of the MFMA stream
#pragma unroll
            for (int i = 0; i < IA; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            // issue order inside a k-pair group: the next group's fragment reads go right behind the first MFMA (their
            // latency hides under this group's remaining MFMAs); nothing may move across groups
            if (kk != 0 && kk != BK / 4) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, IA * JB - 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        do_step(step, Set0{});
        do_step(step + 1, Set1{});
    }
    if (step < nsteps) do_step(step, Set0{});

    // partial sums -> ws[z][m][n]; lane holds rows IA*rl+i (rl = (r&3)+8*(r>>2)+4*(lane>>5)) and columns JB*(lane&31)+j
    float* wz = a.ws + ((long long)blockIdx.z * a.Mp + m0 + wm * 32 * IA) * a.Np + n0 + wn * 32 * JB + fb;
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * khalf;
            float* dst = wz + (long long)(IA * rl + i) * a.Np;
            if (JB == 2) { f32x2 v = {acc[i][0][r], acc[i][JB - 1][r]}; *reinterpret_cast<f32x2*>(dst) = v; }
            else dst[0] = acc[i][0][r];
        }
}

// ---- bf16 matrix-core variant (W2L_PREC_BF16): same GEMM, operands rounded to bf16 on their way into LDS.
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE K values per lane for a fixed row/column, but both operands are K-major
// (pixel-major) in HBM.  The transpose happens in the staging pattern instead of in LDS: a thread owns one column (one
// P channel, or one (tap, cq) of Q) and 8 consecutive pixels, fetches them with 8 dword loads (lanes = consecutive
// channels: 256 B per wave per pixel, coalesced), packs 8 bf16 and issues ONE ds_write_b128 into an [column][K] image with
// 80-B rows (conflict-free for these writes and for the ds_read_b128 fragment reads, exactly as in conv_igemm_bf16_kernel).
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
constexpr int kWBK = 32;            // pixels per K-step
constexpr int kWLD = kWBK + 8;      // bf16 elements per LDS row (80 B)

template <int BM, int BN>
constexpr int wgrad_bf16_lds_bytes() {
    return 2 * (BM + BN) * kWLD * 2 + 2 * kWBK * 16;
}

__device__ __forceinline__ bf16x8w pack8(const float (&v)[8]) {
    bf16x8w o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
    return o;
}

__device__ __forceinline__ float gload1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// ROWOCT: the P grid's width is a multiple of 8, so the 8 pixels of a staging item never straddle an image row: one row-table
// lookup and one row bounds test per item, per-pixel work reduced to an x bounds test and a constant address step.
template <int BM, int BN, int WM, int WN, bool ROWOCT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradKArgs a) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile at least 32x32");
    constexpr int OCT = kWBK / 8;                       // 8-pixel octets per K-step
    constexpr int NA = (BM * OCT + 255) / 256;          // staging items per thread
    constexpr int NB = (BN * OCT + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);             // [2][BM][kWLD]
    __bf16* Bs = As + 2 * BM * kWLD;                          // [2][BN][kWLD]
    int* s_rows = reinterpret_cast<int*>(Bs + 2 * BN * kWLD); // [2][kWBK][4]: q base pixel, iy0, ix0 (x unused)

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = bid % a.tiles_n;
    const int tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int k0 = blockIdx.z * a.chunk;
    const int k1 = min(a.K, k0 + a.chunk);
    const int nsteps = (k1 - k0 + kWBK - 1) / kWBK;
    const int HWp = a.Hp * a.Wp;

    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.p), 0, (int)((((long long)a.K - 1) * a.p_cs + a.CPp) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.q), 0, (int)((((long long)a.N * a.Hq * a.Wq - 1) * a.q_cs + a.CQp) * 4), 0x00020000);

    auto compute_rows = [&](int step) {
        if (t < kWBK) {
            const int pix = k0 + step * kWBK + t;
            int4 e = make_int4(-1, 0, -0x4000, -0x4000);
            if (pix < k1) {
                int n, rem, y, x;
                fast_divmod(pix, HWp, a.inv_hw, n, rem);
                fast_divmod(rem, a.Wp, a.inv_w, y, x);
                e.x = pix;
                e.z = y * a.sy - a.py;
                e.w = x * a.sx - a.px;
                e.y = (n * a.Hq + e.z) * a.Wq + e.w;
            }
            *reinterpret_cast<int4*>(s_rows + ((step & 1) * kWBK + t) * 4) = e;
        }
    };

    // staging items: id = p*256 + t -> column id % B{M,N}, octet id / B{M,N}
    int a_col[NA], a_oct[NA];
    unsigned a_coff[NA];
    bool a_ok[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        const int id = p * 256 + t;
        a_col[p] = id % BM;
        a_oct[p] = id / BM;
        a_ok[p] = (id < BM * OCT) & ((m0 + a_col[p]) < a.CPp);
        a_coff[p] = (unsigned)(m0 + a_col[p]) * 4u;
    }
    int b_col[NB], b_oct[NB], b_ky[NB], b_kx[NB], b_delta[NB];
    unsigned b_coff[NB];
    bool b_ok[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int id = p * 256 + t;
        b_col[p] = id % BN;
        b_oct[p] = id / BN;
        const int nb = n0 + b_col[p];
        b_ok[p] = (id < BN * OCT) & (nb < a.ncols);
        const int tap = b_ok[p] ? nb / a.CQp : 0;
        const int cq = nb - tap * a.CQp;
        b_ky[p] = tap / a.kw;
        b_kx[p] = tap - b_ky[p] * a.kw;
        b_delta[p] = b_ky[p] * a.Wq + b_kx[p];
        b_coff[p] = (unsigned)cq * 4u;
    }

    float ra[2][NA][8], rb[2][NB][8];
    auto gload = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;
        const int* rows = s_rows + (step & 1) * kWBK * 4;
        const int pbase = k0 + step * kWBK;
#pragma unroll
        for (int p = 0; p < NA; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pix = pbase + a_oct[p] * 8 + j;
                const bool ok = a_ok[p] & (pix < k1);
                ra[S][p][j] = gload1(rp, ok ? (unsigned)pix * (unsigned)a.p_cs * 4u + a_coff[p] : kGOob);
                if (j == 7) __builtin_amdgcn_sched_barrier(0);   // keep only one item's 8 address registers live at a time
            }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            if (ROWOCT) {
                const int4 e = *reinterpret_cast<const int4*>(rows + (b_oct[p] * 8) * 4);
                const bool row_ok = b_ok[p] & ((unsigned)(e.z + b_ky[p]) < (unsigned)a.Hq);
                const unsigned base = (unsigned)(e.y + b_delta[p]) * (unsigned)a.q_cs * 4u + b_coff[p];
                const unsigned dstep = (unsigned)(a.sx * a.q_cs) * 4u;
                const int ix = e.w + b_kx[p];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = row_ok & ((unsigned)(ix + j * a.sx) < (unsigned)a.Wq);
                    rb[S][p][j] = gload1(rq, ok ? base + (unsigned)j * dstep : kGOob);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int4 e = *reinterpret_cast<const int4*>(rows + (b_oct[p] * 8 + j) * 4);
                    const bool ok = b_ok[p] & ((unsigned)(e.z + b_ky[p]) < (unsigned)a.Hq) & ((unsigned)(e.w + b_kx[p]) < (unsigned)a.Wq);
                    rb[S][p][j] = gload1(rq, ok ? (unsigned)(e.y + b_delta[p]) * (unsigned)a.q_cs * 4u + b_coff[p] : kGOob);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto lds_store = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        __bf16* Ab = As + buf * BM * kWLD;
        __bf16* Bb = Bs + buf * BN * kWLD;
#pragma unroll
        for (int p = 0; p < NA; ++p)
            if (NA * 256 == BM * OCT || p * 256 + t < BM * OCT)
                *reinterpret_cast<bf16x8w*>(Ab + a_col[p] * kWLD + a_oct[p] * 8) = pack8(ra[S][p]);
#pragma unroll
        for (int p = 0; p < NB; ++p)
            if (NB * 256 == BN * OCT || p * 256 + t < BN * OCT)
                *reinterpret_cast<bf16x8w*>(Bb + b_col[p] * kWLD + b_oct[p] * 8) = pack8(rb[S][p]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    compute_rows(0);
    compute_rows(1);
    __syncthreads();
    gload(0, Set0{});
    gload(1, Set1{});
    __syncthreads();
    compute_rows(2);
    lds_store(0, Set0{});
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;
    auto do_step = [&](int step, auto SET) {
        constexpr int S = decltype(SET)::value;
        using Other = std::integral_constant<int, S ^ 1>;
        const int buf = step & 1;
        const __bf16* Ab = As + buf * BM * kWLD + (wm * TM * 32 + frag_row) * kWLD + frag_k;
        const __bf16* Bb = Bs + buf * BN * kWLD + (wn * TN * 32 + frag_row) * kWLD + frag_k;
        bf16x8w af[2][TM], bfr[2][TN];
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[kq][i] = *reinterpret_cast<const bf16x8w*>(Ab + i * 32 * kWLD + kq * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[kq][j] = *reinterpret_cast<const bf16x8w*>(Bb + j * 32 * kWLD + kq * 16);
        }
        gload(step + 2, SET);
        compute_rows(step + 3);
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i], bfr[kq][j], acc[i][j], 0, 0, 0);
            if (kq == 0) lds_store(buf ^ 1, Other{});
        }
        __syncthreads();
    };
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        do_step(step, Set0{});
        do_step(step + 1, Set1{});
    }
    if (step < nsteps) do_step(step, Set0{});

    // partial sums -> ws[z][m][n]: lane holds column (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5) of each 32x32 tile
    float* wz = a.ws + ((long long)blockIdx.z * a.Mp + m0 + wm * TM * 32) * a.Np + n0 + wn * TN * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wz[(long long)(i * 32 + rl) * a.Np + j * 32] = acc[i][j][r];
            }
}

// Heads with at most 4 P channels (the generator's RGB conv, the discriminator's 1-channel prediction): an MFMA tile
// would be >90 % padding, and the work is a plain HBM-bound reduction.  thread -> (float4 column group of (tap, cq),
// pixel lane); 16 accumulators per thread; pixel lanes combined through LDS; same workspace layout as above (Mp = 4).
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(const WgradKArgs a) {
    __shared__ float red[256][17];
    const int CG = a.ncols >> 2;
    const int RPP = 256 / CG;
    const int t = threadIdx.x;
    const int c4 = t % CG, rl = t / CG;
    float acc[4][4] = {};
    if (rl < RPP) {
        const int nb = c4 * 4;
        const int tap = nb / a.CQp;
        const int cq = nb - tap * a.CQp;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int HWp = a.Hp * a.Wp;
        const int k0 = blockIdx.x * a.chunk;
        const int k1 = min(a.K, k0 + a.chunk);
        for (int pix = k0 + rl; pix < k1; pix += RPP) {
            const int n = pix / HWp;
            const int rem = pix - n * HWp;
            const int y = rem / a.Wp;
            const int x = rem - y * a.Wp;
            const int iy = y * a.sy - a.py + ky, ix = x * a.sx - a.px + kx;
            if ((unsigned)iy >= (unsigned)a.Hq || (unsigned)ix >= (unsigned)a.Wq) continue;
            const f32x4 pv = *reinterpret_cast<const f32x4*>(a.p + (long long)pix * a.p_cs);
            const f32x4 qv = *reinterpret_cast<const f32x4*>(a.q + ((long long)(n * a.Hq + iy) * a.Wq + ix) * a.q_cs + cq);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = fmaf(pv[i], qv[e], acc[i][e]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[t][i * 4 + e] = acc[i][e];
    __syncthreads();
    if (t < CG) {
        float* wz = a.ws + (long long)blockIdx.x * a.Mp * a.Np;
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
                for (int j = 0; j < RPP; ++j) s += red[t + j * CG][i * 4 + e];
                wz[(long long)i * a.Np + t * 4 + e] = s;
            }
    }
}

struct WgradReduceArgs {
    const float* ws;
    float* dw;        // [CP][CQ][ntaps]
    float* colsum;    // optional: unused
    int ksplit, Mp, Np, CP, CQ, CQp, ntaps, ncols;
};

// one thread per (cp, n = (tap, cq)): fixed-order sum over the K splits, scatter into the torch layout
__global__ void wgrad_reduce_kernel(const WgradReduceArgs a) {
    const long long total = (long long)a.CP * a.ncols;
    const long long zs = (long long)a.Mp * a.Np;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cp = (int)(i / a.ncols);
        const int n = (int)(i - (long long)cp * a.ncols);
        const int tap = n / a.CQp;
        const int cq = n - tap * a.CQp;
        if (cq >= a.CQ) continue;
        const float* src = a.ws + (long long)cp * a.Np + n;
        float s = 0.f;
        for (int z = 0; z < a.ksplit; ++z) s += src[z * zs];
        a.dw[((long long)cp * a.CQ + cq) * a.ntaps + tap] = s;
    }
}

float* conv_workspace(hipStream_t stream, size_t bytes);   // conv_igemm.hip: grow-only per-stream scratch of the split-K paths

struct WgradCfg {
    int bm, bn, bk;
    void (*kernel)(const WgradKArgs);
    int lds;
    int wg_per_cu;   // resident workgroups per CU (min of the LDS and VGPR limits): sizes the K split to ONE full wave of workgroups
};
static const WgradCfg kWgradCfgs[] = {
    {128, 128, 32, conv_wgrad_f32_kernel<2, 2, 2, 2, 32>, wgrad_lds_bytes<128, 128, 32>(), 2},   // 0: CP > 64
    {64, 256, 16, conv_wgrad_f32_kernel<2, 2, 1, 4, 16>, wgrad_lds_bytes<64, 256, 16>(), 3},     // 1: 32 < CP <= 64
    {64, 128, 32, conv_wgrad_f32_kernel<2, 1, 1, 4, 32>, wgrad_lds_bytes<64, 128, 32>(), 3},     // 2: same, narrower N tile
    {32, 256, 16, conv_wgrad_f32_kernel<1, 2, 1, 4, 16>, wgrad_lds_bytes<32, 256, 16>(), 3},     // 3: CP <= 32
    {32, 128, 32, conv_wgrad_f32_kernel<1, 1, 1, 4, 32>, wgrad_lds_bytes<32, 128, 32>(), 3},     // 4: same, narrower N tile
};

static const WgradCfg kWgradBf16Cfgs[] = {   // [tile][P-grid width % 8 == 0]
    {128, 64, kWBK, conv_wgrad_bf16_kernel<128, 64, 2, 2, false>, wgrad_bf16_lds_bytes<128, 64>(), 2},   // CP > 64 (128x128 needs > 256 registers)
    {128, 64, kWBK, conv_wgrad_bf16_kernel<128, 64, 2, 2, true>, wgrad_bf16_lds_bytes<128, 64>(), 2},
    {64, 128, kWBK, conv_wgrad_bf16_kernel<64, 128, 1, 4, false>, wgrad_bf16_lds_bytes<64, 128>(), 2},   // 32 < CP <= 64
    {64, 128, kWBK, conv_wgrad_bf16_kernel<64, 128, 1, 4, true>, wgrad_bf16_lds_bytes<64, 128>(), 2},
    {32, 128, kWBK, conv_wgrad_bf16_kernel<32, 128, 1, 4, false>, wgrad_bf16_lds_bytes<32, 128>(), 3},   // CP <= 32
    {32, 128, kWBK, conv_wgrad_bf16_kernel<32, 128, 1, 4, true>, wgrad_bf16_lds_bytes<32, 128>(), 3},
};

static int wgrad_force_cfg = -1;   // W2L_WGRAD_CFG=<id>: force a tile configuration (tuning / tests)
static int wgrad_wino = 1;         // W2L_WINO_WGRAD=0: keep the direct GEMM for the 3x3 s1 p1 layers (A/B runs)

// conv_wino_wgrad.hip
bool wino_wgrad_ok(const w2l_conv_geom* g, int N, int H, int W, int x_cs, int dz_cs);
int wino_wgrad_launch(const w2l_conv_geom* g, hipStream_t s, int N, int H, int W, const float* x, int x_cs, const float* dz,
                      int dz_cs, float* dweight);

int wgrad_reduce_launch(hipStream_t s, const float* ws, float* dw, int ksplit, int Mp, int Np, int CP, int CQ, int CQp, int ntaps) {
    WgradReduceArgs r;
    r.ws = ws; r.dw = dw; r.colsum = nullptr;
    r.ksplit = ksplit; r.Mp = Mp; r.Np = Np; r.CP = CP; r.CQ = CQ; r.CQp = CQp;
    r.ntaps = ntaps; r.ncols = ntaps * CQp;
    long long gr = ((long long)CP * r.ncols + 255) / 256;
    if (gr > 8192) gr = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gr), dim3(256), 0, s, r);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int wgrad_init_attrs() {
    static bool done = false;
    if (done) return W2L_OK;
    for (const WgradCfg& c : kWgradCfgs)
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(c.kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    for (const WgradCfg& c : kWgradBf16Cfgs)
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(c.kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    if (const char* e = getenv("W2L_WGRAD_CFG")) wgrad_force_cfg = atoi(e);
    if (const char* e = getenv("W2L_WINO_WGRAD")) wgrad_wino = atoi(e);
    done = true;
    return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

static int wgrad_impl(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs, const float* dz,
                      int dz_cs, float* dweight, int precision);

extern "C" int w2l_conv_wgrad(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs,
                              const float* dz, int dz_cs, float* dweight) {
    return wgrad_impl(g, stream, N, H, W, x, x_cs, dz, dz_cs, dweight, W2L_PREC_F32);
}

extern "C" int w2l_conv_wgrad_prec(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs,
                                   const float* dz, int dz_cs, float* dweight, int precision) {
    W2L_REQUIRE(precision == W2L_PREC_F32 || precision == W2L_PREC_BF16, "bad precision %d", precision);
    return wgrad_impl(g, stream, N, H, W, x, x_cs, dz, dz_cs, dweight, precision);
}

static int wgrad_impl(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs, const float* dz,
                      int dz_cs, float* dweight, int precision) {
    W2L_REQUIRE(g && x && dz && dweight, "NULL argument");
    W2L_REQUIRE(N >= 1 && H >= 1 && W >= 1, "bad shape N=%d H=%d W=%d", N, H, W);
    int Ho, Wo;
    if (w2l_conv_out_hw(g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    const int cin_p = round_up(g->cin, 4), cout_p = round_up(g->cout, 4);
    W2L_REQUIRE(x_cs >= cin_p && (x_cs & 3) == 0 && dz_cs >= cout_p && (dz_cs & 3) == 0,
                "wgrad: x_cs=%d / dz_cs=%d must be multiples of 4 covering the padded channel counts %d / %d", x_cs, dz_cs,
                cin_p, cout_p);
    W2L_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0, "x and dz must be 16-byte aligned");
    const long long lim = 1ll << 31;
    W2L_REQUIRE((long long)N * H * W * x_cs * 4 < lim && (long long)N * Ho * Wo * dz_cs * 4 < lim,
                "activation buffer larger than 2 GiB: split the batch");
    if (wgrad_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    if (precision == W2L_PREC_F32 && wgrad_wino && wgrad_force_cfg < 0 && wino_wgrad_ok(g, N, H, W, x_cs, dz_cs))
        return wino_wgrad_launch(g, static_cast<hipStream_t>(stream), N, H, W, x, x_cs, dz, dz_cs, dweight);
    WgradKArgs a;
    a.N = N;
    if (!g->transposed) {   // P = dz on the output grid, Q = x
        a.p = dz; a.Hp = Ho; a.Wp = Wo; a.CPp = cout_p; a.p_cs = dz_cs;
        a.q = x; a.Hq = H; a.Wq = W; a.CQp = cin_p; a.q_cs = x_cs;
    } else {                // P = x on the input grid, Q = dz
        a.p = x; a.Hp = H; a.Wp = W; a.CPp = cin_p; a.p_cs = x_cs;
        a.q = dz; a.Hq = Ho; a.Wq = Wo; a.CQp = cout_p; a.q_cs = dz_cs;
    }
    const int CP = g->transposed ? g->cin : g->cout;
    const int CQ = g->transposed ? g->cout : g->cin;
    a.sy = g->sh; a.sx = g->sw; a.py = g->ph; a.px = g->pw;
    a.kh = g->kh; a.kw = g->kw;
    a.K = N * a.Hp * a.Wp;
    a.ncols = g->kh * g->kw * a.CQp;
    a.inv_hw = 1.0f / (float)(a.Hp * a.Wp);
    a.inv_w = 1.0f / (float)a.Wp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.CPp == 4 && a.ncols <= 1024) {   // tiny head: HBM-bound reduction kernel
        a.Mp = 4; a.Np = a.ncols; a.tiles_n = 1;
        const int RPP = 256 / (a.ncols >> 2);
        long long nb = a.K / ((long long)RPP * 64);
        if (nb > 1024) nb = 1024;
        if (nb < 1) nb = 1;
        a.chunk = ceil_div(a.K, (int)nb);
        const int nblk = ceil_div(a.K, a.chunk);
        a.ws = conv_workspace(s, (size_t)nblk * a.Mp * a.Np * sizeof(float));
        if (!a.ws) return W2L_ERR_NOMEM;
        if (flops_counting()) flops_add(2ll * a.Mp * a.Np * a.K, 7);
        hipLaunchKernelGGL(conv_wgrad_small_kernel, dim3(nblk), dim3(256), 0, s, a);
        W2L_HIP_CHECK(hipGetLastError());
        WgradReduceArgs r;
        r.ws = a.ws; r.dw = dweight; r.colsum = nullptr;
        r.ksplit = nblk; r.Mp = a.Mp; r.Np = a.Np; r.CP = CP; r.CQ = CQ; r.CQp = a.CQp;
        r.ntaps = g->kh * g->kw; r.ncols = a.ncols;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(CP * a.ncols, 256)), dim3(256), 0, s, r);
        W2L_HIP_CHECK(hipGetLastError());
        return W2L_OK;
    }
    // tile rows: 32 for CP <= 32, else 64 or 128 — whichever pads the channel axis less (ties: the larger tile);
    // 32/64-row tiles come in two N widths: take the one that pads the (tap, cq) axis less (ties: the wider tile)
    int ci = 0;
    const bool rows64 = a.CPp <= 64 || round_up(a.CPp, 64) * 10 < round_up(a.CPp, 128) * 9;
    if (a.CPp <= 32 || rows64) {
        const int wide = a.CPp <= 32 ? 3 : 1;
        ci = round_up(a.ncols, 128) * 10 < round_up(a.ncols, 256) * 9 ? wide + 1 : wide;
    }
    if (wgrad_force_cfg >= 0 && wgrad_force_cfg < (int)(sizeof(kWgradCfgs) / sizeof(kWgradCfgs[0])) &&
        kWgradCfgs[wgrad_force_cfg].bm >= (a.CPp <= 32 ? 32 : 64))
        ci = wgrad_force_cfg;
    const WgradCfg& cfg = precision == W2L_PREC_BF16
                              ? kWgradBf16Cfgs[2 * (a.CPp <= 32 ? 2 : ((a.CPp <= 64 || rows64) ? 1 : 0)) + ((a.Wp & 7) == 0 ? 1 : 0)]
                              : kWgradCfgs[ci];
    const int tiles_m = ceil_div(a.CPp, cfg.bm);
    a.tiles_n = ceil_div(a.ncols, cfg.bn);
    a.Mp = tiles_m * cfg.bm;
    a.Np = a.tiles_n * cfg.bn;
    const long long tiles = (long long)tiles_m * a.tiles_n;
    // K splits: all workgroups have equal work and run in lock step, so a grid slightly LARGER than the resident capacity
    // (256 CUs x wg_per_cu) costs a whole extra round; size the split to fill exactly one round (at least 8 K-steps per
    // workgroup, workspace capped at 512 MiB)
    const long long slots = 256ll * cfg.wg_per_cu;
    long long ks = tiles >= slots ? 1 : slots / tiles;
    const long long max_by_k = a.K / (8 * cfg.bk) > 0 ? a.K / (8 * cfg.bk) : 1;
    if (ks > max_by_k) ks = max_by_k;
    const long long max_by_ws = (512ll << 20) / ((long long)a.Mp * a.Np * 4);
    if (ks > max_by_ws) ks = max_by_ws;
    if (ks < 1) ks = 1;
    a.chunk = round_up(ceil_div(a.K, (int)ks), cfg.bk);
    const int ksplit = ceil_div(a.K, a.chunk);
    a.ws = conv_workspace(s, (size_t)ksplit * a.Mp * a.Np * sizeof(float));
    if (!a.ws) return W2L_ERR_NOMEM;
    if (flops_counting()) flops_add(2ll * a.Mp * a.Np * (long long)ksplit * a.chunk, precision == W2L_PREC_BF16 ? 4 : 1);
    hipLaunchKernelGGL(cfg.kernel, dim3((unsigned)tiles, 1, ksplit), dim3(256), cfg.lds, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    WgradReduceArgs r;
    r.ws = a.ws; r.dw = dweight; r.colsum = nullptr;
    r.ksplit = ksplit; r.Mp = a.Mp; r.Np = a.Np; r.CP = CP; r.CQ = CQ; r.CQp = a.CQp;
    r.ntaps = g->kh * g->kw; r.ncols = a.ncols;
    long long gr = ((long long)CP * a.ncols + 255) / 256;
    if (gr > 8192) gr = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gr), dim3(256), 0, s, r);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

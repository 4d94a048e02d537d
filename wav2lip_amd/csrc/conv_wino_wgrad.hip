// Winograd F(3x3, 2x2) weight gradient of the 3x3 / stride 1 / pad 1 convolutions on the fp32 matrix cores: the same 2.25x
// multiply saving the forward kernel (conv_wino.hip) gets, for the backward-by-weights pass of those layers
// (autograd of nn.Conv2d under loss.backward(), wav2lip_train.py:229 -> models/conv.py:8).
//
//   dW[co][ci] (3x3) = A^T [ sum over 2x2 output tiles of (G dy G^T) (.) (B^T d B) ] A
//
// with dy the 2x2 tile of the conv-output gradient, d the 4x4 input tile around it, and the F(3,2) matrices
//   G = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]],  A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]]
// (validated against autograd in float64, odd sizes included, before this kernel was written: tests/test_train_gpu.py).
//
// GEMM view: for each of the 16 transform positions an independent GEMM  M_xi[co][ci] += DY_xi[tile][co] * V_xi[tile][ci]
// with K = tiles.  One workgroup (4 waves, one per SIMD, up to 512 registers each) owns 64 couts x 64 cins; every wave owns
// 32 x 32 for ALL 16 positions = 16 accumulators of 32x32 (256 registers), so the output transform A^T M A is per-lane
// register work.  Per K-step of 8 tiles every thread transforms one (tile, channel quad, half) item of the input (3x4 patch
// of float4 -> 8 positions) and of the output gradient (2x2 -> 8 positions) in registers and writes it to LDS as
// [position][tile][channel] rows, so a wave reads its fp32 MFMA operands (one float per lane: row/column = lane & 31,
// k = lane >> 5) as 32 consecutive floats per half-wave — conflict-free.  K is split across workgroups (gridDim.y) and the
// partial 3x3 gradients are reduced in a fixed order by wgrad_reduce_kernel (shared with conv_wgrad.hip).
#include <type_traits>
#include <utility>

#include "w2l_common.h"
#include "w2l_pk.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kQOob = 0x80000000u;
constexpr int kKT = 8;   // tiles per K-step

struct WinoWgradArgs {
    const float* x;    // [N][H][W][x_cs], cin_p readable channels
    const float* dz;   // [N][H][W][dz_cs], cout_p readable channels
    float* ws;         // [ksplit][Mp][Np], Np = 9 * CQp, column = tap * CQp + ci
    int N, H, W, cin, cin_p, x_cs, cout, cout_p, dz_cs;
    int TH, TW, T;     // 2x2 tiles per image and in total
    int tiles_n;       // cin tiles of 64
    int chunk;         // tiles per K split (multiple of kKT)
    int Mp, Np, CQp;
    float inv_thw, inv_tw;
};

__device__ __forceinline__ f32x4 qload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ void divmod_f(int a, int d, float inv_d, int& q, int& r) {
    q = (int)((float)a * inv_d);
    r = a - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;
    q += hi - lo;
    r += (lo - hi) * d;
}

// f(0), f(1), ... f(N-1) with literal arguments (a `#pragma unroll` loop over a body this large may stay rolled, which would
// turn the register arrays indexed by the slice number into scratch memory)
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(I), ...);
}

__global__ __launch_bounds__(256, 1) void conv_wino_wgrad_f32_kernel(const WinoWgradArgs a) {
    constexpr int BM = 64, BN = 64;
    constexpr int POS = kKT * BM;          // floats per position slab ([tile][channel])
    constexpr int OPB = 16 * POS;          // floats per operand buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ds = reinterpret_cast<float*>(smem);   // [2][16][kKT][BM]  transformed output gradients
    float* Vs = Ds + 2 * OPB;                      // [2][16][kKT][BN]  transformed inputs

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x % a.tiles_n;
    const int tile_m = blockIdx.x / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k0 = blockIdx.y * a.chunk;
    const int k1 = min(a.T, k0 + a.chunk);
    const int nsteps = (k1 - k0 + kKT - 1) / kKT;
    const int THW = a.TH * a.TW;
    const long long npix = (long long)a.N * a.H * a.W;

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(((npix - 1) * a.x_cs + a.cin_p) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, (int)(((npix - 1) * a.dz_cs + a.cout_p) * 4), 0x00020000);

    // staging item of this thread: tile slot tl (of 8), channel quad q (of 16), half (rows {0,1} or {2,3} of the transforms)
    const int half = wave >> 1;
    const float sgn = half ? -1.0f : 1.0f;
    const int tl = (t & 127) >> 4;
    const int q = t & 15;
    const bool xq_ok = (n0 + 4 * q) < a.cin_p;
    const bool dq_ok = (m0 + 4 * q) < a.cout_p;

    // ---- staging work of one (tile, channel quad, half) item, cut into small slices that the main loop drops between MFMAs.
    // VALU instructions do not hide behind fp32 MFMAs (profiles/r01/i_mfma_overlap_microbench.txt): the slices are written for
    // instruction count — packed fp32 transforms, tile coordinates advanced incrementally instead of divided out, and
    // "poisoned" row / column offsets instead of per-load masks: an invalid row or column (outside the image, tile past the K
    // chunk, channel quad past the tensor) carries the offset 0x80000000, the load offset is the SATURATING sum row + column,
    // which lands past the end of any tensor < 2 GiB, and the buffer load returns zeros.
    //   G0       next tile of this slot: (n, ty, tx) advanced by kKT tiles
    //   G1 G2    input patch row offsets (3) | column offsets (4)        G3  output-gradient row (2) and column (2) offsets
    //   L0..L3   the 2x2 output-gradient pixels        L4..L15  the 3x4 input patch (row r, column c = (L - 4) / 4, % 4)
    // transforms -> LDS (position = 8*half + j):
    //   D0 D1    rows 2*half, 2*half+1 of G g for tile column c   (G = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]])
    //   D2..D9   column transform (. G^T) of position j = D - 2 and its store
    //   X0..X3   rows 2*half, 2*half+1 of B^T d for patch column c (B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]])
    //   X4..X11  column transform (. B) of position j = X - 4 and its store
    // Tile rows with odd slot index store their 64 channels with the two 32-channel halves swapped, so that the fragment
    // reads of k = 2*kp (lanes 0..31) and k = 2*kp + 1 (lanes 32..63) fall into disjoint banks.
    f32x4 raw[3][4];      // input patch rows (A, B, C) = (d0, d2, d1) for half 0, (d2, d1, d3) for half 1
    f32x4 gq[2][2];       // output-gradient 2x2 tile
    f32x4 ra[4], rb[4], u0[2], u1[2];
    unsigned xro[3], xco[4], dro[2], dco[2];
    const int xcs4 = a.x_cs * 4, dcs4 = a.dz_cs * 4, xrow = a.W * xcs4, drow = a.W * dcs4;
    const int chx = (n0 + 4 * q) * 4, chd = (m0 + 4 * q) * 4;
    const int wsw = (q * 4) ^ ((tl & 1) << 5);
    const int q8 = kKT / a.TW, r8 = kKT % a.TW;      // a step of kKT tiles = q8 tile rows + r8 tiles
    // per-wave constants of the row transforms: (u0, u1) = (ca0 g0 + cb0 g1, ca1 g0 + cb1 g1); rb = sgn * B + C
    const float ca0f = half ? 0.5f : 1.0f, cb0f = half ? -0.5f : 0.0f, ca1f = half ? 0.0f : 0.5f, cb1f = half ? 1.0f : 0.5f;
    const f32x2 ca0 = {ca0f, ca0f}, cb0 = {cb0f, cb0f}, ca1 = {ca1f, ca1f}, cb1 = {cb1f, cb1f}, sgn2 = {sgn, sgn};
    // tile of this staging slot in step 0, then advanced by G0
    int t_tile = k0 + tl, t_n, t_ty, t_tx;
    {
        int rem;
        divmod_f(min(t_tile, a.T - 1), THW, a.inv_thw, t_n, rem);
        divmod_f(rem, a.TW, a.inv_tw, t_ty, t_tx);
    }
    auto g_slice = [&](int sl, bool advance) {
        if (sl == 0) {
            if (advance) {          // kKT tiles further along (tx, then ty, then n): TW, TH >= 3, so one wrap each
                asm volatile("" : "+v"(t_tx));           // keeps the index arithmetic at this slot of the loop
                t_tile += kKT;
                t_tx += r8;
                t_ty += q8;
                const bool cx = t_tx >= a.TW;
                t_tx -= cx ? a.TW : 0;
                t_ty += cx ? 1 : 0;
                const bool cy = t_ty >= a.TH;
                t_ty -= cy ? a.TH : 0;
                t_n += cy ? 1 : 0;
            }
        } else if (sl == 1) {
            int ty = t_ty;
            asm volatile("" : "+v"(ty));
            const unsigned hlim = (t_tile < k1 && xq_ok) ? (unsigned)a.H : 0u;
            const int prow = t_n * a.H + 2 * ty - 1;                     // image row of patch row 0 (may be -1: masked)
            const unsigned b0 = (unsigned)(__mul24(prow, xrow) + chx);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int rowsel = half ? (r == 0 ? 2 : (r == 1 ? 1 : 3)) : (r == 0 ? 0 : (r == 1 ? 2 : 1));
                xro[r] = ((unsigned)(2 * ty - 1 + rowsel) < hlim) ? b0 + (unsigned)(rowsel * xrow) : kQOob;
            }
        } else if (sl == 2) {
            int tx = t_tx;
            asm volatile("" : "+v"(tx));
            const unsigned c1 = (unsigned)__mul24(2 * tx, xcs4);        // column 2*tx: always inside the image
            xco[0] = tx > 0 ? c1 - (unsigned)xcs4 : kQOob;
            xco[1] = c1;
            xco[2] = (2 * tx + 1 < a.W) ? c1 + (unsigned)xcs4 : kQOob;
            xco[3] = (2 * tx + 2 < a.W) ? c1 + (unsigned)(2 * xcs4) : kQOob;
        } else {
            int ty = t_ty, tx = t_tx;
            asm volatile("" : "+v"(ty), "+v"(tx));
            const unsigned hlim = (t_tile < k1 && dq_ok) ? (unsigned)a.H : 0u;
            const unsigned b0 = (unsigned)(__mul24(t_n * a.H + 2 * ty, drow) + chd);
            dro[0] = ((unsigned)(2 * ty) < hlim) ? b0 : kQOob;
            dro[1] = ((unsigned)(2 * ty + 1) < hlim) ? b0 + (unsigned)drow : kQOob;
            const unsigned c0 = (unsigned)__mul24(2 * tx, dcs4);
            dco[0] = c0;
            dco[1] = (2 * tx + 1 < a.W) ? c0 + (unsigned)dcs4 : kQOob;
        }
    };
    auto l_slice = [&](int sl) {
        if (sl < 4) {
            const int r = sl >> 1, c = sl & 1;
            gq[r][c] = qload4(rd, __builtin_elementwise_add_sat(dro[r], dco[c]));
        } else {
            const int r = (sl - 4) >> 2, c = (sl - 4) & 3;
            raw[r][c] = qload4(rx, __builtin_elementwise_add_sat(xro[r], xco[c]));
        }
    };
    auto d_slice = [&](int sl, int buf) {
        if (sl < 2) {
            const int c = sl;
            asm volatile("" : "+v"(gq[0][c]), "+v"(gq[1][c]));
            u0[c] = pk_fma(gq[1][c], cb0, pk_mul(gq[0][c], ca0));
            u1[c] = pk_fma(gq[1][c], cb1, pk_mul(gq[0][c], ca1));
        } else {
            const int j = sl - 2;
            float* dw = Ds + buf * OPB + (half * 8) * POS + tl * BM + wsw;
            const f32x4* uu = (j < 4) ? u0 : u1;
            f32x4 v;
            switch (j & 3) {
                case 0: v = uu[0]; break;
                case 1: v = pk_half(pk_add(uu[0], uu[1])); break;
                case 2: v = pk_half(pk_sub(uu[0], uu[1])); break;
                default: v = uu[1]; break;
            }
            *reinterpret_cast<f32x4*>(dw + j * POS) = v;
        }
    };
    auto x_slice = [&](int sl, int buf) {
        if (sl < 4) {
            const int c = sl;
            asm volatile("" : "+v"(raw[0][c]), "+v"(raw[1][c]), "+v"(raw[2][c]));
            ra[c] = pk_sub(raw[0][c], raw[1][c]);        // half 0: d0 - d2 (row 0)   half 1: d2 - d1 (row 2)
            rb[c] = pk_fma(raw[1][c], sgn2, raw[2][c]);   // half 0: d2 + d1 (row 1)   half 1: d3 - d1 (row 3)
        } else {
            const int j = sl - 4;
            float* vw = Vs + buf * OPB + (half * 8) * POS + tl * BN + wsw;
            const f32x4* rr = (j < 4) ? ra : rb;
            f32x4 v;
            switch (j & 3) {
                case 0: v = pk_sub(rr[0], rr[2]); break;
                case 1: v = pk_add(rr[1], rr[2]); break;
                case 2: v = pk_sub(rr[2], rr[1]); break;
                default: v = pk_sub(rr[3], rr[1]); break;
            }
            *reinterpret_cast<f32x4*>(vw + j * POS) = v;
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // prologue: operands of step 0 -> LDS buffer 0, raw tiles of step 1 in registers
    static_for([&](int sl) { g_slice(sl, false); }, std::make_integer_sequence<int, 4>{});
    static_for([&](int sl) { l_slice(sl); }, std::make_integer_sequence<int, 16>{});
    static_for([&](int sl) { d_slice(sl, 0); }, std::make_integer_sequence<int, 10>{});
    static_for([&](int sl) { x_slice(sl, 0); }, std::make_integer_sequence<int, 12>{});
    static_for([&](int sl) { g_slice(sl, true); }, std::make_integer_sequence<int, 4>{});
    static_for([&](int sl) { l_slice(sl); }, std::make_integer_sequence<int, 16>{});
    __syncthreads();

    // main loop: per K-step 64 MFMA "items" (kp = item >> 4 : k pair, p = item & 15 : position); consecutive items hit
    // different accumulators, so anything may sit between them.  Step s multiplies LDS buffer s & 1 and transforms the raw
    // tiles of step s+1 into the other buffer; every raw register is refilled (tiles of step s+2) soon after its last use, so
    // a load has 29..57 items (2000+ cycles) before its first use:
    //   item  0, 1   D0 D1      2..5  G0..G3      6, 8 .. 20  D2..D9       7, 11, 15, 19  L0..L3 (free since D1)
    //   item 22..25  X0..X3     26, 28 .. 40  X4..X11     27, 30 .. 60  the input patch, column-major (column c free since X_c)
    // The loads and LDS stores of the four waves come at the same items, so they are spread over the step instead of issued
    // back to back (a burst of 48 KB of loads in 12 items runs into the texture unit's 64 B/clk).
    // Fragment reads run kPF items ahead of their MFMA.
    constexpr int kPF = 4;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int rsw_a = (wm * 32 + l31) ^ (khalf << 5), rsw_b = (wn * 32 + l31) ^ (khalf << 5);
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const float* Ab = Ds + buf * OPB + khalf * BM + rsw_a;
        const float* Bb = Vs + buf * OPB + khalf * BN + rsw_b;
        float af[2 * kPF], bf[2 * kPF];
#pragma unroll
        for (int it = 0; it < kPF; ++it) {
            af[it] = Ab[(it & 15) * POS + (it >> 4) * 2 * BM];
            bf[it] = Bb[(it & 15) * POS + (it >> 4) * 2 * BN];
        }
        static_for(
            [&](int it) {
                const int p = it & 15, nx = it + kPF;
                if (nx < 64) {
                    af[nx % (2 * kPF)] = Ab[(nx & 15) * POS + (nx >> 4) * 2 * BM];
                    bf[nx % (2 * kPF)] = Bb[(nx & 15) * POS + (nx >> 4) * 2 * BN];
                }
                if (it < 2) d_slice(it, buf ^ 1);
                if (it >= 2 && it < 6) g_slice(it - 2, true);
                if (it >= 6 && it <= 20 && (it & 1) == 0) d_slice((it - 6) / 2 + 2, buf ^ 1);
                if (it >= 7 && it <= 19 && (it & 3) == 3) l_slice((it - 7) / 4);
                if (it >= 22 && it < 26) x_slice(it - 22, buf ^ 1);
                if (it >= 26 && it <= 40 && (it & 1) == 0) x_slice((it - 26) / 2 + 4, buf ^ 1);
                if (it >= 27 && it <= 60 && (it % 3) == 0) {
                    const int k = (it - 27) / 3;                 // column-major: the columns X0..X3 need first come first
                    l_slice(4 + (k % 3) * 4 + k / 3);
                }
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[it % (2 * kPF)], bf[it % (2 * kPF)], acc[p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            },
            std::make_integer_sequence<int, 64>{});
        __syncthreads();
    }

    // ---- output transform A^T M A per lane: lane holds ci = lane & 31 and co rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ci = n0 + wn * 32 + l31;
    const bool ci_ok = ci < a.cin;
    float* wz = a.ws + (long long)blockIdx.y * a.Mp * a.Np + ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        float t0[4], t1[4], t2[4];     // rows of A^T M: (M0+M1+M2), (M1-M2), (M1+M2+M3) per column pj
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0j = acc[0 + j][r], m1j = acc[4 + j][r], m2j = acc[8 + j][r], m3j = acc[12 + j][r];
            t0[j] = m0j + m1j + m2j;
            t1[j] = m1j - m2j;
            t2[j] = m1j + m2j + m3j;
        }
        float o[9];
        o[0] = t0[0] + t0[1] + t0[2]; o[1] = t0[1] - t0[2]; o[2] = t0[1] + t0[2] + t0[3];
        o[3] = t1[0] + t1[1] + t1[2]; o[4] = t1[1] - t1[2]; o[5] = t1[1] + t1[2] + t1[3];
        o[6] = t2[0] + t2[1] + t2[2]; o[7] = t2[1] - t2[2]; o[8] = t2[1] + t2[2] + t2[3];
        if (ci_ok && co < a.cout) {
            float* dst = wz + (long long)co * a.Np;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) dst[tap * a.CQp] = o[tap];
        }
    }
}

// shared with conv_wgrad.hip
float* conv_workspace(hipStream_t stream, size_t bytes);
int wgrad_reduce_launch(hipStream_t s, const float* ws, float* dw, int ksplit, int Mp, int Np, int CP, int CQ, int CQp, int ntaps);

constexpr int kWinoWgradLds = (2 * 2 * 16 * kKT * 64) * 4;

bool wino_wgrad_ok(const w2l_conv_geom* g, int N, int H, int W, int x_cs, int dz_cs) {
    if (g->transposed || g->kh != 3 || g->kw != 3 || g->sh != 1 || g->sw != 1 || g->ph != 1 || g->pw != 1) return false;
    // the 64 x 64 channel tile is fixed: layers that fill less than 3/4 of their padded tiles (32 -> 32, 80 -> 32) are faster
    // on the direct GEMM's narrower tiles (measured: tools/wgrad_sweep.py, profiles/r01/i_wgrad_sweep.txt)
    const long long padded = (long long)round_up(g->cin, 64) * round_up(g->cout, 64);
    if ((long long)g->cin * g->cout * 4 < padded * 3) return false;
    // index arithmetic of the kernel: single wrap per K-step (tile grid at least 3 x 3), 24-bit multiplies
    if (H < 5 || W < 5 || (long long)N * H >= (1 << 22) || (long long)W * x_cs * 4 >= (1 << 22) || (long long)W * dz_cs * 4 >= (1 << 22))
        return false;
    const long long T = (long long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const long long tiles = (long long)ceil_div(g->cout, 64) * ceil_div(g->cin, 64);
    return T >= 64 * kKT && T / tiles >= 4 * kKT && T < (1ll << 24);    // enough tiles to amortise the 16-accumulator epilogue
}

int wino_wgrad_launch(const w2l_conv_geom* g, hipStream_t s, int N, int H, int W, const float* x, int x_cs, const float* dz,
                      int dz_cs, float* dweight) {
    static bool attr_done = false;
    if (!attr_done) {
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_wgrad_f32_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kWinoWgradLds));
        attr_done = true;
    }
    WinoWgradArgs a;
    a.x = x; a.dz = dz;
    a.N = N; a.H = H; a.W = W;
    a.cin = g->cin; a.cin_p = round_up(g->cin, 4); a.x_cs = x_cs;
    a.cout = g->cout; a.cout_p = round_up(g->cout, 4); a.dz_cs = dz_cs;
    a.TH = (H + 1) / 2; a.TW = (W + 1) / 2; a.T = N * a.TH * a.TW;
    a.inv_thw = 1.0f / (float)(a.TH * a.TW);
    a.inv_tw = 1.0f / (float)a.TW;
    const int tiles_m = ceil_div(g->cout, 64);
    a.tiles_n = ceil_div(g->cin, 64);
    a.CQp = a.cin_p;
    a.Mp = g->cout;
    a.Np = 9 * a.CQp;
    const int tiles = tiles_m * a.tiles_n;
    // one workgroup per CU (512-register waves): split K so that the grid is about one full round of 256 workgroups
    long long ks = tiles >= 256 ? 1 : 256 / tiles;
    const long long max_by_k = a.T / (4 * kKT) > 0 ? a.T / (4 * kKT) : 1;
    if (ks > max_by_k) ks = max_by_k;
    const long long max_by_ws = (512ll << 20) / ((long long)a.Mp * a.Np * 4);
    if (ks > max_by_ws) ks = max_by_ws;
    if (ks < 1) ks = 1;
    a.chunk = round_up(ceil_div(a.T, (int)ks), kKT);
    const int ksplit = ceil_div(a.T, a.chunk);
    a.ws = conv_workspace(s, (size_t)ksplit * a.Mp * a.Np * sizeof(float));
    if (!a.ws) return W2L_ERR_NOMEM;
    // the reduce only reads (co < cout, ci < cin), all of which the kernel writes: no clearing of the workspace
    // 16 transform positions x (64 x 64 channel tile) multiply-adds per 2x2 output tile
    if (flops_counting()) flops_add(2ll * 16 * 64 * 64 * (long long)tiles * ksplit * a.chunk, 2);
    hipLaunchKernelGGL(conv_wino_wgrad_f32_kernel, dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), kWinoWgradLds, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    return wgrad_reduce_launch(s, a.ws, dweight, ksplit, a.Mp, a.Np, g->cout, g->cin, a.CQp, 9);
}

}  // namespace w2l

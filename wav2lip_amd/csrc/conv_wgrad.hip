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
};

__device__ __forceinline__ f32x4 gload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

template <int BM, int BN, int BK>
constexpr int wgrad_lds_bytes() {
    return (2 * BK * BM + 2 * BK * BN) * 4 + 2 * BK * 4 * 4;
}

template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_kernel(const WgradKArgs a) {
    static_assert(WM * WN == 4 && BM == 64 * WM && BN == 64 * WN, "4 waves, 64x64 per wave");
    constexpr int CGA = BM / 4, RA = 256 / CGA, PA = BK / RA;   // float4 column groups, rows per pass, passes
    constexpr int CGB = BN / 4, RB = 256 / CGB, PB = BK / RB;
    static_assert(PA >= 1 && PB >= 1 && PA * RA == BK && PB * RB == BK, "staging must tile the K-step");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);           // [2][BK][BM]
    float* Bs = As + 2 * BK * BM;                         // [2][BK][BN]
    int* s_rows = reinterpret_cast<int*>(Bs + 2 * BK * BN);  // [2][BK][4]: p pixel (or -1), q base pixel, iy0, ix0

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int tile_n = blockIdx.x % a.tiles_n;
    const int tile_m = blockIdx.x / a.tiles_n;
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
                const int n = pix / HWp;
                const int rem = pix - n * HWp;
                const int y = rem / a.Wp;
                const int x = rem - y * a.Wp;
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

    f32x4 ra[PA], rb[PB];
    auto gload = [&](int step) {
        const int* rows = s_rows + (step & 1) * BK * 4;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pix = rows[(ra0 + i * RA) * 4];
            const bool ok = a_col_ok & (pix >= 0);
            ra[i] = gload4(rp, ok ? (unsigned)pix * (unsigned)a.p_cs * 4u + a_col_off : kGOob);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int4 e = *reinterpret_cast<const int4*>(rows + (rb0 + i * RB) * 4);
            const bool ok = b_col_ok & ((unsigned)(e.z + ky) < (unsigned)a.Hq) & ((unsigned)(e.w + kx) < (unsigned)a.Wq);
            rb[i] = gload4(rq, ok ? (unsigned)(e.y + b_delta) * (unsigned)a.q_cs * 4u + b_col_off : kGOob);
        }
    };
    auto lds_store = [&](int buf) {
        float* Ab = As + buf * BK * BM;
        float* Bb = Bs + buf * BK * BN;
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(Ab + (ra0 + i * RA) * BM + ca) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(Bb + (rb0 + i * RB) * BN + cb) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    compute_rows(0);
    __syncthreads();
    gload(0);
    compute_rows(1);
    lds_store(0);
    __syncthreads();

    const int frag = 2 * (lane & 31);
    const int khalf = lane >> 5;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        gload(step + 1);          // rows of step+1 were tabulated one step ago; past the end they are all "-1" -> zeros
        compute_rows(step + 2);   // slot (step & 1): last read by gload(step) during the previous step
        const float* Ab = As + buf * BK * BM + wm * 64 + frag;
        const float* Bb = Bs + buf * BK * BN + wn * 64 + frag;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + khalf;
            const f32x2 av = *reinterpret_cast<const f32x2*>(Ab + k * BM);
            const f32x2 bv = *reinterpret_cast<const f32x2*>(Bb + k * BN);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        lds_store(buf ^ 1);       // idle buffer: its last readers finished before the previous barrier
        __syncthreads();
    }

    // partial sums -> ws[z][m][n]; lane holds rows 2*rl+i (rl = (r&3)+8*(r>>2)+4*(lane>>5)) and columns 2*(lane&31)+j
    float* wz = a.ws + ((long long)blockIdx.z * a.Mp + m0 + wm * 64) * a.Np + n0 + wn * 64 + frag;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * khalf;
            f32x2 v = {acc[i][0][r], acc[i][1][r]};
            *reinterpret_cast<f32x2*>(wz + (long long)(2 * rl + i) * a.Np) = v;
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

float* conv_workspace(size_t bytes);   // conv_igemm.hip: grow-only scratch shared by the split-K paths

struct WgradCfg {
    int bm, bn, bk;
    void (*kernel)(const WgradKArgs);
    int lds;
};
static const WgradCfg kWgradCfgs[] = {
    {128, 128, 32, conv_wgrad_f32_kernel<128, 128, 2, 2, 32>, wgrad_lds_bytes<128, 128, 32>()},
    {64, 256, 16, conv_wgrad_f32_kernel<64, 256, 1, 4, 16>, wgrad_lds_bytes<64, 256, 16>()},
};

int wgrad_init_attrs() {
    static bool done = false;
    if (done) return W2L_OK;
    for (const WgradCfg& c : kWgradCfgs)
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(c.kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    done = true;
    return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

extern "C" int w2l_conv_wgrad(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs,
                              const float* dz, int dz_cs, float* dweight) {
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
    const WgradCfg& cfg = kWgradCfgs[a.CPp <= 64 ? 1 : 0];
    const int tiles_m = ceil_div(a.CPp, cfg.bm);
    a.tiles_n = ceil_div(a.ncols, cfg.bn);
    a.Mp = tiles_m * cfg.bm;
    a.Np = a.tiles_n * cfg.bn;
    const long long tiles = (long long)tiles_m * a.tiles_n;
    // K splits: enough workgroups for ~4 per CU, at least 8 K-steps each, workspace capped at 512 MiB
    long long ks = ceil_div(1024, (int)(tiles < 1024 ? tiles : 1024));
    const long long max_by_k = a.K / (8 * cfg.bk) > 0 ? a.K / (8 * cfg.bk) : 1;
    if (ks > max_by_k) ks = max_by_k;
    const long long max_by_ws = (512ll << 20) / ((long long)a.Mp * a.Np * 4);
    if (ks > max_by_ws) ks = max_by_ws;
    if (ks < 1) ks = 1;
    a.chunk = round_up(ceil_div(a.K, (int)ks), cfg.bk);
    const int ksplit = ceil_div(a.K, a.chunk);
    a.ws = conv_workspace((size_t)ksplit * a.Mp * a.Np * sizeof(float));
    if (!a.ws) return W2L_ERR_NOMEM;
    hipStream_t s = static_cast<hipStream_t>(stream);
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

// Stride-2 transposed 3x3 convolution (nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1) + BN + ReLU: the five upsampling
// layers of the generator's decoder, models/wav2lip.py:63-81 via models/conv.py:33-44) with all four output phases in ONE
// workgroup.
//
// out[2qy+py][2qx+px] = sum over the taps of phase (py, px) of  x[qy+dy][qx+dx] * w[ky][kx]:
//   phase (0,0): (dy,dx,ky,kx) = (0,0,1,1)
//   phase (0,1): (0,1,1,0) (0,0,1,2)
//   phase (1,0): (1,0,0,1) (0,0,2,1)
//   phase (1,1): (1,1,0,0) (1,0,0,2) (0,1,2,0) (0,0,2,2)
// 9 (tap, phase) products over only 4 shifted views of the input.  The implicit-GEMM kernel (conv_igemm.hip) runs the phases
// as separate workgroups: each re-gathers its own A tiles (9 tile loads per input block instead of 4: the layers fetch 4.5-8x
// their input from L2/HBM, profiles/r02/h_pmc3_fetch.txt) and the one- and two-tap phases are 5-10 K-steps long, i.e. mostly
// prologue and epilogue.  Here a workgroup owns a block of input pixels (bh x bw in each of ni images, <= 128 rows) x 64 couts
// x 4 phases; per K-step of 8 channels the input block (+1 halo row / column) is loaded ONCE into LDS and every wave reads its
// four shifted A fragments from there; the weights come straight from L2 in MFMA fragment order, one 1 KB fragment per
// (tap, 32 couts, 8 channels), and feed TWO 32-row blocks each.  Wave (wm, wn) = 64 rows x 32 couts x 4 phases = 8 accumulators
// (128 registers): two workgroups per CU.  Per K-step and wave: 9 fragment loads + 8 LDS reads for 72 MFMAs.
#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kTpOob = 0x80000000u;
constexpr int kTpBM = 128;         // input pixels (GEMM rows) per workgroup
constexpr int kTpBC = 64;          // couts per workgroup
constexpr int kTpKS = 8;           // channels per K-step
constexpr int kTpLDP = 12;         // floats per raw pixel row in LDS (8 used): conflict-free b128 fragment reads
constexpr int kTpRP = 256;         // raw pixels per buffer (2 float4 loads per thread)
constexpr int kTpLDY = kTpBC + 4;
constexpr int kTpStage = kTpBM * 2 * kTpLDY;              // floats: one epilogue round = 128 rows x 2 phases x 64 couts
constexpr int kTpRaw = 2 * kTpRP * kTpLDP;                // floats: double-buffered raw block
constexpr int kTpLdsFloats = kTpStage > kTpRaw ? kTpStage : kTpRaw;
constexpr int kTpLdsBytes = kTpLdsFloats * 4 + kTpBM * 4;
static_assert(2 * kTpLdsBytes <= 160 * 1024, "two workgroups per CU");

struct Tp2KArgs {
    const float* x;
    float* y;
    const float* u;      // packed weights, tp2_pack below
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;      // input; output is N x 2H x 2W
    int cout, y_cs;
    int bh, bw, ni;      // pixel block of a workgroup: bh x bw input pixels in each of ni images (bh*bw*ni <= 128)
    int nby, nbx, ngi;
    int RH, RW, RP;      // raw region per image (bh+1, bw+1) and pixels per K-step ni*RH*RW (<= 256)
    int nks;             // cin / 8
    int tiles_n;         // cout / 64
    long long total;
    int act;
};

// (phase, shift) of tap t; shift d = 2*dy + dx
__device__ __forceinline__ constexpr int tp_phase(int t) { return t == 0 ? 0 : (t < 3 ? 1 : (t < 5 ? 2 : 3)); }
__device__ __forceinline__ constexpr int tp_shift(int t) {
    return (t == 1 || t == 7) ? 1 : ((t == 3 || t == 6) ? 2 : (t == 5 ? 3 : 0));
}

__global__ __launch_bounds__(256, 2) void conv_tp2_f32_kernel(const Tp2KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Rs = reinterpret_cast<float*>(smem);                      // [2][RP][LDP] raw block; later the staging tile
    int* s_opix = reinterpret_cast<int*>(Rs + kTpLdsFloats);         // [128] output pixel (2qy, 2qx) of a row or -1

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);
    const int Wo = 2 * a.W;
    const int bhw = a.bh * a.bw;

    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    for (unsigned jw = blockIdx.x >> 3; jw < per; jw += gw) {
    const unsigned bid = xcd * per + jw;
    if (bid >= total) break;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = (int)(bid % (unsigned)a.tiles_n);
    unsigned mb = bid / (unsigned)a.tiles_n;
    const int bx_i = (int)(mb % (unsigned)a.nbx);
    mb /= (unsigned)a.nbx;
    const int by_i = (int)(mb % (unsigned)a.nby);
    const int gi = (int)(mb / (unsigned)a.nby);
    const int n0 = tile_n * kTpBC;

    if (t < kTpBM) {                 // row table of the epilogue
        const int il = t / bhw, r = t - il * bhw;
        const int qyl = r / a.bw, qxl = r - qyl * a.bw;
        const int n = gi * a.ni + il, qy = by_i * a.bh + qyl, qx = bx_i * a.bw + qxl;
        s_opix[t] = (il < a.ni && n < a.N && qy < a.H && qx < a.W) ? (n * 2 * a.H + 2 * qy) * Wo + 2 * qx : -1;
    }

    // ---- raw block loads: slot e = t + 256*k -> (pixel p = e>>1 of the block's input region, channel quad q = e&1)
    unsigned goff[2];
    int lds_off[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = t + 256 * k;
        const int q = e & 1, p = e >> 1;
        unsigned off = kTpOob;
        if (p < a.RP) {
            const int rxx = p % a.RW, p2 = p / a.RW;
            const int ry = p2 % a.RH, il = p2 / a.RH;
            const int n = gi * a.ni + il;
            const int iy = by_i * a.bh + ry, ix = bx_i * a.bw + rxx;
            if (n < a.N && iy < a.H && ix < a.W)
                off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(q * 4)) * 4u;
        }
        goff[k] = off;
        lds_off[k] = (p < kTpRP ? p : 0) * kTpLDP + q * 4;     // p < 256 always for k = 0, 1
    }
    f32x4 rawreg[2];
    auto raw_gload = [&](int step) {
        const unsigned soff = (unsigned)(step * kTpKS * 4);
#pragma unroll
        for (int k = 0; k < 2; ++k) rawreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)goff[k], (int)soff, 0));
    };
    auto raw_store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<f32x4*>(Rs + buf * (kTpRP * kTpLDP) + lds_off[k]) = rawreg[k];
    };

    // ---- A fragments: row m = wm*64 + b*32 + (lane&31) -> raw pixel of (qy, qx); shift d adds dy*RW + dx pixels
    int abase[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = wm * 64 + b * 32 + (lane & 31);
        const int il = m / bhw, r = m - il * bhw;
        const int qyl = r / a.bw, qxl = r - qyl * a.bw;
        const int p = il < a.ni ? (il * a.RH + qyl) * a.RW + qxl : 0;      // unused row slots read pixel 0: finite, never stored
        abase[b] = p * kTpLDP + (lane >> 5) * 4;
    }
    const int sh1 = kTpLDP, sh2 = a.RW * kTpLDP, sh3 = (a.RW + 1) * kTpLDP;

    // ---- B operand: u[((nb * nks + kc) * 9 + tap) * 256 + (h*32 + n)*4 + e] = w[kc*8 + 4h + e][nb*32 + n][ky(tap)][kx(tap)]
    const int nb = (n0 >> 5) + wn;
    const int F = a.nks * 9;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u + (long long)nb * F * 256), 0, F * 1024, 0x00020000);
    const unsigned bl_lane = (unsigned)(lane * 16);
    auto bload = [&](int kc, int tap) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, (int)bl_lane, (int)((unsigned)(kc * 9 + tap) * 1024u), 0));
    };
    constexpr int RING = 3;
    f32x4 bq[RING];

    f32x16 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][p][r] = 0.f;

    // ---- prologue: raw(0) -> LDS[0]; raw(1) in registers
    const int nsteps = a.cin / kTpKS;
    raw_gload(0);
#pragma unroll
    for (int i = 0; i < RING; ++i) bq[i] = bload(0, i);
    raw_store(0);
    raw_gload(1);
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        // raw(step+1) -> LDS[buf^1] (last read during step-1, a barrier ago), then request raw(step+2)
        raw_store(buf ^ 1);
        raw_gload(step + 2);
        const float* Rb = Rs + buf * (kTpRP * kTpLDP);
        f32x4 af[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            af[b][0] = *reinterpret_cast<const f32x4*>(Rb + abase[b]);
            af[b][1] = *reinterpret_cast<const f32x4*>(Rb + abase[b] + sh1);
            af[b][2] = *reinterpret_cast<const f32x4*>(Rb + abase[b] + sh2);
            af[b][3] = *reinterpret_cast<const f32x4*>(Rb + abase[b] + sh3);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const f32x4 bc = bq[tap % RING];
            // ring of 3: taps 3..8 of this chunk, then 0..2 of the next
            bq[tap % RING] = (tap < 6) ? bload(step, tap + 3) : bload(step + 1, tap - 6);
            constexpr int kDummy = 0;
            (void)kDummy;
            const int ph = tp_phase(tap), sd = tp_shift(tap);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[b][ph] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[b][sd][e], bc[e], acc[b][ph], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue, two rounds of two phases: accumulators -> LDS staging [128 rows][2 phases][LDY] -> float4 rows of y.
    // acc[b][p][r]: row wm*64 + b*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), cout wn*32 + (lane&31), phase p = 2*py + px
    float* Ys = Rs;
    const long long npix = (long long)a.N * 2 * a.H * Wo;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + a.cout) * 4), 0x00020000);
    constexpr int CG = kTpBC / 4;
    const int c4 = t % CG;
    const int ch = n0 + c4 * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
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
                        yrow[(m * 2 + pp) * kTpLDY] = acc[b][2 * round + pp][r];
                    }
        }
        __syncthreads();
        constexpr int NIT = kTpBM * 2 * CG / 256;      // 16 float4 per thread
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 256 + t;
            const int rowp = id / CG;                  // m*2 + px
            const int m = rowp >> 1, px = rowp & 1;
            const int opix = s_opix[m];
            const f32x4 c = *reinterpret_cast<const f32x4*>(Ys + rowp * kTpLDY + c4 * 4);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xv = fmaf(c[e], sc[e], sh[e]);
                v[e] = fmaf(neg_slope, fminf(xv, 0.f), fmaxf(xv, 0.f));
            }
            const int pix = opix + round * Wo + px;
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(u32x4, v), ry,
                (int)(opix >= 0 ? ((unsigned)pix * (unsigned)a.y_cs + (unsigned)ch) * 4u : kTpOob), 0, 0);
        }
        __syncthreads();
    }
    }   // persistent loop
}

// ---- weight packing: nn.ConvTranspose2d layout [cin][cout][3][3] -> fragment order per (cout block, 8-channel chunk, tap)
struct Tp2PackArgs {
    const float* w;
    float* u;
    int cin, cout;
};

__global__ void tp2_pack_kernel(const Tp2PackArgs a) {
    const long long total = (long long)a.cout * a.cin * 9;
    const int nks = a.cin / 8;
    // tap -> (ky, kx), in the kernel's tap order
    const int kyt[9] = {1, 1, 1, 0, 2, 0, 0, 2, 2};
    const int kxt[9] = {1, 0, 2, 1, 1, 0, 2, 0, 2};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int n = (int)((i >> 2) & 31);
        const int h = (int)((i >> 7) & 1);
        const long long rest = i >> 8;
        const int tap = (int)(rest % 9);
        const long long r2 = rest / 9;
        const int kc = (int)(r2 % nks);
        const int nbk = (int)(r2 / nks);
        const int co = nbk * 32 + n;
        const int ci = kc * 8 + 4 * h + e;
        a.u[i] = a.w[(((long long)ci * a.cout + co) * 3 + kyt[tap]) * 3 + kxt[tap]];
    }
}

struct TpBlock { int bh, bw, ni; };
static const TpBlock kTpBlocks[] = {{8, 16, 1}, {16, 8, 1}, {8, 8, 2}, {4, 16, 2}, {4, 8, 4}, {4, 6, 5}, {6, 4, 5}, {2, 12, 5},
                                    {4, 12, 2}, {12, 4, 2}, {6, 6, 3}, {3, 12, 3}, {6, 12, 1}, {12, 6, 1}, {4, 4, 8}, {3, 3, 14},
                                    {2, 4, 16}, {2, 2, 28}, {1, 4, 25}, {1, 1, 64}};

static TpBlock tp2_pick_block(int N, int H, int W) {
    TpBlock best = {1, 1, 1};
    double best_cost = 1e300;
    for (const TpBlock& b : kTpBlocks) {
        if (b.ni * (b.bh + 1) * (b.bw + 1) > kTpRP || b.bh * b.bw * b.ni > kTpBM) continue;
        const double items = (double)ceil_div(H, b.bh) * ceil_div(W, b.bw) * ceil_div(N, b.ni);
        const double halo = (double)(b.bh + 1) * (b.bw + 1) / ((double)b.bh * b.bw);
        const double cost = items * (1.0 + 0.03 * halo);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}

bool tp2_ok(const w2l_conv_geom& g) {
    return g.transposed && g.kh == 3 && g.kw == 3 && g.sh == 2 && g.sw == 2 && g.ph == 1 && g.pw == 1 && g.oph == 1 && g.opw == 1 &&
           g.cin % kTpKS == 0 && g.cout % kTpBC == 0 && g.act != W2L_ACT_SIGMOID;
}

long long tp2_u_floats(int cin, int cout) { return (long long)cout * cin * 9; }

int tp2_pack(const float* w, float* u, int cin, int cout, hipStream_t stream) {
    Tp2PackArgs pa;
    pa.w = w; pa.u = u; pa.cin = cin; pa.cout = cout;
    long long blocks = (tp2_u_floats(cin, cout) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tp2_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int tp2_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tp2_f32_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kTpLdsBytes));
    done = true;
    return W2L_OK;
}

int tp2_launch(const float* x, int x_cs, float* y, int y_cs, const float* u, const float* scale, const float* shift, int N, int H,
               int W, int cin, int cout, int act, hipStream_t stream, long long* flops_out) {
    Tp2KArgs a;
    a.x = x; a.y = y; a.u = u; a.scale = scale; a.shift = shift;
    a.N = N; a.H = H; a.W = W; a.cin = cin; a.x_cs = x_cs; a.cout = cout; a.y_cs = y_cs; a.act = act;
    const TpBlock b = tp2_pick_block(N, H, W);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(H, b.bh);
    a.nbx = ceil_div(W, b.bw);
    a.ngi = ceil_div(N, b.ni);
    a.RH = b.bh + 1;
    a.RW = b.bw + 1;
    a.RP = b.ni * a.RH * a.RW;
    a.nks = cin / 8;
    a.tiles_n = cout / kTpBC;
    a.total = (long long)a.ngi * a.nby * a.nbx * a.tiles_n;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE((long long)N * 4 * H * W < (1ll << 31), "tensor too large");
    if (flops_out) {   // dry run: 9 (tap, phase) GEMMs of [items*128] x [64] x cin
        *flops_out = 2ll * 9 * a.total * kTpBM * kTpBC * cin;
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(conv_tp2_f32_kernel, dim3((unsigned)grid), dim3(256), kTpLdsBytes, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

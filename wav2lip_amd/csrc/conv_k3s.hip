// 3x3 stride-1 pad-1 convolution with 32 output channels + BN + activation (+ residual) (+ the fused 1x1 head of the generator's
// output block: models/wav2lip.py:83-85 - Conv2d(80, 32, 3, 1, 1) then nn.Conv2d(32, 3, 1) + Sigmoid) with split operands on the bf16
// matrix cores: conv_tp2s.hip's recipe on a stride-1 layer.  A workgroup owns <= 256 output pixels (bh x bw in each of ni images) x
// all 32 couts; per K-step of 16 channels the input block with its one-pixel halo is loaded ONCE, split once into three bf16 planes
// [plane][k-half][pixel][8], and the nine taps are nine shifted views of those planes (a shift is a pixel offset of the fragment
// read); six bf16 piece products per product on v_mfma_f32_32x32x16_bf16 (smallest first, fp32 accumulate): an fp32 result with the
// fp32 kernels' error.  Wave w = rows 64w .. 64w+63 x 32 couts (two accumulators); the A fragments of tap t + 1 are requested before
// the MFMAs of tap t; pre-split weights stream from L2 in fragment order through a three-tap ring (all four waves read the same ones).
// Why: the layer ran on F(2x2) Winograd on the fp32 pipe (4 fp32-pipe units per output and (cin, cout) pair); direct split products are
// 9 x 6 / 16 = 3.4, and the kernel leaves half a CU's LDS to the other batches in flight.
#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kK3Oob = 0x80000000u;
constexpr int kK3BM = 256;         // output pixels (GEMM rows) per workgroup
constexpr int kK3BC = 32;          // couts (all of them)
constexpr int kK3KS = 16;          // channels per K-step
constexpr int kK3RP = 384;         // raw pixels per buffer (3 slots of 32 bytes per thread and k-half pair)
constexpr int kK3LDY = kK3BC + 4;
constexpr int kK3KhBytes = kK3RP * 16;
constexpr int kK3PlaneBytes = 2 * kK3KhBytes;
constexpr int kK3BufBytes = 3 * kK3PlaneBytes;            // 36 KB
constexpr int kK3StageBytes = kK3BM * kK3LDY * 4;         // 36 KB
constexpr int kK3MainBytes = 2 * kK3BufBytes;
constexpr int kK3LdsBytes = kK3MainBytes + kK3BM * 4;
static_assert(kK3StageBytes <= kK3MainBytes && 2 * kK3LdsBytes <= 160 * 1024, "two workgroups per CU");

struct K3sKArgs {
    const float* x;
    float* y;
    const float* res;
    const __bf16* u;     // k3s_pack below
    const float* scale;
    const float* shift;
    const float* head_w;     // [head_c][32] or NULL
    const float* head_b;
    int head_c, head_act;
    int N, H, W, cin, x_cs, y_cs, res_cs;
    int bh, bw, ni;      // pixel block: bh x bw output pixels in each of ni images (<= 256 rows)
    int nby, nbx, ngi;
    int RH, RW, RP;      // raw region per image (bh+2, bw+2), pixels per K-step ni*RH*RW (<= 384)
    int nkc;             // cin / 16
    long long total;
    int act;
};

__device__ __forceinline__ unsigned k3_pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void k3_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = k3_pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = k3_pack_bf16x2(r0, r1);
    l = k3_pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ float k3_act(float v, int act) {
    if (act == W2L_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == W2L_ACT_RELU) return act_leaky(v, 0.f);
    if (act == W2L_ACT_LEAKY) return act_leaky(v, 0.01f);
    return v;
}

__global__ __launch_bounds__(256, 2) void conv_k3s_kernel(const K3sKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_opix = reinterpret_cast<int*>(smem + kK3MainBytes);        // [256] output pixel of a row or -1

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);
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
    unsigned mb = bid;
    const int bx_i = (int)(mb % (unsigned)a.nbx);
    mb /= (unsigned)a.nbx;
    const int by_i = (int)(mb % (unsigned)a.nby);
    const int gi = (int)(mb / (unsigned)a.nby);

    {                                // row table of the epilogue: row t -> output pixel
        const int il = t / bhw, r = t - il * bhw;
        const int qyl = r / a.bw, qxl = r - qyl * a.bw;
        const int n = gi * a.ni + il, qy = by_i * a.bh + qyl, qx = bx_i * a.bw + qxl;
        s_opix[t] = (il < a.ni && n < a.N && qy < a.H && qx < a.W) ? (n * a.H + qy) * a.W + qx : -1;
    }

    // ---- raw block loads: slot e = t + 256*k -> (pixel p = e>>1 of the block's input region, k-half kh = e&1: 8 channels = 32 bytes)
    unsigned goff[3];
    int lds_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = t + 256 * k;
        const int kh = e & 1, p = e >> 1;
        unsigned off = kK3Oob;
        if (p < a.RP) {
            const int rxx = p % a.RW, p2 = p / a.RW;
            const int ry = p2 % a.RH, il = p2 / a.RH;
            const int n = gi * a.ni + il;
            const int iy = by_i * a.bh + ry - 1, ix = bx_i * a.bw + rxx - 1;
            if (n < a.N && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(kh * 8)) * 4u;
        }
        goff[k] = off;
        lds_off[k] = kh * kK3KhBytes + p * 16;
    }
    // a wave's slot k covers pixels (256 k + 64 wave) / 2 ..: past the region it has nothing to load, split or store
    bool slot_on[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) slot_on[k] = (256 * k + wave * 64) < 2 * a.RP;
    f32x4 rawreg[3][2];
    auto raw_gload = [&](int step) {
        const unsigned soff = (unsigned)(step * kK3KS * 4);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!slot_on[k]) continue;
            rawreg[k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)goff[k], (int)soff, 0));
            rawreg[k][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(goff[k] + 16u), (int)soff, 0));
        }
    };
    auto raw_store = [&](int buf) {      // split once, three 16-byte stores per slot
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!slot_on[k]) continue;
            unsigned h[4], m[4], l[4];
            k3_split3_pair(rawreg[k][0][0], rawreg[k][0][1], h[0], m[0], l[0]);
            k3_split3_pair(rawreg[k][0][2], rawreg[k][0][3], h[1], m[1], l[1]);
            k3_split3_pair(rawreg[k][1][0], rawreg[k][1][1], h[2], m[2], l[2]);
            k3_split3_pair(rawreg[k][1][2], rawreg[k][1][3], h[3], m[3], l[3]);
            char* d = smem + buf * kK3BufBytes + lds_off[k];
            *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(d + kK3PlaneBytes) = u32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<u32x4*>(d + 2 * kK3PlaneBytes) = u32x4{l[0], l[1], l[2], l[3]};
        }
    };

    // ---- A fragments: row m = wave*64 + b*32 + (lane&31) -> raw pixel of output (qy, qx) under tap (0, 0); tap (dy, dx) adds
    // dy*RW + dx pixels; lane>>5 = k-half
    int abase[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = wave * 64 + b * 32 + (lane & 31);
        const int il = m / bhw, r = m - il * bhw;
        const int qyl = r / a.bw, qxl = r - qyl * a.bw;
        const int p = il < a.ni ? (il * a.RH + qyl) * a.RW + qxl : 0;      // unused row slots read pixel 0: finite, never stored
        abase[b] = p * 16 + (lane >> 5) * kK3KhBytes;
    }
    const int rw16 = a.RW * 16;

    // ---- B operand: u[((kc * 9 + tap) * 3 + plane) * 512 + lane * 8 + e] = piece `plane` of w[lane&31][kc*16 + 8*(lane>>5) + e][tap]
    const int F = a.nkc * 27;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.u), 0, F * 1024, 0x00020000);
    const unsigned bl_lane = (unsigned)(lane * 16);
    auto bload = [&](int kc, int tap, int plane) {       // past-the-end chunks read zero (never used)
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ru, (int)bl_lane, (int)((unsigned)((kc * 9 + tap) * 3 + plane) * 1024u), 0));
    };
    constexpr int RING = 3;
    bf16x8 bq[RING][3];

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // ---- prologue: raw(0) -> LDS[0]; raw(1) in registers
    raw_gload(0);
#pragma unroll
    for (int i = 0; i < RING; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[i][p] = bload(0, i, p);
    raw_store(0);
    raw_gload(1);
    __syncthreads();

    constexpr int kPa[6] = {2, 1, 0, 1, 0, 0};      // the six piece products of a K-chunk, smallest first
    constexpr int kPb[6] = {0, 1, 2, 0, 1, 0};

    for (int step = 0; step < a.nkc; ++step) {
        const int buf = step & 1;
        raw_store(buf ^ 1);              // raw(step+1) -> LDS[buf^1] (last read during step-1, a barrier ago)
        raw_gload(step + 2);
        const char* Ab = smem + buf * kK3BufBytes;
        bf16x8 af[2][2][3];
        auto aload = [&](int set, int tap) {
            const int sh = (tap / 3) * rw16 + (tap % 3) * 16;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[set][b][p] = *reinterpret_cast<const bf16x8*>(Ab + p * kK3PlaneBytes + abase[b] + sh);
        };
        aload(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) aload(cur ^ 1, tap + 1);
            bf16x8 bc[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bc[p] = bq[tap % RING][p];
#pragma unroll
            for (int p = 0; p < 3; ++p)      // ring of 3: taps 3..8 of this chunk, then 0..2 of the next
                bq[tap % RING][p] = (tap < 6) ? bload(step, tap + 3, p) : bload(step + 1, tap - 6, p);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][b][kPa[u]], bc[kPb[u]], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS staging [256 rows][LDY]; one thread per output pixel: scale / shift / residual / activation,
    // then either the 32 channels (float4 stores) or the fused head (head_c channels)
    // acc[b][r]: row wave*64 + b*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), cout lane&31
    float* Ys = reinterpret_cast<float*>(smem);
    {
        float* yrow = Ys + (lane & 31);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wave * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                yrow[m * kK3LDY] = acc[b][r];
            }
    }
    __syncthreads();
    {
        const int opix = s_opix[t];
        const float* row = Ys + t * kK3LDY;
        float hacc[4] = {0.f, 0.f, 0.f, 0.f};
        if (opix >= 0) {
#pragma unroll
            for (int g = 0; g < kK3BC / 4; ++g) {
                f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * g);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * g);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + 4 * g);
                f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                if (a.res) rs = *reinterpret_cast<const f32x4*>(a.res + (long long)opix * a.res_cs + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = k3_act(fmaf(v[e], sc[e], sh[e]) + rs[e], a.act);
                if (a.head_w) {
#pragma unroll
                    for (int o = 0; o < 4; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            hacc[o] = fmaf(v[e], (o < a.head_c) ? a.head_w[o * kK3BC + 4 * g + e] : 0.f, hacc[o]);
                } else {
                    *reinterpret_cast<f32x4*>(a.y + (long long)opix * a.y_cs + 4 * g) = v;
                }
            }
            if (a.head_w) {
                float* dst = a.y + (long long)opix * a.y_cs;
                for (int o = 0; o < a.head_c; ++o) dst[o] = k3_act(hacc[o] + (a.head_b ? a.head_b[o] : 0.f), a.head_act);
            }
        }
    }
    __syncthreads();      // the planes / the row table are rewritten by the next item
    }   // persistent loop
}

// ---- weight packing: nn.Conv2d layout [32][cin][3][3] -> three bf16 pieces per value in fragment order
struct K3sPackArgs {
    const float* w;
    __bf16* u;
    int cin;
};

__global__ void k3s_pack_kernel(const K3sPackArgs a) {
    const long long total = (long long)kK3BC * a.cin * 9;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int ln = (int)((i >> 3) & 63);
        const long long rest = i >> 9;
        const int tap = (int)(rest % 9);
        const int kc = (int)(rest / 9);
        const int co = ln & 31, ci = kc * 16 + 8 * (ln >> 5) + e;
        const float v = a.w[((long long)co * a.cin + ci) * 9 + tap];
        const __bf16 hp = (__bf16)v;
        const float r1 = v - (float)hp;
        const __bf16 mp = (__bf16)r1;
        __bf16* d = a.u + ((long long)kc * 9 + tap) * (3 * 512) + ln * 8 + e;
        d[0] = hp;
        d[512] = mp;
        d[1024] = (__bf16)(r1 - (float)mp);
    }
}

struct K3Block { int bh, bw, ni; };
static const K3Block kK3Blocks[] = {{16, 16, 1}, {8, 16, 2}, {16, 8, 2}, {8, 8, 3}, {4, 16, 3}, {4, 8, 6}, {8, 4, 6}, {4, 4, 10}, {2, 8, 9},
                                    {2, 4, 16}, {2, 2, 24}, {1, 4, 21}, {1, 2, 32}, {1, 1, 42}, {12, 12, 1}, {6, 6, 6}, {3, 3, 15}, {8, 12, 2}};

static K3Block k3s_pick_block(int N, int H, int W) {
    K3Block best = {1, 1, 1};
    double best_cost = 1e300;
    for (const K3Block& b : kK3Blocks) {
        if (b.ni * (b.bh + 2) * (b.bw + 2) > kK3RP || b.bh * b.bw * b.ni > kK3BM) continue;
        const double items = (double)ceil_div(H, b.bh) * ceil_div(W, b.bw) * ceil_div(N, b.ni);
        const double halo = (double)(b.bh + 2) * (b.bw + 2) / ((double)b.bh * b.bw);
        const double cost = items * (1.0 + 0.03 * halo);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}

bool k3s_ok(const w2l_conv_geom& g) {
    return !g.transposed && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 && g.cin % kK3KS == 0 &&
           g.cout == kK3BC;
}

long long k3s_u_elems(int cin) { return (long long)kK3BC * cin * 9 * 3; }

int k3s_pack(const float* w, __bf16* u, int cin, hipStream_t stream) {
    K3sPackArgs pa;
    pa.w = w; pa.u = u; pa.cin = cin;
    long long blocks = ((long long)kK3BC * cin * 9 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k3s_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int k3s_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k3s_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kK3LdsBytes));
    done = true;
    return W2L_OK;
}

// y: 32 channels per pixel, or head_c when head_w != NULL; res (32 channels) may alias x
int k3s_launch(const float* x, int x_cs, float* y, int y_cs, const float* res, int res_cs, const __bf16* u, const float* scale,
               const float* shift, const float* head_w, const float* head_b, int head_c, int head_act, int N, int H, int W, int cin,
               int act, hipStream_t stream, long long* flops_out) {
    K3sKArgs a;
    a.x = x; a.y = y; a.res = res; a.u = u; a.scale = scale; a.shift = shift;
    a.head_w = head_w; a.head_b = head_b; a.head_c = head_c; a.head_act = head_act;
    a.N = N; a.H = H; a.W = W; a.cin = cin; a.x_cs = x_cs; a.y_cs = y_cs; a.res_cs = res_cs; a.act = act;
    const K3Block b = k3s_pick_block(N, H, W);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(H, b.bh);
    a.nbx = ceil_div(W, b.bw);
    a.ngi = ceil_div(N, b.ni);
    a.RH = b.bh + 2;
    a.RW = b.bw + 2;
    a.RP = b.ni * a.RH * a.RW;
    a.nkc = cin / kK3KS;
    a.total = (long long)a.ngi * a.nby * a.nbx;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE(head_w == nullptr || (head_c >= 1 && head_c <= 4), "fused head: 1 .. 4 channels");
    if (flops_out) {   // dry run: 9 tap GEMMs of [items*256] x [32] x cin, six bf16 piece products per product
        *flops_out = 6ll * 2 * 9 * a.total * kK3BM * kK3BC * cin;
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(conv_k3s_kernel, dim3((unsigned)grid), dim3(256), kK3LdsBytes, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

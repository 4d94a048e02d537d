// The generator's first layer - Conv2d(6, 16, kernel 7, stride 1, padding 3) + BN + ReLU at 96x96 (models/wav2lip.py:10-11 via
// models/conv.py:5-19) - with split operands on the bf16 matrix cores (DESIGN 3d's arithmetic: every fp32 operand as the exact sum of
// three bf16 pieces, six piece products per product, fp32 accumulate).
//
// Why a kernel of its own: K = 49 taps x 8 (padded) channels and 16 couts.  The implicit GEMM gathers 49 32-byte pieces per output
// pixel and K-step and runs a 32-cout tile half empty: 0.177 ms for 7.4 GMAC, 3.4 % of the step.  Here a workgroup owns a 16 x 16
// block of output pixels of one image: the 22 x 22 x 8 input region is loaded ONCE, split once into three bf16 planes [plane][pixel][8],
// and the whole contraction runs out of LDS on v_mfma_f32_16x16x32_bf16 - 16 pixels of one block row x 16 couts x (4 taps x 8
// channels) per instruction, the A fragment of lane l being the 8 channels of pixel (row, l & 15) shifted by tap 4c + (l >> 4).  The
// pre-split weights (13 chunks x 3 planes x 1 KB) are copied into LDS once per (persistent) workgroup.  Wave w = block rows 4w .. 4w+3.
#include "w2l_common.h"

namespace w2l {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kS7Oob = 0x80000000u;
constexpr int kS7B = 16;                       // block edge (output pixels)
constexpr int kS7RW = kS7B + 6;                // region edge
constexpr int kS7RP = kS7RW * kS7RW;           // 484 region pixels
constexpr int kS7PlaneBytes = 512 * 16;        // [pixel][8 bf16], padded to 512 pixels
constexpr int kS7Chunks = 13;                  // 4 taps per chunk: taps 49 .. 51 carry zero weights
constexpr int kS7ABytes = 3 * kS7PlaneBytes;   // 24 KB
constexpr int kS7BBytes = kS7Chunks * 3 * 1024;   // 39 KB
constexpr int kS7LdsBytes = kS7ABytes + kS7BBytes;
static_assert(2 * kS7LdsBytes <= 160 * 1024, "two workgroups per CU");

struct Stem7sKArgs {
    const float* x;
    float* y;
    const __bf16* u;     // stem7s_pack below
    const float* scale;
    const float* shift;
    int N, H, W, x_cs, y_cs;
    int nby, nbx;
    long long total;
    int act;
};

__device__ __forceinline__ unsigned s7_pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void s7_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = s7_pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = s7_pack_bf16x2(r0, r1);
    l = s7_pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

__global__ __launch_bounds__(256, 2) void conv_stem7s_kernel(const Stem7sKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                      // [3][512][16 B]
    char* Bs = smem + kS7ABytes;          // [13][3][1 KB]

    const int t0 = threadIdx.x;
    // the pre-split weights, once per workgroup
    for (int i = t0; i < kS7BBytes / 16; i += 256)
        *reinterpret_cast<u32x4*>(Bs + i * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.u) + i * 16);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + 8) * 4), 0x00020000);
    const long long npix = (long long)a.N * a.H * a.W;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix - 1) * a.y_cs + 16) * 4), 0x00020000);

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
        const int bx_i = (int)(bid % (unsigned)a.nbx);
        const unsigned mb = bid / (unsigned)a.nbx;
        const int by_i = (int)(mb % (unsigned)a.nby);
        const int n = (int)(mb / (unsigned)a.nby);
        const int y0 = by_i * kS7B, x0 = bx_i * kS7B;

        // ---- the 22 x 22 x 8 input region -> three bf16 planes (slot e = t + 256 k -> region pixel e)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = t + 256 * k;
            if (e < kS7RP) {
                const int ry_ = e / kS7RW, rx_ = e - ry_ * kS7RW;
                const int iy = y0 + ry_ - 3, ix = x0 + rx_ - 3;
                const unsigned off = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                                         ? (unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs * 4u : kS7Oob;
                const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
                const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off + 16u), 0, 0));
                unsigned h[4], m[4], l[4];
                s7_split3_pair(v0[0], v0[1], h[0], m[0], l[0]);
                s7_split3_pair(v0[2], v0[3], h[1], m[1], l[1]);
                s7_split3_pair(v1[0], v1[1], h[2], m[2], l[2]);
                s7_split3_pair(v1[2], v1[3], h[3], m[3], l[3]);
                char* d = As + e * 16;
                *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
                *reinterpret_cast<u32x4*>(d + kS7PlaneBytes) = u32x4{m[0], m[1], m[2], m[3]};
                *reinterpret_cast<u32x4*>(d + 2 * kS7PlaneBytes) = u32x4{l[0], l[1], l[2], l[3]};
            }
        }
        __syncthreads();

        // ---- 13 chunks x 4 block rows x 6 piece products.  Lane l: A = pixel (row, l & 15) + tap 4c + (l >> 4); B = cout l & 15
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int g = lane >> 4;
        const int apix = (wave * 4) * kS7RW + (lane & 15);      // region pixel of (block row 4w, column l & 15) under tap (0, 0)
        constexpr int kPa[6] = {2, 1, 0, 1, 0, 0};              // smallest piece products first
        constexpr int kPb[6] = {0, 1, 2, 0, 1, 0};
        // software-pipelined over the 52 (chunk, block row) groups of 6 MFMAs: the A fragments of group i + 1 and the B fragments of
        // chunk c + 1 are requested before the MFMAs of group i are issued
        auto tap_shift = [&](int c) {
            int tap = 4 * c + g;
            tap = tap < 49 ? tap : 0;                           // taps 49 .. 51: zero weights, any finite A
            const int dy = (tap * 37) >> 8;                     // tap / 7 for tap < 56
            return (apix + dy * kS7RW + (tap - 7 * dy)) * 16;
        };
        auto aload = [&](int abyte, int j, int p) {
            return *reinterpret_cast<const bf16x8*>(As + p * kS7PlaneBytes + abyte + j * (kS7RW * 16));
        };
        auto bload = [&](int c, int p) { return *reinterpret_cast<const bf16x8*>(Bs + (c * 3 + p) * 1024 + lane * 16); };
        bf16x8 af[2][3], bq[2][3];
        int abyte = tap_shift(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) { bq[0][p] = bload(0, p); af[0][p] = aload(abyte, 0, p); }
#pragma unroll
        for (int i = 0; i < kS7Chunks * 4; ++i) {
            const int c = i >> 2, j = i & 3;
            if (i + 1 < kS7Chunks * 4) {
                if (j == 3) abyte = tap_shift(c + 1);
#pragma unroll
                for (int p = 0; p < 3; ++p) af[(i + 1) & 1][p] = aload(abyte, (i + 1) & 3, p);
            }
            if (j == 1 && c + 1 < kS7Chunks) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[(c + 1) & 1][p] = bload(c + 1, p);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of the MFMAs (the scheduler would sink them to their uses)
#pragma unroll
            for (int u = 0; u < 6; ++u)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i & 1][kPa[u]], bq[c & 1][kPb[u]], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: acc[j][r] = pixel (block row 4w + j, column 4 (l >> 4) + r), cout l & 15
        const int co = lane & 15;
        const float sc = a.scale[co], shf = a.shift[co];
        const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = y0 + wave * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = x0 + 4 * g + r;
                const float xv = fmaf(acc[j][r], sc, shf);
                const float v = fmaf(neg_slope, fminf(xv, 0.f), fmaxf(xv, 0.f));
                const unsigned off = (oy < a.H && ox < a.W) ? ((unsigned)((n * a.H + oy) * a.W + ox) * (unsigned)a.y_cs + (unsigned)co) * 4u : kS7Oob;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)off, 0, 0);
            }
        }
        __syncthreads();      // the planes are rewritten by the next item
    }
}

// ---- weight packing: nn.Conv2d layout [16][cin][7][7] -> u[(c * 3 + plane) * 512 + lane * 8 + e] = piece `plane` of
// w[lane & 15][e][tap 4c + (lane >> 4)] (zero for e >= cin and for taps 49 .. 51)
struct Stem7sPackArgs {
    const float* w;
    __bf16* u;
    int cin;
};

__global__ void stem7s_pack_kernel(const Stem7sPackArgs a) {
    const int total = kS7Chunks * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 7, ln = (i >> 3) & 63, c = i >> 9;
        const int tap = 4 * c + (ln >> 4), co = ln & 15;
        const float v = (e < a.cin && tap < 49) ? a.w[((long long)co * a.cin + e) * 49 + tap] : 0.f;
        const __bf16 hp = (__bf16)v;
        const float r1 = v - (float)hp;
        const __bf16 mp = (__bf16)r1;
        __bf16* d = a.u + (long long)c * (3 * 512) + ln * 8 + e;
        d[0] = hp;
        d[512] = mp;
        d[1024] = (__bf16)(r1 - (float)mp);
    }
}

bool stem7s_ok(const w2l_conv_geom& g) {
    // 5 .. 8 input channels: the engine pads activations to a multiple of 4 channels, the kernel reads 8 per pixel - with cin <= 4 the
    // channels [4, 8) would belong to somebody else (their weights are zero, but 0 x NaN is NaN)
    return !g.transposed && g.kh == 7 && g.kw == 7 && g.sh == 1 && g.sw == 1 && g.ph == 3 && g.pw == 3 && g.cin > 4 && g.cin <= 8 &&
           g.cout == 16 && g.act != W2L_ACT_SIGMOID;
}

long long stem7s_u_elems() { return (long long)kS7Chunks * 3 * 512; }

int stem7s_pack(const float* w, __bf16* u, int cin, hipStream_t stream) {
    Stem7sPackArgs pa;
    pa.w = w; pa.u = u; pa.cin = cin;
    hipLaunchKernelGGL(stem7s_pack_kernel, dim3(26), dim3(256), 0, stream, pa);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int stem7s_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem7s_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kS7LdsBytes));
    done = true;
    return W2L_OK;
}

// x: channels [0, 8) of every pixel are read (the engine pads activations to a multiple of 4 channels with zeros; x_cs >= 8 is checked
// by the caller), y: 16 channels
int stem7s_launch(const float* x, int x_cs, float* y, int y_cs, const __bf16* u, const float* scale, const float* shift, int N, int H,
                  int W, int act, hipStream_t stream, long long* flops_out) {
    Stem7sKArgs a;
    a.x = x; a.y = y; a.u = u; a.scale = scale; a.shift = shift;
    a.N = N; a.H = H; a.W = W; a.x_cs = x_cs; a.y_cs = y_cs; a.act = act;
    a.nby = ceil_div(H, kS7B);
    a.nbx = ceil_div(W, kS7B);
    a.total = (long long)N * a.nby * a.nbx;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    if (flops_out) {   // dry run: [items * 256 pixels] x [16 couts] x [13 chunks * 32], six bf16 piece products per product
        *flops_out = 6ll * 2 * a.total * 256 * 16 * (kS7Chunks * 32);
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(conv_stem7s_kernel, dim3((unsigned)grid), dim3(256), kS7LdsBytes, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

// Stride-2 transposed 3x3 convolutions of the bf16-storage training path with ALL FOUR OUTPUT PHASES in one workgroup: the forward
// of the decoder's upsampling layers (models/wav2lip.py:60-79 `Conv2dTranspose(.., kernel_size=3, stride=2, padding=1,
// output_padding=1)` through models/conv.py:33-44) and the DATA GRADIENT of the stride-2 3x3 convs of all three networks
// (models/wav2lip.py:14-30, models/syncnet.py:12-30; wav2lip_train.py:230 / color_syncnet_train.py:163 `loss.backward()`), which is
// the same geometry: out[2q + p] = sum over the taps of phase p of in[q + d] * w, d in {0, 1} per axis.
//
// Why.  conv_bf16.hip runs such a layer as four independent GEMMs (one per output phase, K = 1 / 2 / 2 / 4 taps x cin): 3..10
// K-steps per workgroup under a full prologue and epilogue, the input fetched once per (phase, tap), every phase writing every other
// pixel of the output.  face_decoder_blocks.6.0 (160 -> 64, 48x48 -> 96x96, 320 frames) took 0.51 ms at 266 TFLOP/s against a byte
// floor of 0.13 ms; the data gradient of face_encoder_blocks.1.0 (32 -> 16 at 96x96) 0.32 ms at 21 TFLOP/s (profiles/r06/a_*).
//
// Here a workgroup (4 waves) owns an 8 x TW block of INPUT pixels (TW = 8 for the 64-cout tile, 16 for the 32-cout tile) and writes
// the 16 x 2TW output pixels it produces, all four phases, BN couts.  Per 32-channel K-chunk the (8+1) x (TW+1) input box (tile + the
// right / bottom halo the taps d = 1 reach) and the nine (phase, tap) weight rows [BN][32] are staged ONCE in LDS - global loads into
// registers one chunk ahead, so that the next chunk's latency runs under this chunk's MFMAs - and every wave issues, per 16-channel
// substep, 4 pixel fragments (one per input offset) + 9 weight fragments for 9 v_mfma_f32_32x32x16_bf16 into its four phase
// accumulators.  The weights are read from the per-phase slabs conv_bf16.hip already packs ([cout_p][kp], K = (tap, c)): no second
// weight format, nothing to re-pack after an optimiser step.
// Roles as in conv_box_bf16.hip: weights are the MFMA's A operand (rows = couts), pixels its B operand (columns), so that a lane ends
// up with 16 couts of ONE input pixel per phase accumulator and the epilogue is register-only (v_permlane32_swap -> 16-byte rows):
// scale / shift (bias), optional residual (an accumulated gradient), activation, bf16 stores, and - forward of a batch-statistics
// block - the per-wave column sums / sums of squares of the rounded outputs that bn_stats_from_partials consumes.
// Not here (the launcher falls back to conv_bf16.hip): other kernel sizes / strides, cin not a multiple of 32, split-K (small
// spatial extents), BatchNorm-backward sums in the epilogue (such a data-gradient launch reports "not fused").
#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kTpOob = 0x80000000u;
constexpr int kTpKC = 32;           // channels per K-chunk
constexpr int kTpRow = 80;          // LDS bytes per (pixel | weight row) and chunk: 64 of data + one pad slot (bank spread of the 16-byte reads)

struct Tp2bArgs {
    const void* x;
    void* y;
    const void* res;
    const void* w;          // conv_bf16.hip's per-phase slabs [cout_p][kp]
    const float* scale;
    const float* shift;
    float* stats;           // NULL or [tiles * PG][2][cout_p]
    long long w_elems;
    int N, H, W, x_cs, y_cs, res_cs, cin_p, cout, cout_p, act;
    int tiles_x, tiles_y, ntiles, cout_tiles;
    // the nine (phase, tap) pairs in the order of kTpPhase / kTpOff: element offset of the pair's first K entry in row 0 of its
    // phase's slab, and that slab's row length
    long long pt_base[9];
    int pt_kp[9];
};

// The nine (phase, tap) pairs of a 3x3 / stride 2 / padding 1 transposed layer as conv_bf16.hip enumerates them (phases (py, px) in
// row-major order, taps by ascending (ky, kx)): the output phase py * 2 + px a pair accumulates into and the input offset
// dy * 2 + dx it reads.  tp2b_ok checks a layer's tap tables against these before the kernel is chosen.
__device__ __host__ constexpr int tp_phase(int i) { return i == 0 ? 0 : (i < 3 ? 1 : (i < 5 ? 2 : 3)); }
__device__ __host__ constexpr int tp_off(int i) {
    return i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? 0 : (i == 3 ? 2 : (i == 4 ? 0 : (i == 5 ? 3 : (i == 6 ? 2 : (i == 7 ? 1 : 0)))))));
}

// Workgroup = 8 waves.  Wave g owns GROUP b * 8 + g: a 4 x 8 block of input pixels of one image (groups enumerate (image, block
// row, block column) linearly, so extents that are no multiple of a tile waste nothing) and all BN couts of the workgroup's cout
// tile: 4 phases x BN / 32 accumulators.  LDS per 32-channel chunk: eight 5 x 9 pixel boxes (block + right / bottom halo) and
// the nine weight-row sets [BN][32 ch]; the weights are fetched once per 256 input pixels.
template <int BN>
__global__ __launch_bounds__(512) void conv_tp2b_bf16_kernel(const Tp2bArgs a) {
    constexpr int NG = 8;                             // groups (= waves) per workgroup
    constexpr int CT = BN / 32;                       // 32-cout accumulator tiles per wave
    constexpr int GPX = 5 * 9;                        // box pixels per group
    constexpr int NTHR = NG * 64;
    constexpr int BOXB = NG * GPX * kTpRow;
    constexpr int NBOX = (NG * GPX * 4 + NTHR - 1) / NTHR;   // 16-byte box items per thread
    constexpr int NWT = (9 * BN * 4 + NTHR - 1) / NTHR;      // 16-byte weight items per thread
    __shared__ __attribute__((aligned(16))) char Box[BOXB];
    __shared__ __attribute__((aligned(16))) char Wl[9 * BN * kTpRow];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5, n = lane & 31;

    const int ctile = blockIdx.x % a.cout_tiles;      // cout tile fastest: the workgroups that share the input boxes run together
    const int gblock = blockIdx.x / a.cout_tiles;
    const int n0 = ctile * BN;
    const int per = a.tiles_x * a.tiles_y;            // groups per image

    const long long npix_in = (long long)a.N * a.H * a.W;
    const long long npix_out = npix_in * 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)(((npix_in - 1) * a.x_cs + a.cin_p) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)(a.w_elems * 2), 0x00020000);
    const int cout8 = (a.cout + 7) & ~7;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(((npix_out - 1) * a.y_cs + cout8) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix_out - 1) * a.res_cs + cout8) * 2) : 0, 0x00020000);

    // ---- this thread's staging items (chunk-independent parts)
    unsigned box_src[NBOX];       // byte offset of the item's pixel + slot in x at chunk 0, or kTpOob
    int box_dst[NBOX];            // LDS byte offset, or -1
#pragma unroll
    for (int j = 0; j < NBOX; ++j) {
        const int item = t + NTHR * j;
        const int bp = item >> 2, slot = item & 3;    // box pixel over all groups
        const int g = bp / GPX, q = bp - g * GPX;
        const int by = q / 9, bx = q - by * 9;
        const int grp = gblock * NG + g;
        const bool in_box = g < NG;
        const int img = grp / per, rem = grp - img * per;
        const int iy = (rem / a.tiles_x) * 4 + by, ix = (rem % a.tiles_x) * 8 + bx;
        const bool ok = in_box && grp < a.ntiles && iy < a.H && ix < a.W;
        box_src[j] = ok ? (unsigned)((((img * a.H + iy) * a.W + ix) * a.x_cs + slot * 8) * 2) : kTpOob;
        box_dst[j] = in_box ? bp * kTpRow + slot * 16 : -1;
    }
    unsigned w_src[NWT];
    int w_dst[NWT];
#pragma unroll
    for (int j = 0; j < NWT; ++j) {
        const int item = t + NTHR * j;
        const bool in_w = item < 9 * BN * 4;
        const int pt = in_w ? item / (BN * 4) : 0;
        const int rem = item - pt * (BN * 4);
        const int row = rem >> 2, slot = rem & 3;
        const bool ok = in_w && n0 + row < a.cout_p;
        w_src[j] = ok ? (unsigned)((a.pt_base[pt] + (long long)(n0 + row) * a.pt_kp[pt] + slot * 8) * 2) : kTpOob;
        w_dst[j] = in_w ? (pt * BN + row) * kTpRow + slot * 16 : -1;
    }

    u32x4 breg[NBOX], wreg[NWT];
    auto fetch = [&](int chunk) {
        const unsigned cb = (unsigned)(chunk * kTpKC * 2);
#pragma unroll
        for (int j = 0; j < NBOX; ++j)
            breg[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(box_src[j] == kTpOob ? kTpOob : box_src[j] + cb), 0, 0);
#pragma unroll
        for (int j = 0; j < NWT; ++j)
            wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(w_src[j] == kTpOob ? kTpOob : w_src[j] + cb), 0, 0);
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < NBOX; ++j)
            if (box_dst[j] >= 0) *reinterpret_cast<u32x4*>(Box + box_dst[j]) = breg[j];
#pragma unroll
        for (int j = 0; j < NWT; ++j)
            if (w_dst[j] >= 0) *reinterpret_cast<u32x4*>(Wl + w_dst[j]) = wreg[j];
    };

    // this wave's group, this lane's input pixel inside it (B operand column) and its weight row (A operand row)
    const int grp = gblock * NG + wave;
    const int img = grp / per, grem = grp - img * per;
    const int pr = n >> 3, pc = n & 7;
    const int iy = (grem / a.tiles_x) * 4 + pr, ix = (grem % a.tiles_x) * 8 + pc;
    const int pbase = (wave * GPX + pr * 9 + pc) * kTpRow + h * 16;
    const int wbase = n * kTpRow + h * 16;

    f32x16 acc[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.f;

    const int nchunks = a.cin_p / kTpKC;
    fetch(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                      // the previous chunk's fragment reads are done
        stash();
        __syncthreads();
        if (chunk + 1 < nchunks) fetch(chunk + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                xf[o] = *reinterpret_cast<const bf16x8*>(Box + pbase + ((o >> 1) * 9 + (o & 1)) * kTpRow + ks * 32);
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wl + wbase + (i * BN + c * 32) * kTpRow + ks * 32);
                    acc[tp_phase(i)][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[tp_off(i)], acc[tp_phase(i)][c], 0, 0, 0);
                }
        }
    }

    // ---- epilogue.  Accumulator register 4 g + e of lane (n, h) = cout 32 c + 8 g + 4 h + e of input pixel n; after the swaps
    // (conv_box_bf16.hip) lane (n, h) holds couts 16 k + 8 h + {0..7} in registers 8 k + {0..7}: two 16-byte rows per accumulator.
    const bool pix_ok = (grp < a.ntiles) & (iy < a.H) & (ix < a.W);
    const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
    const bool is_sigmoid = a.act == W2L_ACT_SIGMOID;
    const bool has_res = a.res != nullptr;
    const bool want_stats = a.stats != nullptr;
    float st0[CT][2][8], st1[CT][2][8];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) { st0[c][k][j] = 0.f; st1[c][k][j] = 0.f; }
    const int Wo = 2 * a.W;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned opix = (unsigned)((img * 2 * a.H + 2 * iy + (p >> 1)) * Wo + 2 * ix + (p & 1));
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            u32x4 rv[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rv[k] = u32x4{0u, 0u, 0u, 0u};
                const int c0 = n0 + c * 32 + 16 * k + 8 * h;
                if (has_res)
                    rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(pix_ok && c0 < cout8 ? (opix * (unsigned)a.res_cs + (unsigned)c0) * 2u : kTpOob), 0, 0);
            }
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[r] = acc[p][c][r];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(vv[8 * k + e]), "+v"(vv[8 * k + 4 + e]));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c0 = n0 + c * 32 + 16 * k + 8 * h;      // this lane's 8 consecutive couts
                const bf16x8 rb = __builtin_bit_cast(bf16x8, rv[k]);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool live = c0 + j < a.cout;
                    const float sc = (a.scale && live) ? a.scale[c0 + j] : 1.f;
                    const float sh = (a.shift && live) ? a.shift[c0 + j] : 0.f;
                    float v = vv[8 * k + j] * sc + sh + (float)rb[j];
                    if (is_sigmoid) v = 1.0f / (1.0f + expf(-v));
                    else v = act_leaky(v, neg_slope);
                    o[j] = (__bf16)(live ? v : 0.f);
                    if (want_stats) {
                        const float vr = pix_ok ? (float)o[j] : 0.f;
                        st0[c][k][j] += vr;
                        st1[c][k][j] += vr * vr;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry,
                                                       (int)(pix_ok && c0 < cout8 ? (opix * (unsigned)a.y_cs + (unsigned)c0) * 2u : kTpOob), 0, 0);
            }
        }
    }
    if (want_stats && grp < a.ntiles) {
        // fold the 32 pixel lanes of each half wave, then lanes 0 and 32 hold the wave's sums of their couts: partial row = the group,
        // channels of this workgroup's cout tile - the rows bn_stats_from_partials sums over
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int m = 1; m < 32; m <<= 1) {
                        st0[c][k][j] += __shfl_xor(st0[c][k][j], m);
                        st1[c][k][j] += __shfl_xor(st1[c][k][j], m);
                    }
        if (n == 0) {
            float* dst = a.stats + (long long)grp * 2 * a.cout_p;
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c0 = n0 + c * 32 + 16 * k + 8 * h;
                    if (c0 < a.cout_p) {
                        *reinterpret_cast<f32x4*>(dst + c0) = f32x4{st0[c][k][0], st0[c][k][1], st0[c][k][2], st0[c][k][3]};
                        *reinterpret_cast<f32x4*>(dst + c0 + 4) = f32x4{st0[c][k][4], st0[c][k][5], st0[c][k][6], st0[c][k][7]};
                        *reinterpret_cast<f32x4*>(dst + a.cout_p + c0) = f32x4{st1[c][k][0], st1[c][k][1], st1[c][k][2], st1[c][k][3]};
                        *reinterpret_cast<f32x4*>(dst + a.cout_p + c0 + 4) = f32x4{st1[c][k][4], st1[c][k][5], st1[c][k][6], st1[c][k][7]};
                    }
                }
        }
    }
}

// ---- host side (called from conv_bf16.hip's launcher)
// The geometry the kernel implements: a transposed 3x3 / stride 2 / padding 1 layer whose output is exactly 2H x 2W (output padding
// 1: the decoder's layers, and the data gradient of a 3x3 / stride 2 / padding 1 conv over an even extent), four phases whose taps
// reach input offsets 0 / 1 only, cin a multiple of the 32-channel chunk.  A shape-only rule (bit-reproducible): the launch must
// fill the chip (>= 512 workgroups) - smaller ones keep the implicit GEMM and its split-K.
static int tp2b_level() {
    static const int level = [] { const char* e = getenv("W2L_CONVB_TP2B"); return e ? atoi(e) : 1; }();
    return level;
}
// 32-cout tiles: 128 registers, two workgroups per CU; the 64-cout tile (206 registers, one workgroup per CU) only at level 2 (A/B);
// level 3: every layer on 32-cout tiles
int tp2b_tile(int cout_p) { return (cout_p <= 64 || tp2b_level() >= 3) ? 32 : 64; }

static long long tp2b_groups(int N, int H, int W) { return (long long)N * ((H + 3) / 4) * ((W + 7) / 8); }

bool tp2b_ok(int transposed, int kh, int kw, int sh, int sw, int ph, int pw, int nphase, const ConvPhase* phs, const int* taps_host, int cin_p,
             int cout_p, int N, int H, int W, int Ho, int Wo) {
    if (!(transposed && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && nphase == 4 && Ho == 2 * H && Wo == 2 * W)) return false;
    if (cin_p % kTpKC != 0) return false;
    int i = 0;
    for (int p = 0; p < 4; ++p) {
        if (phs[p].po_y * 2 + phs[p].po_x != p) return false;
        for (int t = 0; t < phs[p].ntaps; ++t, ++i) {
            if (i >= 9) return false;
            const int tv = taps_host[phs[p].tap_off + t];
            const int dy = (int)(short)(tv & 0xffff), dx = tv >> 16;
            if (tp_phase(i) != p || tp_off(i) != dy * 2 + dx) return false;      // the kernel's compile-time pair table
        }
    }
    if (i != 9) return false;
    const int bn = tp2b_tile(cout_p);
    const long long wgs = (tp2b_groups(N, H, W) + 7) / 8 * ((cout_p + bn - 1) / bn);
    if (bn == 64 && tp2b_level() < 2) return false;
    // measured (tools/tp2b_bench.py, profiles/r06/h_*): with 33..64 couts (two 32-cout tiles over the same boxes) the kernel is ahead
    // of the four-phase implicit GEMM only on large extents (160 -> 64 at 48x48 x 320 frames: 0.35 against 0.42 ms; 128 -> 64 at
    // 12x12: 0.030 against 0.023)
    if (cout_p > 32 && tp2b_level() < 3 && (long long)N * H * W < 400000) return false;
    return wgs >= 256;
}

int tp2b_npart(int cout_p, int N, int H, int W) { (void)cout_p; return (int)tp2b_groups(N, H, W); }

int tp2b_launch(hipStream_t stream, const void* x, int x_cs, void* y, int y_cs, const void* res, int res_cs, const void* w, long long w_elems,
                const float* scale, const float* shift, float* stats, const ConvPhase* phs, const int* taps_host, int N, int H, int W,
                int cin_p, int cout, int cout_p, int act) {
    Tp2bArgs a;
    a.x = x; a.y = y; a.res = res; a.w = w; a.scale = scale; a.shift = shift; a.stats = stats; a.w_elems = w_elems;
    a.N = N; a.H = H; a.W = W; a.x_cs = x_cs; a.y_cs = y_cs; a.res_cs = res_cs; a.cin_p = cin_p; a.cout = cout; a.cout_p = cout_p; a.act = act;
    const int bn = tp2b_tile(cout_p);
    a.tiles_x = (W + 7) / 8;              // 4 x 8 pixel groups per image row / column
    a.tiles_y = (H + 3) / 4;
    const long long groups = tp2b_groups(N, H, W);
    a.cout_tiles = (cout_p + bn - 1) / bn;
    const long long gblocks = (groups + 7) / 8;
    W2L_REQUIRE(gblocks * a.cout_tiles < (1ll << 31) && groups < (1ll << 28), "grid too large");
    a.ntiles = (int)groups;
    int i = 0;
    for (int p = 0; p < 4; ++p)
        for (int t = 0; t < phs[p].ntaps; ++t, ++i) {
            a.pt_base[i] = phs[p].w_off + (long long)t * cin_p;
            a.pt_kp[i] = phs[p].kp;
        }
    (void)taps_host;
    const dim3 grid((unsigned)(gblocks * a.cout_tiles)), block(512);
    if (bn == 64) hipLaunchKernelGGL(conv_tp2b_bf16_kernel<64>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(conv_tp2b_bf16_kernel<32>, grid, block, 0, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

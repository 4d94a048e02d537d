// Winograd F(2x2, 3x3) convolution for gfx950, second generation: the same arithmetic as conv_wino.hip
//   y = act( A^T [ sum_c (G g G^T)[xi] * (B^T d B)[xi] ] A * scale + shift (+ res) )
// re-tiled around what the first kernel's measurements said costs time on this part (profiles/r01/i_*.txt):
//
//  * fp32 MFMAs share their issue/execution resources with every other vector instruction of the SAME wave (a VALU costs
//    ~5 cycles of MFMA time with one wave per SIMD, ~2.7 with two; a buffer_load_dwordx4 ~40) and a workgroup that is alone
//    on its CU exposes all of its prologue / epilogue latency (9-11 us per work item, 30 % of the 64-channel 96x96 layers).
//    => TWO workgroups per CU (two waves per SIMD, desynchronised): a wave keeps only 8 of the 16 transform positions
//       (8 accumulators = 128 registers, 256 in total), so one workgroup's epilogue / prologue runs under the other's MFMAs.
//  * 28 vector-memory instructions per 64 MFMAs (16 weight-fragment loads + a 3x4 pixel patch per thread, every input pixel
//    fetched 6 times per workgroup) were the largest non-MFMA cost, and 2.5x fetch amplification on top.
//    => the input block of a workgroup is loaded ONCE into LDS (a (2bh+2) x (2bw+2) pixel region per image for a bh x bw
//       block of tiles: 1.4-3 float4 loads per thread per K-step instead of 12) and the B^T d B transform reads it from there;
//       a wave loads 8 instead of 16 weight fragments per K-step (only its half of the positions).
//
// Work decomposition.  Workgroup = 4 waves = 32 tiles x 64 couts x 16 positions; wave (wn, ph) owns 32 tiles x 32 couts
// (cout half wn) x the 8 positions (i, j) with j in {2*ph, 2*ph + 1}.  The inverse transform A^T M A splits along j:
//   t0[j] = M[0][j] + M[1][j] + M[2][j],  t1[j] = M[1][j] - M[2][j] - M[3][j]          (per lane, over the wave's own i)
//   ph 0:  P = ( t0[0] + t0[1],  t0[1],  t1[0] + t1[1],  t1[1] )
//   ph 1:  Q = ( t0[2], -t0[2] - t0[3],  t1[2], -t1[2] - t1[3] )         out[k] = P[k] + Q[k],  k = (dy, dx) of the 2x2 tile
// Both halves go to an LDS staging tile; the float4 pass that adds the residual and stores also adds P + Q.
// The 32 tiles of a workgroup are a bh x bw block of tiles in each of ni consecutive images (bh*bw*ni <= 32, chosen on the
// host per layer): 4x8x1 at 96x96 / 48x48, 4x4x2 at 24x24, 2x2x8 at 12x12, ...
//
// Per K-step (8 input channels), per thread:  <= 3 global float4 loads (raw block, two steps ahead) -> registers -> LDS raw
// buffer; 8 LDS reads of the raw block + 32 VALU + 4 LDS writes (one row of B^T d B of one tile and channel quad: thread =
// (row i = wave, tile, quad)); 8 weight-fragment loads from L2 (ring of 4); 8 LDS fragment reads; 32 MFMAs.
#include <stdlib.h>

#include "w2l_common.h"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kW2Oob = 0x80000000u;
constexpr int kW2KS = 8;           // channels per K-step
constexpr int kW2LDK = kW2KS + 4;  // V row stride (floats): conflict-free b128 fragment reads

// Two shapes of the same kernel.  MT = 32-tile blocks per workgroup, BC = couts per workgroup; a wave always owns 32 tiles x
// 32 couts x 8 positions:
//   <1, 64>  32 tiles x 64 couts, waves (wn, ph), 74 KB of LDS -> TWO workgroups per CU   (cout % 64 == 0 layers)
//   <2, 32>  64 tiles x 32 couts, waves (wm, ph), 123 KB of LDS -> one workgroup per CU   (cout % 32 == 0: the 32-channel
//            residual blocks and the generator's 80->32 output block, whose 1x1 + sigmoid head is fused into the last pass)
template <int MT, int BC>
struct W2 {
    static constexpr int BT = 32 * MT;                // tiles per workgroup
    static constexpr int VPOS = BT * kW2LDK;          // floats per position slab
    static constexpr int VBUF = 16 * VPOS;            // floats per V buffer
    static constexpr int NRAW = MT == 1 ? 3 : 4;      // raw-block float4 loads per thread per K-step
    static constexpr int RAW4 = 256 * NRAW;           // float4 slots per raw buffer
    static constexpr int LDY = BC + 4;                // staging row stride (floats)
    static constexpr int LdsFloats = 2 * VBUF + 2 * RAW4 * 4;
    static constexpr int LdsBytes = LdsFloats * 4 + 2 * BT * 4;
    static constexpr int WGS = MT == 1 ? 2 : 1;       // workgroups per CU
    static_assert(2 * BT * 4 * LDY <= LdsFloats, "P/Q staging tiles must fit in the V + raw buffers");
    static_assert(WGS * LdsBytes <= 160 * 1024, "LDS budget");
    static_assert(BC == 64 ? MT == 1 : (BC == 32 && MT == 2), "wave grid is 2 (tile or cout halves) x 2 (position halves)");
};

__device__ __forceinline__ f32x4 w2_buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(f32x4, v);
}

struct Wino2KArgs {
    const float* x;
    float* y;
    const float* res;
    const float* u;      // transformed weights in MFMA B-fragment order (wino_pack, conv_wino.hip)
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;
    int cout, y_cs, res_cs;
    int TH, TW;          // 2x2 output tiles per image
    int bh, bw, ni;      // tile block of a workgroup: bh x bw tiles in each of ni images
    int nby, nbx, ngi;   // blocks per image (y, x) and image groups: ceil(TH/bh), ceil(TW/bw), ceil(N/ni)
    int RH, RW, R4;      // raw region per image (pixels) and float4 slots per K-step: ni*RH*RW*2
    int nks;             // cin / 8
    int tiles_n;         // cout / BC
    long long total;     // work items: ngi*nby*nbx*tiles_n
    int act;
    // fused 1x1 head (<2, 32> only, cout == 32): y[pix][o] = head_act( sum_c head_w[o][c] * act(...)[c] + head_b[o] ), o < head_c
    const float* head_w;
    const float* head_b;
    int head_c, head_act;
    int stagger_mode;    // which workgroups start late: 0 none, 1: blockIdx >= gridDim/2, 2: odd (blockIdx >> 3)
    int stagger_sleeps;  // how late: this many s_sleep(127) (8128 cycles each) ~ half a work item
};

__device__ __forceinline__ float w2_act(float v, int act) {
    if (act == W2L_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == W2L_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == W2L_ACT_LEAKY) return v > 0.0f ? v : 0.01f * v;
    return v;
}

template <int MT, int BC, bool HEAD>
__global__ __launch_bounds__(256, (MT == 1 ? 2 : 1)) void conv_wino2_f32_kernel(const Wino2KArgs a) {
    using T = W2<MT, BC>;
    constexpr int kW2BT = T::BT, kW2BC = BC, kW2VPOS = T::VPOS, kW2VBUF = T::VBUF, kW2RAW4 = T::RAW4, kW2LDY = T::LDY;
    constexpr int NRAW = T::NRAW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);                  // [2][16][BT][LDK]
    float* Rs = Vs + 2 * kW2VBUF;                                // [2][RAW4] float4 slots, linear in the load index
    int* s_opix = reinterpret_cast<int*>(Rs + 2 * kW2RAW4 * 4);  // [BT] output pixel of (2ty, 2tx) or -1
    int* s_oflag = s_opix + kW2BT;                               // [BT] bit0: column 2tx+1 exists, bit1: row 2ty+1 exists

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);

    // persistent workgroups, two per CU; XCD x (hardware ids x, x+8, ...) walks one contiguous range of work items, so
    // neighbouring blocks (shared halo rows / columns, same weights) meet in one L2
    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    // Experiment switch (W2L_WINO2_STAGGER, off by default): half of the workgroups start half a work item late, so that the two
    // workgroups of a CU are never in their prologue / epilogue at the same time.  Measured: no effect on any layer
    // (profiles/r02/d_wino2_stagger.txt) - one wave per SIMD already keeps the fp32 pipe ~88 % as busy as two do
    // (profiles/r02/e_wino2_grid.txt: 256 vs 512 workgroups), the fixed phases are issue work, not exposed latency.
    if (a.stagger_sleeps > 0 &&
        ((a.stagger_mode == 1 && blockIdx.x >= (gridDim.x >> 1)) || (a.stagger_mode == 2 && ((blockIdx.x >> 3) & 1u)))) {
        for (int i = 0; i < a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // The raw block of channel chunk 0 of the NEXT work item is requested before this item's epilogue (its registers are free
    // there), so an item starts with its first input block already on the way: rawreg / goff / the item coordinates carry over.
    unsigned jw = blockIdx.x >> 3;
    if (jw >= per || xcd * per + jw >= total) return;
    int tile_n, bx_i, by_i, gi;
    unsigned goff[NRAW];
    f32x4 rawreg[NRAW];
    auto item_begin = [&](unsigned bid, int tt) {   // decode a work item, the byte offsets of its raw block, request chunk 0
        tile_n = (int)(bid % (unsigned)a.tiles_n);
        unsigned mb = bid / (unsigned)a.tiles_n;
        bx_i = (int)(mb % (unsigned)a.nbx);
        mb /= (unsigned)a.nbx;
        by_i = (int)(mb % (unsigned)a.nby);
        gi = (int)(mb / (unsigned)a.nby);
        // slot e = t + 256*k  ->  (image il, row ry, column rx, channel quad q) of the block's input region
#pragma unroll
        for (int k = 0; k < NRAW; ++k) {
            const int e = tt + 256 * k;
            unsigned off = kW2Oob;
            if (e < a.R4) {
                const int q = e & 1, p = e >> 1;
                const int rxx = p % a.RW, p2 = p / a.RW;
                const int ry = p2 % a.RH, il = p2 / a.RH;
                const int n = gi * a.ni + il;
                const int iy = 2 * by_i * a.bh - 1 + ry, ix = 2 * bx_i * a.bw - 1 + rxx;
                if (n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(q * 4)) * 4u;
            }
            goff[k] = off;
        }
#pragma unroll
        for (int k = 0; k < NRAW; ++k) rawreg[k] = w2_buf_load4(rx, goff[k], 0u);
    };
    item_begin(xcd * per + jw, (int)threadIdx.x);
    for (;;) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));     // per-item coordinates are re-derived from an opaque copy: nothing stays live across items
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = BC == 64 ? (wave & 1) : 0;   // cout half of this wave   (<1, 64>)
    const int wm = BC == 64 ? 0 : (wave & 1);   // tile half of this wave   (<2, 32>)
    const int ph = wave >> 1;                   // position half: j in {2ph, 2ph+1}
    const int n0 = tile_n * kW2BC;
    const int bhw = a.bh * a.bw;

    if (t < kW2BT) {                 // tile table of the epilogue
        const int il = t / bhw, r = t - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int n = gi * a.ni + il, ty = by_i * a.bh + tyl, tx = bx_i * a.bw + txl;
        int o = -1, f = 0;
        if (il < a.ni && n < a.N && ty < a.TH && tx < a.TW) {
            o = (n * a.H + 2 * ty) * a.W + 2 * tx;
            f = ((2 * tx + 1 < a.W) ? 1 : 0) | ((2 * ty + 1 < a.H) ? 2 : 0);
        }
        s_opix[t] = o;
        s_oflag[t] = f;
    }

    // ---- raw block loads (goff: item_begin above)
    auto raw_gload = [&](int step) {            // channels [8*step, 8*step+8); past-the-end steps read zero (descriptor bound)
        const unsigned soff = (unsigned)(step * kW2KS * 4);
#pragma unroll
        for (int k = 0; k < NRAW; ++k) rawreg[k] = w2_buf_load4(rx, goff[k], soff);
    };
    auto raw_store = [&](int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(Rs) + buf * kW2RAW4 + t;
#pragma unroll
        for (int k = 0; k < NRAW; ++k) dst[256 * k] = rawreg[k];
    };

    // ---- transform item of this thread: row i = wave of B^T d B for (tile tl, channel quad q)
    //   row i of B^T d:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3   ==  d[ra] + sg * d[rb]
    const int q = lane & 1;
    const int ra = (wave == 0) ? 0 : (wave == 2 ? 2 : 1);
    const int rb = (wave == 0) ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = (wave == 1) ? 1.0f : -1.0f;
    int row_a[MT], row_b[MT];                    // float4 slots of this thread's two raw rows, per tile (lane>>1) + 32*m
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int tl = (lane >> 1) + 32 * m;
        const int il = tl / bhw, r = tl - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int ilc = il < a.ni ? il : 0;      // unused tile slots (bh*bw*ni < BT) read image 0's region: finite, never stored
        const int base = ((ilc * a.RH + 2 * tyl) * a.RW + 2 * txl) * 2 + q;
        row_a[m] = base + ra * a.RW * 2;
        row_b[m] = base + rb * a.RW * 2;
    }
    float* const vwr = Vs + (wave * 4) * kW2VPOS + (lane >> 1) * kW2LDK + q * 4;
    f32x4 da[MT][4], db[MT][4];
    auto tf_load = [&](int buf) {
        const f32x4* src = reinterpret_cast<const f32x4*>(Rs) + buf * kW2RAW4;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                da[m][c] = src[row_a[m] + 2 * c];
                db[m][c] = src[row_b[m] + 2 * c];
            }
    };
    auto tf_rows = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) da[m][c][e] = fmaf(sg, db[m][c][e], da[m][c][e]);
    };
    auto tf_store = [&](int buf, int j) {       // position (i, j) of B^T d B
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 v;
            switch (j) {
                case 0: v = da[m][0] - da[m][2]; break;
                case 1: v = da[m][1] + da[m][2]; break;
                case 2: v = da[m][2] - da[m][1]; break;
                default: v = da[m][1] - da[m][3]; break;
            }
            *reinterpret_cast<f32x4*>(vwr + buf * kW2VBUF + j * kW2VPOS + m * 32 * kW2LDK) = v;
        }
    };

    // ---- B operand: u[((nb * nks + kc) * 16 + pos) * 256 + (h*32 + n)*4 + e] = U_pos[nb*32 + n][kc*8 + 4h + e];
    // this wave reads positions pos(s) = 4*(s>>1) + 2*ph + (s&1), s = 0..7, of every chunk kc
    const int nb = (n0 >> 5) + wn;
    const int F = a.nks * 16;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u + (long long)nb * F * 256), 0, F * 1024, 0x00020000);
    const unsigned bl_lane = (unsigned)(lane * 16);
    const unsigned bl_ph = (unsigned)(2 * ph) * 1024u;
    auto bload = [&](int kc, int s) {            // s is a compile-time constant at every call site
        const unsigned soff = (unsigned)kc * 16384u + bl_ph + (unsigned)(4 * (s >> 1) + (s & 1)) * 1024u;
        return w2_buf_load4(ru, bl_lane, soff);  // past-the-end chunks read zero (never used)
    };
    constexpr int RING = 4;
    f32x4 bq[RING];

    f32x16 acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // ---- prologue: raw(0) (requested by item_begin) -> LDS, raw(1) in flight, V(0) from raw(0), raw(1) -> LDS
    const int nsteps = a.cin / kW2KS;
#pragma unroll
    for (int i = 0; i < RING; ++i) bq[i] = bload(0, i);
    raw_store(0);
    raw_gload(1);
    __syncthreads();                 // raw[0], tile table
    tf_load(0);
    tf_rows();
#pragma unroll
    for (int j = 0; j < 4; ++j) tf_store(0, j);
    raw_store(1);
    __syncthreads();                 // V[0], raw[1]

    const float* Abase = Vs + (2 * ph) * kW2VPOS + (wm * 32 + (lane & 31)) * kW2LDK + (lane >> 5) * 4;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const float* Ab = Abase + buf * kW2VBUF;
        f32x4 af = *reinterpret_cast<const f32x4*>(Ab);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 ac = af;
            if (s < 7) af = *reinterpret_cast<const f32x4*>(Ab + (4 * ((s + 1) >> 1) + ((s + 1) & 1)) * kW2VPOS);
            const f32x4 bc = bq[s % RING];
            // the ring runs 4 slots ahead: slots 4..7 of this chunk, then 0..3 of the next
            bq[s % RING] = (s < 4) ? bload(step, s + 4) : bload(step + 1, s - 4);
            // everything else of the K-step rides between the MFMA groups, one piece per slot:
            //   slot 0: request the raw block of step+2; read this thread's two raw rows of step+1 from LDS
            //   slot 1: row transform; slots 2-5: one transformed position each -> V[buf^1]; slot 6: raw(step+2) -> LDS
            if (s == 0) {
                raw_gload(step + 2);
                tf_load(buf ^ 1);
            } else if (s == 1) {
                tf_rows();
            } else if (s < 6) {
                tf_store(buf ^ 1, s - 2);
            } else if (s == 6) {
                raw_store(buf);      // raw[buf] was last read at slot 0 of the PREVIOUS step (barrier in between)
            }
            // the 4 MFMAs of a slot chain through one accumulator: issue them back to back (an instruction between two
            // MFMAs on the same accumulator costs ~43 cycles, between different ones ~6)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[e], bc[e], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue.  acc[s][r]: position (i = s>>1, j = 2ph + (s&1)), cout lane&31, tile (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Ys = Vs;                   // [2 (ph)][BT tiles][4 pixels][LDY]; V / raw buffers are dead after the last barrier
    constexpr int CG = kW2BC / 4;                    // float4 column groups per pixel
    constexpr int NIT = kW2BT * 4 * CG / 256;        // 8 items per thread
    const long long npix = (long long)a.N * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0, 0x00020000);
    const int c4 = t % CG;
    const int ch = n0 + c4 * 4;
    // the residual of the whole output tile is requested before the partial inverse transforms are formed and staged, the next
    // item's first raw block right behind it: both travel under ~1.5k cycles of VALU / LDS work and the barrier
    int pixv[NIT];
    f32x4 rv[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int id = i * 256 + t;
        const int px = (id / CG) & 3;
        const int tile = id / (CG * 4);
        const int opix = s_opix[tile];
        const int fl = s_oflag[tile];
        const bool ok = (opix >= 0) & (((px & 1) == 0) | ((fl & 1) != 0)) & (((px & 2) == 0) | ((fl & 2) != 0));
        const int pix = ok ? opix + (px & 1) + (px >> 1) * a.W : -1;
        pixv[i] = pix;
        u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
            rr, (int)(pix >= 0 ? ((unsigned)pix * (unsigned)a.res_cs + (unsigned)ch) * 4u : kW2Oob), 0, 0);
        rv[i] = __builtin_bit_cast(f32x4, raw);
    }
    jw += gw;
    const bool have_next = jw < per && xcd * per + jw < total;
    if (have_next) item_begin(xcd * per + jw, t);
    {
        float* yrow = Ys + ph * (kW2BT * 4 * kW2LDY) + wn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t0[2], t1[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                t0[jj] = acc[0 + jj][r] + acc[2 + jj][r] + acc[4 + jj][r];
                t1[jj] = acc[2 + jj][r] - acc[4 + jj][r] - acc[6 + jj][r];
            }
            float o[4];
            if (ph == 0) {
                o[0] = t0[0] + t0[1]; o[1] = t0[1]; o[2] = t1[0] + t1[1]; o[3] = t1[1];
            } else {
                o[0] = t0[0]; o[1] = -t0[0] - t0[1]; o[2] = t1[0]; o[3] = -t1[0] - t1[1];
            }
            const int tlr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int k = 0; k < 4; ++k) yrow[(tlr * 4 + k) * kW2LDY] = o[k];
        }
    }
    __syncthreads();
    {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            a.y, 0, (int)(((npix - 1) * a.y_cs + (HEAD ? a.head_c : a.cout)) * 4), 0x00020000);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
        const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 256 + t;
            const float* src = Ys + (id / CG) * kW2LDY + c4 * 4;
            const f32x4 p = *reinterpret_cast<const f32x4*>(src);
            const f32x4 qv = *reinterpret_cast<const f32x4*>(src + kW2BT * 4 * kW2LDY);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // none / ReLU / LeakyReLU(0.01) without a branch: max(x,0) + slope * min(x,0) is exact for all three
                const float x = fmaf(p[e] + qv[e], sc[e], sh[e]) + rv[i][e];
                v[e] = fmaf(neg_slope, fminf(x, 0.f), fmaxf(x, 0.f));
            }
            if (!HEAD) {
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, v), ry,
                    (int)(pixv[i] >= 0 ? ((unsigned)pixv[i] * (unsigned)a.y_cs + (unsigned)ch) * 4u : kW2Oob), 0, 0);
            } else {
                // fused head, step 1: the activated channels go back into the P staging tile (each thread overwrites exactly
                // the float4 it just read); step 2 below contracts whole pixels
                *reinterpret_cast<f32x4*>(const_cast<float*>(src)) = v;
            }
        }
        if (HEAD) {
            // step 2: one thread per output pixel (BT * 4 = 256 of them): its BC = 32 activated channels (8 LDS reads of the
            // staging row) times the [head_c][32] matrix (wave-uniform scalar loads), bias, activation, head_c scalar stores
            __syncthreads();
            const int tile = t >> 2, px = t & 3;
            const int opix = s_opix[tile];
            const int fl = s_oflag[tile];
            const bool ok = (opix >= 0) & (((px & 1) == 0) | ((fl & 1) != 0)) & (((px & 2) == 0) | ((fl & 2) != 0));
            const float* row = Ys + t * kW2LDY;
            float hacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < kW2BC / 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * g);
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        hacc[o] = fmaf(v[e], (o < a.head_c) ? a.head_w[o * a.cout + 4 * g + e] : 0.f, hacc[o]);
            }
            if (ok) {
                float* dst = a.y + (long long)(opix + (px & 1) + (px >> 1) * a.W) * a.y_cs;
                for (int o = 0; o < a.head_c; ++o) dst[o] = w2_act(hacc[o] + (a.head_b ? a.head_b[o] : 0.f), a.head_act);
            }
        }
    }
    __syncthreads();   // staging tiles / tile table are rewritten by the next work item
    if (!have_next) break;
    }   // persistent loop
}

// ---- third shape: 32 tiles x 32 couts, the 16 positions cut 2 x 2 over the four waves (wave (I, J) owns i in {2I, 2I+1},
// j in {2J, 2J+1}: 4 accumulators = 64 registers), 74 KB of LDS -> TWO workgroups per CU.  For the 32-cout layers (the 80->32
// output block with its fused 1x1 + sigmoid head, the 32-channel encoder blocks) the <2, 32> shape above runs one workgroup per
// CU, i.e. one wave per SIMD, where every non-MFMA instruction costs twice the MFMA time and nothing hides a work item's
// prologue / epilogue.  A^T M A is bilinear in the position blocks, as in conv_wino4.hip: every wave forms
//   P[a][b] = sum_il sum_jl AT[a][2I+il] * M[il][jl] * AT[b][2J+jl],    AT = | 1  1  1  0 |
//                                                                            | 0  1 -1 -1 |
// for its 32 tiles x 32 couts; the four partials meet in the LDS staging tile and the float4 output pass adds them.
// Same transformed weights (wino_pack), raw block, B^T d B item (row i = wave) and K-step as <1, 64>; 4 MFMA slots per K-step.
constexpr int kW2qBT = 32, kW2qBC = 32;
constexpr int kW2qVPOS = kW2qBT * kW2LDK, kW2qVBUF = 16 * kW2qVPOS, kW2qNRAW = 3, kW2qRAW4 = 256 * kW2qNRAW, kW2qLDY = kW2qBC + 4;
constexpr int kW2qLdsFloats = 2 * kW2qVBUF + 2 * kW2qRAW4 * 4;
constexpr int kW2qLdsBytes = kW2qLdsFloats * 4 + 2 * kW2qBT * 4;
static_assert(4 * kW2qBT * 4 * kW2qLDY <= kW2qLdsFloats, "the four partial staging tiles must fit in the V + raw buffers");
static_assert(2 * kW2qLdsBytes <= 160 * 1024, "two workgroups per CU");

template <bool HEAD>
__global__ __launch_bounds__(256, 2) void conv_wino2q_f32_kernel(const Wino2KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);                    // [2][16][32][LDK]
    float* Rs = Vs + 2 * kW2qVBUF;                                 // [2][RAW4] float4 slots, linear in the load index
    int* s_opix = reinterpret_cast<int*>(Rs + 2 * kW2qRAW4 * 4);   // [32] output pixel of (2ty, 2tx) or -1
    int* s_oflag = s_opix + kW2qBT;                                // [32] bit0: column 2tx+1 exists, bit1: row 2ty+1 exists

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)((((long long)a.N * a.H * a.W - 1) * a.x_cs + a.cin) * 4), 0x00020000);
    const unsigned total = (unsigned)a.total;
    const unsigned per = (total + 7u) / 8u;
    const unsigned xcd = blockIdx.x & 7u, gw = gridDim.x >> 3;
    unsigned jw = blockIdx.x >> 3;
    if (jw >= per || xcd * per + jw >= total) return;
    int tile_n, bx_i, by_i, gi;
    unsigned goff[kW2qNRAW];
    f32x4 rawreg[kW2qNRAW];
    auto item_begin = [&](unsigned bid, int tt) {   // decode a work item, the byte offsets of its raw block, request chunk 0
        tile_n = (int)(bid % (unsigned)a.tiles_n);
        unsigned mb = bid / (unsigned)a.tiles_n;
        bx_i = (int)(mb % (unsigned)a.nbx);
        mb /= (unsigned)a.nbx;
        by_i = (int)(mb % (unsigned)a.nby);
        gi = (int)(mb / (unsigned)a.nby);
#pragma unroll
        for (int k = 0; k < kW2qNRAW; ++k) {
            const int e = tt + 256 * k;
            unsigned off = kW2Oob;
            if (e < a.R4) {
                const int q = e & 1, p = e >> 1;
                const int rxx = p % a.RW, p2 = p / a.RW;
                const int ry = p2 % a.RH, il = p2 / a.RH;
                const int n = gi * a.ni + il;
                const int iy = 2 * by_i * a.bh - 1 + ry, ix = 2 * bx_i * a.bw - 1 + rxx;
                if (n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cs + (unsigned)(q * 4)) * 4u;
            }
            goff[k] = off;
        }
#pragma unroll
        for (int k = 0; k < kW2qNRAW; ++k) rawreg[k] = w2_buf_load4(rx, goff[k], 0u);
    };
    item_begin(xcd * per + jw, (int)threadIdx.x);
    for (;;) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));     // per-item coordinates are re-derived from an opaque copy: nothing stays live across items
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int PI = wave >> 1, PJ = wave & 1;   // position block of this wave
    const int n0 = tile_n * kW2qBC;
    const int bhw = a.bh * a.bw;

    if (t < kW2qBT) {                // tile table of the epilogue
        const int il = t / bhw, r = t - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int n = gi * a.ni + il, ty = by_i * a.bh + tyl, tx = bx_i * a.bw + txl;
        int o = -1, f = 0;
        if (il < a.ni && n < a.N && ty < a.TH && tx < a.TW) {
            o = (n * a.H + 2 * ty) * a.W + 2 * tx;
            f = ((2 * tx + 1 < a.W) ? 1 : 0) | ((2 * ty + 1 < a.H) ? 2 : 0);
        }
        s_opix[t] = o;
        s_oflag[t] = f;
    }

    auto raw_gload = [&](int step) {            // channels [8*step, 8*step+8); past-the-end steps read zero (descriptor bound)
        const unsigned soff = (unsigned)(step * kW2KS * 4);
#pragma unroll
        for (int k = 0; k < kW2qNRAW; ++k) rawreg[k] = w2_buf_load4(rx, goff[k], soff);
    };
    auto raw_store = [&](int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(Rs) + buf * kW2qRAW4 + t;
#pragma unroll
        for (int k = 0; k < kW2qNRAW; ++k) dst[256 * k] = rawreg[k];
    };

    // ---- transform item of this thread: row i = wave of B^T d B for (tile lane>>1, channel quad q)
    //   row i of B^T d:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3   ==  d[ra] + sg * d[rb]
    const int q = lane & 1;
    const int ra = (wave == 0) ? 0 : (wave == 2 ? 2 : 1);
    const int rb = (wave == 0) ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = (wave == 1) ? 1.0f : -1.0f;
    int row_a, row_b;
    {
        const int tl = lane >> 1;
        const int il = tl / bhw, r = tl - il * bhw;
        const int tyl = r / a.bw, txl = r - tyl * a.bw;
        const int ilc = il < a.ni ? il : 0;      // unused tile slots read image 0's region: finite, never stored
        const int base = ((ilc * a.RH + 2 * tyl) * a.RW + 2 * txl) * 2 + q;
        row_a = base + ra * a.RW * 2;
        row_b = base + rb * a.RW * 2;
    }
    float* const vwr = Vs + (wave * 4) * kW2qVPOS + (lane >> 1) * kW2LDK + q * 4;
    f32x4 da[4], db[4];
    auto tf_load = [&](int buf) {
        const f32x4* src = reinterpret_cast<const f32x4*>(Rs) + buf * kW2qRAW4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            da[c] = src[row_a + 2 * c];
            db[c] = src[row_b + 2 * c];
        }
    };
    auto tf_rows = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) da[c][e] = fmaf(sg, db[c][e], da[c][e]);
    };
    auto tf_store = [&](int buf, int j) {       // position (i, j) of B^T d B
        f32x4 v;
        switch (j) {
            case 0: v = da[0] - da[2]; break;
            case 1: v = da[1] + da[2]; break;
            case 2: v = da[2] - da[1]; break;
            default: v = da[1] - da[3]; break;
        }
        *reinterpret_cast<f32x4*>(vwr + buf * kW2qVBUF + j * kW2qVPOS) = v;
    };

    // ---- B operand: u[((nb * nks + kc) * 16 + pos) * 256 + (h*32 + n)*4 + e]; this wave reads pos(s) = 4*(2I + (s>>1)) + 2J + (s&1)
    const int F = a.nks * 16;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u + (long long)(n0 >> 5) * F * 256), 0, F * 1024, 0x00020000);
    const unsigned bl_lane = (unsigned)(lane * 16);
    const int pos0 = 8 * PI + 2 * PJ;
    const unsigned bl_pb = (unsigned)pos0 * 1024u;
    auto bload = [&](int kc, int s) {            // s is a compile-time constant at every call site
        const unsigned soff = (unsigned)kc * 16384u + bl_pb + (unsigned)(4 * (s >> 1) + (s & 1)) * 1024u;
        return w2_buf_load4(ru, bl_lane, soff);  // past-the-end chunks read zero (never used)
    };
    f32x4 bq[4];                                 // fragment of slot s, refilled for the next K-step right after its use

    f32x16 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // ---- prologue: raw(0) (requested by item_begin) -> LDS, raw(1) in flight, V(0) from raw(0), raw(1) -> LDS
    const int nsteps = a.cin / kW2KS;
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = bload(0, i);
    raw_store(0);
    raw_gload(1);
    __syncthreads();                 // raw[0], tile table
    tf_load(0);
    tf_rows();
#pragma unroll
    for (int j = 0; j < 4; ++j) tf_store(0, j);
    raw_store(1);
    __syncthreads();                 // V[0], raw[1]

    const float* Abase = Vs + pos0 * kW2qVPOS + (lane & 31) * kW2LDK + (lane >> 5) * 4;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const float* Ab = Abase + buf * kW2qVBUF;
        f32x4 af = *reinterpret_cast<const f32x4*>(Ab);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 ac = af;
            if (s < 3) af = *reinterpret_cast<const f32x4*>(Ab + (4 * ((s + 1) >> 1) + ((s + 1) & 1)) * kW2qVPOS);
            const f32x4 bc = bq[s];
            bq[s] = bload(step + 1, s);
            // the rest of the K-step rides between the four MFMA groups
            if (s == 0) {
                raw_gload(step + 2);
                tf_load(buf ^ 1);
            } else if (s == 1) {
                tf_rows();
                tf_store(buf ^ 1, 0);
                tf_store(buf ^ 1, 1);
            } else if (s == 2) {
                tf_store(buf ^ 1, 2);
                tf_store(buf ^ 1, 3);
            } else {
                raw_store(buf);      // raw[buf] was last read at slot 0 of the PREVIOUS step (barrier in between)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[e], bc[e], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue.  acc[2*il + jl][r]: position (2I + il, 2J + jl), cout lane&31, tile (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Ys = Vs;                   // [4 (wave)][32 tiles][4 pixels][LDY]; V / raw buffers are dead after the last barrier
    constexpr int CG = kW2qBC / 4;                     // float4 column groups per pixel
    constexpr int NIT = kW2qBT * 4 * CG / 256;         // 4 items per thread
    constexpr int kPart = kW2qBT * 4 * kW2qLDY;        // floats per partial staging tile
    const long long npix = (long long)a.N * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res ? a.res : a.y), 0, a.res ? (int)(((npix - 1) * a.res_cs + a.cout) * 4) : 0, 0x00020000);
    const int c4 = t % CG;
    const int ch = n0 + c4 * 4;
    int pixv[NIT];
    f32x4 rv[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {   // residual of the whole output tile, requested before the partial transforms are staged
        const int id = i * 256 + t;
        const int px = (id / CG) & 3;
        const int tile = id / (CG * 4);
        const int opix = s_opix[tile];
        const int fl = s_oflag[tile];
        const bool ok = (opix >= 0) & (((px & 1) == 0) | ((fl & 1) != 0)) & (((px & 2) == 0) | ((fl & 2) != 0));
        const int pix = ok ? opix + (px & 1) + (px >> 1) * a.W : -1;
        pixv[i] = pix;
        u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
            rr, (int)(pix >= 0 ? ((unsigned)pix * (unsigned)a.res_cs + (unsigned)ch) * 4u : kW2Oob), 0, 0);
        rv[i] = __builtin_bit_cast(f32x4, raw);
    }
    jw += gw;
    const bool have_next = jw < per && xcd * per + jw < total;
    if (have_next) item_begin(xcd * per + jw, t);
    {
        // wave-uniform rows of A^T for this wave's blocks: (1, 1) / (0, 1) for block 0, (1, 0) / (-1, -1) for block 1
        const float ta0 = 1.f, ta1 = PI == 0 ? 1.f : 0.f, tb0 = PI == 0 ? 0.f : -1.f, tb1 = PI == 0 ? 1.f : -1.f;
        const float da0 = 1.f, da1 = PJ == 0 ? 1.f : 0.f, db0 = PJ == 0 ? 0.f : -1.f, db1 = PJ == 0 ? 1.f : -1.f;
        float* yrow = Ys + wave * kPart + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // t[a][jl] = sum_il AT[a][2I+il] * M[il][jl]
            const float t00 = fmaf(ta0, acc[0][r], ta1 * acc[2][r]), t01 = fmaf(ta0, acc[1][r], ta1 * acc[3][r]);
            const float t10 = fmaf(tb0, acc[0][r], tb1 * acc[2][r]), t11 = fmaf(tb0, acc[1][r], tb1 * acc[3][r]);
            const int tlr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            yrow[(tlr * 4 + 0) * kW2qLDY] = fmaf(da0, t00, da1 * t01);     // pixel (a, b) = (0, 0)
            yrow[(tlr * 4 + 1) * kW2qLDY] = fmaf(db0, t00, db1 * t01);     // (0, 1)
            yrow[(tlr * 4 + 2) * kW2qLDY] = fmaf(da0, t10, da1 * t11);     // (1, 0)
            yrow[(tlr * 4 + 3) * kW2qLDY] = fmaf(db0, t10, db1 * t11);     // (1, 1)
        }
    }
    __syncthreads();
    {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            a.y, 0, (int)(((npix - 1) * a.y_cs + (HEAD ? a.head_c : a.cout)) * 4), 0x00020000);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + ch);
        const float neg_slope = a.act == W2L_ACT_RELU ? 0.f : (a.act == W2L_ACT_LEAKY ? 0.01f : 1.f);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = i * 256 + t;
            const float* src = Ys + (id / CG) * kW2qLDY + c4 * 4;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(src);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(src + kPart);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(src + 2 * kPart);
            const f32x4 p3 = *reinterpret_cast<const f32x4*>(src + 3 * kPart);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // none / ReLU / LeakyReLU(0.01) without a branch: max(x,0) + slope * min(x,0) is exact for all three
                const float x = fmaf((p0[e] + p1[e]) + (p2[e] + p3[e]), sc[e], sh[e]) + rv[i][e];
                v[e] = fmaf(neg_slope, fminf(x, 0.f), fmaxf(x, 0.f));
            }
            if (!HEAD) {
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, v), ry,
                    (int)(pixv[i] >= 0 ? ((unsigned)pixv[i] * (unsigned)a.y_cs + (unsigned)ch) * 4u : kW2Oob), 0, 0);
            } else {
                // fused head, step 1: the activated channels go back into the first staging tile (each thread overwrites exactly
                // the float4 it just read); step 2 below contracts whole pixels
                *reinterpret_cast<f32x4*>(const_cast<float*>(src)) = v;
            }
        }
        if (HEAD) {
            // step 2: one thread per output pixel (32 tiles x 4 = 128 of them): its 32 activated channels (8 LDS reads of the
            // staging row) times the [head_c][32] matrix (wave-uniform scalar loads), bias, activation, head_c scalar stores
            __syncthreads();
            if (t < kW2qBT * 4) {
                const int tile = t >> 2, px = t & 3;
                const int opix = s_opix[tile];
                const int fl = s_oflag[tile];
                const bool ok = (opix >= 0) & (((px & 1) == 0) | ((fl & 1) != 0)) & (((px & 2) == 0) | ((fl & 2) != 0));
                const float* row = Ys + t * kW2qLDY;
                float hacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < kW2qBC / 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * g);
#pragma unroll
                    for (int o = 0; o < 4; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            hacc[o] = fmaf(v[e], (o < a.head_c) ? a.head_w[o * a.cout + 4 * g + e] : 0.f, hacc[o]);
                }
                if (ok) {
                    float* dst = a.y + (long long)(opix + (px & 1) + (px >> 1) * a.W) * a.y_cs;
                    for (int o = 0; o < a.head_c; ++o) dst[o] = w2_act(hacc[o] + (a.head_b ? a.head_b[o] : 0.f), a.head_act);
                }
            }
        }
    }
    __syncthreads();   // staging tiles / tile table are rewritten by the next work item
    if (!have_next) break;
    }   // persistent loop
}

// ---- host side --------------------------------------------------------------------------------
struct W2Block { int bh, bw, ni; };
// candidate tile blocks: bh*bw*ni <= BT tiles, raw region ni*(2bh+2)*(2bw+2)*2 <= RAW4 float4 slots
static const W2Block kW2Blocks[] = {{8, 8, 1}, {4, 8, 2}, {8, 4, 2}, {4, 8, 1}, {8, 4, 1}, {4, 4, 4}, {4, 4, 2}, {2, 8, 2},
                                    {2, 4, 8}, {2, 4, 4}, {4, 2, 4}, {2, 2, 16}, {2, 2, 8}, {3, 3, 7}, {3, 3, 3}, {1, 4, 8},
                                    {1, 2, 16}, {2, 1, 16}, {1, 1, 32}, {1, 1, 24}};

static W2Block wino2_pick_block(int N, int TH, int TW, int BT, int RAW4) {
    W2Block best = {1, 1, 1};
    double best_cost = 1e300;
    for (const W2Block& b : kW2Blocks) {
        if (b.ni * (2 * b.bh + 2) * (2 * b.bw + 2) * 2 > RAW4 || b.bh * b.bw * b.ni > BT) continue;
        // work items x (MFMA work per item is fixed) + a small preference for blocks with more halo sharing
        const double items = (double)ceil_div(TH, b.bh) * ceil_div(TW, b.bw) * ceil_div(N, b.ni);
        const double halo = (double)b.ni * (2 * b.bh + 2) * (2 * b.bw + 2) / (4.0 * b.bh * b.bw * b.ni);
        const double cost = items * (1.0 + 0.05 * halo);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}

struct W2Cfg {
    int bt, bc, raw4, lds, wgs;
    void (*kernel)(const Wino2KArgs);
    void (*kernel_head)(const Wino2KArgs);
};
static const W2Cfg kW2Cfgs[] = {
    {W2<1, 64>::BT, 64, W2<1, 64>::RAW4, W2<1, 64>::LdsBytes, 2, conv_wino2_f32_kernel<1, 64, false>, nullptr},
    {W2<2, 32>::BT, 32, W2<2, 32>::RAW4, W2<2, 32>::LdsBytes, 1, conv_wino2_f32_kernel<2, 32, false>,
     conv_wino2_f32_kernel<2, 32, true>},
};
constexpr int kNumW2 = sizeof(kW2Cfgs) / sizeof(kW2Cfgs[0]);

int wino2_num_cfgs() { return kNumW2; }

// head_c > 0: the layer carries a fused 1x1 head (only the 32-cout shape implements it, and only with all couts in one tile)
bool wino2_ok(int cfg, int cin, int cout, int head_c) {
    if (cfg < 0 || cfg >= kNumW2 || cin % kW2KS != 0 || cout % kW2Cfgs[cfg].bc != 0) return false;
    if (head_c > 0) return kW2Cfgs[cfg].kernel_head != nullptr && cout == kW2Cfgs[cfg].bc && head_c <= 4;
    return true;
}

int wino2_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    for (int i = 0; i < kNumW2; ++i) {
        W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kW2Cfgs[i].kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kW2Cfgs[i].lds));
        if (kW2Cfgs[i].kernel_head)
            W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kW2Cfgs[i].kernel_head),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kW2Cfgs[i].lds));
    }
    done = true;
    return W2L_OK;
}

int wino2_launch(int cfg, const WinoKArgs& w, const float* head_w, const float* head_b, int head_c, int head_act,
                 hipStream_t stream, long long* flops_out) {
    const W2Cfg& wc = kW2Cfgs[cfg];
    Wino2KArgs a;
    a.x = w.x; a.y = w.y; a.res = w.res; a.u = w.u; a.scale = w.scale; a.shift = w.shift;
    a.N = w.N; a.H = w.H; a.W = w.W; a.cin = w.cin; a.x_cs = w.x_cs;
    a.cout = w.cout; a.y_cs = w.y_cs; a.res_cs = w.res_cs; a.act = w.act;
    a.head_w = head_w; a.head_b = head_b; a.head_c = head_c; a.head_act = head_act;
    a.TH = (a.H + 1) / 2;
    a.TW = (a.W + 1) / 2;
    const W2Block b = wino2_pick_block(a.N, a.TH, a.TW, wc.bt, wc.raw4);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(a.TH, b.bh);
    a.nbx = ceil_div(a.TW, b.bw);
    a.ngi = ceil_div(a.N, b.ni);
    a.RH = 2 * b.bh + 2;
    a.RW = 2 * b.bw + 2;
    a.R4 = b.ni * a.RH * a.RW * 2;
    a.nks = a.cin / 8;
    a.tiles_n = a.cout / wc.bc;
    a.total = (long long)a.ngi * a.nby * a.nbx * a.tiles_n;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE((long long)a.N * a.H * a.W < (1ll << 31), "tensor too large");
    W2L_REQUIRE(head_w == nullptr || (wc.kernel_head != nullptr && a.tiles_n == 1), "fused head needs the 32-cout shape");
    if (flops_out) {   // dry run: 16 position-GEMMs of [items*BT] x [BC] x cin
        *flops_out = 2ll * 16 * a.total * wc.bt * wc.bc * a.cin;
        return W2L_OK;
    }
    // persistent: wgs workgroups per CU, a multiple of 8 so that every XCD gets the same number
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 256 * wc.wgs) grid = 256 * wc.wgs;
    a.stagger_mode = 0;        // start-offset experiment (profiles/r02/d_wino2_stagger.txt): no effect, kept switchable
    a.stagger_sleeps = 0;
    static const int mode_env = []() { const char* e = getenv("W2L_WINO2_STAGGER"); return e ? atoi(e) : 0; }();
    if (mode_env && wc.wgs == 2 && grid == 512 && a.total >= 1024) {
        a.stagger_mode = mode_env;
        a.stagger_sleeps = (int)(0.5 * (8.8 + 2.1 * (a.cin / kW2KS)) / 3.4 + 0.5);
    }
    hipLaunchKernelGGL(head_w ? wc.kernel_head : wc.kernel, dim3((unsigned)grid), dim3(256), wc.lds, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---- quarter-split shape (conv_wino2q_f32_kernel): its own configuration id behind conv_wino4's (conv_igemm.hip)
bool wino2q_ok(int cin, int cout, int head_c) {
    if (cin % kW2KS != 0 || cout % kW2qBC != 0) return false;
    return head_c <= 0 || (cout == kW2qBC && head_c <= 4);
}

int wino2q_init_attrs() {   // called under the lock of init_kernel_attrs (conv_igemm.hip)
    static bool done = false;
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2q_f32_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kW2qLdsBytes));
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2q_f32_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kW2qLdsBytes));
    done = true;
    return W2L_OK;
}

int wino2q_launch(const WinoKArgs& w, const float* head_w, const float* head_b, int head_c, int head_act, hipStream_t stream,
                  long long* flops_out) {
    Wino2KArgs a;
    a.x = w.x; a.y = w.y; a.res = w.res; a.u = w.u; a.scale = w.scale; a.shift = w.shift;
    a.N = w.N; a.H = w.H; a.W = w.W; a.cin = w.cin; a.x_cs = w.x_cs;
    a.cout = w.cout; a.y_cs = w.y_cs; a.res_cs = w.res_cs; a.act = w.act;
    a.head_w = head_w; a.head_b = head_b; a.head_c = head_c; a.head_act = head_act;
    a.TH = (a.H + 1) / 2;
    a.TW = (a.W + 1) / 2;
    const W2Block b = wino2_pick_block(a.N, a.TH, a.TW, kW2qBT, kW2qRAW4);
    a.bh = b.bh; a.bw = b.bw; a.ni = b.ni;
    a.nby = ceil_div(a.TH, b.bh);
    a.nbx = ceil_div(a.TW, b.bw);
    a.ngi = ceil_div(a.N, b.ni);
    a.RH = 2 * b.bh + 2;
    a.RW = 2 * b.bw + 2;
    a.R4 = b.ni * a.RH * a.RW * 2;
    a.nks = a.cin / 8;
    a.tiles_n = a.cout / kW2qBC;
    a.total = (long long)a.ngi * a.nby * a.nbx * a.tiles_n;
    a.stagger_mode = 0;
    a.stagger_sleeps = 0;
    W2L_REQUIRE(a.total < (1ll << 31), "grid too large");
    W2L_REQUIRE((long long)a.N * a.H * a.W < (1ll << 31), "tensor too large");
    W2L_REQUIRE(head_w == nullptr || a.tiles_n == 1, "fused head needs all couts in one tile");
    if (flops_out) {   // dry run: 16 position-GEMMs of [items*32] x [32] x cin
        *flops_out = 2ll * 16 * a.total * kW2qBT * kW2qBC * a.cin;
        return W2L_OK;
    }
    long long grid = (a.total + 7) / 8 * 8;
    if (grid > 512) grid = 512;        // persistent, two workgroups per CU
    if (head_w) hipLaunchKernelGGL(conv_wino2q_f32_kernel<true>, dim3((unsigned)grid), dim3(256), kW2qLdsBytes, stream, a);
    else hipLaunchKernelGGL(conv_wino2q_f32_kernel<false>, dim3((unsigned)grid), dim3(256), kW2qLdsBytes, stream, a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // namespace w2l

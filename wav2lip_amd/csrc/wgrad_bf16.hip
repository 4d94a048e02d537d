// Weight gradient of nn.Conv2d / nn.ConvTranspose2d (models/conv.py:8,24,36) for the bf16-STORAGE training path:
//     dW[cp][cq][ky][kx] = sum_pix P[pix][cp] * Q[pix*s - p + (ky,kx)][cq]
// P = the tensor on the coarse grid (conv: dz; transposed conv: x), Q = the one on the fine grid; both NHWC bf16 in HBM, dW fp32
// in torch layout (which is [CP][CQ][kh][kw] for both layer kinds).  GEMM view: M = CP, N = (tap, cq), K = pixels, on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// What round 2's bf16 weight gradient did wrong (Weak 8): it read the Q operand once per tap through L2 (9x for a
// 3x3 layer) with 8 dword loads per 8 K-elements per lane, because K = pixels is the STRIDED direction of an NHWC tensor.  Here:
//   * a workgroup walks boxes of <= 160 P pixels (bh x bw pixels of one image, or ni whole small images).  The P box and the
//     Q box with its halo ((bh-1)*s+kh) x ((bw-1)*s+kw) are staged ONCE, by LDS-DMA (no registers, no ds_write; out-of-range
//     pixels / channels arrive as zeros = padding), as [pixel][32 channels] rows of 64 bytes in up to two 32-channel planes;
//   * every tap's shifted view of Q is then an LDS row offset: Q row of P slot k for tap (ky,kx) = qtab[k] + ky*qbw + kx;
//   * the MFMA wants 8 consecutive K (pixels) of one channel per lane - the transposed direction of the staged rows.  gfx950's
//     ds_read_b64_tr_b16 does that transpose in the LDS read path: a 16-lane group reads a [4 pixels][16 channels] block, every
//     lane supplying the address of its own 8 bytes (so the 4 pixels need not be neighbours: stride-2 layers and ragged boxes
//     work the same way), and lane i receives channel i of the four pixels (probed: tools/microbench/probe_lds.hip, P1).  With
//     64-byte rows the four rows of a group's read are 256 contiguous bytes for stride-1 layers: conflict-free, no swizzle;
//   * one workgroup = 64 cp x (taps of one group x 64 cq) with 9 accumulator tiles per wave (144 registers), so that a staged
//     pixel feeds up to 18 MFMA columns-tiles; K (= boxes) is split over gridDim.y to fill the chip, partial sums are reduced
//     in a fixed order by wgrad_bf16_reduce_kernel (bit-reproducible, no atomics).
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "w2l_common.h"

namespace w2l {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWgMaxKsub = 10;     // K-substeps (16 P pixels each) per box: <= 160 P pixels
constexpr int kWgRowB = 64;        // bytes per LDS row: 32 channels of one pixel
constexpr int kWgTiles = 9;        // 32x32 accumulators per wave
constexpr int kWgPPass = 5;        // DMA passes of a wave over the P planes: ceil(2 * 10 / 4)
constexpr int kWgQPass = 10;       // ... over the Q planes: <= 640 rows in total (qp * q_rows_pad)
constexpr int kWgQRowsMax = 640;
constexpr unsigned kWgOob = 0x80000000u;

struct WgB {
    const void* P;
    const void* Q;
    float* ws;            // [split][CP][wcols] fp32
    int N, Hp, Wp, p_cs, CP, CPp;
    int Hq, Wq, q_cs, CQ, CQp;
    int kw, ntaps, sy, sx, py, px;
    int ni, bh, bw, nslots, ksubs;
    int qbh, qbw, qrows;
    int boxes_y, boxes_x, nboxes, boxes_per_split;
    int mt, qp;           // 32-channel planes of P / Q per workgroup
    int tg, ncq, ntg;     // taps per group, number of cq slices, number of tap groups
    int wcols;            // ntaps * CQp
    int p_rows_pad, q_rows_pad;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16s_kernel(const WgB a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((address_space(3))) bf16x4* lds_v4_t;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    int tid = blockIdx.x;
    const int itg = tid % a.ntg;
    tid /= a.ntg;
    const int icq = tid % a.ncq;
    const int icp = tid / a.ncq;
    const int cp0 = icp * 64, cq0 = icq * 64;
    const int cpw = min(64, a.CPp - cp0);          // multiples of 8
    const int cqw = min(64, a.CQp - cq0);
    const int tap0 = itg * a.tg;
    const int tgn = min(a.tg, a.ntaps - tap0);     // taps of this group
    const int ncols = tgn * cqw;
    const int ntile = (ncols + 31) >> 5;

    const int p_plane = a.p_rows_pad * kWgRowB;
    const int q_plane = a.q_rows_pad * kWgRowB;
    char* Pl = smem;
    char* Ql = smem + a.mt * p_plane;
    int* qtab = reinterpret_cast<int*>(Ql + a.qp * q_plane);     // [160] byte offset of the tap-(0,0) Q row of every P slot

    // ---- constant tables: slot -> Q row
    const int box_pix = a.bh * a.bw;
    for (int k = t; k < kWgMaxKsub * 16; k += 256) {
        int q = 0;
        if (k < a.nslots) {
            const int i = k / box_pix, rem = k - i * box_pix;
            const int y = rem / a.bw, x = rem - y * a.bw;
            q = ((i * a.qbh + y * a.sy) * a.qbw + x * a.sx) * kWgRowB;
        }
        qtab[k] = q;
    }

    // ---- wave roles: M-tile wm, N-tiles wn + j * nwn
    const int wm = wave % a.mt;
    const int wn = wave / a.mt;
    const int nwn = 4 / a.mt;
    // source-lane geometry of a tr-read: lane supplies pixel sub-row (lane & 15) >> 2 (+ 8 * (lane >> 5)) and the 4 channels
    // 16 * ((lane >> 4) & 1) + 4 * (lane & 3) .. + 3 of the tile's 32
    const int lane_pix = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int lane_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned a_addr = lds0 + wm * p_plane + lane_pix * kWgRowB + lane_ch * 2;
    unsigned toff[kWgTiles];
#pragma unroll
    for (int j = 0; j < kWgTiles; ++j) {
        const int tile = j * nwn + wn;
        const int col = tile * 32 + lane_ch;
        unsigned o = 0;
        if (tile < ntile && col < ncols) {
            const int ti = col / cqw, c = col - ti * cqw;
            const int gt = tap0 + ti;
            const int ky = gt / a.kw, kx = gt - ky * a.kw;
            o = (unsigned)((ky * a.qbw + kx) * kWgRowB + (c >> 5) * q_plane + (c & 31) * 2);
        }
        toff[j] = o;
    }
    const unsigned q_addr = lds0 + a.mt * p_plane;
    const int nt_w = ntile > wn ? min(kWgTiles, (ntile - wn + nwn - 1) / nwn) : 0;      // tiles j * nwn + wn < ntile of this wave

    // ---- DMA coordinates, box-independent: per pass the (image, y, x) of this lane's row inside the box and its byte offset
    // relative to the box origin
    const int drow = lane >> 2;          // row of a 16-row DMA instruction
    const int dchunk = lane & 3;         // 16-byte chunk = 8 channels
    int p_rel[kWgPPass], p_crd[kWgPPass], q_rel[kWgQPass], q_crd[kWgQPass];
    const int p_instr = a.mt * (a.p_rows_pad >> 4), q_instr = a.qp * (a.q_rows_pad >> 4);
#pragma unroll
    for (int ps = 0; ps < kWgPPass; ++ps) {
        const int ii = ps * 4 + wave;
        int rel = 0, crd = -1;
        if (ii < p_instr) {
            const int plane = ii / (a.p_rows_pad >> 4), rb = ii - plane * (a.p_rows_pad >> 4);
            const int row = rb * 16 + drow;
            const int ch = plane * 32 + dchunk * 8;
            if (row < a.nslots && ch < cpw) {
                const int i = row / box_pix, rem = row - i * box_pix;
                const int y = rem / a.bw, x = rem - y * a.bw;
                rel = (((i * a.Hp + y) * a.Wp + x) * a.p_cs + cp0 + ch) * 2;
                crd = i | (y << 8) | (x << 20);
            }
        }
        p_rel[ps] = rel;
        p_crd[ps] = crd;
    }
#pragma unroll
    for (int ps = 0; ps < kWgQPass; ++ps) {
        const int ii = ps * 4 + wave;
        int rel = 0, crd = -1;
        if (ii < q_instr) {
            const int plane = ii / (a.q_rows_pad >> 4), rb = ii - plane * (a.q_rows_pad >> 4);
            const int row = rb * 16 + drow;
            const int ch = plane * 32 + dchunk * 8;
            if (row < a.qrows && ch < cqw) {
                const int i = row / (a.qbh * a.qbw), rem = row - i * (a.qbh * a.qbw);
                const int y = rem / a.qbw, x = rem - y * a.qbw;
                rel = (((i * a.Hq + y) * a.Wq + x) * a.q_cs + cq0 + ch) * 2;
                crd = i | (y << 8) | (x << 20);
            }
        }
        q_rel[ps] = rel;
        q_crd[ps] = crd;
    }
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.P), 0, (int)((((long long)a.N * a.Hp * a.Wp - 1) * a.p_cs + a.CPp) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.Q), 0, (int)((((long long)a.N * a.Hq * a.Wq - 1) * a.q_cs + a.CQp) * 2), 0x00020000);

    f32x16 acc[kWgTiles];
#pragma unroll
    for (int j = 0; j < kWgTiles; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int b_first = blockIdx.y * a.boxes_per_split;
    const int b_last = min(a.nboxes, b_first + a.boxes_per_split);
    const int bxy = a.boxes_y * a.boxes_x;
    for (int b = b_first; b < b_last; ++b) {
        const int ig = b / bxy, rem = b - ig * bxy;
        const int by = rem / a.boxes_x, bx = rem - by * a.boxes_x;
        const int n0 = ig * a.ni, y0 = by * a.bh, x0 = bx * a.bw;
        const int qy0 = y0 * a.sy - a.py, qx0 = x0 * a.sx - a.px;
        const unsigned p_org = (unsigned)(((n0 * a.Hp + y0) * a.Wp + x0) * a.p_cs * 2);
        const unsigned q_org = (unsigned)(((n0 * a.Hq + qy0) * a.Wq + qx0) * a.q_cs * 2);   // may wrap below zero: valid sums do not
        __syncthreads();   // the previous box's fragments have been read (and, first time, qtab is written)
#pragma unroll
        for (int ps = 0; ps < kWgPPass; ++ps) {
            const int ii = ps * 4 + wave;
            if (ii < p_instr) {
                const int crd = p_crd[ps];
                const bool ok = (crd >= 0) & (n0 + (crd & 0xff) < a.N) & (y0 + ((crd >> 8) & 0xfff) < a.Hp) & (x0 + (crd >> 20) < a.Wp);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(Pl + ii * 1024), 16, (int)(ok ? p_org + (unsigned)p_rel[ps] : kWgOob), 0, 0, 0);
            }
        }
#pragma unroll
        for (int ps = 0; ps < kWgQPass; ++ps) {
            const int ii = ps * 4 + wave;
            if (ii < q_instr) {
                const int crd = q_crd[ps];
                const bool ok = (crd >= 0) & (n0 + (crd & 0xff) < a.N) & ((unsigned)(qy0 + ((crd >> 8) & 0xfff)) < (unsigned)a.Hq) &
                                ((unsigned)(qx0 + (crd >> 20)) < (unsigned)a.Wq);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)(Ql + ii * 1024), 16, (int)(ok ? q_org + (unsigned)q_rel[ps] : kWgOob), 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // K loop over the box's 16-pixel substeps, software-pipelined by hand: the (substep, tile) sequence is straight-line code
        // over all nine tile slots; the B fragment three steps ahead - of this substep, or of the next one past the ninth slot - is
        // requested before each MFMA into a ring of three register pairs, the next substep's A fragment and Q-row table entries
        // likewise, and only the MFMA itself sits behind the wave-uniform "this wave has a tile j" test (slots without a tile read
        // a valid LDS address and are not accumulated).  The first version read two fragments, waited for them and issued one
        // MFMA, nine times per substep: every MFMA paid a full LDS latency, MFMA busy 0.2.
        auto rd = [&](unsigned addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(size_t)addr); };
        constexpr int D = 3;
        unsigned q0 = q_addr + (unsigned)qtab[lane_pix], q1 = q_addr + (unsigned)qtab[lane_pix + 4];
        bf16x4 a0 = rd(a_addr), a1 = rd(a_addr + 256);
        bf16x4 r0[D], r1[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { r0[d] = rd(q0 + toff[d]); r1[d] = rd(q1 + toff[d]); }
        for (int ks = 0; ks < a.ksubs; ++ks) {
            const int kn = ks + 1 < a.ksubs ? ks + 1 : ks;             // the last substep prefetches itself again (unused)
            const unsigned q0n = q_addr + (unsigned)qtab[kn * 16 + lane_pix];
            const unsigned q1n = q_addr + (unsigned)qtab[kn * 16 + lane_pix + 4];
            const bf16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            a0 = rd(a_addr + kn * 1024);
            a1 = rd(a_addr + kn * 1024 + 256);
#pragma unroll
            for (int j = 0; j < kWgTiles; ++j) {
                const bf16x8 bfr = __builtin_shufflevector(r0[j % D], r1[j % D], 0, 1, 2, 3, 4, 5, 6, 7);
                if (j + D < kWgTiles) {
                    r0[j % D] = rd(q0 + toff[j + D]);
                    r1[j % D] = rd(q1 + toff[j + D]);
                } else {
                    r0[j % D] = rd(q0n + toff[j + D - kWgTiles]);
                    r1[j % D] = rd(q1n + toff[j + D - kWgTiles]);
                }
                if (j < nt_w) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
            }
            q0 = q0n;
            q1 = q1n;
        }
    }

    // ---- partial sums -> ws[split][cp][(tap, cq)]: lane holds column (lane & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* wsz = a.ws + (long long)blockIdx.y * a.CP * a.wcols;
#pragma unroll
    for (int j = 0; j < kWgTiles; ++j) {
        const int tile = j * nwn + wn;
        if (tile >= ntile) continue;
        const int col = tile * 32 + (lane & 31);
        if (col >= ncols) continue;
        const int ti = col / cqw, c = col - ti * cqw;
        if (cq0 + c >= a.CQ) continue;
        const int gcol = (tap0 + ti) * a.CQp + cq0 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cp = cp0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (cp < a.CP) wsz[(long long)cp * a.wcols + gcol] = acc[j][r];
        }
    }
}

// dW[cp][cq][tap] = sum_split ws[split][cp][tap * CQp + cq].  One workgroup = 64 elements x 4 split lanes: lane q adds the
// splits z = q, q+4, ... with four loads in flight, the four lane sums meet in LDS in a fixed order (bit-reproducible; the
// first version walked up to 512 splits serially per thread and cost 3 ms of a 30 ms step on dependent L2 misses)
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int CP,
                                                                int CQ, int CQp, int ntaps, int wcols) {
    __shared__ float red[4][64];
    const long long total = (long long)CP * wcols;
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        const long long i = base + el;
        float s = 0.f;
        if (i < total) {
            const float* p = ws + i;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            int z = q;
            for (; z + 12 < nsplit; z += 16) {
                t0 += p[(long long)z * total];
                t1 += p[(long long)(z + 4) * total];
                t2 += p[(long long)(z + 8) * total];
                t3 += p[(long long)(z + 12) * total];
            }
            for (; z < nsplit; z += 4) t0 += p[(long long)z * total];
            s = (t0 + t1) + (t2 + t3);
        }
        red[q][el] = s;
        __syncthreads();
        if (q == 0 && i < total) {
            const int cp = (int)(i / wcols);
            const int g = (int)(i - (long long)cp * wcols);
            const int tap = g / CQp, cq = g - tap * CQp;
            if (cq < CQ) dw[((long long)cp * CQ + cq) * ntaps + tap] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
        }
        __syncthreads();
    }
}

float* conv_workspace(hipStream_t stream, size_t bytes);   // conv_igemm.hip: grow-only scratch, one per stream

static int wgb_init_attrs() {
    static std::mutex m;
    static bool done = false;
    std::lock_guard<std::mutex> lock(m);
    if (done) return W2L_OK;
    W2L_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16s_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    done = true;
    return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

extern "C" int w2l_conv_wgrad_bf16(const w2l_conv_geom* g, void* stream, int N, int H, int W, const void* x, int x_cs,
                                    const void* dz, int dz_cs, float* dweight) {
    W2L_REQUIRE(g && x && dz && dweight, "NULL argument");
    W2L_REQUIRE(N >= 1 && H >= 1 && W >= 1, "bad shape N=%d H=%d W=%d", N, H, W);
    int Ho, Wo;
    if (w2l_conv_out_hw(g, H, W, &Ho, &Wo) != W2L_OK) return W2L_ERR_ARG;
    W2L_REQUIRE(Ho >= 1 && Wo >= 1, "empty output %dx%d", Ho, Wo);
    const int cin8 = round_up(g->cin, 8), cout8 = round_up(g->cout, 8);
    W2L_REQUIRE(x_cs >= cin8 && (x_cs & 7) == 0 && dz_cs >= cout8 && (dz_cs & 7) == 0,
                "channel strides must be multiples of 8 covering the padded channels (x_cs=%d, dz_cs=%d)", x_cs, dz_cs);
    W2L_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0, "x / dz must be 16-byte aligned");
    const long long lim = 1ll << 31;
    W2L_REQUIRE(((long long)N * H * W * x_cs) * 2 < lim && ((long long)N * Ho * Wo * dz_cs) * 2 < lim,
                "activation buffer larger than 2 GiB: split the batch");
    if (wgb_init_attrs() != W2L_OK) return W2L_ERR_HIP;
    WgB a;
    a.N = N;
    if (!g->transposed) {   // P = dz on the output grid, Q = x
        a.P = dz; a.Hp = Ho; a.Wp = Wo; a.p_cs = dz_cs; a.CP = g->cout;
        a.Q = x; a.Hq = H; a.Wq = W; a.q_cs = x_cs; a.CQ = g->cin;
    } else {                // P = x on the input grid, Q = dz
        a.P = x; a.Hp = H; a.Wp = W; a.p_cs = x_cs; a.CP = g->cin;
        a.Q = dz; a.Hq = Ho; a.Wq = Wo; a.q_cs = dz_cs; a.CQ = g->cout;
    }
    a.CPp = round_up(a.CP, 8);
    a.CQp = round_up(a.CQ, 8);
    a.kw = g->kw; a.ntaps = g->kh * g->kw; a.sy = g->sh; a.sx = g->sw; a.py = g->ph; a.px = g->pw;
    a.mt = a.CPp > 32 ? 2 : 1;
    a.qp = a.CQp > 32 ? 2 : 1;
    const int qrow_cap = kWgQRowsMax / a.qp;
    // ---- the box: ni whole images when an image has <= 80 pixels, else bh x bw pixels of one image; least (boxes x K-substeps)
    // among the shapes whose Q halo fits the LDS budget
    const int hw = a.Hp * a.Wp;
    long long best_cost = -1;
    a.ni = 1; a.bh = 1; a.bw = 1;
    auto consider = [&](int ni, int bh, int bw) {
        const int slots = ni * bh * bw;
        if (slots > kWgMaxKsub * 16 || ni > 255) return;
        const int qbh = (bh - 1) * a.sy + g->kh, qbw = (bw - 1) * a.sx + g->kw;
        if (qbh > 4095 || qbw > 2047 || (long long)ni * qbh * qbw > qrow_cap) return;
        const long long boxes = (long long)ceil_div(N, ni) * ceil_div(a.Hp, bh) * ceil_div(a.Wp, bw);
        // staged bytes per box count too (a thin box re-reads its halo): cost = boxes * (MFMA substeps + Q rows / 16)
        const long long cost = boxes * (4 * ceil_div(slots, 16) + ceil_div(ni * qbh * qbw, 16));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; a.ni = ni; a.bh = bh; a.bw = bw; }
    };
    if (hw <= 80)
        for (int ni = 1; ni <= kWgMaxKsub * 16 / hw && ni <= N; ++ni) consider(ni, a.Hp, a.Wp);
    for (int bw = 1; bw <= a.Wp && bw <= 160; ++bw)
        for (int bh = 1; bh <= a.Hp && bh * bw <= kWgMaxKsub * 16; ++bh) consider(1, bh, bw);
    W2L_REQUIRE(best_cost >= 0, "weight gradient: no pixel box fits (kernel %dx%d, stride %dx%d)", g->kh, g->kw, g->sh, g->sw);
    a.nslots = a.ni * a.bh * a.bw;
    a.ksubs = ceil_div(a.nslots, 16);
    a.qbh = (a.bh - 1) * a.sy + g->kh;
    a.qbw = (a.bw - 1) * a.sx + g->kw;
    a.qrows = a.ni * a.qbh * a.qbw;
    a.boxes_y = ceil_div(a.Hp, a.bh);
    a.boxes_x = ceil_div(a.Wp, a.bw);
    a.nboxes = ceil_div(N, a.ni) * a.boxes_y * a.boxes_x;
    a.p_rows_pad = a.ksubs * 16;
    a.q_rows_pad = round_up(a.qrows, 16);
    // ---- tiles: 64 cp x (tap group x 64 cq); a wave holds 9 accumulators, 4 / mt waves share the N range
    const int cols_max = 32 * kWgTiles * (4 / a.mt);
    const int cqw_max = a.CQp < 64 ? a.CQp : 64;
    a.tg = cols_max / cqw_max;      // taps per group: tg * cqw columns fit the workgroup's 9 * (4 / mt) tiles
    if (a.tg > a.ntaps) a.tg = a.ntaps;
    W2L_REQUIRE(a.tg >= 1, "weight gradient: tap group empty");
    a.ntg = ceil_div(a.ntaps, a.tg);
    a.ncq = ceil_div(a.CQp, 64);
    const int ncp = ceil_div(a.CPp, 64);
    const int ntiles = ncp * a.ncq * a.ntg;
    a.wcols = a.ntaps * a.CQp;
    int splits = 512 / ntiles;
    if (splits < 1) splits = 1;
    if (splits > a.nboxes) splits = a.nboxes;
    a.boxes_per_split = ceil_div(a.nboxes, splits);
    splits = ceil_div(a.nboxes, a.boxes_per_split);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t ws_bytes = (size_t)splits * a.CP * a.wcols * sizeof(float);
    a.ws = conv_workspace(s, ws_bytes);
    if (!a.ws) return W2L_ERR_NOMEM;
    const int lds = a.mt * a.p_rows_pad * kWgRowB + a.qp * a.q_rows_pad * kWgRowB + kWgMaxKsub * 16 * 4;
    W2L_REQUIRE(lds <= 80 * 1024, "weight gradient: LDS budget exceeded (%d bytes)", lds);
    if (flops_counting()) {
        // per box and workgroup: ksubs K-substeps x mt M-tiles x (9 * 4 / mt) N-tile slots of which the valid ones issue an MFMA
        long long nt_sum = 0;
        for (int icq = 0; icq < a.ncq; ++icq)
            for (int itg = 0; itg < a.ntg; ++itg) {
                const int cqw = a.CQp - icq * 64 < 64 ? a.CQp - icq * 64 : 64;
                const int tgn = a.ntaps - itg * a.tg < a.tg ? a.ntaps - itg * a.tg : a.tg;
                nt_sum += (tgn * cqw + 31) / 32;
            }
        flops_add(2ll * 32 * 32 * 16 * a.ksubs * a.mt * nt_sum * ncp * (long long)a.nboxes, 6);
    }
    hipLaunchKernelGGL(conv_wgrad_bf16s_kernel, dim3(ntiles, splits), dim3(256), lds, s, a);
    W2L_HIP_CHECK(hipGetLastError());
    const long long total = (long long)a.CP * a.wcols;
    int rb = (int)((total + 63) / 64);
    if (rb > 8192) rb = 8192;
    hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(rb), dim3(256), 0, s, a.ws, dweight, splits, a.CP, a.CQ, a.CQp, a.ntaps, a.wcols);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

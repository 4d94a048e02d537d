// Internal helpers shared by the HIP translation units of libw2l_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/w2l_hip.h"

namespace w2l {

void set_error(const char* fmt, ...);
// executed-FLOP counter (api.hip): families 0 = fp32 conv (forward / data gradient), 1 = fp32 weight gradient (direct GEMM),
// 2 = fp32 Winograd weight gradient, 3 = bf16c conv, 4 = bf16c weight gradient, 5 = bf16-storage conv, 6 = bf16-storage weight
// gradient, 7 = non-matrix-core reduction kernels (tiny heads)
bool flops_counting();
void flops_add(long long flops, int family);

#define W2L_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            w2l::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                           __FILE__, __LINE__);                                          \
            return W2L_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define W2L_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            w2l::set_error(__VA_ARGS__);      \
            return W2L_ERR_ARG;               \
        }                                     \
    } while (0)

// The ReLU / LeakyReLU / identity family as ONE expression of the negative-side slope (0, 0.01, 1): a select, not
// fma(neg, min(v, 0), max(v, 0)) - that form turned ReLU(-inf) into 0 * -inf = NaN and a NaN input into 0 (min / max drop the
// NaN), which hides a diverged training step.  Here NaN propagates (v < 0 is false: v itself), ReLU(-inf) = 0 (the one product
// that is not a number is 0 * -inf) and LeakyReLU(-inf) = -inf, as torch.  Used by every training-path kernel, split or not.
__device__ __forceinline__ float act_leaky(float v, float neg) {
    const float nv = neg * v;
    return v < 0.f ? (nv == nv ? nv : 0.f) : v;
}

// Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  This maps the hardware id to a
// logical id such that every XCD walks one CONTIGUOUS range of logical ids: neighbouring tiles (which share input halos
// and A/B operand tiles) then meet in the same L2 instead of being fetched from HBM once per XCD.
#ifndef W2L_NO_XCD_REMAP
__host__ __device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {
    constexpr unsigned kXcd = 8;
    const unsigned xcd = id % kXcd, local = id / kXcd;
    const unsigned base = n / kXcd, rem = n % kXcd;
    return xcd * base + (xcd < rem ? xcd : rem) + local;
}
#else
__host__ __device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned) { return id; }
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// HBM bytes a launch fetches under the two tile orders of a (phase-major) grid whose 8 XCDs each walk a contiguous range of tiles,
// x = input bytes, w = weight bytes, tn = cout-tiles, passes = phases: cout-tiles FASTEST - every XCD holds 1/8 of the M-tiles and
// streams all of the weights; cout-tile SLOWEST - an XCD's range is tn / 8 of the M-tiles of the one or two cout-tiles it touches.
// The model ranks every launch of the batch-128 plan as FETCH_SIZE measures them (profiles/r05/l_*).
static inline long long fetch_cout_fastest(long long x, long long w, int tn, int passes) { return passes * x + 8 * w; }
static inline long long fetch_cout_slowest(long long x, long long w, int tn, int passes) {
    const int touched = tn >= 8 ? tn : 8 + (8 % tn ? tn - 1 : 0);   // (XCD, cout-tile) pairs
    return passes * x * (tn < 8 ? tn : 8) + w * touched / tn;
}

// ---- conv implicit-GEMM kernel arguments ---------------------------------------------------
constexpr int kMaxPhases = 9;   // convT k3 s1 p0 on 1x1 -> 3x3 has 9 single-tap phases
constexpr int kMaxTaps = 49;    // 7x7 stems
constexpr int kBK = 32;         // K-step of the implicit GEMM (floats)

struct ConvPhase {
    int ntaps;      // taps contributing to this output phase
    int kp;         // padded K length of this phase's weight slab = round_up(ntaps*cin_p, kBK)
    int po_y, po_x; // output position offset of the phase
    int tap_off;    // first entry of this phase in the tap table
    int pad_;
    long long w_off;  // float offset of this phase's slab [cout_p][kp] in the packed weights
};

struct ConvKArgs {
    const float* x;
    float* y;
    const float* res;
    const float* w;
    const __bf16* wsplit;   // the packed weights as three bf16 planes per phase (split-operand kernel), else unused
    const float* scale;
    const float* shift;
    const int* taps;  // packed (dy & 0xffff) | (dx << 16), relative input offsets per tap
    int N, H, W, cin_p, x_cs;
    int Ho, Wo, cout, cout_p, y_cs, res_cs;
    int Hq, Wq;       // q-grid: one implicit-GEMM row per (n, qy, qx)
    int sy, sx;       // input pixel = q*s + d(tap)
    int omy, omx;     // output pixel = q*om + po(phase)
    int act;
    int ncols;        // valid GEMM columns: cout, or pair*cout for an x-paired variant
    int pair;         // 1, or 2: GEMM column j is channel j % cout of output pixel ox + j / cout (x-paired small-cout conv)
    // fused 1x1 head (models/wav2lip.py:84-85): out[o] = head_act( sum_c head_w[o][c] * act(...)[c] + head_b[o] ), o < head_c;
    // when head_w != NULL the kernel writes head_c channels per pixel to y instead of cout
    const float* head_w;
    const float* head_b;
    int head_c, head_act;
    int vec_epilogue;  // 1: float4 epilogue (cout, strides and pointers 16-byte friendly)
    int M;            // N*Hq*Wq
    int tiles_m, tiles_n;
    int ksplit;           // gridDim.z: K-steps are cut into ksplit ranges of steps_per_split
    int steps_per_split;
    float* ws;            // split-K partial sums [ksplit][N*Ho*Wo][cout_p] (ksplit > 1)
    int order, order_r;   // workgroup -> (phase, tile_m, tile_n) order, see igemm_block_coords
    ConvPhase ph[kMaxPhases];
};

// Which (phase, M-tile, cout-tile) a workgroup of the implicit-GEMM grid (x = tiles, y = phases; z = K-splits, untouched) computes.
// Workgroups are handed to the 8 XCDs round-robin in dispatch order and every XCD has its own 4 MB L2, so WHAT runs side by side and
// back to back on one XCD decides how often an operand is fetched from HBM (measured per launch with FETCH_SIZE, profiles/r05/l_*):
//   kOrderPhaseMajor:   the plain grid order - all tiles of phase 0, then all of phase 1, ...; inside a phase every XCD walks a
//                       contiguous range of M-tiles, cout-tiles fastest.  A stride-2 transposed layer makes four passes over
//                       its input this way (512 -> 128 at 24x24, 128 frames: 151 MB of input, 763 MB fetched).
//   kOrderCoutSlowest:  phase-major too, but inside a phase the cout-tile is the SLOWEST index: an XCD's range covers one or two
//                       cout-tiles and streams its share of the M-tiles past their weights - for layers whose weights outweigh
//                       their input (512 -> 512 at 3x3: every XCD fetched all of the weights, 126 MB; now 1/8 of them each, 30 MB).
//   kOrderPhaseBlocked: groups of order_r M-tiles; an XCD runs phase 0 of a group (all cout-tiles), then phase 1 of the SAME group,
//                       ...: the group's input rows are still in L2 for phases 1-3 (438 MB fetched, same duration).  Needs several
//                       groups per XCD: phases differ in length (1 / 2 / 2 / 4 taps), and an XCD that holds one group's light phases
//                       while its neighbour holds the heavy ones finishes early (1024 -> 384 at 6x6: +19 %).
// What did NOT work, measured (profiles/r05/l_*): the phases of one M-tile SIDE BY SIDE on an XCD - phase fastest, or groups shorter
// than a round of 32 workgroups - fetches less but runs 30-40 % longer on every transposed layer: workgroups of different phases
// differ 1 : 2 : 2 : 4 in length, and the chip runs best when the workgroups in flight are alike.
enum { kOrderPhaseMajor = 0, kOrderCoutSlowest = 1, kOrderPhaseBlocked = 2 };
__host__ __device__ __forceinline__ void igemm_block_decode(int order, int order_r, int tiles_m, int tiles_n, unsigned bx, unsigned by,
                                                            unsigned nx, unsigned ny, int& phase, int& tile_m, int& tile_n) {
    const unsigned tm = (unsigned)tiles_m, tn = (unsigned)tiles_n;
    if (order != kOrderPhaseBlocked) {
        const unsigned b = xcd_remap(bx, nx);
        phase = (int)by;
        if (order == kOrderCoutSlowest) { tile_m = (int)(b % tm); tile_n = (int)(b / tm); }
        else { tile_n = (int)(b % tn); tile_m = (int)(b / tn); }
        return;
    }
    const unsigned j = xcd_remap(by * nx + bx, nx * ny);
    const unsigned R = (unsigned)order_r, G = R * tn * ny;
    const unsigned g = j / G, r = j - g * G;
    const unsigned left = tm - g * R, Re = left < R ? left : R;
    phase = (int)(r / (Re * tn));
    const unsigned rr = r % (Re * tn);
    tile_n = (int)(rr % tn);
    tile_m = (int)(g * R + rr / tn);
}
__device__ __forceinline__ void igemm_block_coords(const ConvKArgs& a, int& phase, int& tile_m, int& tile_n) {
    igemm_block_decode(a.order, a.order_r, a.tiles_m, a.tiles_n, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, phase, tile_m, tile_n);
}

// ---- Winograd F(2x2,3x3) kernel arguments (conv_wino.hip) -----------------------------------
struct WinoKArgs {
    const float* x;
    float* y;
    const float* res;
    const float* u;      // transformed weights in MFMA B-fragment order (wino_pack)
    const float* scale;
    const float* shift;
    int N, H, W, cin, x_cs;   // output is N x H x W too (3x3, stride 1, pad 1)
    int cout, y_cs, res_cs;
    int TH, TW, M;       // 2x2 output tiles per image and in total (filled by wino_launch)
    int nks;             // cin / 8
    int tiles_n, tiles_m;
    int act;
    int m_fastest;       // conv_wino.hip only: work items run M-tile fastest (weight-heavy layers), else cout-tile fastest
};

int wino_num_cfgs();
int wino_init_attrs();
bool wino_cfg_ok(int cfg, int cin, int cout);
long long wino_u_floats(int cin, int cout);
int wino_pack(const float* w, float* u, int cin, int cout, int transposed, hipStream_t stream);
int wino_launch(int cfg, WinoKArgs a, hipStream_t stream, long long* flops_out);
// second-generation kernel (conv_wino2.hip): configuration ids wino_num_cfgs() + [0, wino2_num_cfgs()) of the Winograd family
int wino2_num_cfgs();
bool wino2_ok(int cfg, int cin, int cout, int head_c);
int wino2_init_attrs();
int wino2_launch(int cfg, const WinoKArgs& a, const float* head_w, const float* head_b, int head_c, int head_act,
                 hipStream_t stream, long long* flops_out);

// quarter-split shape of the second-generation kernel (32 tiles x 32 couts, two workgroups per CU; conv_wino2.hip): the
// configuration id after conv_wino4's; same transformed weights as the other F(2x2) kernels
bool wino2q_ok(int cin, int cout, int head_c);
int wino2q_init_attrs();
int wino2q_launch(const WinoKArgs& a, const float* head_w, const float* head_b, int head_c, int head_act, hipStream_t stream,
                  long long* flops_out);

// F(4x4,3x3) Winograd kernel (conv_wino4.hip): the configuration id after conv_tp2's; its own 36-position weight transform
bool wino4_ok(int cin, int cout);
long long wino4_u_floats(int cin, int cout);
int wino4_pack(const float* w, float* u, int cin, int cout, int transposed, hipStream_t stream);
int wino4_init_attrs();
int wino4_launch(const WinoKArgs& a, const float* u4, hipStream_t stream, long long* flops_out);

// split-operand F(2x2,3x3) Winograd kernel (conv_wino2s.hip): fp32 result from the bf16 matrix cores; the LAST configuration id
// (behind the split-operand implicit-GEMM ids); its own pre-split 16-position weight transform, split from the F(2x2) fp32 weights on first use
bool wino2s_ok(int cin, int cout);
long long wino2s_u_elems(int cin, int cout);
int wino2s_pack(const float* wino_u32, __bf16* u, int cin, int cout, hipStream_t stream);   // from wino_pack's fp32 U
int wino2s_launch(const WinoKArgs& a, const __bf16* u, hipStream_t stream, long long* flops_out);
void wino2s_plane_geom(int bh, int bw, int ni, int* pitch, int* istride);

// fused-phase stride-2 transposed 3x3 convolution (conv_tp2.hip): the configuration id after the Winograd families
bool tp2_ok(const w2l_conv_geom& g);
long long tp2_u_floats(int cin, int cout);
int tp2_pack(const float* w, float* u, int cin, int cout, hipStream_t stream);
int tp2_init_attrs();
// direct 3x3 stride-1 convolution with split operands for 32-cout layers, optional fused 1x1 head (conv_k3s.hip)
bool k3s_ok(const w2l_conv_geom& g);
long long k3s_u_elems(int cin);
int k3s_pack(const float* w, __bf16* u, int cin, hipStream_t stream);
int k3s_init_attrs();
int k3s_launch(const float* x, int x_cs, float* y, int y_cs, const float* res, int res_cs, const __bf16* u, const float* scale,
               const float* shift, const float* head_w, const float* head_b, int head_c, int head_act, int N, int H, int W, int cin,
               int act, hipStream_t stream, long long* flops_out);
// the generator's 7x7 first layer with split operands (conv_stem7s.hip): region staged and split once, contraction out of LDS
bool stem7s_ok(const w2l_conv_geom& g);
long long stem7s_u_elems();
int stem7s_pack(const float* w, __bf16* u, int cin, hipStream_t stream);
int stem7s_init_attrs();
int stem7s_launch(const float* x, int x_cs, float* y, int y_cs, const __bf16* u, const float* scale, const float* shift, int N, int H,
                  int W, int act, hipStream_t stream, long long* flops_out);
// the same layer shape with split operands on the bf16 matrix cores (conv_tp2s.hip): pre-split weights built from tp2_pack's output
bool tp2s_ok(const w2l_conv_geom& g);
long long tp2s_u_elems(int cin, int cout);
int tp2s_pack(const float* tp2_u32, __bf16* u, int cin, int cout, hipStream_t stream);
int tp2s_init_attrs();
int tp2s_launch(const float* x, int x_cs, float* y, int y_cs, const __bf16* u, const float* scale, const float* shift, int N, int H,
                int W, int cin, int cout, int act, int ksplit, float* ws, int* ksplit_out, hipStream_t stream, long long* flops_out);
int tp2_launch(const float* x, int x_cs, float* y, int y_cs, const float* u, const float* scale, const float* shift, int N, int H,
               int W, int cin, int cout, int act, hipStream_t stream, long long* flops_out);

}  // namespace w2l

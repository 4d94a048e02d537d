// C-ABI glue of libw2l_hip.so: error channel, device queries, the small HBM-bound kernels of the data
// path (BN fold, layout changes, datagen pack, uint8 frames, L2-norm / cosine / BCE) and launch plans.
#include <stdarg.h>

#include <atomic>
#include <new>
#include <vector>

#include "w2l_common.h"

namespace w2l {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int conv_forward_impl(const w2l_conv* c, hipStream_t stream, int N, int H, int W, const float* x,
                      int x_cs, float* y, int y_cs, const float* res, int res_cs, int force_tile, int force_ksplit,
                      long long* flops_out, int* cfg_out);
int conv_num_tiles();
int conv_num_igemm_tiles();
void tune_store_launch(const w2l_conv* c, int N, int H, int W, bool has_res, int tile, int ksplit);

// ---- executed-FLOP counter (w2l_flops_begin / w2l_flops_end): while counting, every conv / weight-gradient launch of the
// PROCESS adds the multiply-add work its matrix cores execute (padded tiles and K, Winograd products) - launches still happen.
// Process-wide, not per thread: torch runs backward() on its autograd thread.
static std::atomic<bool> g_flops_on{false};
static std::atomic<long long> g_flops{0};
static std::atomic<long long> g_flops_by[8];
bool flops_counting() { return g_flops_on.load(std::memory_order_relaxed); }
void flops_add(long long f, int family) {
    g_flops.fetch_add(f, std::memory_order_relaxed);
    if (family >= 0 && family < 8) g_flops_by[family].fetch_add(f, std::memory_order_relaxed);
}

// ---- shader-clock probe (w2l_clock_probe): one wave reads s_memtime (shader clock) and s_memrealtime (100 MHz) around a spin
// of `spin_us` microseconds.  Launched on its own stream NEXT TO a workload it reports the clock the chip sustains under that
// workload - on gfx950 fp32 MFMA kernels run at ~2.05-2.1 GHz, not the 2.4 GHz the peak figures are quoted at (EXPERIMENTS.md).
__global__ void clock_probe_kernel(unsigned long long* out, unsigned spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

static inline int grid_for(long long work, int block, int cap = 8192) {
    long long g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------- BN fold
__global__ void bn_fold_kernel(int C, const float* bias, const float* gamma, const float* beta,
                               const float* mean, const float* var, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    const float mu = mean ? mean[c] : 0.f;
    // 1/sqrt in fp32 with IEEE sqrt and division, the same two roundings torch's eval batch_norm performs
    const float inv = var ? 1.0f / sqrtf(var[c] + eps) : 1.f;
    const float s = g * inv;
    scale[c] = s;
    shift[c] = ((bias ? bias[c] : 0.f) - mu) * s + b;
}

// ---------------------------------------------------------------- layout
// NCHW -> NHWC through a 32x33 LDS tile over (C, HW): coalesced on both sides
__global__ void nchw_to_nhwc_kernel(int C, int HW, const float* __restrict__ x, float* __restrict__ y,
                                    int y_cs, int c_zero_to) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? x[((long long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < c_zero_to) y[((long long)n * HW + p) * y_cs + c] = tile[tx][j];
    }
}

__global__ void nhwc_to_nchw_kernel(int C, int HW, const float* __restrict__ x, int x_cs,
                                    float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        tile[j][tx] = (c < C && p < HW) ? x[((long long)n * HW + p) * x_cs + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (c < C && p < HW) y[((long long)n * C + c) * HW + p] = tile[tx][j];
    }
}

// ---------------------------------------------------------------- datagen
// one thread per pixel: 3 bytes in, c_zero_to floats out (vector stores when the row is 16-B aligned)
__global__ void datagen_pack_kernel(long long npix, int S, const uint8_t* __restrict__ faces,
                                    float* __restrict__ y, int y_cs, int c_zero_to) {
    const int half = S / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)((i / S) % S);
        const uint8_t* f = faces + i * 3;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)((double)f[c] / 255.0);  // f64 divide, then f32 round
        const bool masked = row >= half;
        float* o = y + i * y_cs;
        float out[8] = {masked ? 0.f : v[0], masked ? 0.f : v[1], masked ? 0.f : v[2], v[0], v[1], v[2], 0.f, 0.f};
        if (c_zero_to == 8 && (y_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
            reinterpret_cast<float4*>(o)[0] = make_float4(out[0], out[1], out[2], out[3]);
            reinterpret_cast<float4*>(o)[1] = make_float4(out[4], out[5], out[6], out[7]);
        } else {
            for (int c = 0; c < c_zero_to; ++c) o[c] = c < 6 ? out[c] : 0.f;
        }
    }
}

__global__ void frames_to_u8_kernel(long long npix, const float* __restrict__ x, int x_cs,
                                    uint8_t* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (long long)gridDim.x * blockDim.x) {
        const float* p = x + i * x_cs;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // numpy: f32 array * python float 255. stays f32; astype(uint8) truncates toward zero
            const float v = p[c] * 255.0f;
            y[i * 3 + c] = (uint8_t)(int)v;
        }
    }
}

// ---------------------------------------------------------------- SyncNet tail
// one wave per row
__global__ void l2norm_rows_kernel(int N, int C, const float* __restrict__ x, int x_cs, float* __restrict__ y) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* p = x + (long long)row * x_cs;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += p[c] * p[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float d = fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < C; c += 64) y[(long long)row * C + c] = p[c] / d;
}

__global__ void cosine_rows_kernel(int N, int C, const float* __restrict__ a, const float* __restrict__ v,
                                   float* __restrict__ cos_out) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* pa = a + (long long)row * C;
    const float* pv = v + (long long)row * C;
    float dot = 0.f, na = 0.f, nv = 0.f;
    for (int c = lane; c < C; c += 64) {
        dot += pa[c] * pv[c];
        na += pa[c] * pa[c];
        nv += pv[c] * pv[c];
    }
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o);
        na += __shfl_xor(na, o);
        nv += __shfl_xor(nv, o);
    }
    // F.cosine_similarity (ATen): x.y / sqrt(clamp_min(|x|^2 |y|^2, eps^2)), eps = 1e-8
    if (lane == 0) cos_out[row] = dot / sqrtf(fmaxf(na * nv, 1e-16f));
}

// single block: mean over rows of BCE(cos, y) with torch's log clamp at -100
__global__ void bce_mean_kernel(int N, const float* __restrict__ p, const float* __restrict__ y,
                                float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float lp = fmaxf(logf(p[i]), -100.f);
        const float lq = fmaxf(logf(1.f - p[i]), -100.f);
        s -= y[i] * lp + (1.f - y[i]) * lq;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)N;
}

}  // namespace w2l

using namespace w2l;

struct PlanItem {
    const w2l_conv* c;
    int N, H, W;
    const float* x;
    int x_cs;
    float* y;
    int y_cs;
    const float* res;
    int res_cs;
    int tile;     // -1: heuristic
    int ksplit;
};
struct w2l_plan {
    std::vector<PlanItem> items;
};

extern "C" {

const char* w2l_last_error(void) { return g_err; }
int w2l_abi_version(void) { return 1; }

int w2l_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int w2l_device_arch(int dev, char* buf, size_t buflen) {
    W2L_REQUIRE(buf && buflen > 0, "NULL buffer");
    hipDeviceProp_t p;
    W2L_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s", p.gcnArchName);
    return W2L_OK;
}

int w2l_bn_fold(void* stream, int C, const float* bias, const float* gamma, const float* beta,
                const float* mean, const float* var, float eps, float* scale, float* shift) {
    W2L_REQUIRE(C >= 1 && scale && shift, "bad bn_fold arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), C,
                       bias, gamma, beta, mean, var, eps, scale, shift);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_nchw_to_nhwc(void* stream, int N, int C, int H, int W, const float* x, float* y, int y_cs, int c_zero_to) {
    W2L_REQUIRE(x && y && N >= 1 && C >= 1 && H >= 1 && W >= 1, "bad nchw_to_nhwc arguments");
    if (c_zero_to < C) c_zero_to = C;
    W2L_REQUIRE(y_cs >= c_zero_to, "y_cs=%d < %d", y_cs, c_zero_to);
    W2L_REQUIRE(N <= 65535, "N too large for one launch");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ceil_div(HW, 32), ceil_div(c_zero_to, 32), N), dim3(256), 0,
                       static_cast<hipStream_t>(stream), C, HW, x, y, y_cs, c_zero_to);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_nhwc_to_nchw(void* stream, int N, int C, int H, int W, const float* x, int x_cs, float* y) {
    W2L_REQUIRE(x && y && N >= 1 && C >= 1 && H >= 1 && W >= 1 && x_cs >= C, "bad nhwc_to_nchw arguments");
    W2L_REQUIRE(N <= 65535, "N too large for one launch");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), N), dim3(256), 0,
                       static_cast<hipStream_t>(stream), C, HW, x, x_cs, y);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_datagen_pack(void* stream, int N, int S, const uint8_t* faces, float* y, int y_cs, int c_zero_to) {
    W2L_REQUIRE(faces && y && N >= 1 && S >= 2, "bad datagen_pack arguments");
    if (c_zero_to < 6) c_zero_to = 6;
    W2L_REQUIRE(c_zero_to <= 8 && y_cs >= c_zero_to, "datagen_pack: need 6 <= c_zero_to <= 8 <= y_cs");
    const long long npix = (long long)N * S * S;
    hipLaunchKernelGGL(datagen_pack_kernel, dim3(grid_for(npix, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), npix, S, faces, y, y_cs, c_zero_to);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_frames_to_u8(void* stream, int N, int H, int W, const float* x, int x_cs, uint8_t* y) {
    W2L_REQUIRE(x && y && N >= 1 && H >= 1 && W >= 1 && x_cs >= 3, "bad frames_to_u8 arguments");
    const long long npix = (long long)N * H * W;
    hipLaunchKernelGGL(frames_to_u8_kernel, dim3(grid_for(npix, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), npix, x, x_cs, y);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_l2norm_rows(void* stream, int N, int C, const float* x, int x_cs, float* y) {
    W2L_REQUIRE(x && y && N >= 1 && C >= 1 && x_cs >= C, "bad l2norm arguments");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), N, C,
                       x, x_cs, y);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_cosine_bce(void* stream, int N, int C, const float* a, const float* v, const float* y, float* cos_out,
                   float* loss_out) {
    W2L_REQUIRE(a && v && cos_out && N >= 1 && C >= 1, "bad cosine_bce arguments");
    W2L_REQUIRE(y == nullptr || loss_out != nullptr, "loss_out required when y is given");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cosine_rows_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, s, N, C, a, v, cos_out);
    W2L_HIP_CHECK(hipGetLastError());
    if (y) {
        hipLaunchKernelGGL(bce_mean_kernel, dim3(1), dim3(256), 0, s, N, cos_out, y, loss_out);
        W2L_HIP_CHECK(hipGetLastError());
    }
    return W2L_OK;
}

int w2l_bce_mean(void* stream, int N, const float* p, const float* y, float* loss_out) {
    W2L_REQUIRE(p && y && loss_out && N >= 1, "bad bce_mean arguments");
    hipLaunchKernelGGL(bce_mean_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), N, p, y, loss_out);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

// ---------------------------------------------------------------- plans
int w2l_plan_create(w2l_plan_t** out) {
    W2L_REQUIRE(out, "NULL out");
    w2l_plan* p = new (std::nothrow) w2l_plan();
    if (!p) { set_error("out of host memory"); return W2L_ERR_NOMEM; }
    *out = p;
    return W2L_OK;
}

int w2l_plan_destroy(w2l_plan_t* p) {
    delete p;
    return W2L_OK;
}

int w2l_plan_add_conv(w2l_plan_t* p, const w2l_conv_t* c, int N, int H, int W, const float* x, int x_cs, float* y,
                      int y_cs, const float* res, int res_cs) {
    W2L_REQUIRE(p && c && x && y, "NULL argument");
    p->items.push_back(PlanItem{c, N, H, W, x, x_cs, y, y_cs, res, res_cs, -1, 1});
    return W2L_OK;
}

int w2l_plan_copy_item(w2l_plan_t* dst, const w2l_plan_t* src, int index) {
    W2L_REQUIRE(dst && src && index >= 0 && index < (int)src->items.size(), "bad plan_copy_item arguments");
    dst->items.push_back(src->items[index]);
    return W2L_OK;
}
int w2l_plan_size(const w2l_plan_t* p) { return p ? (int)p->items.size() : 0; }

int w2l_plan_run(const w2l_plan_t* p, void* stream) {
    W2L_REQUIRE(p, "NULL plan");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (const PlanItem& it : p->items) {
        const int rc = conv_forward_impl(it.c, s, it.N, it.H, it.W, it.x, it.x_cs, it.y, it.y_cs, it.res, it.res_cs, it.tile, it.ksplit, nullptr, nullptr);
        if (rc != W2L_OK) return rc;
    }
    return W2L_OK;
}

int w2l_plan_executed_flops(const w2l_plan_t* p, long long* flops_out, int* config_out) {
    W2L_REQUIRE(p && flops_out, "bad plan_executed_flops arguments");
    for (size_t i = 0; i < p->items.size(); ++i) {
        const PlanItem& it = p->items[i];
        const int rc = conv_forward_impl(it.c, nullptr, it.N, it.H, it.W, it.x, it.x_cs, it.y, it.y_cs, it.res, it.res_cs,
                                         it.tile, it.ksplit, &flops_out[i], config_out ? config_out + 2 * i : nullptr);
        if (rc != W2L_OK) return rc;
    }
    return W2L_OK;
}

int w2l_conv_num_igemm_tiles(void) { return conv_num_igemm_tiles(); }

int w2l_flops_begin(void) {
    g_flops = 0;
    for (auto& v : g_flops_by) v = 0;
    g_flops_on = true;
    return W2L_OK;
}
int w2l_igemm_block_order(int order, int order_r, int tiles_m, int tiles_n, int nphase, int* out) {
    W2L_REQUIRE(out && order >= 0 && order <= 2 && order_r >= 1 && tiles_m >= 1 && tiles_n >= 1 && nphase >= 1 && nphase <= kMaxPhases &&
                    (long long)tiles_m * tiles_n * nphase < (1ll << 24), "igemm_block_order: bad arguments");
    const unsigned nx = (unsigned)(tiles_m * tiles_n), ny = (unsigned)nphase;
    for (unsigned by = 0; by < ny; ++by)
        for (unsigned bx = 0; bx < nx; ++bx) {
            int* o = out + 3 * (size_t)(by * nx + bx);
            igemm_block_decode(order, order_r, tiles_m, tiles_n, bx, by, nx, ny, o[0], o[1], o[2]);
        }
    return W2L_OK;
}

int w2l_clock_probe(void* stream, int spin_us, unsigned long long* out2_dev) {
    W2L_REQUIRE(out2_dev != nullptr && spin_us > 0 && spin_us <= 100000, "clock_probe: out2_dev must be a device buffer of two uint64, spin 1..100000 us");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out2_dev, (unsigned)spin_us * 100u);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}
long long w2l_flops_end(long long* by_family) {
    g_flops_on = false;
    if (by_family)
        for (int i = 0; i < 8; ++i) by_family[i] = g_flops_by[i].load();
    return g_flops.load();
}

int w2l_conv_config_family(int id) {
    if (id < 0 || id >= conv_num_tiles()) return -1;
    if (id < conv_num_igemm_tiles()) return 0;
    if (id < conv_num_igemm_tiles() + wino_num_cfgs()) return 1;
    if (id < conv_num_igemm_tiles() + wino_num_cfgs() + wino2_num_cfgs()) return 2;
    const int tp2 = conv_num_igemm_tiles() + wino_num_cfgs() + wino2_num_cfgs();
    if (id == conv_num_tiles() - 1) return 9;         // direct 3x3 kernel with split operands for 32-cout layers (conv_k3s.hip), the last id
    if (id == conv_num_tiles() - 2) return 8;         // the 7x7 first-layer kernel with split operands (conv_stem7s.hip)
    if (id == conv_num_tiles() - 3) return 7;         // fused-phase stride-2 transposed kernel with split operands (conv_tp2s.hip)
    if (id == conv_num_tiles() - 4) return 6;         // split-operand F(2x2) Winograd (conv_wino2s.hip)
    if (id >= tp2 + 3) return 5;                      // implicit-GEMM tile id - (tp2 + 3) with split fp32 operands
    return id == tp2 ? 3 : (id == tp2 + 1 ? 4 : 2);   // the id behind conv_wino4's is the quarter-split conv_wino2 shape
}

// Kernel families a caller has switched off (bit mask over w2l_conv_config_family values): the autotune does not time them and
// a table / forced id of such a family falls through to the next rule (conv_forward_impl).  Used to build and to run the
// "exact" launch table (no F(4x4) Winograd: half the rounding error of the default table).
static int g_excluded_families = 0;
int w2l_conv_exclude_families(int mask) {
    W2L_REQUIRE(mask >= 0 && mask < 1024 && (mask & 1) == 0, "bad family mask %d (the implicit GEMM cannot be excluded)", mask);
    g_excluded_families = mask;
    return W2L_OK;
}
}  // extern "C"
namespace w2l {
bool conv_family_excluded(int id) {
    const int f = w2l_conv_config_family(id);
    return f >= 0 && ((g_excluded_families >> f) & 1);
}
}  // namespace w2l
extern "C" {

// Time every (tile, split-K) candidate of every recorded launch on the real buffers and keep the fastest.
int w2l_plan_autotune(w2l_plan_t* p, void* stream, int reps) {
    W2L_REQUIRE(p && reps >= 1, "bad plan_autotune arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    W2L_HIP_CHECK(hipEventCreate(&e0));
    W2L_HIP_CHECK(hipEventCreate(&e1));
    int rc = W2L_OK;
    const int ksplits[] = {1, 2, 4, 8, 16};
    for (PlanItem& it : p->items) {
        float best = 1e30f;
        int best_tile = -1, best_ks = 1;
        for (int tile = 0; tile < conv_num_tiles() && rc == W2L_OK; ++tile) {
            float t1 = 1e30f;   // time of this tile without split-K: deeper splits are only tried while they help
            int last_ks = -1;
            if (conv_family_excluded(tile)) continue;
            for (int ks : ksplits) {
                // A candidate the layer cannot run (a Winograd id on a strided layer, split-K on a kernel without it, ...)
                // silently resolves to something else: timing it would record noise under a configuration that never ran.
                // Resolve first (dry run, launches nothing) and time only candidates that resolve to themselves.
                long long fl = 0;
                int resolved[2] = {-1, -1};
                rc = conv_forward_impl(it.c, nullptr, it.N, it.H, it.W, it.x, it.x_cs, it.y, it.y_cs, it.res, it.res_cs, tile, ks,
                                       &fl, resolved);
                if (rc != W2L_OK) break;
                if (resolved[0] != tile) break;        // another kernel would run: no split of it is a candidate either
                const int rks = resolved[1];           // split-K as it resolves (clamped to the K-steps, dropped by kernels without it)
                if (rks == last_ks) continue;          // the same launch as the previous candidate
                last_ks = rks;
                float tmin = 1e30f;
                for (int r = 0; r <= reps && rc == W2L_OK; ++r) {   // r == 0: warm-up
                    (void)hipEventRecord(e0, s);
                    rc = conv_forward_impl(it.c, s, it.N, it.H, it.W, it.x, it.x_cs, it.y, it.y_cs, it.res, it.res_cs,
                                           tile, rks, nullptr, nullptr);
                    (void)hipEventRecord(e1, s);
                    if (hipEventSynchronize(e1) != hipSuccess) { set_error("sync failed in plan_autotune"); rc = W2L_ERR_HIP; }
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (r > 0 && ms < tmin) tmin = ms;
                }
                if (rc != W2L_OK) break;
                if (tmin < best) { best = tmin; best_tile = tile; best_ks = rks; }   // the RESOLVED configuration is what is recorded
                if (ks == 1) t1 = tmin;
                if (tmin > 1.15f * t1 || t1 > 0.25f) break;   // splitting stopped paying, or the launch is long anyway
            }
        }
        if (rc != W2L_OK) break;
        if (best_tile < 0) { set_error("plan_autotune: no configuration id resolves to itself for a recorded launch"); rc = W2L_ERR_ARG; break; }
        it.tile = best_tile;
        it.ksplit = best_ks;
        // later launches of this shape through ANY handle / plan of this process replay the same choice (w2l_tune_export
        // turns a tuning session into the committed table)
        tune_store_launch(it.c, it.N, it.H, it.W, it.res != nullptr, best_tile, best_ks);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int w2l_plan_get_config(const w2l_plan_t* p, int index, int* tile, int* ksplit) {
    W2L_REQUIRE(p && tile && ksplit && index >= 0 && index < (int)p->items.size(), "bad plan_get_config arguments");
    *tile = p->items[index].tile;
    *ksplit = p->items[index].ksplit;
    return W2L_OK;
}

int w2l_plan_set_config(w2l_plan_t* p, int index, int tile, int ksplit) {
    W2L_REQUIRE(p && index >= 0 && index < (int)p->items.size(), "bad plan_set_config arguments");
    W2L_REQUIRE(tile >= -1 && tile < conv_num_tiles() && ksplit >= 1 && ksplit <= 64, "bad config (%d, %d)", tile, ksplit);
    p->items[index].tile = tile;
    p->items[index].ksplit = ksplit;
    return W2L_OK;
}

int w2l_plan_profile(const w2l_plan_t* p, void* stream, int reps, float* ms_out) {
    W2L_REQUIRE(p && ms_out && reps >= 1, "bad plan_profile arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = p->items.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (size_t i = 0; i <= n; ++i) W2L_HIP_CHECK(hipEventCreate(&ev[i]));
    for (size_t i = 0; i < n; ++i) ms_out[i] = 0.f;
    int rc = W2L_OK;
    for (int r = 0; r < reps && rc == W2L_OK; ++r) {
        (void)hipEventRecord(ev[0], s);
        for (size_t i = 0; i < n && rc == W2L_OK; ++i) {
            const PlanItem& it = p->items[i];
            rc = conv_forward_impl(it.c, s, it.N, it.H, it.W, it.x, it.x_cs, it.y, it.y_cs, it.res, it.res_cs, it.tile, it.ksplit, nullptr, nullptr);
            (void)hipEventRecord(ev[i + 1], s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) { set_error("sync failed in plan_profile"); rc = W2L_ERR_HIP; }
        for (size_t i = 0; i < n && rc == W2L_OK; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            ms_out[i] += ms / reps;
        }
    }
    for (size_t i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

}  // extern "C"

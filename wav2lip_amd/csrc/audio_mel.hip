// audio.melspectrogram on gfx950: one workgroup per STFT frame, everything fused:
//   pre-emphasis (f64) -> reflect pad -> periodic-Hann window (f64) -> 800-point real DFT (f64, direct, twiddle
//   table in LDS) -> cast to complex64 -> |.| (f32) -> 80-bin Slaney mel (f32) -> 20*log10(max(1e-5,.)) - 20
//   -> clip(8*((S+100)/100) - 4, -4, 4).
// Follows audio.py:45-51 (melspectrogram), :20-23 (preemphasis = scipy lfilter([1,-k],[1])), :57-61 (_stft =
// librosa.stft(n_fft=800, hop=200, win=800): center=True, reflect pad, f64 FFT stored as complex64), :92-105,
// :110-116 of the reference with hparams.py:32-69.  The numerics mirror oracle/audio_ref.py step by step.
// HBM-bound in principle (200 new samples in, 80 floats out per frame); at clip lengths the launch is latency-bound.
#include <math.h>

#include <new>
#include <vector>

#include "w2l_common.h"

namespace w2l {

constexpr int kNfft = 800;
constexpr int kHop = 200;
constexpr int kBins = kNfft / 2 + 1;  // 401
constexpr int kMels = 80;
constexpr double kPreemph = 0.97;

struct MelKArgs {
    const float* wav;
    long long nsamples;
    const float* basis;    // [80][401]
    const double* window;  // [800]
    const double2* twiddle;  // [800] (cos, sin)(2*pi*j/800)
    float* mel;            // [80][T]
    int T;
};

__device__ __forceinline__ double preemph_sample(const float* wav, long long j) {
    // scipy lfilter (direct form II transposed), b = [1, -k]: y[n] = x[n] + (-k * x[n-1]), products and sums
    // rounded separately in f64 (no fused multiply-add)
    const double x0 = (double)wav[j];
    if (j == 0) return x0;
    return __dadd_rn(x0, __dmul_rn(-kPreemph, (double)wav[j - 1]));
}

__global__ __launch_bounds__(256) void mel_frame_kernel(const MelKArgs a) {
    __shared__ double s_frame[kNfft];
    __shared__ double2 s_tw[kNfft];
    __shared__ float s_mag[kBins + 3];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const long long pad = kNfft / 2;

    for (int n = tid; n < kNfft; n += 256) {
        long long j = (long long)t * kHop + n - pad;
        if (j < 0) j = -j;                                   // np.pad(mode='reflect'): no edge repeat
        if (j >= a.nsamples) j = 2 * (a.nsamples - 1) - j;
        s_frame[n] = __dmul_rn(a.window[n], preemph_sample(a.wav, j));
        s_tw[n] = a.twiddle[n];
    }
    __syncthreads();

    for (int k = tid; k < kBins; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;  // (k*n) mod 800
        for (int n = 0; n < kNfft; ++n) {
            const double2 w = s_tw[idx];
            const double v = s_frame[n];
            re = fma(v, w.x, re);
            im = fma(-v, w.y, im);
            idx += k;
            if (idx >= kNfft) idx -= kNfft;
        }
        // complex64 storage, then np.abs in f32
        s_mag[k] = hypotf((float)re, (float)im);
    }
    __syncthreads();

    if (tid < kMels) {
        const float* b = a.basis + tid * kBins;
        float acc = 0.f;
        for (int k = 0; k < kBins; ++k) acc = fmaf(b[k], s_mag[k], acc);
        // _amp_to_db: min_level = exp(-100/20*ln 10) = 1e-5 ; 20*log10(max(min_level, x)) ; then - ref_level_db (20)
        const float min_level = 1e-5f;
        float S = 20.0f * log10f(fmaxf(min_level, acc)) - 20.0f;
        // _normalize, symmetric + clipping: clip(2*4*((S+100)/100) - 4, -4, 4)
        float v = 8.0f * ((S + 100.0f) / 100.0f) - 4.0f;
        v = fminf(fmaxf(v, -4.0f), 4.0f);
        a.mel[(long long)tid * a.T + t] = v;
    }
}

__global__ void mel_gather_kernel(const float* __restrict__ mel, int T, const int* __restrict__ starts, int B,
                                  float* __restrict__ out, int out_cs, int c_zero_to) {
    const long long total = (long long)B * kMels * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const int m = (int)((i >> 4) % kMels);
        const int b = (int)(i / (16 * kMels));
        const int s = starts[b] + j;
        float* o = out + i * out_cs;
        o[0] = (s >= 0 && s < T) ? mel[(long long)m * T + s] : 0.f;
        for (int c = 1; c < c_zero_to; ++c) o[c] = 0.f;
    }
}

}  // namespace w2l

using namespace w2l;

struct w2l_mel {
    float* basis = nullptr;
    double* window = nullptr;
    double2* twiddle = nullptr;
};

// ---------------------------------------------------------------- sample-rate conversion (audio.py:9-10 -> librosa.load)
// resampy's band-limited sinc interpolation (resample_f): one thread per output sample walks the left wing (x[n], x[n-1], ...)
// and then the right wing (x[n+1], ...) of the Kaiser-windowed sinc, in that order, rounding the float32 accumulator after
// every term exactly as the reference's numba loop does (y is float32, the weights float64).  __dmul_rn / __dadd_rn keep the
// compiler from contracting the products into FMAs (numpy rounds the product, then the sum).
struct ResampleKArgs {
    const float* x;
    const double* tr;     // time register of every output sample (host: repeated float64 addition)
    const double* win;    // half window [nwin] (already scaled by the ratio when downsampling)
    const double* delta;  // its first differences, last entry 0
    float* y;
    int n_in, n_out, nwin, num_table, index_step;
    double scale;
};

__global__ void resample_sinc_kernel(const ResampleKArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_out) return;
    const double time_register = a.tr[t];
    const int n = (int)time_register;
    float acc = 0.f;
    double frac = __dmul_rn(a.scale, time_register - (double)n);
    double index_frac = __dmul_rn(frac, (double)a.num_table);
    int offset = (int)index_frac;
    double eta = index_frac - (double)offset;
    const int i_max = min(n + 1, (a.nwin - offset) / a.index_step);
    for (int i = 0; i < i_max; ++i) {
        const int idx = offset + i * a.index_step;
        const double w = __dadd_rn(a.win[idx], __dmul_rn(eta, a.delta[idx]));
        acc = (float)__dadd_rn((double)acc, __dmul_rn(w, (double)a.x[n - i]));
    }
    frac = a.scale - frac;
    index_frac = __dmul_rn(frac, (double)a.num_table);
    offset = (int)index_frac;
    eta = index_frac - (double)offset;
    const int k_max = min(a.n_in - n - 1, (a.nwin - offset) / a.index_step);
    for (int k = 0; k < k_max; ++k) {
        const int idx = offset + k * a.index_step;
        const double w = __dadd_rn(a.win[idx], __dmul_rn(eta, a.delta[idx]));
        acc = (float)__dadd_rn((double)acc, __dmul_rn(w, (double)a.x[n + k + 1]));
    }
    a.y[t] = acc;
}

extern "C" {

int w2l_mel_num_frames(long long nsamples) { return (int)(1 + nsamples / kHop); }

int w2l_mel_create(const float* mel_basis_host, const double* window_host, w2l_mel_t** out) {
    W2L_REQUIRE(mel_basis_host && window_host && out, "NULL argument");
    w2l_mel* m = new (std::nothrow) w2l_mel();
    if (!m) { set_error("out of host memory"); return W2L_ERR_NOMEM; }
    std::vector<double2> tw(kNfft);
    for (int j = 0; j < kNfft; ++j) {
        const double ang = 2.0 * M_PI * (double)j / (double)kNfft;
        tw[j].x = cos(ang);
        tw[j].y = sin(ang);
    }
    int rc = W2L_OK;
    if (hipMalloc(&m->basis, sizeof(float) * kMels * kBins) != hipSuccess ||
        hipMalloc(&m->window, sizeof(double) * kNfft) != hipSuccess ||
        hipMalloc(&m->twiddle, sizeof(double2) * kNfft) != hipSuccess) {
        set_error("hipMalloc failed in w2l_mel_create");
        rc = W2L_ERR_NOMEM;
    } else if (hipMemcpy(m->basis, mel_basis_host, sizeof(float) * kMels * kBins, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(m->window, window_host, sizeof(double) * kNfft, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(m->twiddle, tw.data(), sizeof(double2) * kNfft, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy failed in w2l_mel_create");
        rc = W2L_ERR_HIP;
    }
    if (rc != W2L_OK) { w2l_mel_destroy(m); return rc; }
    *out = m;
    return W2L_OK;
}

int w2l_mel_destroy(w2l_mel_t* m) {
    if (!m) return W2L_OK;
    if (m->basis) (void)hipFree(m->basis);
    if (m->window) (void)hipFree(m->window);
    if (m->twiddle) (void)hipFree(m->twiddle);
    delete m;
    return W2L_OK;
}

int w2l_melspectrogram(const w2l_mel_t* m, void* stream, const float* wav, long long nsamples, float* mel) {
    W2L_REQUIRE(m && wav && mel, "NULL argument");
    W2L_REQUIRE(nsamples > kNfft / 2, "need more than %d samples for reflect padding (got %lld)", kNfft / 2, nsamples);
    MelKArgs a;
    a.wav = wav; a.nsamples = nsamples; a.basis = m->basis; a.window = m->window; a.twiddle = m->twiddle;
    a.mel = mel; a.T = w2l_mel_num_frames(nsamples);
    hipLaunchKernelGGL(mel_frame_kernel, dim3(a.T), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_mel_gather(void* stream, const float* mel, int T, const int32_t* starts, int B, float* out, int out_cs,
                   int c_zero_to) {
    W2L_REQUIRE(mel && starts && out && T >= 16 && B >= 1, "bad mel_gather arguments");
    if (c_zero_to < 1) c_zero_to = 1;
    W2L_REQUIRE(out_cs >= c_zero_to, "out_cs=%d < %d", out_cs, c_zero_to);
    const long long total = (long long)B * kMels * 16;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(mel_gather_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), mel, T,
                       starts, B, out, out_cs, c_zero_to);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

int w2l_resample_sinc(void* stream, const float* x, int n_in, const double* tr, int n_out, double sample_ratio,
                      const double* win, const double* delta, int nwin, int num_table, float* y) {
    W2L_REQUIRE(x && tr && win && delta && y, "NULL argument");
    W2L_REQUIRE(n_in >= 1 && n_out >= 1 && nwin >= 2 && num_table >= 1 && sample_ratio > 0.0, "bad resample arguments");
    ResampleKArgs a;
    a.x = x; a.tr = tr; a.win = win; a.delta = delta; a.y = y;
    a.n_in = n_in; a.n_out = n_out; a.nwin = nwin; a.num_table = num_table;
    a.scale = sample_ratio < 1.0 ? sample_ratio : 1.0;
    a.index_step = (int)(a.scale * num_table);
    W2L_REQUIRE(a.index_step >= 1, "sample ratio %g too small for a table of %d samples per zero crossing", sample_ratio, num_table);
    hipLaunchKernelGGL(resample_sinc_kernel, dim3((unsigned)((n_out + 127) / 128)), dim3(128), 0, static_cast<hipStream_t>(stream), a);
    W2L_HIP_CHECK(hipGetLastError());
    return W2L_OK;
}

}  // extern "C"

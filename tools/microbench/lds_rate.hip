// LDS instruction throughput on gfx950, per CU, with 8 waves resident (two 256-thread workgroups per CU on every CU): bytes per
// shader clock for ds_read_b64, ds_read_b128, ds_read_b64_tr_b16 (conflict-free 64-byte-row pattern of wgrad_bf16.hip) and
// ds_write_b32 / b64 / b128.  Prints plain text.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/bin/lds_rate tools/microbench/lds_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define AS3 __attribute__((address_space(3)))

constexpr int kIters = 2048;

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel(unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    // per-lane base inside a 16 KB window of this wave
    char* base = lds + wave * 16384;
    unsigned addr;
    if (MODE == 2) {   // tr-read: lane supplies pixel sub-row ((lane & 15) >> 2) + 8 * (lane >> 5), 8 bytes at channel group
        const int pix = 8 * (lane >> 5) + ((lane & 15) >> 2), ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        addr = (unsigned)(size_t)(AS3 char*)(base + pix * 64 + ch * 2);
    } else if (MODE == 0 || MODE == 4) {
        addr = (unsigned)(size_t)(AS3 char*)(base + lane * 8);
    } else if (MODE == 3) {
        addr = (unsigned)(size_t)(AS3 char*)(base + lane * 4);
    } else {
        addr = (unsigned)(size_t)(AS3 char*)(base + lane * 16);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < kIters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned a = addr + ((i + u) & 7) * 1024;
            if (MODE == 0) {
                f32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); acc.x += 0.f * v.x;
            } else if (MODE == 1) {
                f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); acc.x += 0.f * v.x;
            } else if (MODE == 2) {
                f32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a)); acc.x += 0.f * v.x;
            } else if (MODE == 3) {
                asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(acc.y) : "memory");
            } else if (MODE == 4) {
                f32x2 v = {acc.y, acc.z}; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory");
            } else {
                asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(acc) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (t == 0) out[blockIdx.x] = c1 - c0;
    if (acc.x == 12345.f) sink[0] = acc.x;
}

template <int MODE>
static void run(const char* name, int bytes_per_lane) {
    unsigned long long* out;
    float* sink;
    const int nwg = 512;
    CK(hipMalloc(&out, nwg * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(nwg), dim3(256), 0, 0, out, sink);
    CK(hipDeviceSynchronize());
    unsigned long long h[512];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < nwg; ++i) avg += (double)h[i];
    avg /= nwg;
    // two workgroups (8 waves) share a CU: bytes moved per CU while one workgroup runs its loop = 2 x 4 waves x 64 lanes x ...
    const double bytes = 2.0 * 4 * 64 * (double)bytes_per_lane * kIters;
    printf("%-22s %8.0f cycles per %d instructions per wave -> %.1f B/clk/CU (8 waves resident)\n", name, avg, kIters, bytes / avg);
    CK(hipFree(out));
    CK(hipFree(sink));
}

int main() {
    run<0>("ds_read_b64", 8);
    run<1>("ds_read_b128", 16);
    run<2>("ds_read_b64_tr_b16", 8);
    run<3>("ds_write_b32", 4);
    run<4>("ds_write_b64", 8);
    run<5>("ds_write_b128", 16);
    return 0;
}

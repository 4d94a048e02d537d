// Micro-benchmark (gfx950): does a wave's VALU / LDS work overlap with MFMAs in flight — its own, or those of the other wave
// on the same SIMD?  Each wave loops over { v_mfma_f32_32x32x2_f32 on one of 8 accumulators ; NV independent v_fma_f32 ;
// NL ds_read_b32 }.  Reported: wall-clock cycles (at the measured clock) per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap tools/microbench/mfma_overlap.hip && ./mfma_overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NL, int NP = 0, int NS = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    __shared__ float lds[4096];
    const int t = threadIdx.x;
    lds[t] = (float)t;
    lds[t + 512] = 1.0f;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float x[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    float a = (float)(t & 7), b = 0.5f, c1 = 1.0001f, c2 = 0.0001f;
    float l[4] = {0.f, 0.f, 0.f, 0.f};
    f32x2 xp[8], cp1 = {1.0001f, 1.0001f}, cp2 = {0.0001f, 0.0001f};
    for (int v = 0; v < 8; ++v) xp[v] = f32x2{(float)v, (float)t};
    unsigned sx[4] = {0, 1, 2, 3};
    const float* lp = lds + (t & 63);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c1), "v"(c2));
#pragma unroll
            for (int v = 0; v < NP; ++v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[v & 7]) : "v"(cp1), "v"(cp2));
#pragma unroll
            for (int v = 0; v < NS; ++v) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx[v & 3]));
#pragma unroll
            for (int v = 0; v < NL; ++v) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(l[v & 3]) : "v"((int)(size_t)lp * 0 + (t & 63) * 4), "n"(256 * (v + 1)));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][r];
    for (int v = 0; v < 8; ++v) s += x[v];
    for (int v = 0; v < 4; ++v) s += l[v] + (float)sx[v];
    for (int v = 0; v < 8; ++v) s += xp[v].x + xp[v].y;
    out[blockIdx.x * blockDim.x + t] = s;
}

template <int NV, int NL, int NP = 0, int NS = 0>
static void run(float* out, int threads, double ghz) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NL, NP, NS>), dim3(grid), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NL, NP, NS>), dim3(grid), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = iters * 8.0 * waves_per_simd;
    printf("waves/SIMD %.0f  VALU/MFMA %2d  pk_fma/MFMA %2d  SALU/MFMA %2d  ds_read/MFMA %d : %7.1f cycles per MFMA per SIMD (%.3f ms)\n", waves_per_simd, NV, NP, NS, NL,
           ms * 1e-3 * ghz * 1e9 / mfma_per_simd, ms);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz (a 32x32x2 fp32 MFMA occupies the matrix pipe for 64 cycles)\n", ghz);
    for (int threads : {256, 512}) {
        run<0, 0>(out, threads, ghz);
        run<4, 0>(out, threads, ghz);
        run<8, 0>(out, threads, ghz);
        run<12, 0>(out, threads, ghz);
        run<16, 0>(out, threads, ghz);
        run<24, 0>(out, threads, ghz);
        run<32, 0>(out, threads, ghz);
        run<0, 2>(out, threads, ghz);
        run<8, 2>(out, threads, ghz);
        run<0, 4>(out, threads, ghz);
        run<0, 0, 4>(out, threads, ghz);
        run<0, 0, 8>(out, threads, ghz);
        run<0, 0, 16>(out, threads, ghz);
        run<0, 0, 0, 8>(out, threads, ghz);
        run<0, 0, 0, 16>(out, threads, ghz);
    }
    return 0;
}

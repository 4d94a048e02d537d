// Micro-benchmark (gfx950), bf16 companion of mfma_overlap.hip: what does vector / LDS work cost next to v_mfma_f32_32x32x16_bf16
// (8 passes = 32 cycles of the matrix pipe)?  Each wave loops over { one MFMA on one of 4 accumulators ; NV independent
// v_fma_f32 ; NC v_cvt_pk_bf16_f32 ; NR ds_read_b128 ; NW ds_write_b64 }.  Reported: cycles per MFMA per SIMD at 1, 2, 4 waves per
// SIMD.  The split-operand implicit GEMM carries ~5 VALU, 0.75 ds_read_b128 and 0.4 LDS stores per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap_bf16 tools/microbench/mfma_overlap_bf16.hip && ./mfma_overlap_bf16
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NC, int NR, int NW, bool FRESH = true>
__global__ __launch_bounds__(1024, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int t = threadIdx.x;
    for (int i = t; i < 16384; i += blockDim.x) lds[i] = (float)(i & 255);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)((t + e) & 3); b[e] = (__bf16)0.5f; }
    float x[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    unsigned cv[4] = {0, 0, 0, 0};
    float c1 = 1.0001f, c2 = 0.0001f;
    double old = 3.0 + t;                                   // FRESH = false: the stores send a value no instruction of the loop writes
    f32x4 l[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const int rd = (t & 63) * 16 + (t >> 6) * 1024;        // conflict-free 16-byte reads, one KB per wave
    const int wr = 32768 + (t & 63) * 8 + (t >> 6) * 512;  // conflict-free 8-byte stores
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[p], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c1), "v"(c2));
#pragma unroll
            for (int v = 0; v < NC; ++v) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[v & 3]) : "v"(x[v & 7]), "v"(x[(v + 1) & 7]));
#pragma unroll
            for (int v = 0; v < NR; ++v) asm volatile("ds_read_b128 %0, %1" : "=v"(l[v & 1]) : "v"(rd));
#pragma unroll
            for (int v = 0; v < NW; ++v) {
                if (FRESH) asm volatile("ds_write_b64 %0, %1" : : "v"(wr), "v"(*reinterpret_cast<double*>(&x[0])) : "memory");
                else asm volatile("ds_write_b64 %0, %1" : : "v"(wr), "v"(old) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NR || NW) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][r];
    for (int v = 0; v < 8; ++v) s += x[v];
    for (int v = 0; v < 4; ++v) s += (float)cv[v];
    s += l[0][0] + l[1][1] + (float)old;
    out[blockIdx.x * blockDim.x + t] = s;
}

template <int NV, int NC, int NR, int NW, bool FRESH = true>
static void run(float* out, int threads, double ghz) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NC, NR, NW, FRESH>), dim3(grid), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NC, NR, NW, FRESH>), dim3(grid), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = iters * 4.0 * waves_per_simd;
    printf("waves/SIMD %.0f  v_fma %2d  cvt_pk %2d  ds_read_b128 %d  ds_write_b64 %d%s : %7.1f cycles per MFMA per SIMD (%.3f ms)\n",
           waves_per_simd, NV, NC, NR, NW, NW ? (FRESH ? " (fresh value)" : " (old value)  ") : "", ms * 1e-3 * ghz * 1e9 / mfma_per_simd, ms);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz (a 32x32x16 bf16 MFMA occupies the matrix pipe for 32 cycles)\n", ghz);
    for (int threads : {256, 512, 1024}) {
        run<0, 0, 0, 0>(out, threads, ghz);
        run<2, 0, 0, 0>(out, threads, ghz);
        run<4, 0, 0, 0>(out, threads, ghz);
        run<6, 0, 0, 0>(out, threads, ghz);
        run<8, 0, 0, 0>(out, threads, ghz);
        run<12, 0, 0, 0>(out, threads, ghz);
        run<0, 4, 0, 0>(out, threads, ghz);
        run<4, 2, 0, 0>(out, threads, ghz);
        run<0, 0, 1, 0>(out, threads, ghz);
        run<0, 0, 2, 0>(out, threads, ghz);
        run<0, 0, 0, 1>(out, threads, ghz);
        run<4, 1, 1, 1>(out, threads, ghz);
        run<4, 1, 1, 1, false>(out, threads, ghz);
        run<4, 1, 0, 1>(out, threads, ghz);
        run<4, 1, 0, 1, false>(out, threads, ghz);
        run<4, 1, 1, 0>(out, threads, ghz);
        run<6, 0, 0, 1, false>(out, threads, ghz);
    }
    return 0;
}

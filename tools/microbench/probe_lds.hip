// Probes (gfx950) for the questions the bf16 training kernels rest on; prints plain text, no pass/fail:
//   P1  lane <-> element mapping of ds_read_b64_tr_b16 (which LDS halfword lands in which lane / slot)
//   P2  buffer_load_dwordx4 ... lds (LDS-DMA): destination = M0 base + lane*16 ? what do out-of-range lanes write ?
//   P3  ds_read_b128 from addresses that are only 2-/4-/8-byte aligned: correct ? how much slower ?
//   P4  A/B fragment K order of v_mfma_f32_32x32x16_bf16 (lane l: row l&31, k = 8*(l>>5)+j)
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/bin/probe_lds tools/microbench/probe_lds.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- P1
__global__ void p1_kernel(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<char*>(lds) + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

// ---- P2
__global__ void p2_kernel(const float* src, int src_bytes, const int* voff, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -7.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64), 16,
                                             voff[threadIdx.x], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

// ---- P3
template <int MIS>
__global__ void p3_kernel(float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // rows of 144 B (conflict-free for aligned b128 reads), lane reads 16 B at row lane&31, +16*(lane>>5), + MIS bytes
    const char* base = reinterpret_cast<const char*>(lds) + (lane & 31) * 144 + (lane >> 5) * 16 + MIS;
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32x4 v;
            const unsigned a = (unsigned)(size_t)(base + u * 144 * 32 / 8 * 0 + ((it + u) & 3) * 4608);
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
    // first read's content for the correctness print
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        u32x4 v;
        const unsigned a = (unsigned)(size_t)base;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        unsigned* o = reinterpret_cast<unsigned*>(out + 65536) + threadIdx.x * 4;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
}

// ---- P4: D = A * B with A[i][k] = (i == 3 ? code(k) : 0), B[k][j] = (j == 5 ? 1 : 0)  ->  D[3][5] = sum_k code(k); with
// code(k) = 2^k-ish distinct values we instead multiply one-hot: for each kk, A[i][k] = (i==3 && k==kk), B[k][j] = (k==kk2 && j==5)
__global__ void p4_kernel(float* out) {
    const int lane = threadIdx.x;
    for (int kk = 0; kk < 16; ++kk) {
        for (int kk2 = 0; kk2 < 16; ++kk2) {
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * (lane >> 5) + j;
                a[j] = (__bf16)(((lane & 31) == 3 && k == kk) ? 1.0f : 0.0f);
                b[j] = (__bf16)(((lane & 31) == 5 && k == kk2) ? 1.0f : 0.0f);
            }
            f32x16 c;
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
            // D[3][5]: col = lane&31 = 5, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = 3 -> lane 5, r = 3
            if (lane == 5) out[kk * 16 + kk2] = c[3];
        }
    }
}

int main() {
    // ---------------- P1
    {
        int* d_addr; unsigned short* d_out;
        CK(hipMalloc(&d_addr, 64 * 4)); CK(hipMalloc(&d_out, 64 * 4 * 2));
        const char* names[3] = {"addr = lane*8 (contiguous)", "addr = (lane&15)*64 + (lane>>4)*8 (16 rows of 64 B per 16-lane group)",
                                "addr = (lane&3)*64 + ((lane>>2)&3)*8 + (lane>>4)*512"};
        for (int pat = 0; pat < 3; ++pat) {
            std::vector<int> addr(64);
            for (int l = 0; l < 64; ++l)
                addr[l] = pat == 0 ? l * 8 : pat == 1 ? (l & 15) * 64 + (l >> 4) * 8 : (l & 3) * 64 + ((l >> 2) & 3) * 8 + (l >> 4) * 512;
            CK(hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(p1_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_out);
            CK(hipDeviceSynchronize());
            std::vector<unsigned short> o(256);
            CK(hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost));
            printf("P1 pattern %d: %s\n  lane: byte addr -> halfword indices received (each lane's own 4 halfwords are addr/2 .. addr/2+3)\n", pat, names[pat]);
            for (int l = 0; l < 64; ++l)
                printf("  %2d: %4d -> %4d %4d %4d %4d\n", l, addr[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
        }
    }
    // ---------------- P2
    {
        std::vector<float> src(4096);
        for (int i = 0; i < 4096; ++i) src[i] = (float)i;
        float* d_src; int* d_voff; float* d_out;
        CK(hipMalloc(&d_src, 4096 * 4)); CK(hipMalloc(&d_voff, 256)); CK(hipMalloc(&d_out, 4096));
        CK(hipMemcpy(d_src, src.data(), 4096 * 4, hipMemcpyHostToDevice));
        std::vector<int> voff(64);
        for (int l = 0; l < 64; ++l) voff[l] = (l % 5 == 4) ? (int)0x80000000u : ((63 - l) * 64 + 16);  // reversed rows, some OOB
        CK(hipMemcpy(d_voff, voff.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p2_kernel, dim3(1), dim3(64), 0, 0, d_src, 4096 * 4, d_voff, d_out);
        CK(hipDeviceSynchronize());
        std::vector<float> o(1024);
        CK(hipMemcpy(o.data(), d_out, 4096, hipMemcpyDeviceToHost));
        printf("P2 LDS-DMA b128: lds prefilled -7; dma base = float 64; lane l source float index (63-l)*16+4 (l%%5==4: out of range)\n");
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d: lds[%4d..] = %6.0f %6.0f %6.0f %6.0f   (expect %d.. or OOB)\n", l, 64 + l * 4, o[64 + l * 4], o[65 + l * 4],
                   o[66 + l * 4], o[67 + l * 4], (63 - l) * 16 + 4);
        printf("  untouched guard: lds[60..63] = %g %g %g %g, lds[320..323] = %g %g %g %g\n", o[60], o[61], o[62], o[63], o[320], o[321], o[322], o[323]);
    }
    // ---------------- P3
    {
        float* d_out; long long* d_cyc;
        CK(hipMalloc(&d_out, (65536 + 1024) * 4)); CK(hipMalloc(&d_cyc, 8));
        const int iters = 2000;
        for (int mis = 0; mis < 4; ++mis) {
            const int M[4] = {0, 2, 4, 8};
            for (int rep = 0; rep < 2; ++rep) {
                switch (mis) {
                    case 0: hipLaunchKernelGGL(p3_kernel<0>, dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc); break;
                    case 1: hipLaunchKernelGGL(p3_kernel<2>, dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc); break;
                    case 2: hipLaunchKernelGGL(p3_kernel<4>, dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc); break;
                    default: hipLaunchKernelGGL(p3_kernel<8>, dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc); break;
                }
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) { printf("P3 misalign %d: launch failed: %s\n", M[mis], hipGetErrorString(e)); break; }
            }
            long long cyc; std::vector<unsigned> o(256);
            CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(o.data(), d_out + 65536, 1024, hipMemcpyDeviceToHost));
            // expected first dword of lane l: halfwords at byte (l&31)*144 + (l>>5)*16 + MIS
            int bad = 0;
            for (int l = 0; l < 64; ++l) {
                const int hb = ((l & 31) * 144 + (l >> 5) * 16 + M[mis]) / 2;
                for (int d = 0; d < 4; ++d) {
                    const unsigned want = (unsigned)(hb + 2 * d) | ((unsigned)(hb + 2 * d + 1) << 16);
                    if (o[l * 4 + d] != want) ++bad;
                }
            }
            printf("P3 ds_read_b128 misaligned by %d B: %d wrong dwords of 256; %.1f wave-clock cycles per dependent read (4 waves per CU)\n",
                   M[mis], bad, (double)cyc / (iters * 8));
        }
    }
    // ---------------- P4
    {
        float* d_out; CK(hipMalloc(&d_out, 1024));
        hipLaunchKernelGGL(p4_kernel, dim3(1), dim3(64), 0, 0, d_out);
        CK(hipDeviceSynchronize());
        std::vector<float> o(256);
        CK(hipMemcpy(o.data(), d_out, 1024, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) if (o[a * 16 + b] != (a == b ? 1.f : 0.f)) ok = 0;
        printf("P4 mfma_f32_32x32x16_bf16 operand K order: lane l holds row l&31, k = 8*(l>>5)+j for BOTH operands: %s\n", ok ? "confirmed" : "NOT as assumed");
    }
    return 0;
}

// lane map of v_permlane32_swap_b32 on gfx950 (probed before conv_box_bf16.hip's epilogue used it):
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/permlane32_swap.hip -o /tmp/pl && /tmp/pl
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 1000 + lane, b = 2000 + lane;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned* d;
    unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("first  (a = 1000 + lane): lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[0], h[31], h[32], h[63]);
    printf("second (b = 2000 + lane): lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[64], h[95], h[96], h[127]);
    return 0;
}

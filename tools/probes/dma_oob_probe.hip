// Probe: what does an LDS-DMA (buffer_load_dwordx4 ... lds) leave in LDS for lanes whose buffer offset is out of range?
// (and for lanes that are masked off by EXEC)   Build: hipcc --offload-arch=gfx950 -O3 dma_oob_probe.hip -o dma_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const float *src, float *out, int n) {
    __shared__ float lds[3 * 256];
    for (int i = threadIdx.x; i < 768; i += 64) lds[i] = 7.f;
    __syncthreads();
    rsrc_t S = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    // piece 0: odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(S, (__attribute__((address_space(3))) void *)lds, 16, (lane & 1) ? 0x7ffffff0 : lane * 16, 0, 0, 0);
    // piece 1: odd lanes masked by EXEC
    if (!(lane & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(S, (__attribute__((address_space(3))) void *)(lds + 256), 16, lane * 16, 0, 0, 0);
    // piece 2: ALL lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(S, (__attribute__((address_space(3))) void *)(lds + 512), 16, 0x7ffffff0, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out; hipMalloc(&src, 1024 * 4); hipMalloc(&out, 768 * 4);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(src, out, 1024); float o[768]; hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    for (int p = 0; p < 3; ++p) { printf("piece %d:", p); for (int i = 0; i < 16; ++i) printf(" %g", o[p * 256 + i]); printf("\n"); }
    return 0;
}

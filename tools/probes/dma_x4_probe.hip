// Probe: does buffer_load_dwordx4 ... lds accept 4-byte-aligned (not 16-byte-aligned) global and LDS addresses?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const float* src, float* out, int n, int goff, int loff) {
  __shared__ float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1.f;
  __syncthreads();
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n*4, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + loff), 16, (int)(threadIdx.x*16 + goff*4), 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  const int n = 4096; std::vector<float> h(n); for (int i=0;i<n;++i) h[i]=(float)i;
  float *d, *o; hipMalloc(&d, n*4); hipMalloc(&o, 512*4); hipMemcpy(d, h.data(), n*4, hipMemcpyHostToDevice);
  for (int goff = 0; goff < 4; ++goff) for (int loff = 0; loff < 4; ++loff) {
    k<<<1,64>>>(d, o, n, goff, loff);
    std::vector<float> r(512); hipMemcpy(r.data(), o, 512*4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i=0;i<256;++i) if (r[loff+i] != (float)(goff+i)) ++bad;
    printf("goff=%d loff=%d bad=%d first=%g %g %g %g %g\n", goff, loff, bad, r[0], r[1], r[2], r[3], r[4]);
  }
  return 0;
}

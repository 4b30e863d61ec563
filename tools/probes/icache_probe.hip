// Probe for the pool's slow boxes: does the cost of INSTRUCTION FETCH differ between boxes?
// Round 5 eliminated clocks, the power cap, the matrix rate, HBM -> LDS and L2 -> LDS tile fills (any stride) and the copy rate; what
// the slow box loses orders by (code size x workgroup churn): the launches of big unrolled kernels made of many short-lived
// workgroups lose 30-50 %, loops in long-lived workgroups 10-15 %, tiny streaming kernels 4 %.  Two kernels do the SAME arithmetic
// (~8080 dependent fmas per thread): `big` as straight-line instructions with distinct constants (~70 KB of code: more than
// the 64 KB instruction cache a pair of CUs shares, executed once per short-lived workgroup), `small` as
// a 16-instruction loop body.  Their time ratio is the price of fetching code; a box whose ratio is far above the pool's is slow there.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 icache_probe.hip -o icache_probe    (prints us per launch and the ratio)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int I> struct Step {
    __device__ static __forceinline__ float run(float v, float s) {
        v = __builtin_fmaf(v, 1.0f + (float)(I % 97) * 1.0e-6f, s * (float)((I * 7) % 13));
        return Step<I - 1>::run(v, s);
    }
};
template <> struct Step<0> { __device__ static __forceinline__ float run(float v, float) { return v; } };

// ~8080 straight-line fmas in 16 chunks of ~505 (template depth)
__global__ __launch_bounds__(256) void big(float *out, float s) {
    float v = (float)threadIdx.x;
    v = Step<512>::run(v, s); v = Step<511>::run(v, s + 1.f); v = Step<510>::run(v, s + 2.f); v = Step<509>::run(v, s + 3.f);
    v = Step<508>::run(v, s + 4.f); v = Step<507>::run(v, s + 5.f); v = Step<506>::run(v, s + 6.f); v = Step<505>::run(v, s + 7.f);
    v = Step<504>::run(v, s + 8.f); v = Step<503>::run(v, s + 9.f); v = Step<502>::run(v, s + 10.f); v = Step<501>::run(v, s + 11.f);
    v = Step<500>::run(v, s + 12.f); v = Step<499>::run(v, s + 13.f); v = Step<498>::run(v, s + 14.f); v = Step<497>::run(v, s + 15.f);
    if (v == 12345.678f) out[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void small(float *out, float s) {
    float v = (float)threadIdx.x;
    for (int it = 0; it < 505; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v = __builtin_fmaf(v, 1.0f + (float)k * 1.0e-6f, s * (float)k);
    }
    if (v == 12345.678f) out[blockIdx.x] = v;
}
template <typename K> static float time_us(K kern, float *out, int grid, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 0.5f);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main() {
    float *out; hipMalloc(&out, 1 << 20);
    for (int grid : {256, 3104, 24832}) {
        const float tb = time_us(big, out, grid, 50), ts = time_us(small, out, grid, 50);
        printf("grid %6d workgroups of 256 threads, ~8080 fmas per thread: straight-line code %8.1f us, 16-instruction loop %8.1f us, ratio %.2f\n",
               grid, tb, ts, tb / ts);
    }
    return 0;
}

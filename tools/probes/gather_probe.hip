// Probe for the pool's slow boxes: the ATTENTION-BLOCK PROLOGUE of the strip kernels in isolation.
// Every gmap / gmap3 workgroup starts by loading its strip's L x L attention block straight into MFMA fragments: lane (ln, lg) of
// a wavefront reads 16 bytes at (row 16 t + ln, column 32 ks + 8 lg [+ 4]) -- 16 DIFFERENT rows per load instruction, 32 bytes of
// each -- and the transposed jobs (dk, dv) gather 4 bytes per lane from 64 different rows.  Round 5 ruled out everything else that
// was measurable on a slow box (clock, power, matrix rate, LDS-DMA fills from HBM and from L2, address translation, copy); these
// per-lane, non-coalescing register loads are what its worst launches (+30 .. 49 %) have most of.  Three ways to bring the same
// 97 x 97 fp32 block (37 KB) of a column strip into registers, 768 workgroups x `reps` strips each, out of the 58 MB attention tensor
// of the headline shape: `frag16` (the kernels' pattern), `gather4` (the transposed jobs' pattern), `rows` (whole rows, coalesced:
// what an LDS-staged prologue would issue).  Prints us per block and GB/s for each.  A box on which frag16 / gather4 are far slower
// relative to `rows` than on the pool's normal boxes is slow THERE.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 97, W = 97, S = H + W, B = 8;

template <int MODE>
__global__ __launch_bounds__(256, 3) void k(const float *__restrict__ T, float *sink, int reps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
    float acc = 0.f;
    unsigned seed = 2654435761u * (blockIdx.x + 1);
    for (int r = 0; r < reps; ++r) {
        seed = seed * 1664525u + 1013904223u;
        const int strip = (seed >> 8) % (B * W), b = strip / W, w = strip - b * W;          // column strip (b, :, w)
        const float *base = T + ((size_t)b * H * W + w) * S;                                // row i of the block: base + i * W * S, 97 floats
        if (MODE == 0) {                 // the kernels' fragments: tiles t = wv, wv + 4; k-steps 0..2; two 16-byte loads each
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const int row = 16 * (wv + 4 * a) + ln, col = 32 * ks + 8 * lg;
                    if (row < H && col + 8 <= 96) {
                        const f32x4 u = *reinterpret_cast<const f32x4 *>(base + (size_t)row * W * S + col);
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (size_t)row * W * S + col + 4);
                        acc += u[0] + u[3] + v[1] + v[2];
                    }
                }
        } else if (MODE == 1) {          // the transposed gather: 4 bytes per lane, lane <-> row
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = 32 * ks + 8 * lg + e, col = 16 * (wv + 4 * a) + ln;
                        if (row < H && col < H) acc += base[(size_t)row * W * S + col];
                    }
        } else {                         // whole rows: 97 floats = 24 x 16 bytes + 1, one row per 32 lanes
            for (int row = 2 * wv + (lane >> 5); row < H; row += 8) {
                const int c4 = lane & 31;
                if (c4 < 24) {
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(base + (size_t)row * W * S + 4 * c4);
                    acc += u[0] + u[1] + u[2] + u[3];
                }
            }
        }
    }
    if (acc == 12345.678f) sink[blockIdx.x] = acc;
}
template <int MODE> static void run(const char *name, const float *T, float *sink, double bytes_per_block) {
    const int grid = 768, reps = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, T, sink, reps);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, T, sink, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / 10, blocks = (double)grid * reps;
    printf("%-8s %9.1f us per launch of %d x %d blocks: %6.3f us per block per workgroup slot, %7.1f GB/s\n", name, us, grid, reps,
           us / reps, blocks * bytes_per_block / (us * 1e-6) / 1e9);
}
int main() {
    float *T, *sink; const size_t n = (size_t)B * H * W * S;
    hipMalloc(&T, n * 4); hipMalloc(&sink, 1 << 16); hipMemset(T, 0, n * 4);
    run<0>("frag16", T, sink, 97.0 * 96 * 4);
    run<1>("gather4", T, sink, 96.0 * 97 * 4);
    run<2>("rows", T, sink, 97.0 * 96 * 4);
    return 0;
}

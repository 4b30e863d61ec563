// Probe for the row-band decomposition: a pixel-major tensor v[B][H][W][C] (C = 512 fp32, 2 KB per
// pixel); a workgroup (image b, channel group cg of CC channels, band of R rows) needs, for the column part of
// its output rows, EVERY pixel of the (b, cg) slice -- 97 x 97 segments of CC*4 bytes -- so the H/R band workgroups
// of one (b, cg) read the same 1.2 MB.  Question: when those workgroups are neighbours in the (XCD-aware) launch
// order, what does the chip deliver -- L2 -> LDS bytes per second -- and how many bytes come from HBM?
//   mode 0: every band workgroup streams the whole slice column by column (LDS-DMA, 3 columns in flight)
//   mode 1: the same, workgroup order NOT remapped (neighbours land on different XCDs)
//   mode 2: reference: each workgroup streams only its own 1/nb of the slice (no sharing; plain 154 MB stream)
// Build: hipcc --offload-arch=gfx950 -O3 band_share_probe.hip -o band_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int H = 97, W = 97, C = 512;

__device__ inline rsrc_t mk(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}

template <int CC>
__global__ __launch_bounds__(256) void band_kernel(const float *v, int ncg, int nb, int mode, float *sink) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ float lds[];
    constexpr int LPP = CC / 4;                       // 16-byte lanes per pixel
    constexpr int PPI = 64 / LPP;                     // pixels per wave instruction
    constexpr int NPIECE = (H + PPI - 1) / PPI;       // DMA instructions per column
    constexpr int PPW = (NPIECE + 3) / 4;             // per wave (padded: masked lanes fetch nothing)
    constexpr int COLF = PPW * 4 * 256;               // floats per column buffer
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    if (mode != 1) { const int q = nwg >> 3, r = nwg & 7, x = id & 7, i = id >> 3; id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
    const int band = id % nb, cg = (id / nb) % ncg, b = id / (nb * ncg);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t S = mk(v + (size_t)b * H * W * C, (unsigned)(H * W * C * 4));
    const int w_lo = mode == 2 ? band * W / nb : 0, w_hi = mode == 2 ? (band + 1) * W / nb : W;
    float acc = 0.f;
    auto issue = [&](int w, int buf) {
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const int piece = wv * PPW + p, j = piece * PPI + lane / LPP;
            const int voff = j < H ? ((j * W + w) * C + cg * CC + 4 * (lane % LPP)) * 4 : 0x7ffffff0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(S, (__attribute__((address_space(3))) void *)(lds + buf * COLF + piece * 256),
                                                     16, voff, 0, 0, 0);
        }
    };
    int n = 0;
    for (int w = w_lo; w < w_lo + 2 && w < w_hi; ++w) issue(w, n++ % 3);
    n = 0;
    for (int w = w_lo; w < w_hi; ++w, ++n) {
        if (w + 2 < w_hi) issue(w + 2, (n + 2) % 3);
        else { issue(w_hi - 1, (n + 2) % 3); }                       // keep the instruction count uniform
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        acc += lds[(n % 3) * COLF + threadIdx.x];                     // touch the landed column
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 123.456f) sink[0] = acc;
#endif
}

template <int CC>
void run(const float *v, float *sink, int B, int nb, int mode, int iters) {
    const int ncg = C / CC, grid = B * ncg * nb;
    constexpr int LPP = CC / 4, PPI = 64 / LPP, NPIECE = (H + PPI - 1) / PPI, PPW = (NPIECE + 3) / 4, COLF = PPW * 4 * 256;
    const size_t lds = 3 * COLF * 4;
    hipFuncSetAttribute((const void *)band_kernel<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) band_kernel<CC><<<grid, 256, lds>>>(v, ncg, nb, mode, sink);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) band_kernel<CC><<<grid, 256, lds>>>(v, ncg, nb, mode, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    const double us = ms * 1e3 / iters;
    const double slice = (double)B * H * W * C * 4;
    const double moved = mode == 2 ? slice : slice * nb;
    printf("CC=%3d bands=%2d mode=%d grid=%5d lds=%6zu : %8.1f us   L2->LDS %7.1f GB/s   (unique bytes %.0f MB -> %7.1f GB/s)\n",
           CC, nb, mode, grid, lds, us, moved / us * 1e-3, slice * 1e-6, slice / us * 1e-3);
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const size_t n = (size_t)B * H * W * C;
    float *v, *sink;
    hipMalloc(&v, n * 4); hipMalloc(&sink, 64);
    hipMemset(v, 0, n * 4);
    printf("B=%d: %.0f MB tensor, pixel-major, %d B per pixel\n", B, n * 4e-6, C * 4);
    for (int mode : {0, 1, 2}) {
        run<32>(v, sink, B, 7, mode, 10);
        run<32>(v, sink, B, 13, mode, 10);
        run<64>(v, sink, B, 7, mode, 10);
        run<64>(v, sink, B, 13, mode, 10);
    }
    return 0;
}

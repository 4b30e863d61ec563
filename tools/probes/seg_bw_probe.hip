// Probe: streaming bandwidth of a (planes, 97, 97) fp32 tensor when a workgroup touches, per (plane, row),
// only a SEG-byte segment (the column-strip pattern of cca_map.hpp: NS strips * 4 B) versus whole rows.
//   mode 0 = read (buffer_load_dwordx4 -> registers), 1 = write, 2 = copy, 3 = read by LDS-DMA
// Build: hipcc --offload-arch=gfx950 -O3 seg_bw_probe.hip -o seg_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int H = 97, W = 97, HW = H * W;

__device__ inline rsrc_t mk(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}

// COLS adjacent columns per workgroup (COLS = 0: ROWS_PER_WG whole rows instead); U planes in flight per wave
template <int COLS, int MODE, int U>
__global__ __launch_bounds__(512) void seg_kernel(const float *src, const float *src2, float *dst, int planes_per_image, int tiles,
                                                  int nsplit, int planes_per_block, float *sink, int remap) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ float lds[];
    int id = blockIdx.x;
    if (remap) { const int per = (gridDim.x + 7) / 8; id = (blockIdx.x % 8) * per + blockIdx.x / 8; if (id >= (int)gridDim.x) return; }
    const int b = id / (tiles * nsplit), rem = id % (tiles * nsplit);
    const int split = rem / tiles, tile = rem % tiles;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p_begin = split * planes_per_block;
    int p_end = p_begin + planes_per_block;
    if (p_end > planes_per_image) p_end = planes_per_image;
    const size_t img = (size_t)b * planes_per_image * HW;
    const rsrc_t S = mk(src + img, (unsigned)(planes_per_image * HW * 4));
    const rsrc_t D = mk(dst + img, (unsigned)(planes_per_image * HW * 4));
    const rsrc_t S2 = mk(src2 + img, (unsigned)(planes_per_image * HW * 4));
    constexpr int ROWS = 8;                        // row mode: 8 whole rows = 776 floats per plane
    constexpr int LPR = COLS ? COLS / 4 : 1;       // lanes per row segment
    constexpr int RPI = 64 / LPR;                  // rows per instruction
    constexpr int NI = COLS ? (H + RPI - 1) / RPI : (ROWS * W + 255) / 256;
    int voff[NI];
    bool ok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (COLS) {
            const int h = i * RPI + lane / LPR, w = tile * COLS + 4 * (lane % LPR);
            ok[i] = h < H && w + 3 < W;
            voff[i] = (h * W + w) * 4;
        } else {
            const int e = i * 256 + lane * 4;
            ok[i] = e + 3 < ROWS * W && tile * ROWS * W + e + 3 < HW;
            voff[i] = (tile * ROWS * W + e) * 4;
        }
    }
    u4 acc = {0, 0, 0, 0};
    for (int p0 = p_begin + wv * U; p0 < p_end; p0 += 8 * U) {
        u4 v[U][NI];
        u4 v2[U][NI];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u < p_end ? p0 + u : p_end - 1;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (MODE == 0 || MODE == 2 || MODE >= 4) {
                    v[u][i] = ok[i] ? __builtin_amdgcn_raw_buffer_load_b128(S, voff[i], p * HW * 4, 0) : u4{0, 0, 0, 0};
                    if (MODE == 4 || MODE == 5) v2[u][i] = ok[i] ? __builtin_amdgcn_raw_buffer_load_b128(S2, voff[i], p * HW * 4, 0) : u4{0, 0, 0, 0};
                } else if (MODE == 3) {
                    if (ok[i])
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(S, (__attribute__((address_space(3))) void *)(lds + ((wv * U + u) * NI + i) * 256),
                                                                 16, voff[i], p * HW * 4, 0, 0);
                } else {
                    v[u][i] = u4{(unsigned)p, (unsigned)i, (unsigned)lane, 7u};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u < p_end ? p0 + u : p_end - 1;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (MODE == 0) acc ^= v[u][i];
                if (MODE == 5) acc ^= v[u][i] ^ v2[u][i];
                if (MODE == 6 && COLS == 8) { const int e0 = i * 256 + (threadIdx.x & 63) * 4, k = e0 >> 6, r = e0 & 63, rem = H - k * 8, nr = rem < 8 ? rem : 8; if (r < 8 * nr && p0 + u < p_end) __builtin_amdgcn_raw_buffer_store_b128(v[u][i], D, 4 * (k * 8 * W + tile * 8 * nr + r), p * HW * 4, 0); }
                if (MODE == 7) { const int lin = (tile * 4096 + i * 256 + (threadIdx.x & 63) * 4) % (HW - 4); if (ok[i] && p0 + u < p_end) __builtin_amdgcn_raw_buffer_store_b128(v[u][i], D, lin * 4, p * HW * 4, 0); }
                if (MODE == 4) { const int lin = (tile * 4096 + i * 256 + (threadIdx.x & 63) * 4) % (HW - 4); if (ok[i] && p0 + u < p_end) __builtin_amdgcn_raw_buffer_store_b128(v[u][i] ^ v2[u][i], D, lin * 4, p * HW * 4, 0); }
                if (MODE == 1 || MODE == 2)
                    if (ok[i] && p0 + u < p_end) __builtin_amdgcn_raw_buffer_store_b128(v[u][i], D, voff[i], p * HW * 4, 0);
            }
        }
        if (MODE == 3) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);   // vmcnt(0)
    }
    if (MODE == 3) acc[0] = __float_as_uint(lds[threadIdx.x]);
    if (acc[0] == 0x12345u && acc[1] == 0x777u) sink[0] = 1.f;
#endif
}

template <int COLS, int MODE, int U>
double run(const float *src, const float *src2, float *dst, float *sink, int B, int C, int target_blocks, size_t lds_bytes, int iters, int remap = 1) {
    const int tiles = COLS ? (W + COLS - 1) / COLS : (H + 7) / 8;
    int nsplit = target_blocks / (B * tiles);
    if (nsplit < 1) nsplit = 1;
    const int ppb = (C + nsplit - 1) / nsplit;
    const int grid = B * tiles * nsplit;
    hipFuncSetAttribute((const void *)seg_kernel<COLS, MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) seg_kernel<COLS, MODE, U><<<grid, 512, lds_bytes>>>(src, src2, dst, C, tiles, nsplit, ppb, sink, remap);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) seg_kernel<COLS, MODE, U><<<grid, 512, lds_bytes>>>(src, src2, dst, C, tiles, nsplit, ppb, sink, remap);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    // bytes actually touched (column mode drops the last partial 16-byte piece of a row; close enough to count HW)
    const double cols_touched = COLS ? (double)((W / 4) * 4) : W;
    const double bytes = (double)B * C * H * cols_touched * 4 * ((MODE == 2 || MODE == 5 || MODE == 6 || MODE == 7) ? 2 : MODE == 4 ? 3 : 1);
    const double us = ms * 1e3 / iters;
    printf("cols=%2d mode=%d U=%d remap=%d grid=%4d lds=%6zu : %8.1f us  %7.1f GB/s\n", COLS, MODE, U, remap, grid, lds_bytes, us, bytes / us * 1e-3);
    return us;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = 512;
    const size_t n = (size_t)B * C * HW;
    float *src, *src2, *dst, *sink;
    hipMalloc(&src, n * 4); hipMalloc(&src2, n * 4); hipMalloc(&dst, n * 4); hipMalloc(&sink, 64);
    hipMemset(src, 0, n * 4); hipMemset(src2, 0, n * 4); hipMemset(dst, 0, n * 4);
    const int iters = 10;
    const size_t lds = (size_t)160 * 1024 - 512;
    printf("B=%d: %.0f MB per tensor\n", B, n * 4e-6);
    for (int tb : {26 * B, 52 * B}) {
        printf("-- target blocks %d\n", tb);
        run<8, 0, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<16, 0, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<32, 0, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<0, 0, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<8, 1, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<0, 1, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<8, 2, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<0, 2, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<8, 5, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<0, 5, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<8, 6, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<8, 7, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<0, 7, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<8, 4, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<16, 4, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
        run<0, 4, 2>(src, src2, dst, sink, B, C, tb, lds, iters);
    }
    return 0;
}

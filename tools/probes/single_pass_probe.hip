// single_pass_probe.hip -- VERDICT r3 item 3: can a SINGLE-PASS, XCD-resident aggregation beat the two strip passes?
// MOVE-ONLY probe of its data movement (no MFMA, no attention, no x / y): the floor of any such kernel.
//
// Candidate: workgroup = a TH x TW pixel tile of one image; it computes, for its pixels, the column AND the row contribution of
// functions.py:46-47 and writes y once (no column -> row partial in HBM, v read from HBM once).  For that it must see, per
// 64-channel group, the TW column strips and the TH row strips that cross the tile -- TH + TW strip tiles of 97 positions
// x 256 B (hi | lo planes) -- of which it uses one 16-row MFMA tile each.  All tiles of an image run on ONE XCD (ids 8 apart), so
// the 2.4 MB (image, channel group) slice is fetched from HBM once and re-read (97/TH + 97/TW) times out of that XCD's L2.
//   mode 0  the candidate's reads: grid = B * tiles, image b on XCD b, per workgroup 8 groups x (TH + TW) strip tiles through a
//           three-slot LDS ring (1 KiB LDS-DMA pieces, counted vmcnt: the library's fill machinery)
//   mode 1  the same, NOT XCD-pinned (tiles of an image spread over the 8 L2s)
//   mode 2  reference, what the two strip passes read today: every column strip and every row strip once (2 x 154 MB from HBM)
// Output: us per launch, bytes into LDS per second, and the time the two real passes take for comparison (HISTORY.md 3.6).
// Build: hipcc --offload-arch=gfx950 -O3 single_pass_probe.hip -o single_pass_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int H = 97, W = 97, C = 512, NCG = C / 64, PIECES = 2 * 13, SLOT = PIECES * 256;   // floats per ring slot (26 KiB)

__device__ inline rsrc_t mk(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000); }

// one strip tile (97 positions x (hi | lo) x 64 channels of group cg) -> LDS slot; 4 waves share the 26 pieces
__device__ inline void fill(const rsrc_t &S, float *slot, int pix0, int pstep, int cg, int wv, int lane) {
    for (int it = wv; it < PIECES; it += 4) {
        const int plane = it >= 13, piece = it - 13 * plane, p = lane >> 3, i = 8 * piece + p, q = lane & 7;
        const int off = i < 97 ? ((pix0 + i * pstep) * 2 * C + plane * C + cg * 64 + 8 * q) * 2 : 0x7ffffff0;
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(slot + it * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(lds_addr), "s"(S) : "memory");
    }
}

template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void probe_kernel(const unsigned short *v, int B, int mode, float *sink) {
#if __HIP_DEVICE_COMPILE__
    __shared__ float lds[3 * SLOT];
    constexpr int NTY = (H + TH - 1) / TH, NTX = (W + TW - 1) / TW, NTILE = NTY * NTX;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b, nstrip, pix0[TH + TW], pstep[TH + TW];
    if (mode == 2) {                       // one strip per workgroup: id = b * 194 + r
        b = blockIdx.x / (H + W);
        const int r = blockIdx.x % (H + W);
        nstrip = 1;
        pix0[0] = r < W ? r : (r - W) * W;
        pstep[0] = r < W ? W : 1;
    } else {
        int tile;
        if (mode == 0) { b = blockIdx.x & 7; tile = blockIdx.x >> 3; if (b >= B) return; }
        else           { b = blockIdx.x / NTILE; tile = blockIdx.x % NTILE; }
        if (tile >= NTILE) return;
        const int ty = tile / NTX, tx = tile % NTX;
        nstrip = 0;
        for (int w = tx * TW; w < (tx + 1) * TW && w < W; ++w) { pix0[nstrip] = w; pstep[nstrip++] = W; }          // column strips
        for (int h = ty * TH; h < (ty + 1) * TH && h < H; ++h) { pix0[nstrip] = h * W; pstep[nstrip++] = 1; }      // row strips
    }
    const rsrc_t S = mk(v + (size_t)b * H * W * 2 * C, (unsigned)(H * W * 2 * C * 2));
    const int total = NCG * nstrip, npw = (PIECES - wv + 3) / 4;
    float acc = 0.f;
    auto issue = [&](int n) { fill(S, lds + (n % 3) * SLOT, pix0[n % nstrip], pstep[n % nstrip], n / nstrip, wv, lane); };
    issue(0);
    if (total > 1) issue(1);
    for (int n = 0; n < total; ++n) {
        if (n + 1 < total) { if (npw == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (n + 2 < total) issue(n + 2);
        acc += lds[(n % 3) * SLOT + threadIdx.x * 4];                 // touch the landed tile
    }
    if (acc == 123.456f) sink[0] = acc;
#endif
}

template <int TH, int TW>
void run(const unsigned short *v, float *sink, int B, int mode, int iters) {
    constexpr int NTILE = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const int grid = mode == 2 ? B * (H + W) : mode == 0 ? 8 * NTILE : B * NTILE;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe_kernel<TH, TW><<<grid, 256>>>(v, B, mode, sink);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) probe_kernel<TH, TW><<<grid, 256>>>(v, B, mode, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    const double us = ms * 1e3 / iters, tile_bytes = 97.0 * 256;              // one strip tile of one channel group: 97 positions x (128 B hi + 128 B lo)
    const double moved = mode == 2 ? (double)B * (H + W) * NCG * tile_bytes
                                   : (double)B * NCG * tile_bytes * (((H + TH - 1) / TH) * (double)W + ((W + TW - 1) / TW) * (double)H);
    printf("tile %2dx%2d mode %d grid %5d: %8.1f us   into LDS %6.2f GB -> %7.1f GB/s\n", TH, TW, mode, grid, us, moved * 1e-9, moved / us * 1e-3);
}

int main(int argc, char **argv) {
    const int B = 8, iters = argc > 1 ? atoi(argv[1]) : 20, only = argc > 2 ? atoi(argv[2]) : -1;      // only: one mode at 14 x 14 (PMC runs)
    unsigned short *v; float *sink;
    const size_t n = (size_t)B * H * W * 2 * C;
    hipMalloc(&v, n * 2); hipMalloc(&sink, 64);
    hipMemset(v, 0x11, n * 2);
    printf("single-pass aggregation, data movement only; (8,512,97,97): v = 154 MB; today's two strip passes: 66 + 146 us incl. MFMA, A, partial, x, y\n");
    for (int mode = 0; mode < 3; ++mode) {
        if (only >= 0 && mode != only) continue;
        run<14, 14>(v, sink, B, mode, iters);
        if (mode == 2 || only >= 0) continue;
        run<16, 16>(v, sink, B, mode, iters);
        run<25, 25>(v, sink, B, mode, iters);
        run<33, 33>(v, sink, B, mode, iters);
    }
    hipFree(v); hipFree(sink);
    return 0;
}

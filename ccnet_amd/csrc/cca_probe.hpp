// cca_probe.hpp -- in-band probes of the DEVICE STATE a measurement was taken in (ccnet_cca_probe_* in include/ccnet_cca.h).
//
// Why they exist: boxes of one MI355X pool ran the same step 0.665 .. 0.820 ms (profiles/r04*_bench.json, BENCH_r04.json) at the same
// sclk as rocm-smi reports it, with a float4 copy within 6 % -- the launches that keep every pipe of a CU busy lost 33-50 %.  rocm-smi
// shows the REQUESTED clock level; what a wave really ran at is the ratio of two counters it can read itself (s_memtime: shader
// cycles, s_memrealtime: 100 MHz).  bench.py runs these probes next to the timed step and prints them on the metric's line, so a
// slow box explains itself: effective clock idle / under a matrix-pipe burn / WHILE THE STEP RUNS, the matrix rate, the issue ->
// landed latency of the 25 KB LDS-DMA tile every strip kernel waits for (idle and with the whole chip streaming).
// Nothing here is on the product path.
#pragma once
#include "cca_gmap.hpp"

namespace cca {

// One wave per workgroup samples (shader clock, reference clock) every ``interval`` reference ticks (10 ns each), ``nsamples``
// times; out[(wg * nsamples + s) * 2 + {0, 1}].  Launched on a stream of its own NEXT TO whatever is to be observed (a queue of
// replayed steps, the burn kernel, nothing): it needs one wave slot and no LDS.  out[2 * nwg * nsamples + wg] = the XCC it ran on.
__global__ __launch_bounds__(64) void probe_clock_kernel(unsigned long long *out, int nsamples, int interval) {
    if (lane_id() != 0) return;
    const int wg = blockIdx.x;
    unsigned long long *o = out + (size_t)wg * nsamples * 2;
    const uint64_t t0 = ref_clock();
    for (int s = 0; s < nsamples; ++s) {
        const uint64_t due = t0 + (uint64_t)s * (uint64_t)interval;
        while (ref_clock() < due) short_sleep();
        // (the two reads are a few cycles apart: noise of < 0.1 % at intervals of >= 10 us)
        const uint64_t sc = shader_clock(), rc = ref_clock();
        o[2 * s] = sc;
        o[2 * s + 1] = rc;
    }
    out[(size_t)gridDim.x * nsamples * 2 + wg] = (unsigned long long)xcc_id();
}

// Matrix-pipe burn: every wave issues ``iters`` x 8 independent v_mfma_f32_16x16x32_bf16 (the instruction of the split-bf16
// contractions); wave 0 of a workgroup records (shader, reference) stamps around its loop: clk[wg * 4 + {0..3}] = shader start,
// reference start, shader end, reference end.  flops per wave = iters * 8 * 16384.  ``sink`` (one float per thread) keeps the
// accumulators alive.
__global__ __launch_bounds__(256, 2) void probe_mfma_kernel(unsigned long long *clk, float *sink, int iters) {
    const int lane = lane_id();
    u32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = 0x3f803f80u + (uint32_t)lane; b[e] = 0x3f803f80u ^ ((uint32_t)lane << 2); }
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint64_t s0 = shader_clock(), r0 = ref_clock();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = mfma_bf16_16x16x32(a, b, acc[t]);
    }
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) keep += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    const uint64_t s1 = shader_clock(), r1 = ref_clock();
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = keep;
    if (threadIdx.x == 0) {
        unsigned long long *o = clk + (size_t)blockIdx.x * 4;
        o[0] = s0; o[1] = r0; o[2] = s1; o[3] = r1;
    }
}

// The tile fill every strip kernel waits for: 25 LDS-DMA pieces of 1 KiB (4 rows of 256 B, rows ``row_stride`` bytes apart -- a
// column strip of an fp32 pixel-major tensor) into LDS, 4 waves, issue -> landed (vmcnt(0) + barrier) in shader cycles, ``reps``
// times on tiles spread over ``span_rows`` rows of the source.  clk[wg * 4 + {0..3}] = sum of the fill latencies, their maximum,
// reference ticks start / end of the whole loop (bytes moved = reps * 25 KiB per workgroup).  WPC workgroups per CU: 1 with a
// one-workgroup grid = the idle latency, 3 x 256 workgroups = the loaded latency and the rate the whole chip streams such tiles at.
constexpr int PROBE_TILE_PIECES = 25;
__global__ __launch_bounds__(GS_THREADS, 3) void probe_dma_kernel(const float *__restrict__ src, size_t src_bytes, unsigned long long *clk,
                                                                  int reps, int row_stride, int span_rows) {
    __shared__ __attribute__((aligned(16))) float tile[PROBE_TILE_PIECES * 256];
    CCA_LDS_REGISTER(tile);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const FBuf Sb = make_fbuf(src, src_bytes);
    uint64_t sum = 0, mx = 0;
    const uint64_t r0 = ref_clock();
    unsigned seed = 2654435761u * (blockIdx.x + 1);
    // (rows of 256 B out of a row_stride-byte pitch: the tile's column offset is random too, so that the fills of a large source
    //  come from HBM and not from a 0.5 MB working set that lives in the L2)
    const unsigned ncol = row_stride >= 512 ? (unsigned)(row_stride / 256) : 1u;
    for (int r = 0; r < reps; ++r) {
        seed = seed * 1664525u + 1013904223u;
        const int row0 = (int)((seed >> 8) % (unsigned)(span_rows > 100 ? span_rows - 100 : 1));
        seed = seed * 1664525u + 1013904223u;
        const int col = 256 * (int)((seed >> 8) % ncol);
        barrier_dma_keep<0>();
        const uint64_t s0 = shader_clock();
        for (int it = wv; it < PROBE_TILE_PIECES; it += GS_WAVES) {
            const int row = row0 + 4 * it + (lane >> 4);
            fbuf_load_to_lds_x4_uncounted(Sb, tile + it * 256, row * row_stride + col + 16 * (lane & 15));
        }
        barrier_dma_keep<0>();
        const uint64_t d = shader_clock() - s0;
        sum += d;
        mx = d > mx ? d : mx;
    }
    const uint64_t r1 = ref_clock();
    if (tid == 0) {
        unsigned long long *o = clk + (size_t)blockIdx.x * 4;
        o[0] = sum; o[1] = mx; o[2] = r0; o[3] = r1;
        // (keep the tile observable: the DMA is opaque to the compiler, this read is not)
        if (CCA_LDS_LD(tile) == 12345.678f) o[0] = 0;
    }
}

}  // namespace cca

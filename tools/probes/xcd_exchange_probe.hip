// Probe for DESIGN.md 9 (the XCD-cooperative aggregation): the EXCHANGE PRIMITIVE that design stands on, alone.
// Per XCD one "image": 97 workgroups (slots claimed at run time from the hardware XCC id, so nothing is assumed about the
// dispatcher) play a column-strip AND a row-strip role for PHASES channel groups of 32 channels:
//   produce  : workgroup s writes its column's slice of the partial -- 97 pieces of 128 B, one per row h, at [slot][h][s][32]
//   signal   : all stores acknowledged (s_waitcnt vmcnt(0)), one relaxed device-scope atomic add on the XCD's flag of the phase
//   wait     : spin (bounded: a time-out sets an error word instead of hanging the box) until the flag says 97
//   consume  : workgroup h reads row h of the partial -- 97 x 128 B contiguous, written by 97 DIFFERENT workgroups -- after a
//              buffer_inv (the vector L1 may hold the slot's lines of three phases ago), and checks every value
// with three partial slots (a slot is rewritten two phases after its readers).  MODE 1 adds the traffic the real phase has around the
// exchange (two v tiles and an x tile read, a y tile written per workgroup and phase, from / to a large buffer): the L2 pressure.
// MODE 2 is the baseline without any exchange: the same streaming, every workgroup re-reads its OWN slice, no flag; MODE 3 the streaming and the flag without any partial data.
// Prints per mode: us per launch and per phase, mean / max shader cycles a workgroup spent in the flag wait per phase, mismatching
// values (stale or unwritten data), time-outs, workgroups seen per XCC.  Under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE the launch's
// fabric traffic tells whether the partial stayed in the L2 (tools/gpu_round6.sh, stage "xprobe").
// Build: hipcc --offload-arch=gfx950 -O3 xcd_exchange_probe.hip -o xcd_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NP = 97, CG = 32, NSLOT = 3, NXCD = 8, WG_PER_XCD = 128, THREADS = 256;
constexpr size_t SLICE = (size_t)NP * NP * CG;                     // floats of one partial slot of one XCD (1.2 MB)

__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u); }
__device__ __forceinline__ float val(int xcc, int ph, int h, int w, int q) { return (float)(((xcc * 131 + ph) * 101 + h) * 97 + w) + 0.125f * q; }

// BAR = 0: one atomic counter per XCD and phase, every workgroup spins on it.  BAR = 1: one arrival WORD per workgroup -- a workgroup
// stores the phase number into its own word and wavefront 0 polls all 97 words with two wave-wide loads: no atomic, no shared line written
// by more than one workgroup.
template <int MODE, int BAR>
__global__ __launch_bounds__(THREADS, 4) void k(float *partial, unsigned *flags, unsigned *reg, unsigned *err, unsigned long long *stats,
                                                const float *stream_in, float *stream_out, size_t stream_floats, int phases) {
    __shared__ int s_slot, s_xcc;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_xcc = xcc_id() & 7;
        s_slot = (int)atomicAdd(&reg[s_xcc], 1u);
        atomicAdd(&stats[2 + s_xcc], 1ull);                        // (every workgroup: what the dispatcher really did)
    }
    __syncthreads();
    const int xcc = s_xcc, slot = s_slot;
    if (slot >= NP) return;
    float *P = partial + (size_t)xcc * NSLOT * SLICE;
    unsigned *F = flags + xcc * 64;
    unsigned long long wait_cycles = 0, wait_max = 0, prod_cycles = 0, cons_cycles = 0;
    unsigned bad = 0;
    float sink = 0.f;
    const size_t tile = (size_t)NP * CG;                           // floats of one 97 x 32 tile (12.4 KB)
    for (int ph = 0; ph < phases; ++ph) {
        float *S = P + (size_t)(ph % NSLOT) * SLICE;
        if (MODE != 0) {                                           // the phase's streaming: 3 tiles in, 1 out, at addresses of their own
            const size_t base = (((size_t)(xcc * phases + ph) * NP + slot) * 4 * tile) % (stream_floats - 4 * tile);
            for (int e = tid; e < (int)(3 * tile / 4); e += THREADS) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(stream_in + base + 4 * (size_t)e);
                sink += u[0] + u[3];
            }
            for (int e = tid; e < (int)(tile / 4); e += THREADS)
                *reinterpret_cast<f32x4 *>(stream_out + base + 4 * (size_t)e) = f32x4{sink, 1.f, 2.f, 3.f};
        }
        // produce: column role, [h][w = slot][32]: 8 lanes x 16 B per row h
        const unsigned long long tp = clock64();
        if (MODE != 3)
        for (int e = tid; e < NP * 8; e += THREADS) {
            const int h = e >> 3, q = e & 7;
            const float v = val(xcc, ph, h, slot, q);
            *reinterpret_cast<f32x4 *>(S + ((size_t)h * NP + slot) * CG + 4 * q) = f32x4{v, v + 1.f, v + 2.f, v + 3.f};
        }
        if (MODE != 2) {
            __builtin_amdgcn_s_waitcnt(0);                         // (vmcnt(0): every store of this wavefront acknowledged by the L2)
            __syncthreads();
            prod_cycles += clock64() - tp;
            if (BAR == 0 && tid == 0) {
                (void)__hip_atomic_fetch_add(&F[ph], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t0 = clock64();
                unsigned spins = 0;
                while (__hip_atomic_load(&F[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NP) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 22)) { atomicAdd(err, 1u); break; }
                }
                const unsigned long long dt = clock64() - t0;
                wait_cycles += dt;
                wait_max = dt > wait_max ? dt : wait_max;
            }
            if (BAR >= 1 && tid < 64) {
                // BAR = 2: the words are written and polled the way the DATA is -- plain stores (acknowledged by the L2), plain loads after
                // a buffer_inv of the vector L1 -- so that the XCD's own L2 serves the poll; BAR = 1: device-scope atomics (the fabric does)
                volatile unsigned *Aw = flags + NXCD * 64 + xcc * 128;    // 97 arrival words of this XCD
                if (tid == 0) {
                    if (BAR == 2) { Aw[slot] = (unsigned)(ph + 1); __builtin_amdgcn_s_waitcnt(0); }
                    else __hip_atomic_store(const_cast<unsigned *>(&Aw[slot]), (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned long long t0 = clock64();
                unsigned spins = 0;
                for (;;) {
                    unsigned a, b;
                    if (BAR == 2) {
                        asm volatile("buffer_inv sc1" ::: "memory");
                        a = Aw[tid];
                        b = tid + 64 < NP ? Aw[tid + 64] : 0xffffffffu;
                    } else {
                        a = __hip_atomic_load(const_cast<unsigned *>(&Aw[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        b = tid + 64 < NP ? __hip_atomic_load(const_cast<unsigned *>(&Aw[tid + 64]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    }
                    const bool ok = a >= (unsigned)(ph + 1) && b >= (unsigned)(ph + 1);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 22)) { if (tid == 0) atomicAdd(err, 1u); break; }
                }
                if (tid == 0) {
                    const unsigned long long dt = clock64() - t0;
                    wait_cycles += dt;
                    wait_max = dt > wait_max ? dt : wait_max;
                }
            }
            __syncthreads();
            asm volatile("buffer_inv sc1" ::: "memory");
        } else {
            __syncthreads();
        }
        // consume: row role (MODE 2: the workgroup's own column slice again)
        const unsigned long long tc = clock64();
        if (MODE != 3)
        for (int e = tid; e < NP * 8; e += THREADS) {
            const int w = e >> 3, q = e & 7;
            const int hh = MODE == 2 ? w : slot, ww = MODE == 2 ? slot : w;
            const f32x4 u = *reinterpret_cast<const f32x4 *>(S + ((size_t)hh * NP + ww) * CG + 4 * q);
            const float v = val(xcc, ph, hh, ww, q);
            bad += (u[0] != v) + (u[1] != v + 1.f) + (u[2] != v + 2.f) + (u[3] != v + 3.f);
        }
        __syncthreads();
        cons_cycles += clock64() - tc;
    }
    if (bad) atomicAdd(err + 1, bad);
    if (sink == 12345.678f) stream_out[0] = sink;
    if (tid == 0) {
        atomicAdd(&stats[0], wait_cycles);
        atomicMax(&stats[1], wait_max);
        atomicAdd(&stats[10], prod_cycles);
        atomicAdd(&stats[11], cons_cycles);
    }
}

template <int MODE, int BAR = 0> static void run(const char *name, float *partial, unsigned *flags, unsigned *reg, unsigned *err, unsigned long long *stats,
                                    const float *sin, float *sout, size_t sfloats, int phases, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float total = 0;
    unsigned herr[2] = {0, 0};
    unsigned long long hst[12] = {0};
    for (int i = -2; i < iters; ++i) {
        hipMemsetAsync(flags, 0, NXCD * (64 + 128) * 4, 0); hipMemsetAsync(reg, 0, NXCD * 4, 0);
        if (i == 0) { hipMemsetAsync(err, 0, 8, 0); hipMemsetAsync(stats, 0, 96, 0); }
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, BAR>), dim3(NXCD * WG_PER_XCD), dim3(THREADS), 0, 0, partial, flags, reg, err, stats, sin, sout, sfloats, phases);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (i >= 0) total += ms;
    }
    hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost); hipMemcpy(hst, stats, 96, hipMemcpyDeviceToHost);
    const double us = total * 1e3 / iters, nwg = (double)NXCD * NP * iters;
    printf("%-34s %8.1f us per launch, %6.2f us per phase | flag wait per workgroup and phase: mean %7.0f cycles, max %8llu | mismatching values %u, "
           "time-outs %u | workgroups per XCC:", name, us, us / phases, (double)hst[0] / (nwg * phases), hst[1], herr[1], herr[0]);
    for (int x = 0; x < 8; ++x) printf(" %llu", hst[2 + x] / (unsigned long long)iters);
    printf(" | cycles per workgroup and phase: produce + store acknowledgement %.0f, consume %.0f\n", (double)hst[10] / (nwg * phases), (double)hst[11] / (nwg * phases));
}

int main(int argc, char **argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 16, iters = argc > 2 ? atoi(argv[2]) : 20;
    float *partial, *sin, *sout; unsigned *flags, *reg, *err; unsigned long long *stats;
    const size_t sfloats = (size_t)160 << 20;                      // 640 MB each way: beyond the 256 MiB Infinity Cache
    hipMalloc(&partial, NXCD * NSLOT * SLICE * 4); hipMalloc(&flags, NXCD * (64 + 128) * 4); hipMalloc(&reg, NXCD * 4); hipMalloc(&err, 8);
    hipMalloc(&stats, 96); hipMalloc(&sin, sfloats * 4); hipMalloc(&sout, sfloats * 4);
    hipMemset(sin, 0, sfloats * 4); hipMemset(partial, 0xff, NXCD * NSLOT * SLICE * 4);
    printf("XCD exchange probe: %d workgroups per XCD claim %d strip-pair slots, %d phases of 32 channels, partial slice %.2f MB per XCD and slot\n",
           WG_PER_XCD, NP, phases, SLICE * 4 / 1e6);
    run<0>("exchange only", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<1>("exchange + the phase's streaming", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<2>("streaming, no exchange (baseline)", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<3>("streaming + flag, no partial data", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    printf("-- the same with one arrival word per workgroup instead of one atomic counter --\n");
    run<3, 1>("streaming + flag, no partial data", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<0, 1>("exchange only", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<1, 1>("exchange + the phase's streaming", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    printf("-- arrival words written and polled like the data: plain stores, buffer_inv + plain loads (served by the XCD's L2) --\n");
    run<3, 2>("streaming + flag, no partial data", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<0, 2>("exchange only", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    run<1, 2>("exchange + the phase's streaming", partial, flags, reg, err, stats, sin, sout, sfloats, phases, iters);
    return 0;
}

// cca_platform.hpp -- gfx950 implementations of the device primitives the kernels are written against:
// wave shuffles, the MFMA wrappers, buffer-resource loads / stores, LDS-DMA, counted barriers, the launch macro.
// (The CPU test-suite provides a header of the same name under tests/emu/ that implements the same primitives in a
// SIMT emulator; the product never sees it.)
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

namespace cca {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 8 packed bf16 (element e in dword e/2, low half = even e)
constexpr int kWave = 64;                 // CDNA wavefront


__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32: two fp32 -> two bf16 (round to nearest even) in one dword, first operand in the low half
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo_elem, float hi_elem) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo_elem, hi_elem}, bf16x2_t));
}
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Wait states after an exact-f32 MFMA whose result is read straight away.  hipcc (ROCm 7.2) pads only 8 states between a
// v_mfma_f32_16x16x4_f32 at the end of one basic block and a v_accvgpr_read of its last result register at the top of
// the next one, three short of what the 8-pass instruction needs: the read returned the accumulator WITHOUT that MFMA's
// contribution (found on MI355X with the row-band kernel: band rows 3, 7, 11, 15 lost their k-tail).  Call this
// between such an MFMA and the first reader when they may end up in different basic blocks.
__device__ __forceinline__ void mfma_f32_result_fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// scheduling fence: the compiler may not move instructions across it (it still places the s_waitcnt each use needs)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// tell the compiler a value is wave-uniform (it is: derived from the wave id) so it lives in SGPRs
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Opaque copy of a per-lane value: everything computed from the result is recomputed where it is used instead of
// being hoisted out of the surrounding loop (loop-invariant address arithmetic of the tile-store phase would
// otherwise occupy VGPRs across the MFMA phase, where every register is spoken for).
__device__ __forceinline__ int recompute_here(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Buffer-resource view of one image's worth of a tensor: buffer_load_dword v, voff, s[rsrc], soff offen
// keeps ONE 32-bit VGPR offset per lane plus a scalar offset per load, instead of a 64-bit VGPR
// address per load (which is what plain pointer arithmetic compiles to, and what spilled).
typedef __amdgpu_buffer_rsrc_t FBuf;
__device__ __forceinline__ FBuf make_fbuf(const float *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float fbuf_load(const FBuf &b, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, voff_bytes, soff_bytes, 0));
}
__device__ __forceinline__ f32x4 fbuf_load_x4(const FBuf &b, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, voff_bytes, soff_bytes, 0));
}
// stores whose per-lane offset is out of range (kOobOffset) are dropped by the buffer range check
__device__ __forceinline__ void fbuf_store(const FBuf &b, float v, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), b, voff_bytes, soff_bytes, 0);
}
// 16-byte store (global address needs only 4-byte alignment, like the 16-byte loads)
__device__ __forceinline__ void fbuf_store_x4(const FBuf &b, f32x4 v, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, voff_bytes, soff_bytes, 0);
}
// 8-byte store (two dwords; 4-byte alignment suffices)
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fbuf_store_x2(const FBuf &b, uint32_t v0, uint32_t v1, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2s{v0, v1}, b, voff_bytes, soff_bytes, 0);
}
// ds_read_b128: p must be 16-byte aligned
__device__ __forceinline__ f32x4 lds_load_x4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
// ds_write_b128: p must be 16-byte aligned
__device__ __forceinline__ void lds_store_x4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
// LDS-DMA (buffer_load_dword ... lds): no staging VGPRs, no ds_write pass; the 64 dwords of the wave land
// contiguously at the wave-uniform LDS address (M0) + lane * 4.  Completion is tracked by vmcnt; the
// compiler drains it before the next __syncthreads(), which is exactly the double-buffer hand-over.
__device__ __forceinline__ void fbuf_load_to_lds(const FBuf &b, float *lds_wave_base, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void *)lds_wave_base, 4,
                                             voff_bytes, soff_bytes, 0, 0);
}

// 16-byte form (buffer_load_dwordx4 ... lds): 1 KiB per wave instruction.  Neither the global nor the LDS
// address needs more than 4-byte alignment (probed on MI355X: tools/probes/dma_x4_probe.hip).
__device__ __forceinline__ void fbuf_load_to_lds_x4(const FBuf &b, float *lds_wave_base, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void *)lds_wave_base, 16,
                                             voff_bytes, soff_bytes, 0, 0);
}

// ds_read_b64_tr_b16 (gfx950 transposing LDS read): every lane passes the 8-byte-aligned LDS address of 4 consecutive
// 16-bit elements; inside each group of 16 lanes, lane i receives element (i & 3) of the 8 bytes addressed by lane
// 4 j + (i >> 2), j = 0..3 -- i.e. the group reads a [4 rows][16 columns] block (row r supplied as four 8-byte pieces by
// lanes 4 r .. 4 r + 3) and lane i gets column i, rows 0..3 packed as two dwords (row 0 in the low half of the first).
// Probed on MI355X with arbitrary per-lane addresses: tools/probes/tr16_probe.hip.  Wave-collective.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_read_tr16_b64(const void *p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)p));
}

// LDS-DMA that the COMPILER DOES NOT SEE (inline asm; same instruction as fbuf_load_to_lds_x4 with soffset 0).  hipcc
// tracks builtin LDS-DMAs as LDS stores and protects later LDS reads it cannot prove disjoint from them with a
// `s_waitcnt vmcnt` of its own -- for the transposing reads and the 16-bit gathers of cca_gmap.hpp that wait drained the
// fill of the NEXT tile before every group's first fragment read (ISA: vmcnt(0) / vmcnt(#newer loads); tools/isa_waits.py),
// i.e. the double buffer never overlapped anything.  An asm statement is opaque: no wait is inserted for it and it is not in
// the compiler's vmcnt bookkeeping (its waits for other loads become slightly stricter, never weaker: completion is in issue
// order).  The kernel itself waits with counted barriers (barrier_dma_keep*) before reading a tile -- as it always did.
// M0 (LDS destination base) is saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void fbuf_load_to_lds_x4_uncounted(const FBuf &b, float *lds_wave_base, int voff_bytes) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(lds_addr), "s"(b) : "memory");
}

// LDS reads outside the compiler's lgkmcnt bookkeeping, for a register ping-pong (cca_gemm.hpp): across a loop back edge hipcc
// waits with lgkmcnt(0) before the first use of ANY tracked read -- also for the reads of the NEXT half step requested just
// before, which are the ones the MFMAs are supposed to hide.  The kernel waits itself: lds_wait_keep<N>() returns once at
// most N of this wave's LDS reads are outstanding (they return in order) and carries the fragment registers as operands,
// so no use of them can be scheduled above it.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_x4_uncounted(const void *p) {
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF) : "memory");
    return r;
}
template <int KEEP>
__device__ __forceinline__ void lds_wait_keep(u32x4 (&b)[4], u32x4 (&a)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                 : "n"(KEEP) : "memory");
}

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt) but NOT for
// its outstanding global stores / loads (vmcnt).  __syncthreads() drains vmcnt as well whenever an LDS-DMA
// has been issued, which would stall every chunk on the acknowledgement of the tile stores.
__device__ __forceinline__ void barrier_lds_only() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Workgroup barrier for a DMA pipeline that is more than one chunk deep: waits until at most KEEP of this
// wave's vector-memory operations (the LDS-DMA pieces of the newest chunk) are still in flight, then
// synchronises.  __syncthreads() would drain vmcnt to 0 and stall on the chunk that was only just requested
// (cdna_hip_programming.md, "Pipelining across barriers": counted vmcnt + raw s_barrier).
template <int KEEP>
__device__ __forceinline__ void barrier_dma_keep() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// this wave's outstanding vector-memory operations (LDS-DMA pieces, loads, stores) have all completed
__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Cycle counters (tools: ccnet_cca_probe_*).  s_memtime ticks at the SHADER clock (MI355X_MICROARCH.md: "tick = shader cycle"),
// s_memrealtime at the constant 100 MHz reference clock: the ratio of two deltas is the clock the wave really ran at --
// what rocm-smi's "sclk" (the requested level) does not show when a power / thermal limit stretches the clock.
__device__ __forceinline__ uint64_t shader_clock() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ uint64_t ref_clock() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void short_sleep() { __builtin_amdgcn_s_sleep(8); }
// XCC (= XCD) the wave runs on: HW_REG_XCC_ID (id 20), bits [3:0]
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u); }

// device-scope atomic max on a 32-bit word (order-independent: the result does not depend on which wave gets there first)
__device__ __forceinline__ void atomic_max_u32(unsigned *p, uint32_t v) {
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define CCA_LDS_REGISTER(arr) do { } while (0)
#define CCA_LDS_LD(p) (*(p))
#define CCA_LDS_ST(p, v) do { *(p) = (v); } while (0)


}  // namespace cca

// ---- launch profiler (ccnet_cca_profile_begin / _end): when armed, every launch is bracketed by a HIP-event pair on
// ---- its own stream, so a tool can read each launch's duration INSIDE a step (bench.py: gpu_kernel_sum_ms, idle_ms)
// ---- without an external profiler.  Disarmed (the default) it costs one relaxed atomic load per launch.
#include <atomic>
#include <mutex>
#include <stdio.h>
namespace cca_prof {
constexpr int kMaxLaunches = 256;
struct State {
    std::atomic<int> armed{0};
    int n = 0, cap = 0;
    hipEvent_t e0[kMaxLaunches], e1[kMaxLaunches];
    const char *name[kMaxLaunches];
};
inline State &state() { static State s; return s; }
inline int before(const char *name, hipStream_t stream) {
    State &s = state();
    if (!s.armed.load(std::memory_order_relaxed) || s.n >= s.cap) return -1;
    const int i = s.n++;
    s.name[i] = name;
    (void)hipEventRecord(s.e0[i], stream);
    return i;
}
inline void after(int i, hipStream_t stream) {
    if (i >= 0) (void)hipEventRecord(state().e1[i], stream);
}
// arm: returns nullptr or the reason it could not
inline const char *begin(int max_launches) {
    State &s = state();
    if (s.armed.load()) return "profile_begin: already armed";
    if (max_launches <= 0 || max_launches > kMaxLaunches) return "profile_begin: 1..256 launches";
    for (int i = s.cap; i < max_launches; ++i) {              // events are created once and kept
        if (hipEventCreate(&s.e0[i]) != hipSuccess || hipEventCreate(&s.e1[i]) != hipSuccess) return "profile_begin: hipEventCreate failed";
        s.cap = i + 1;
    }
    s.n = 0;
    s.armed.store(1);
    return nullptr;
}
// disarm, wait for the recorded launches, hand out durations and names; returns the count
inline int end(float *ms, char *names, int name_stride, int cap, const char **why) {
    State &s = state();
    if (!s.armed.load()) { *why = "profile_end: not armed"; return 0; }
    s.armed.store(0);
    const int n = s.n < cap ? s.n : cap;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(s.e1[i]) != hipSuccess || hipEventElapsedTime(&t, s.e0[i], s.e1[i]) != hipSuccess) {
            *why = "profile_end: reading an event pair failed";
            return 0;
        }
        if (ms) ms[i] = t;
        if (names && name_stride > 0) snprintf(names + (size_t)i * name_stride, (size_t)name_stride, "%s", s.name[i]);
    }
    return n;
}
}  // namespace cca_prof

// A library-owned second stream per device for launches that are independent of the caller's chain (the dv passes next to
// softmax-backward -> dq | dk of the split-plane backward): fork() makes the side stream wait for everything the caller's
// stream holds so far and hands it out, join() makes the caller's stream wait for the side stream.  Event record / wait pairs
// only: under stream capture the side stream joins the capture and comes back before it ends, i.e. the step stays one graph.
namespace cca_side {
struct Dev {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
// ``may_create`` false: hand out the device's side stream only if it exists already (a stream must not be created while the
// caller's stream is being captured: stream / event creation is not a capturable operation and may invalidate the capture)
inline Dev *dev_state(bool may_create = true) {
    static Dev devs[64];
    static std::mutex mu;
    const int d = [] { int x = 0; return hipGetDevice(&x) == hipSuccess && x >= 0 && x < 64 ? x : -1; }();
    if (d < 0) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Dev &v = devs[d];
    if (!v.s && !may_create) return nullptr;
    if (!v.s) {
        hipStream_t s = nullptr;
        hipEvent_t f = nullptr, j = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&j, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        v.fork = f; v.join = j; v.s = s;
    }
    return &v;
}
// The record + wait pair of a fork (of a join) is issued under one lock: host threads driving different streams of one device
// share the side stream and its two events, and a record of thread B between thread A's record and wait would make A's side
// work wait for the wrong point.  (Sharing costs such callers parallelism, never ordering: the side stream then waits for
// both callers, and a join waits for all side work enqueued so far.)
inline std::mutex &pair_lock() { static std::mutex mu; return mu; }
// nullptr when the side stream cannot be had (the caller then stays on its own stream)
inline hipStream_t fork(hipStream_t main) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    Dev *v = dev_state(cap == hipStreamCaptureStatusNone);      // first use inside a capture: stay on the caller's stream
    if (!v) return nullptr;
    std::lock_guard<std::mutex> lock(pair_lock());
    if (hipEventRecord(v->fork, main) != hipSuccess || hipStreamWaitEvent(v->s, v->fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return v->s;
}
inline bool join(hipStream_t main) {
    Dev *v = dev_state();
    if (!v) return false;
    std::lock_guard<std::mutex> lock(pair_lock());
    return hipEventRecord(v->join, v->s) == hipSuccess && hipStreamWaitEvent(main, v->join, 0) == hipSuccess;
}
}  // namespace cca_side

// kernel launch on a caller-given stream; a stale error of an earlier, unrelated HIP call is cleared first so that
// launch_status() reports this launch and nothing else
#define CCA_LAUNCH(kern, grid, block, stream, ...)                                        \
    do {                                                                                   \
        (void)hipGetLastError();                                                           \
        const int cca_prof_i_ = cca_prof::before(#kern, (hipStream_t)(stream));            \
        hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__); \
        cca_prof::after(cca_prof_i_, (hipStream_t)(stream));                               \
    } while (0)

// number of compute units of the CURRENT device (the channel splits are balanced for it)
inline int cca_current_device_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return prop.multiProcessorCount;
}
inline int cca_current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

// hip_emu.hpp -- a tiny host-side SIMT emulator used ONLY by the CPU test-suite.
//
// The build container has no GPU, so the kernels in ccnet_amd/csrc/ would otherwise first execute
// on the (scarce) MI355X box.  This shim lets the *same kernel source* be compiled by the host
// clang++ and executed on the CPU: every HIP thread of a workgroup is a fiber; __syncthreads(),
// wave shuffles and the f32 MFMA are rendez-vous points between fibers.  Fibers run until they
// block, so a missing barrier shows up as a wrong answer instead of going unnoticed, LDS is
// poisoned with NaNs per workgroup, and LDS bank conflicts of ds_read/ds_write_b32 are counted.
//
// It is test infrastructure: nothing in the product (ccnet_amd/, cc_attention/) links or loads it,
// and the library it produces (tests/emu/libcca_emu.so) is never on the GPU path.
#pragma once
#ifndef CCNET_EMU
#define CCNET_EMU 1
#endif

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
enum { hipMemcpyDeviceToHost = 2 };
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { std::memcpy(d, s, n); return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

namespace emu {

struct Lane {
    dim3 tid;
};
extern dim3 g_block, g_bdim, g_gdim;
extern Lane *g_cur;

void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void block_barrier();
// exchange one 64-bit payload per lane across the calling lane's wave; returns pointer to the
// wave's 64 payload slots (valid until the lane's next collective).
const uint64_t *wave_exchange(uint64_t mine);
// wider payloads: up to 32 bytes per lane; returns the wave's 64 slots of 32 bytes each
const unsigned char *wave_exchange_bytes(const void *mine, size_t nbytes);
int lane_id();

// LDS instrumentation: kernels register their __shared__ array so it can be poisoned and so
// that bank conflicts of 4-byte accesses can be counted per wave-instruction.
void lds_register(void *base, size_t bytes);
void lds_note_read(const void *addr, int site);
void lds_note_write(const void *addr, int site);
struct Stats {
    unsigned long long lds_read_instr, lds_read_cycles, lds_write_instr, lds_write_cycles, mfma, launches;
};
Stats &stats();

}  // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_block)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)
inline void __syncthreads() { emu::block_barrier(); }

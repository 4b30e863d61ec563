// cca_platform.hpp (tests/emu) -- SIMT-emulator implementations of the device primitives of
// ccnet_amd/csrc/cca_platform.hpp, so that the CPU test-suite can execute the kernel sources without a GPU.
// Test infrastructure only: the emulator build puts this directory FIRST on the include path; the product build
// never does.
#pragma once
#include "hip_emu.hpp"

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace cca {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 8 packed bf16 (element e in dword e/2, low half = even e)
constexpr int kWave = 64;                 // CDNA wavefront


__device__ inline int lane_id() { return emu::lane_id(); }

__device__ inline float shfl_xor(float v, int mask) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    const uint64_t *s = emu::wave_exchange(bits);
    uint32_t o = uint32_t(s[emu::lane_id() ^ mask]);
    float r;
    memcpy(&r, &o, 4);
    return r;
}

// D = A(16x4) * B(4x16) + C, v_mfma_f32_16x16x4_f32 layout (cdna_hip_programming.md section 3):
//   a: lane l holds A[i = l & 15][k = l >> 4];  b: lane l holds B[k = l >> 4][j = l & 15]
//   c/d: lane l, reg r holds D[row = 4 * (l >> 4) + r][col = l & 15]
// bit-for-bit a k-ordered fmaf chain.
__device__ inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    uint32_t ab[2];
    memcpy(&ab[0], &a, 4);
    memcpy(&ab[1], &b, 4);
    uint64_t payload = uint64_t(ab[0]) | (uint64_t(ab[1]) << 32);
    const uint64_t *s = emu::wave_exchange(payload);
    const int l = emu::lane_id(), col = l & 15, rg = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            uint32_t ua = uint32_t(s[k * 16 + row]), ub = uint32_t(s[k * 16 + col] >> 32);
            float fa, fb;
            memcpy(&fa, &ua, 4);
            memcpy(&fb, &ub, 4);
            acc = fmaf(fa, fb, acc);
        }
        d[r] = acc;
    }
    emu::stats().mfma++;
    return d;
}

// cycle counters of the probe kernels: the emulator has no clocks; a process-wide tick keeps the probe loops finite
__device__ inline uint64_t shader_clock() { static uint64_t t = 0; return t += 24; }
__device__ inline uint64_t ref_clock() { static uint64_t t = 0; return ++t; }
__device__ inline void short_sleep() {}
__device__ inline int xcc_id() { return 0; }
__device__ inline void atomic_max_u32(unsigned *p, uint32_t v) { if (v > *p) *p = v; }
__device__ inline void mfma_f32_result_fence() {}
__device__ inline void sched_fence() {}

__device__ inline void wait_vmem_all() {}
__device__ inline int uniform(int v) { return v; }
__device__ inline int recompute_here(int v) { return v; }

// round-to-nearest-even fp32 -> bf16 (as the device's v_cvt_pk_bf16_f32), two values into one dword
__device__ inline uint32_t emu_bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;     // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ inline uint32_t cvt_pk_bf16(float lo_elem, float hi_elem) {
    return emu_bf16_rne(lo_elem) | (emu_bf16_rne(hi_elem) << 16);
}
__device__ inline float emu_bf16_to_f32(uint32_t h) {
    uint32_t u = h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// D = A(16x32) * B(32x16) + C for v_mfma_f32_16x16x32_bf16:
//   a: lane l holds A[i = l & 15][k = 8 (l >> 4) + e], e = 0..7;   b: lane l holds B[k = 8 (l >> 4) + e][j = l & 15]
//   c/d as the f32 16x16 forms.  Products are exact in fp32; the emulator sums them in double.
__device__ inline f32x4 mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    uint32_t mine[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const unsigned char *s = emu::wave_exchange_bytes(mine, 32);
    const int l = emu::lane_id(), col = l & 15, rg = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        double acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            uint32_t wa[8], wb[8];
            memcpy(wa, s + size_t(kg * 16 + row) * 32, 32);
            memcpy(wb, s + size_t(kg * 16 + col) * 32, 32);
            for (int e = 0; e < 8; ++e) {
                const uint32_t ha = (wa[e / 2] >> (16 * (e & 1))) & 0xffffu;
                const uint32_t hb = (wb[4 + e / 2] >> (16 * (e & 1))) & 0xffffu;
                acc += (double)emu_bf16_to_f32(ha) * (double)emu_bf16_to_f32(hb);
            }
        }
        d[r] = (float)acc;
    }
    emu::stats().mfma++;
    return d;
}

// Read-only view of one image's worth of a tensor, addressed by (per-lane byte offset) +
// (wave-uniform byte offset).  Out-of-range reads return 0 like a raw buffer resource.
struct FBuf {
    const char *base;
    uint32_t bytes;
};
__device__ inline FBuf make_fbuf(const float *p, size_t bytes) { return FBuf{(const char *)p, (uint32_t)bytes}; }
__device__ inline float fbuf_load(const FBuf &b, int voff_bytes, int soff_bytes) {
    const uint32_t o = (uint32_t)voff_bytes + (uint32_t)soff_bytes;
    if ((uint32_t)voff_bytes >= b.bytes || (size_t)o + 4 > b.bytes) return 0.f;
    float r;
    memcpy(&r, b.base + o, 4);
    return r;
}
__device__ inline f32x4 fbuf_load_x4(const FBuf &b, int voff_bytes, int soff_bytes) {
    f32x4 v;
    for (int e = 0; e < 4; ++e) v[e] = fbuf_load(b, voff_bytes + 4 * e, soff_bytes);
    return v;
}
__device__ inline void fbuf_store(const FBuf &b, float v, int voff_bytes, int soff_bytes) {
    const uint32_t o = (uint32_t)voff_bytes + (uint32_t)soff_bytes;
    if ((uint32_t)voff_bytes >= b.bytes || (size_t)o + 4 > b.bytes) return;      // out-of-range stores are dropped
    memcpy(const_cast<char *>(b.base) + o, &v, 4);
}
__device__ inline void fbuf_store_x4(const FBuf &b, f32x4 v, int voff_bytes, int soff_bytes) {
    for (int e = 0; e < 4; ++e) fbuf_store(b, v[e], voff_bytes + 4 * e, soff_bytes);
}
__device__ inline void fbuf_store_x2(const FBuf &b, uint32_t v0, uint32_t v1, int voff_bytes, int soff_bytes) {
    float f0, f1;
    memcpy(&f0, &v0, 4);
    memcpy(&f1, &v1, 4);
    fbuf_store(b, f0, voff_bytes, soff_bytes);
    fbuf_store(b, f1, voff_bytes + 4, soff_bytes);
}
__device__ inline f32x4 lds_load_x4(const float *p) {
    f32x4 v;
    memcpy(&v, p, 16);
    return v;
}
__device__ inline void lds_store_x4(float *p, f32x4 v) { memcpy(p, &v, 16); }
// LDS-DMA: every lane fetches one dword and the wave deposits the 64 dwords CONTIGUOUSLY at
// lds_wave_base + lane (buffer_load_dword ... lds).  The emulator completes it synchronously.
__device__ inline void fbuf_load_to_lds(const FBuf &b, float *lds_wave_base, int voff_bytes, int soff_bytes) {
    lds_wave_base[emu::lane_id()] = fbuf_load(b, voff_bytes, soff_bytes);
}
// 16-byte form: every lane moves 4 consecutive dwords to lds_wave_base + 4 * lane
__device__ inline void fbuf_load_to_lds_x4(const FBuf &b, float *lds_wave_base, int voff_bytes, int soff_bytes) {
    for (int e = 0; e < 4; ++e)
        lds_wave_base[4 * emu::lane_id() + e] = fbuf_load(b, voff_bytes + 4 * e, soff_bytes);
}

__device__ inline void fbuf_load_to_lds_x4_uncounted(const FBuf &b, float *lds_wave_base, int voff_bytes) {
    fbuf_load_to_lds_x4(b, lds_wave_base, voff_bytes, 0);
}

// ds_read_b64_tr_b16: see the product header.  Lane i of each 16-lane group receives, for j = 0..3, element (i & 3) of
// the 8 bytes addressed by lane 4 j + (i >> 2) of its group.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ inline u32x2 lds_read_tr16_b64(const void *p) {
    const uint64_t *s = emu::wave_exchange((uint64_t)(uintptr_t)p);
    const int l = emu::lane_id(), grp = l & ~15, i = l & 15;
    uint16_t e[4];
    for (int j = 0; j < 4; ++j) {
        const uint16_t *src = (const uint16_t *)(uintptr_t)s[grp + 4 * j + (i >> 2)];
        memcpy(&e[j], src + (i & 3), 2);
    }
    return u32x2{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16)};
}

template <int OFF>
__device__ inline u32x4 lds_read_x4_uncounted(const void *p) {
    u32x4 r;
    memcpy(&r, (const char *)p + OFF, 16);
    return r;
}
template <int KEEP>
__device__ inline void lds_wait_keep(u32x4 (&)[4], u32x4 (&)[4]) {}

__device__ inline void barrier_lds_only() { __syncthreads(); }
template <int KEEP>
__device__ inline void barrier_dma_keep() { __syncthreads(); }

#define CCA_LDS_REGISTER(arr) do { emu::lds_register((void *)(arr), sizeof(arr)); __syncthreads(); } while (0)
#define CCA_LDS_LD(p) (emu::lds_note_read((const void *)(p), __LINE__), *(p))
#define CCA_LDS_ST(p, v) do { emu::lds_note_write((const void *)(p), __LINE__); *(p) = (v); } while (0)


}  // namespace cca

#define CCA_LAUNCH(kern, grid, block, stream, ...) emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })

// the launch profiler measures HIP events: nothing to measure in the emulator
namespace cca_prof {
inline const char *begin(int) { return "profile_begin: not available in the emulator build"; }
inline int end(float *, char *, int, int, const char **why) { *why = "profile_end: not available in the emulator build"; return 0; }
}  // namespace cca_prof

// the emulator runs every launch to completion: there is nothing to overlap, the "side stream" is the caller's
namespace cca_side {
inline hipStream_t fork(hipStream_t) { return nullptr; }
inline bool join(hipStream_t) { return true; }
}  // namespace cca_side

inline int cca_current_device_cus() { return 0; }      // the host default (256) applies
inline int cca_current_device() { return 0; }

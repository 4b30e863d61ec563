// cca_common.hpp -- device-side helpers shared by the criss-cross attention kernels (gfx950).
//
// Geometry vocabulary used throughout csrc/:
//   strip     one column (b, :, w) or one row (b, h, :) of the feature map; the unit a wavefront
//             owns.  Column strips have length H and there are W of them per image; row strips
//             have length W and there are H of them.
//   slot      index into the (H+W)-wide attention axis (reference ``concate`` order,
//             /root/reference/cc_attention/functions.py:38-40): slot j < H  <-> source pixel (j, w),
//             slot H + j <-> source pixel (h, j).
#pragma once

#ifdef CCNET_EMU
#include "hip_emu.hpp"
#else
#include <hip/hip_runtime.h>
#endif

#include <math.h>
#include <stdint.h>

namespace cca {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;                 // CDNA wavefront
constexpr int kStripsPerBlock = 8;        // one wavefront per strip, 8 strips (=8 adjacent w or h) per workgroup
constexpr int kBlock = kWave * kStripsPerBlock;
constexpr int kTile = 16;                 // v_mfma_f32_16x16x4_f32 output tile
constexpr int kMaxTiles = 7;              // strips up to 112 long in the W-kernel
constexpr int kMaxStrip = 100;            // strip-stationary kernels hold 25 k-steps x 7 n-tiles of attention

// Branch geometry: how a (strip g, position i) pair maps to feature / attention addresses.
struct Branch {
    int L;        // strip length (H for the column branch, W for the row branch)
    int G;        // strips per image (W resp. H)
    int fs_i;     // feature stride (elements) along the strip            (W resp. 1)
    int fs_g;     // feature stride between adjacent strips               (1 resp. W)
    int as_q;     // attention stride (elements) of the query position    (W*S resp. S)
    int as_g;     // attention stride between adjacent strips             (S resp. W*S)
    int a_off;    // first slot of this branch                            (0 resp. H)
};

__host__ __device__ inline Branch make_branch(bool row, int H, int W) {
    const int S = H + W;
    Branch g;
    if (row) { g.L = W; g.G = H; g.fs_i = 1; g.fs_g = W; g.as_q = S; g.as_g = W * S; g.a_off = H; }
    else     { g.L = H; g.G = W; g.fs_i = W; g.fs_g = 1; g.as_q = W * S; g.as_g = S; g.a_off = 0; }
    return g;
}

// ---------------------------------------------------------------------------------------------
// wave-level primitives.  On the device these are single instructions; under CCNET_EMU they are
// rendez-vous points of the fiber scheduler (tests/emu/hip_emu.cpp).
// ---------------------------------------------------------------------------------------------
#ifdef CCNET_EMU

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

__device__ inline int uniform(int v) { return v; }

// Read-only view of one image's worth of a tensor, addressed by (per-lane byte offset) +
// (wave-uniform byte offset).  Out-of-range reads return 0 like a raw buffer resource.
struct FBuf {
    const char *base;
    uint32_t bytes;
};
__device__ inline FBuf make_fbuf(const float *p, size_t bytes) { return FBuf{(const char *)p, (uint32_t)bytes}; }
__device__ inline float fbuf_load(const FBuf &b, int voff_bytes, int soff_bytes) {
    const uint32_t o = (uint32_t)voff_bytes + (uint32_t)soff_bytes;
    if ((size_t)o + 4 > b.bytes) return 0.f;
    float r;
    memcpy(&r, b.base + o, 4);
    return r;
}

#define CCA_LDS_REGISTER(arr) do { emu::lds_register((void *)(arr), sizeof(arr)); __syncthreads(); } while (0)
#define CCA_LDS_LD(p) (emu::lds_note_read((const void *)(p), __LINE__), *(p))
#define CCA_LDS_ST(p, v) do { emu::lds_note_write((const void *)(p), __LINE__); *(p) = (v); } while (0)

#else  // ----- real gfx950 -----

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// tell the compiler a value is wave-uniform (it is: derived from the wave id) so it lives in SGPRs
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

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

#define CCA_LDS_REGISTER(arr) do { } while (0)
#define CCA_LDS_LD(p) (*(p))
#define CCA_LDS_ST(p, v) do { *(p) = (v); } while (0)

#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

}  // namespace cca

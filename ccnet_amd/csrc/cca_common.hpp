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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 8 packed bf16 (element e in dword e/2, low half = even e)

constexpr int kWave = 64;                 // CDNA wavefront
// Strip kernels are templated on NS = strips per workgroup (one wavefront per strip, NS adjacent w or h):
//   NS = 8: 512 threads, one workgroup per CU;  NS = 4: 256 threads, two independent workgroups per CU
constexpr int kMaxStripsPerBlock = 8;
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
__device__ inline void fbuf_store(const FBuf &b, float v, int voff_bytes, int soff_bytes) {
    const uint32_t o = (uint32_t)voff_bytes + (uint32_t)soff_bytes;
    if ((uint32_t)voff_bytes >= b.bytes || (size_t)o + 4 > b.bytes) return;      // out-of-range stores are dropped
    memcpy(const_cast<char *>(b.base) + o, &v, 4);
}
__device__ inline void fbuf_store_x4(const FBuf &b, f32x4 v, int voff_bytes, int soff_bytes) {
    for (int e = 0; e < 4; ++e) fbuf_store(b, v[e], voff_bytes + 4 * e, soff_bytes);
}
__device__ inline f32x4 lds_load_x4(const float *p) {
    f32x4 v;
    memcpy(&v, p, 16);
    return v;
}
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

__device__ inline void barrier_lds_only() { __syncthreads(); }
template <int KEEP>
__device__ inline void barrier_dma_keep() { __syncthreads(); }

#define CCA_LDS_REGISTER(arr) do { emu::lds_register((void *)(arr), sizeof(arr)); __syncthreads(); } while (0)
#define CCA_LDS_LD(p) (emu::lds_note_read((const void *)(p), __LINE__), *(p))
#define CCA_LDS_ST(p, v) do { emu::lds_note_write((const void *)(p), __LINE__); *(p) = (v); } while (0)

#else  // ----- real gfx950 -----

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
// stores whose per-lane offset is out of range (kOobOffset) are dropped by the buffer range check
__device__ __forceinline__ void fbuf_store(const FBuf &b, float v, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), b, voff_bytes, soff_bytes, 0);
}
// 16-byte store (global address needs only 4-byte alignment, like the 16-byte loads)
__device__ __forceinline__ void fbuf_store_x4(const FBuf &b, f32x4 v, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, voff_bytes, soff_bytes, 0);
}
// ds_read_b128: p must be 16-byte aligned
__device__ __forceinline__ f32x4 lds_load_x4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
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

// Split-bf16: x = hi + lo + O(2^-17 |x|) with hi = bf16_rne(x), lo = bf16_rne(x - hi).  A product a*b is then
// replaced by a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix pipe (16x the fp32 MFMA rate, 3 products:
// ~5x net), fp32 accumulate; the dropped terms are O(2^-16 |a b|)  (SURVEY.md section 0, fact 5).
struct BfSplit {
    u32x4 hi, lo;
};
__device__ __forceinline__ BfSplit bf16_split8(const float (&x)[8]) {
    BfSplit s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t h = cvt_pk_bf16(x[2 * p], x[2 * p + 1]);
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
        s.hi[p] = h;
        s.lo[p] = cvt_pk_bf16(x[2 * p] - h0, x[2 * p + 1] - h1);
    }
    return s;
}

// "Packed" split-bf16: the four products of ONE channel are laid along the K axis of a single
// v_mfma_f32_16x16x32_bf16 (8 channels x 4 k-slots):
//     A side (a_hi, a_lo, a_hi, a_lo)   B side (b_hi, b_hi, b_lo, b_lo)   ->  (a_hi + a_lo) * (b_hi + b_lo)
// so a K = 8-channel chunk needs one MFMA instead of two f32 k-steps (4x fewer matrix-pipe cycles) while the LDS
// image stays 8 channels deep; what is lost is only the 2^-18 |x| residual of the two-term split of each operand.
// A lane holds the slots of channels 2 (l >> 4) and 2 (l >> 4) + 1.
__device__ __forceinline__ u32x4 bf16_pack_a(float x0, float x1) {
    const uint32_t h0 = cvt_pk_bf16(x0, x0), h1 = cvt_pk_bf16(x1, x1);                   // (hi, hi)
    const float r0 = x0 - __builtin_bit_cast(float, h0 << 16), r1 = x1 - __builtin_bit_cast(float, h1 << 16);
    const uint32_t l0 = cvt_pk_bf16(r0, r0), l1 = cvt_pk_bf16(r1, r1);                    // (lo, lo)
    const uint32_t d0 = (h0 & 0xffffu) | (l0 << 16), d1 = (h1 & 0xffffu) | (l1 << 16);    // (hi, lo)
    return u32x4{d0, d0, d1, d1};
}
__device__ __forceinline__ u32x4 bf16_pack_b(float y0, float y1) {
    const uint32_t h0 = cvt_pk_bf16(y0, y0), h1 = cvt_pk_bf16(y1, y1);                   // (hi, hi)
    const float r0 = y0 - __builtin_bit_cast(float, h0 << 16), r1 = y1 - __builtin_bit_cast(float, h1 << 16);
    return u32x4{h0, cvt_pk_bf16(r0, r0), h1, cvt_pk_bf16(r1, r1)};                       // (hi, hi), (lo, lo)
}

// XCD-aware workgroup order.  The dispatcher places workgroup L on XCD L % 8, each with a private 4 MiB L2
// (MI355X_MICROARCH.md, workgroup dispatch).  Neighbouring strip tiles of one image share every 128-byte
// line of the column branch (a tile only uses 4*NS bytes of it), so they must sit on the SAME XCD or each
// L2 re-fetches the line from HBM.  This bijection hands every XCD a contiguous range of logical ids
// (cdna_hip_programming.md T1); the kernels then decode image-major / tile-fastest from the logical id.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_logical_id(int linear, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = linear & 7, idx = linear >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---------------------------------------------------------------------------------------------
// Storage types of the feature tensors.  Arithmetic is fp32 everywhere; bf16 tensors (BASELINE configs[4]) are
// raw 16-bit patterns widened on load and rounded to nearest-even on store (NaN stays NaN), the same rounding as
// v_cvt_pk_bf16_f32 and torch's float -> bfloat16 conversion.
// ---------------------------------------------------------------------------------------------
struct bf16_t { uint16_t bits; };
__host__ __device__ inline float load_f32(const float *p) { return *p; }
__host__ __device__ inline void store_f32(float *p, float v) { *p = v; }
__host__ __device__ inline float load_f32(const bf16_t *p) {
    const uint32_t u = (uint32_t)p->bits << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline void store_f32(bf16_t *p, float v) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    p->bits = (uint16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16));
}

// ---------------------------------------------------------------------------------------------
// Strip-tile geometry shared by the strip kernels: the NS strips x L positions of ONE channel form a
// linear LDS image, element (position i, strip gg) at index p:
//   column branch  p = i * NS + (gg ^ swz(i))    NS consecutive w per position; the swizzle only swaps the two
//                  16-byte halves (swz = 4 * ((i >> 2) & 1), NS = 8) so that a 16-byte DMA lane stays contiguous
//                  in memory while the stride-NS fragment reads spread over more banks
//   row branch     p = gg * L + i                the NS rows are contiguous in memory
// The image is filled by 16-byte LDS-DMA pieces of 256 floats (lane l moves p = 256 m + 4 l .. + 3: 16 B of one
// (c, i) position in the column branch, 16 B of a row in the row branch).  Lanes whose data would lie outside the
// strip tile are masked.  (How the map kernel drains its result images is described in cca_map.hpp.)
// ---------------------------------------------------------------------------------------------
constexpr int kOobOffset = 0x7ffffff0;            // per-lane byte offset that is out of range for every view

__host__ __device__ constexpr int strip_pieces_c(int ns) { return (ns * kMaxStrip + 63) / 64; }      // 13 / 7 dword pieces
__host__ __device__ constexpr int strip_pieces4_c(int ns) { return (ns * kMaxStrip + 255) / 256; }   // 4 / 2 16-byte pieces

template <int NS>
__device__ __forceinline__ int col_swizzle(int i) {
    return NS == 8 ? 4 * ((i >> 2) & 1) : 0;
}

// LDS index of (position i, strip gg) inside a channel image
template <int NS, bool ROW>
__device__ __forceinline__ int strip_lds_index(int i, int gg, int L) {
    return ROW ? gg * L + i : i * NS + (gg ^ col_swizzle<NS>(i));
}

// Per-lane addressing of the 16-byte DMA pieces.  Piece m covers LDS indices [256 m, 256 m + 256); lane l moves
// the 4 consecutive elements starting at plane[vb + piece_soff(m)] when valid(m).  A valid lane may fetch up to
// 3 elements past its strip tile (next strips / next row: mapped memory inside the view, or 0 beyond it); they
// land in LDS slots of strips whose results are never stored.
template <int NS, bool ROW>
struct StripLanes4 {
    int vb;
    bool okg;
    int li;
    int lim;

    __device__ __forceinline__ void init(int lane, int L, int W, int g0, int gvalid) {
        if (ROW) {
            li = 4 * lane;
            lim = gvalid * L;
            vb = 4 * (g0 * W + 4 * lane);
            okg = true;
        } else {
            constexpr int GPP = NS / 4;                    // 16-byte groups per position (2 for NS = 8, 1 for NS = 4)
            li = lane / GPP;                               // position inside the piece
            const int grp = lane % GPP;                    // LDS half
            const int gg0 = 4 * grp ^ col_swizzle<NS>(li); // first strip of the group in memory
            okg = gg0 < gvalid;
            vb = 4 * (li * W + g0 + gg0);
            lim = L;
        }
    }
    __device__ __forceinline__ int piece_soff(int m, int W) const {
        return ROW ? m * 1024 : m * (256 / NS) * W * 4;
    }
    __device__ __forceinline__ bool valid(int m) const {
        return ROW ? (m * 256 + li < lim) : (okg && m * (256 / NS) + li < lim);
    }
};

// ---------------------------------------------------------------------------------------------
// Layout of the column -> row PARTIAL SUM of the map kernels (cca_map.hpp).  The column launch owns NS adjacent
// columns, the row launch NS adjacent rows; in the tensor's natural layout the column launch would write 4*NS-byte
// segments (32 B: measured 2.7 TB/s for a copy against ~5 TB/s for whole rows, tools/probes/seg_bw_probe.hip).
// The partial sums are internal -- the row launch rewrites the same memory with the final values -- so they are
// stored PERMUTED inside each band of NS rows [k*NS, k*NS + nr), which is exactly the memory a row workgroup
// owns:   (h, w) -> k*NS*W + (w/4)*4*nr + (h - k*NS)*nw + w%4,   nr = rows in the band, nw = min(4, W - 4*(w/4)).
// 16-byte granules are 4 consecutive w of one row (what the row launch stores), and the 4*NS granules of a
// (band, NS columns) tile are contiguous (NS*NS*4 = 256 B at NS = 8) -- what the column launch stores.
// ---------------------------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ int blocked_offset(int h, int w, int H, int W) {
    const int k = h / NS, hh = h - k * NS;
    const int nr = (H - k * NS < NS) ? H - k * NS : NS;
    const int g = w >> 2;
    const int nw = (W - 4 * g < 4) ? W - 4 * g : 4;
    return k * NS * W + g * 4 * nr + hh * nw + (w & 3);
}

// LDS-DMA of one channel plane slice into its image (FULL: npieces4 is the compile-time maximum)
template <int NS, bool ROW, bool FULL>
__device__ __forceinline__ void strip_dma_channel(const FBuf &src, float *dst, int soff, int npieces4, int W,
                                                  const StripLanes4<NS, ROW> &sl) {
#pragma unroll
    for (int m = 0; m < strip_pieces4_c(NS); ++m)
        if (FULL || m < npieces4)
            if (sl.valid(m)) fbuf_load_to_lds_x4(src, dst + m * 256, sl.vb, soff + sl.piece_soff(m, W));
}

}  // namespace cca

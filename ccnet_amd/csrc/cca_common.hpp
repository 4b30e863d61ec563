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

// The device primitives (wave shuffles, MFMA wrappers, buffer resources, LDS-DMA, counted barriers, the launch
// macro) live in <cca_platform.hpp>.  The product build finds ccnet_amd/csrc/cca_platform.hpp (gfx950 intrinsics);
// the CPU test-suite puts tests/emu/ first on the include path and gets the SIMT-emulator implementations of the
// same names -- nothing emulator-related lives in this directory.
#include <cca_platform.hpp>

#include <math.h>
#include <stdint.h>

namespace cca {

// f32x4 / u32x4 (8 packed bf16: element e in dword e/2, low half = even e) and kWave = 64 come from the platform header
// Strip kernels are templated on NS = strips per workgroup (one wavefront per strip, NS adjacent w or h):
//   NS = 8: 512 threads, one workgroup per CU;  NS = 4: 256 threads, two independent workgroups per CU
constexpr int kMaxStripsPerBlock = 8;
constexpr int kTile = 16;                 // v_mfma_f32_16x16x4_f32 output tile
constexpr int kMaxTiles = 7;              // strips up to 112 long in the W-kernel
constexpr int kMaxStrip = 100;            // strip-stationary kernels hold 25 k-steps x 7 n-tiles of attention
// Strips at least this long run the compile-time-shaped ("FULL") bodies: all 7 tiles / 25 k-steps are computed whatever
// the length (positions beyond the strip are zero operands), every DMA piece and tile store is ISSUED whatever the
// length (lanes beyond the strip carry an out-of-range buffer offset: such a load deposits zeros in LDS, such a store
// is dropped -- tools/probes/dma_oob_probe.hip), so the counted-vmcnt pipeline and the 16-byte store paths hold for
// 96, 80, 65 ... as they do for 97..100.  Shorter strips take the run-time-shaped bodies.
constexpr int kFullMinStrip = 49;

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

// Three-way split: x = hi + mid + lo EXACTLY up to 2^-25 |x| (three 8-bit significands cover fp32's 24 bits).  A product is
// then the six terms of weight >= 2^-16 -- hi hi, hi mid, mid hi, hi lo, mid mid, lo hi -- on the bf16 matrix pipe with fp32
// accumulation: what is dropped is 2^-24 of the product, i.e. fp32 arithmetic (ca_backward, whose gradients scale with the
// logits: the three-product form leaves ~1.2e-5 of the gradient's magnitude, tests: logit-scale sweep).  6 x 16 cycles per
// k-step of 32 against 8 x 32 for v_mfma_f32_16x16x4_f32.
struct BfSplit3 {
    u32x4 hi, mid, lo;
};
__device__ __forceinline__ BfSplit3 bf16_split8x3(const float (&x)[8]) {
    BfSplit3 s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t h = cvt_pk_bf16(x[2 * p], x[2 * p + 1]);
        const float r0 = x[2 * p] - __builtin_bit_cast(float, h << 16), r1 = x[2 * p + 1] - __builtin_bit_cast(float, h & 0xffff0000u);
        const uint32_t m = cvt_pk_bf16(r0, r1);
        s.hi[p] = h;
        s.mid[p] = m;
        s.lo[p] = cvt_pk_bf16(r0 - __builtin_bit_cast(float, m << 16), r1 - __builtin_bit_cast(float, m & 0xffff0000u));
    }
    return s;
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
    bool inimg_last;                 // this lane's slot of the LAST piece lies inside the channel image

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
        inimg_last = (strip_pieces4_c(NS) - 1) * 256 + 4 * lane < strip_pieces_c(NS) * 64;
    }
    __device__ __forceinline__ int piece_soff(int m, int W) const {
        return ROW ? m * 1024 : m * (256 / NS) * W * 4;
    }
    __device__ __forceinline__ bool valid(int m) const {
        return ROW ? (m * 256 + li < lim) : (okg && m * (256 / NS) + li < lim);
    }
    // compile-time-shaped bodies issue EVERY piece: a lane whose 16 bytes lie inside the channel image but outside the
    // strip tile fetches from an out-of-range offset (zeros land in LDS); lanes past the image (the last piece overhangs
    // it: the next channel's image starts there) stay masked -- never a whole piece, so the instruction count is fixed
    // (only the last piece overhangs: strip_pieces_c * 64 > (strip_pieces4_c - 1) * 256; m is a compile-time constant
    // at every call site, so the earlier pieces carry no mask at all)
    __device__ __forceinline__ bool in_image(int m, int) const { return m < strip_pieces4_c(NS) - 1 || inimg_last; }
    __device__ __forceinline__ int full_offset(int m) const { return valid(m) ? vb : kOobOffset; }
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

// LDS-DMA of one channel plane slice into its image (FULL: npieces4 is the compile-time maximum; EXACT: strips
// 97..100 long, where every piece of a full tile holds lanes of the tile and plain lane masks keep the instruction count)
template <int NS, bool ROW, bool FULL, bool EXACT = false>
__device__ __forceinline__ void strip_dma_channel(const FBuf &src, float *dst, int soff, int npieces4, int W,
                                                  const StripLanes4<NS, ROW> &sl) {
#pragma unroll
    for (int m = 0; m < strip_pieces4_c(NS); ++m)
        if (FULL && !EXACT) {
            if (sl.in_image(m, lane_id())) fbuf_load_to_lds_x4(src, dst + m * 256, sl.full_offset(m), soff + sl.piece_soff(m, W));
        } else if (FULL || m < npieces4) {
            if (sl.valid(m)) fbuf_load_to_lds_x4(src, dst + m * 256, sl.vb, soff + sl.piece_soff(m, W));
        }
}

// counted barrier with a run-time (wave-uniform) count.  The vmcnt field holds 0 .. 63; a count above 63 waits with 63
// (keeping FEWER operations in flight than allowed is always safe), a negative one drains.
__device__ __forceinline__ void barrier_dma_keep_n(int n) {
    if (n > 63) n = 63;
    switch (n) {
#define CCA_KEEP(N_) case N_: barrier_dma_keep<N_>(); break;
#define CCA_KEEP8(B_) CCA_KEEP(B_) CCA_KEEP(B_ + 1) CCA_KEEP(B_ + 2) CCA_KEEP(B_ + 3) CCA_KEEP(B_ + 4) CCA_KEEP(B_ + 5) CCA_KEEP(B_ + 6) CCA_KEEP(B_ + 7)
        CCA_KEEP8(0) CCA_KEEP8(8) CCA_KEEP8(16) CCA_KEEP8(24) CCA_KEEP8(32) CCA_KEEP8(40) CCA_KEEP8(48) CCA_KEEP8(56)
#undef CCA_KEEP8
#undef CCA_KEEP
        default: barrier_dma_keep<0>(); break;
    }
}

}  // namespace cca

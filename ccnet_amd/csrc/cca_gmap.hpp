// cca_gmap.hpp -- the strip contractions on PIXEL-MAJOR features: one strip per workgroup, channel groups of 64
// streaming through LDS.  Feature element type fp32 or bf16 (BASELINE configs[4]: bf16 I/O, fp32 attention / softmax /
// accumulation).
//
//   gmap_kernel     TRANS = false   out[pixel(i, g), c] (+)= alpha * sum_j P_g[i][j] * F[pixel(j, g), c]
//                   TRANS = true    out[pixel(j, g), c] (+)= alpha * sum_i P_g[i][j] * F[pixel(i, g), c]
//                   with P_g[i][j] = T[b, pixel(i, g), a_off + j]  (cca_map.hpp has the same contractions on NCHW)
//   gweight_kernel  T[b, pixel(i, g), a_off + j] = sum_c X[pixel(i, g), c] * Y[pixel(j, g), c]
//
// Why another kernel family: in NCHW the column branch of a strip tile is a stream of 32-byte segments (8 strips x
// 4 B), which the L1 / TA path serves at a third of the row rate.  With the features PIXEL-MAJOR (B, H*W, pixel
// stride) -- the layout of a channels_last tensor, and of a projection computed as x^T W^T -- a pixel's 64 channels
// are one 256-byte (fp32) / 128-byte (bf16) segment for column strips and row strips alike, the channel axis is the
// contiguous one (so the K = channel contraction of gweight reads MFMA fragments straight out of LDS), and nothing
// depends on the strip length being 97..100.
//
// gmap (MI355X): workgroup = one strip g of one image, 4 wavefronts, two workgroups per CU.  ATTENTION-STATIONARY:
// wavefront w owns the 16-position M tiles w, w + 4, w + 8 of the strip and loads their slice of P_g straight from
// global memory into MFMA fragments -- split once into bf16 hi + lo (cca_common.hpp), every k-step, 96 VGPRs at 132
// positions -- where they stay for the strip's whole life; no attention image in LDS, so a workgroup needs only the
// streaming tiles (71 KB) and two fit a CU, one's prologue and stores hiding behind the other's MFMA phase.  Per
// 64-channel group: the L x 64 feature tile arrives by LDS-DMA (double-buffered, out-of-range lanes zero-fill the
// padding); a wavefront gathers the feature fragment of each 16-channel N tile (bf16: 8 ds_read_u16 + pack, exact;
// fp32: 8 ds_read_b32 + hi/lo split) and issues v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (D^T = F^T P^T: 2 per
// tile for bf16 features, 3 for fp32; + one exact f32 step for a k remainder <= 4), so that a lane ends up with 4
// consecutive channels of one position: one ds_write_b128 into the fp32 pixel-major output image.  The image leaves as
// whole pixel rows (256 / 128 bytes), each lane adding the fp32 addend (and bf16 residual) slice it loaded into registers
// before the MFMA phase; rounded to bf16 once.  A group's stores stay in flight across the next group's barrier (counted
// vmcnt).  Measured (profiles/r02f_*): the 512-channel launches move their bytes at 4.2 - 4.6 TB/s.
#pragma once
#include "cca_common.hpp"

#include <type_traits>

namespace cca {

// k-step plan of a contraction of length L: nbf split-bf16 steps of 32, then (tail) one exact f32 step of 4 at 32 * nbf
struct BandK {
    int nbf;
    bool tail;
};
__device__ __forceinline__ BandK band_ksteps(int L) {
    const int nfull = L >> 5, rem = L & 31;
    BandK k;
    k.tail = rem > 0 && rem <= 4;
    k.nbf = nfull + (rem > 4 ? 1 : 0);
    return k;
}

constexpr int GM_CG = 64;                       // channels per group = four MFMA N tiles
constexpr int GM_WAVES = 8;                     // gweight
constexpr int GM_THREADS = GM_WAVES * 64;
constexpr int GS_WAVES = 4;                     // gmap: two workgroups per CU
constexpr int GS_THREADS = GS_WAVES * 64;
constexpr int GM_PP = 4 * GM_CG + 8;            // dwords per 4-pixel piece of an fp32 tile (+8: bank spread)
constexpr int GM_PB = 8 * GM_CG / 2 + 8;        // dwords per 8-pixel piece of a bf16 tile (+8)
constexpr int GM_BP = 68;                       // dwords per row of a bf16 attention image (136 bf16: 128 + pad)

// ---- bf16 tiles: 8 pixels x 64 channels per 1 KiB DMA piece; the 16-byte chunk q of pixel p sits at chunk
// ---- position q ^ (p & 7) (the DMA lane fetches the permuted source chunk), so that fragment reads of 16
// ---- consecutive pixels at one chunk spread over the banks
template <typename FT>
struct GTile {
    static constexpr bool BF = std::is_same<FT, bf16_t>::value;
    static constexpr int PIX = BF ? 8 : 4;                          // pixels per DMA piece
    static constexpr int PITCH = BF ? GM_PB : GM_PP;                // dwords per piece
    __host__ __device__ static constexpr int pieces(int P) { return (P + PIX - 1) / PIX; }
    __host__ __device__ static constexpr int size(int P) { return pieces(P) * PITCH; }       // dwords
};

// one DMA piece of a pixel-major tile: pixels pix0 + i * pstep (i < n), channels c0 .. c0 + 63 of a tensor with
// pixel stride ps (elements of FT); lane -> (pixel, 16-byte chunk).  Lanes beyond the strip / the channel count are
// issued out of range: the DMA deposits zeros for them, so a tile is complete (and finite) after every fill
template <typename FT>
__device__ __forceinline__ void gtile_dma_piece(const FBuf &src, float *img, int piece, int lane, int pix0, int pstep, int n,
                                                int ps, int c0, int C, int plane_off = 0) {
    if constexpr (GTile<FT>::BF) {
        const int p = lane >> 3, i = 8 * piece + p, q = (lane & 7) ^ (p & 7), c = c0 + 8 * q;
        fbuf_load_to_lds_x4_uncounted(src, img + piece * GM_PB, (i < n && c < C) ? ((pix0 + i * pstep) * ps + plane_off + c) * 2 : kOobOffset);
    } else {
        const int i = 4 * piece + (lane >> 4), c = c0 + 4 * ((lane & 15) ^ (i & 1));
        fbuf_load_to_lds_x4_uncounted(src, img + piece * GM_PP, (i < n && c < C) ? ((pix0 + i * pstep) * ps + c) * 4 : kOobOffset);
    }
}
// fp32 tiles: 4 pixels x 64 channels per 1 KiB DMA piece; the 16-byte chunk q of pixel p sits at chunk position
// q ^ (p & 1): 16 consecutive pixels reading one chunk (gweight fragments) then meet every bank group twice, not 4 times.
// dword index of channel c (0..63) of line position j inside an fp32 tile
__device__ __forceinline__ int gtile_f32_idx(int j, int c) {
    return (j >> 2) * GM_PP + (j & 3) * GM_CG + ((((c >> 2) ^ (j & 1)) << 2) | (c & 3));
}
// byte offset of channel c (0..63) of line position j inside a bf16 tile
__device__ __forceinline__ int gtile_bf_byte(int j, int c) {
    return ((j >> 3) * GM_PB + (j & 7) * 32) * 4 + ((((c >> 3) ^ (j & 7)) << 4) | ((c & 7) << 1));
}
__device__ __forceinline__ uint32_t lds_load_u16(const float *base, int byte_off) {
    return *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(base) + byte_off);
}

// ---------------------------------------------------------------------------------------------------------------
// SPLIT PLANES: an fp32 feature tensor stored PRE-SPLIT as two bf16 planes per pixel, (B, H*W, 2, C):
//     hi = bf16_rne(x) at [b][p][0][c],   lo = bf16_rne(x - hi) at [b][p][1][c]          (x = hi + lo + O(2^-17 |x|))
// Same bytes as fp32, but the consumers run the bf16 matrix pipe with NO per-fragment split in their inner loops
// ("pre-split once, consume many"): the producer of the tensor splits it once (pm_split_kernel for the value
// slice of the projection, nchw_to_planes_kernel for the module's NCHW dy).  A pixel's 64 channels of one plane are one
// 128-byte segment, for column strips and row strips alike.  Pointers are to the hi plane; the lo plane of a C-channel
// tensor starts C elements later; the pixel stride (elements) is >= 2 C.
// ---------------------------------------------------------------------------------------------------------------
struct bf16p_t { uint16_t bits; };

// "T16" tile of ONE plane: P positions x 64 channels as 1 KiB pieces of 8 positions x 128 B, no padding.  The 16-byte
// chunk q of position p sits at chunk slot q ^ (p & 7) ^ (4 * ((p >> 3) & 1)): built for ds_read_b64_tr_b16 fragment
// reads ([4 positions][16 channels] blocks per 16-lane group, two groups per LDS pass) -- the 8 rows a pass touches
// (positions r and r + 8 of two neighbouring pieces, r = 0..3 or 4..7) then cover all 64 banks exactly once.
constexpr int T16_PIECE = 256;                                                                // dwords
__host__ __device__ constexpr int t16_pieces(int P) { return (P + 7) / 8; }
__host__ __device__ constexpr int t16_size(int P) { return t16_pieces(P) * T16_PIECE; }      // dwords per plane tile
// FLIP = false drops the piece-parity term: the layout for tiles that are read by ROWS (ds_read_b128 of 8 consecutive channels
// of 16 consecutive positions, the K-contiguous fragments of the dA contraction) -- with the flip, positions p and p + 12 of a
// b128 lane group meet on the same banks (50 % LDS conflict cycles in gweight, profiles/r03g_pmc_step_summary.json).
template <bool FLIP = true>
__device__ __forceinline__ int t16_byte(int j, int c) {
    const int piece = j >> 3, r = j & 7;
    return (piece * T16_PIECE + r * 32) * 4 + ((((c >> 3) ^ r ^ (FLIP ? (piece & 1) << 2 : 0)) << 4) | ((c & 7) << 1));
}
// one DMA piece of a plane tile: positions 8 piece .. + 7 (pixels pix0 + i * pstep), channels c0 .. c0 + 63 of the plane
// that starts ``plane_off`` elements into a pixel; ps = pixel stride in elements; lanes beyond the strip / the channel
// count fetch out of range (zero fill)
template <bool FLIP = true>
__device__ __forceinline__ void t16_dma_piece(const FBuf &src, float *img, int piece, int lane, int pix0, int pstep, int n,
                                              int ps, int c0, int C, int plane_off) {
    const int p = lane >> 3, i = 8 * piece + p, q = (lane & 7) ^ (p & 7) ^ (FLIP ? (piece & 1) << 2 : 0), c = c0 + 8 * q;
    fbuf_load_to_lds_x4_uncounted(src, img + piece * T16_PIECE, (i < n && c < C) ? ((pix0 + i * pstep) * ps + plane_off + c) * 2 : kOobOffset);
}
// MFMA fragment whose K axis runs over tile positions 32 ks + 8 (lane >> 4) + e, e < 8, at channel 16 nt + (lane & 15):
// two transposing reads (cca_platform.hpp: lane i of a 16-lane group supplies the address of row i >> 2, columns
// 4 (i & 3) .. + 3 of a [4][16] block and receives column i)
__device__ __forceinline__ u32x4 t16_frag(const float *tile, int ks, int nt, int lane) {
    const int i = lane & 15, kg = lane >> 4, r = i >> 2, c = 16 * nt + 4 * (i & 3);
    const char *b = reinterpret_cast<const char *>(tile);
    const u32x2 lo = lds_read_tr16_b64(b + t16_byte(32 * ks + 8 * kg + r, c));
    const u32x2 hi = lds_read_tr16_b64(b + t16_byte(32 * ks + 8 * kg + 4 + r, c));
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// ---------------------------------------------------------------------------------------------------------------
// "F32T" tile: P positions x 64 channels of an fp32 PIXEL-MAJOR tensor as it lies in memory -- the value slice of the packed
// projection, read WITHOUT a split pass (round 4: the v -> planes pass was 308 MB of traffic and 50 us of every forward) -- for
// the dA contraction, whose fragments are ROWS of the tile (K = the channels of one position: two ds_read_b128 per lane; 8
// consecutive lanes = 8 consecutive positions at one chunk).  Unpadded 1 KiB DMA pieces of 4 positions x 256 B (a position's 64
// channels = one row of exactly 64 banks); the 16-byte chunk c4 of position j sits at slot c4 ^ f(j), f(j) = (j & 7) ^
// (((j >> 3) & 3) << 2): eight consecutive positions meet eight different slots.  (The aggregation kernels, whose fragments run
// down COLUMNS of the tile, read fp32 features through the padded GTile<float> pieces: constant strides, fewer address registers.)
// The hi | lo split happens per fragment, in registers (v_cvt_pk_bf16_f32 + subtract + v_cvt_pk): the launches that consume the
// tiles are HBM-bound with the matrix and vector pipes mostly idle.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int f32t_pieces(int P) { return (P + 3) / 4; }
__host__ __device__ constexpr int f32t_size(int P) { return f32t_pieces(P) * 256; }          // dwords
__device__ __forceinline__ int f32t_swz(int j) { return (j & 7) ^ (((j >> 3) & 3) << 2); }
__device__ __forceinline__ int f32t_idx(int j, int c) {                                       // dword index of (position j, channel c)
    return j * 64 + ((((c >> 2) ^ f32t_swz(j)) << 2) | (c & 3));
}
__device__ __forceinline__ void f32t_dma_piece(const FBuf &src, float *img, int piece, int lane, int pix0, int pstep, int n,
                                               int ps, int c0, int C) {
    const int j = 4 * piece + (lane >> 4), c = c0 + 4 * ((lane & 15) ^ f32t_swz(j));
    fbuf_load_to_lds_x4_uncounted(src, img + piece * 256, (j < n && c < C) ? ((pix0 + j * pstep) * ps + c) * 4 : kOobOffset);
}
// FT: feature element, OT: output element; the addend (ADD) is fp32 pixel-major with its own strides; resid (may be
// null) has the output's type.  4 wavefronts, two workgroups per CU: wavefront w owns the M tiles (16
// strip positions each) t = w, w + 4, w + 8 of all four 16-channel N tiles, and keeps their attention fragments -- hi and
// lo bf16 halves, every k-step -- in registers for the whole strip (A-stationary: 96 VGPRs at 132 positions).
// [channel][position] output image of the NCHW epilogue: pitch P + 4; the 4-position granule g of channel ch sits at granule
// g ^ (((ch >> 2) & 1) << 2) (whole groups of 8 granules only): the 4 lane groups of an accumulator write (channels 4 apart)
// then meet two bank halves instead of one
template <int P>
__device__ __forceinline__ int oimg_nchw_idx(int ch, int i) {
    int g = i >> 2;
    if (g < (P + 4) / 32 * 8) g ^= ((ch >> 2) & 1) << 2;
    return ch * (P + 4) + 4 * g + (i & 3);
}

// NCHW (fp32 row strips with an addend only): ``resid`` and ``out`` are NCHW tensors (batch strides rbs / obs in elements):
// the addend slices are loaded in the accumulator layout, the output image is kept [channel][position] and leaves as runs
// of W floats per channel -- the module's x and y never exist pixel-major.
// DUAL (ca_backward in one launch per branch): blockIdx.y = 0 runs the job of the ordinary arguments non-transposed (dq: features
// k), blockIdx.y = 1 runs job ``j1`` transposed (dk: features q) on the same T = dE -- twice the workgroups per launch (what
// 1-2 images per GPU need) and half the launch boundaries.  Only the prologue differs between the two.
// The two jobs of a strip read the same block of T: they are dispatched next to each other AND onto the same XCD (workgroup
// ids 8 apart -- ids go round-robin over the 8 XCDs), so that the second read of the block is served by that XCD's L2:
// linear id = 16 m + 8 job + r  <->  strip-workgroup 8 m + r.  ``nwg`` = number of strip-workgroups (ids beyond it exit).
// (Measured against "all of job 0, then all of job 1": dq | dk row pass 45.0 -> 42.9 us, step -2 us; profiles/r03r_family_compare.txt)
template <typename FT, typename OT>
struct GmapJob {
    const FT *F;
    const float *addend;
    OT *out;
    long fbs, obs;
    int fps, ops;
    int nwg;
    int nb, jblk;     // LONG (strips of 133 .. 4 x 132 positions): blocks per strip, the key block of this launch
    // DUAL: a fixed-order sum that rides on this launch (the dgamma partials of softmax-backward, which precedes the dq | dk
    // column pass on the stream): workgroup 0 adds red_n floats at red_src into red_dst[0] -- one launch less per backward
    const float *red_src;
    int red_n;
    float *red_dst;
    int xcd;          // > 0: strips per XCD of the XCD-aware strip decode (non-DUAL launches; see the kernel)
    // P3 (three-plane output, see the kernel): elements between the planes of a pixel's row, the column-sum partials of job 0
    // (``cs``: row stride ``cs_stride`` floats, pre-offset to the job's first channel) and of job 1 (DUAL)
    int p3_plane;
    float *cs, *cs1;
    int cs_stride;
};
// LONG strips: a strip of L > P positions is cut into nb blocks of ``long_block(L, nb)`` positions (the last one shorter).  A
// workgroup then owns the QUERY block I of a strip and contracts over the KEY block J: out_I (+)= F_J . A_{I,J}^T; the key blocks
// are separate launches chained through the addend (the partial is updated in place), the last one runs the epilogue.
__host__ __device__ inline int long_block(int L, int nb) { return (((L + nb - 1) / nb) + 3) & ~3; }
inline unsigned gmap_dual_grid(int nwg) { return 16u * (unsigned)((nwg + 7) / 8); }

// WPC: workgroups per CU the LDS budget is checked for (2 everywhere except the 132-position kernels on fp32 / split-plane
// features, whose two feature tiles + output image take 104 KB: one workgroup per CU)
// ABF (bf16 features, round 5): the addend -- the column partial -- is bf16 (see gmap3_kernel, OT): one 16-byte load of 8 channels
template <int P, bool ROW, bool TRANS, bool ADD, typename FT, typename OT, bool NCHW = false, bool DUAL = false, int WPC = 2, bool LONG = false,
          bool SIX = false, bool ABF = false, bool P3 = false>
__global__ __launch_bounds__(GS_THREADS, WPC) void gmap_kernel(const float *__restrict__ T, const FT *__restrict__ F,
                                                              const float *__restrict__ addend,
                                                              const OT *__restrict__ resid,
                                                              const float *__restrict__ gamma, OT *out,
                                                              int C, int H, int W, long fbs, int fps, long abs_, int aps,
                                                              long rbs, int rps, long obs, int ops, int n_whole, int split,
                                                              GmapJob<FT, OT> j1) {
    constexpr bool BF = GTile<FT>::BF, OBF = std::is_same<OT, bf16_t>::value;
    constexpr bool PL = std::is_same<FT, bf16p_t>::value;             // split planes: hi tile | lo tile, T16 geometry
    // SIX (option "dqdk_exact", the default of ca_backward on fp32 q | k): every product as the SIX bf16 terms of a three-way split
    // (bf16_split8x3) -- fp32-equivalent (2^-24), where the three-term form leaves ~1.2e-5 of the gradient's magnitude on dq / dk
    // (tests: logit-scale sweep) -- at 3/8 of the matrix time of v_mfma_f32_16x16x4_f32.  The attention block stays in registers
    // as the fp32 values it is (one channel group per strip at C/8 <= 64: nothing is reused) and is split per k-step; the
    // accumulators start from the column partial (ca_backward has no gamma), so the row pass needs no addend registers.
    // (Rounds 4-5 ran an exact-f32 MFMA form behind a device-side gate on max |dq|, |dk|: two extra launches per step and a step
    //  time that depended on the data; profiles/r06a_ab_row_pieces_six_terms.txt.)
    constexpr bool X6 = SIX;
    static_assert(!SIX || (DUAL && !BF && !PL), "gmap: the six-term form exists for ca_backward on fp32 features");
    // P3 (round 6): the FINAL pass of a gradient writes it as THREE bf16 PLANES per pixel -- hi | lo | hi at plane
    // stride j1.p3_plane inside rows of ``ops`` bf16 elements (CCNET_PLANES_HLH: the K-concatenated operand of the module's
    // split-bf16 GEMMs dx = W^T dqkv^T and dW = dqkv^T x) -- instead of fp32, and one row of COLUMN SUMS per strip-workgroup (the
    // bias gradients, added up in a fixed order by colsum_reduce_kernel): the module's backward no longer reads dqkv back to
    // split it (ccnet_cca_split_planes_colsum_f32: 192 MB read, 289 MB written, 95 us at (8,512,97,97)).  ``out`` / ``obs`` / ``ops``
    // are then in bf16 elements.
    static_assert(!P3 || (ROW && ADD && !NCHW && !OBF && !LONG && !std::is_same<FT, bf16_t>::value), "gmap: the three-plane output belongs to the final fp32 row passes");
    constexpr int TSP = t16_size(P);
    const int dual_id = (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7));
    const bool job1 = DUAL && ((blockIdx.x >> 3) & 1) != 0;           // (wave-uniform)
    const bool trans = DUAL ? job1 : TRANS;
    if (DUAL && dual_id >= j1.nwg) return;      // (padding of the last 16-block)
    if (job1) {
        F = j1.F; addend = j1.addend; out = j1.out; fbs = j1.fbs; fps = j1.fps; obs = j1.obs; ops = j1.ops;
    }
    constexpr int NT = (P + 15) / 16, TPW = (NT + GS_WAVES - 1) / GS_WAVES, NKS = P / 32;
    // pixel-major output image: one pixel = 64 channels + 4 dwords of pad, so that the 16 lanes of a ds_write_b128 pass
    // (16 consecutive positions, the same 4 channels) fall into 16 different bank quads (PMC: 45 % conflict cycles in the dv
    // row pass with the 4-pixel pieces of the feature-tile geometry, whose pad moved the bank once per 4 pixels)
    constexpr int OPX = GM_CG + 4;
    constexpr int FSZ = PL ? 2 * TSP : GTile<FT>::size(P), OSZ = P * OPX;
    constexpr int NPF = PL ? 2 * t16_pieces(P) : GTile<FT>::pieces(P);
    constexpr int SPX = OBF ? 8 : 4;                                  // pixels per store instruction
    constexpr int NSI = NCHW ? 1 : ((P + SPX - 1) / SPX + GS_WAVES - 1) / GS_WAVES;   // store instructions per wave and group (max)
    constexpr int PO = P + 4, OIMG = NCHW ? GM_CG * PO : OSZ;         // NCHW: [channel][position] image, pitch PO
    constexpr int CPI = P <= 128 ? 2 : 1, LPC = kWave / CPI;          // NCHW: channels per store instruction, lanes per channel (4 positions each)
    constexpr int NSX = GM_CG / CPI / GS_WAVES;                       // NCHW: store instructions per wave and group
    static_assert(!NCHW || (ROW && ADD && !OBF && !TRANS && !DUAL), "gmap: the NCHW epilogue belongs to the final fp32 row pass");
    // (LONG: the key blocks of a strip are separate launches chained through the addend; the FIRST block of a column pass starts the
    //  partial and has none)
    static_assert(!LONG || ADD || !ROW, "gmap: the key blocks of a blocked row strip are chained through the addend");
    static_assert(!PL || !OBF, "gmap: split-plane features produce fp32 outputs");
    // WPC = 3 (ca_backward at C/8 <= 64: ONE channel group per strip, so nothing is ever prefetched): a single feature slot + the
    // output image = 53.6 KB, three workgroups per CU, <= 168 VGPRs (the four N tiles accumulated two at a time, no residual slices)
    // (at 101 .. 132 positions the same form is WPC = 2 -- 70.7 KB, 180 / 228 VGPRs -- where the two-slot one ran ONE workgroup per CU)
    constexpr bool ONEG = WPC == 3 || (DUAL && !BF && !PL && !LONG && P > 100 && WPC == 2);
    static_assert(!ONEG || (DUAL && !BF && !PL), "gmap: the one-group form exists for ca_backward on fp32 q | k");
    constexpr int NSLOT = ONEG ? 1 : 2;
    // the residual slices (x of functions.py:49 in the output's layout) exist for the final row passes of the all-pixel-major families
    constexpr bool RES = ROW && ADD && !NCHW && !DUAL && !PL;
    static_assert(P % 4 == 0 && (NSLOT * FSZ + OIMG) * 4 * WPC <= 163840, "gmap: LDS of WPC workgroups per CU");
    __shared__ __attribute__((aligned(16))) float lds[NSLOT * FSZ + OIMG];
    CCA_LDS_REGISTER(lds);
    float *const FB = lds, *const oimg = lds + NSLOT * FSZ;
    const int HW = H * W, S = H + W;
    const int L = ROW ? W : H, G = ROW ? H : W;
    // workgroups are dispatched in index order, two per CU: the first n_whole take a whole strip each, the remaining
    // strips (fewer than one round) are cut into `split` channel ranges so that the last round is a short one
    const int ncg = (C + GM_CG - 1) / GM_CG;
    int id = DUAL ? dual_id : (int)blockIdx.x, cg0 = 0, cg1 = ncg;
    if (!DUAL && j1.xcd > 0) {
        // XCD-aware decode (j1.xcd = strips per XCD; the host offers it only when strips and n_whole divide by 8): workgroup ids
        // go round-robin over the 8 XCDs, so XCD x = id & 7 takes the strips [x * xcd, (x + 1) * xcd) -- at 8 images of 97 rows
        // one image per XCD -- the whole ones first in ITS dispatch order, then its share of the cut ones.  Neighbouring NCHW
        // rows share their boundary cache lines (a row of 97 floats is 388 B at arbitrary alignment): on one XCD the second
        // touch of such a line is an L2 hit and the two partial writes merge there (forward row pass 155 -> 148 us, same run:
        // profiles/r04m_ab_*.txt).
        const int x = id & 7, idx = id >> 3, nw8 = n_whole >> 3;
        if (idx < nw8) {
            id = x * j1.xcd + idx;
        } else {
            const int r = idx - nw8, part = r % split;
            id = x * j1.xcd + nw8 + r / split;
            cg0 = part * ncg / split;
            cg1 = (part + 1) * ncg / split;
        }
    } else if (id >= n_whole) {
        const int r = id - n_whole, part = r % split;
        id = n_whole + r / split;
        cg0 = part * ncg / split;
        cg1 = (part + 1) * ncg / split;
    }
    const int id0 = id;                         // (the strip: the row of the P3 column-sum partials)
    // LONG: id = (strip, query block); this launch contracts over key block j1.jblk
    const int qblk = LONG ? id % j1.nb : 0;
    if (LONG) id /= j1.nb;
    const int b = id / G, g = id - b * G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    if (DUAL && blockIdx.x == 0 && wv == 0 && j1.red_dst) {                            // (see GmapJob::red_*)
        float t = 0.f;
        for (int i = lane; i < j1.red_n; i += kWave) t += j1.red_src[i];
        t = wave_sum(t);
        if (lane == 0) j1.red_dst[0] = t;
    }
    const int pix0 = ROW ? g * W : g, pstep = ROW ? 1 : W;                             // pixel(i) = pix0 + i * pstep
    const int a_off = ROW ? H : 0;
    // query side (attention rows, outputs, addend, residual): positions i0 .. i0 + Lm; key side (attention columns, features):
    // j0 .. j0 + Lk.  Whole strips: both are the strip.
    const int lb = LONG ? long_block(L, j1.nb) : 0;
    const int i0 = LONG ? qblk * lb : 0, j0 = LONG ? j1.jblk * lb : 0;
    const int Lm = LONG ? (L - i0 < lb ? L - i0 : lb) : L, Lk = LONG ? (L - j0 < lb ? L - j0 : lb) : L;
    // (the attention block P_g[m][k] is T[pixel(query)][slot(key)]: non-transposed the queries are the M side, transposed the K side)
    const int pixM = pix0 + i0 * pstep, pixK = pix0 + j0 * pstep, aK = a_off + j0, aM = a_off + i0;

    const FBuf Tb = make_fbuf(T + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
    const FBuf Fb = make_fbuf(reinterpret_cast<const float *>(F + (size_t)b * fbs), ((size_t)(HW - 1) * fps + (PL ? 2 : 1) * C) * sizeof(FT));
    const FBuf Ob = P3 ? make_fbuf(reinterpret_cast<const float *>(reinterpret_cast<const char *>(out) + (size_t)b * obs * 2), (size_t)HW * ops * 2)
                       : make_fbuf(reinterpret_cast<const float *>(out + (size_t)b * obs),
                                   NCHW ? (size_t)C * HW * sizeof(OT) : ((size_t)(HW - 1) * ops + C) * sizeof(OT));
    const FBuf Rb = make_fbuf(reinterpret_cast<const float *>(resid ? resid + (size_t)b * rbs : out),
                              !resid ? 4 : NCHW ? (size_t)C * HW * sizeof(OT) : ((size_t)(HW - 1) * rps + C) * sizeof(OT));
    static_assert(!ABF || (ADD && OBF && !NCHW && !DUAL), "gmap: a bf16 addend feeds the bf16 family's final row passes");
    const FBuf Db = make_fbuf(ADD ? (ABF ? reinterpret_cast<const float *>(reinterpret_cast<const uint16_t *>(addend) + (size_t)b * abs_)
                                         : addend + (size_t)b * abs_) : T,
                              ADD ? ((size_t)(HW - 1) * aps + C) * (ABF ? 2 : sizeof(float)) : 4);
    const float alpha = gamma ? gamma[0] : 1.f;
    const BandK kp = band_ksteps(Lk);

    auto issue_feat = [&](int cg) {
        for (int it = wv; it < NPF; it += GS_WAVES) {
            if constexpr (PL) {
                const int plane = it >= NPF / 2;
                t16_dma_piece(Fb, FB + ((cg - cg0) & 1) * FSZ + plane * TSP, it - plane * (NPF / 2), lane, pixK, pstep, Lk, fps,
                              cg * GM_CG, C, plane ? C : 0);
            } else {
                gtile_dma_piece<FT>(Fb, FB + (ONEG ? 0 : (cg - cg0) & 1) * FSZ, it, lane, pixK, pstep, Lk, fps, cg * GM_CG, C);
            }
        }
    };
    issue_feat(cg0);

    // ---- the strip's attention block -> MFMA fragments in registers.  Fragment (tile t, k-step ks) of lane (ln, lg):
    // ---- P_g[m][32 ks + 8 lg + e] (TRANS: P_g[32 ks + 8 lg + e][m]), m = 16 t + ln, e < 8; zero beyond the strip
    u32x4 ah[TPW][X6 ? 1 : NKS], al[TPW][X6 ? 1 : NKS];
    float ax[TPW][X6 ? NKS : 1][8];                                   // X6: the block's fp32 values, fragment order
    float at[TPW];
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        const int t = wv + GS_WAVES * a, m = 16 * t + ln;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float x[8];
            if (ks < kp.nbf && 16 * t < Lm) {                         // wave-uniform
                const int k0 = 32 * ks + 8 * lg;
                if (!trans) {
                    const int base = ((pixM + m * pstep) * S + aK + k0) * 4;
                    const f32x4 u = fbuf_load_x4(Tb, (m < Lm && k0 < Lk) ? base : kOobOffset, 0);
                    const f32x4 v = fbuf_load_x4(Tb, (m < Lm && k0 + 4 < Lk) ? base + 16 : kOobOffset, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = k0 + e < Lk ? u[e] : 0.f; x[4 + e] = k0 + 4 + e < Lk ? v[e] : 0.f; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = fbuf_load(Tb, (m < Lm && k0 + e < Lk) ? ((pixK + (k0 + e) * pstep) * S + aM + m) * 4 : kOobOffset, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = 0.f;
            }
            if constexpr (X6) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ax[a][ks][e] = x[e];
            } else {
                const BfSplit sp = bf16_split8(x);
                ah[a][ks] = sp.hi;
                al[a][ks] = sp.lo;
            }
        }
        const int kt = 32 * kp.nbf + lg;
        at[a] = fbuf_load(Tb, (kp.tail && m < Lm && kt < Lk) ? (trans ? ((pixK + kt * pstep) * S + aM + m) * 4
                                                                       : ((pixM + m * pstep) * S + aK + kt) * 4) : kOobOffset, 0);
    }

    // this lane's slice of store instruction k of its wave: pixel position and first channel (within the group)
    const int nsi_total = (Lm + SPX - 1) / SPX;
    const int nstore = (nsi_total - wv + GS_WAVES - 1) / GS_WAVES;          // store instructions this wave issues per group
    auto st_pos = [&](int k) { return SPX * (wv + GS_WAVES * k) + (OBF ? lane >> 3 : lane >> 4); };
    const int st_c = OBF ? 8 * (lane & 7) : 4 * (lane & 15);
    int nstore_nchw = 0;

    for (int cg = cg0; cg < cg1; ++cg) {
        const float *img = FB + (ONEG ? 0 : (cg - cg0) & 1) * FSZ;
        // tile cg landed, every wave is done with group cg - 1; the stores of group cg - 1 (the most recent vector
        // memory operations of this wave) may stay in flight
        if (cg == cg0) barrier_dma_keep<0>();
        else           barrier_dma_keep_n(NCHW ? nstore_nchw : P3 ? 3 * nstore + (wv == 0 ? 1 : 0) : nstore);
        if (!ONEG && cg + 1 < cg1) issue_feat(cg + 1);
        // the fp32 addend / bf16 residual slices this lane will store over: in registers by the time the tiles are done
        f32x4 add0[NSI], add1[NSI];
        u32x4 res[NSI];
        f32x4 addp[NCHW ? TPW : 1][4], resx[NCHW ? NSX : 1];
        if constexpr (NCHW) {
#pragma unroll
            for (int a = 0; a < TPW; ++a) {                 // addend in the accumulator layout: 4 channels of position i
                const int i = 16 * (wv + GS_WAVES * a) + ln;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int c = cg * GM_CG + 16 * nt + 4 * lg;
                    addp[a][nt] = fbuf_load_x4(Db, (i < Lm && c < C) ? ((pixM + i) * aps + c) * 4 : kOobOffset, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < NSX; ++k) {                 // residual in the store layout: 4 positions of channel c
                const int c = cg * GM_CG + CPI * (wv + GS_WAVES * k) + (lane / LPC), w4 = lane & (LPC - 1);
                const int w0 = (4 * w4 + 3 < Lm || Lm < 4) ? 4 * w4 : Lm - 4;      // the last granule is shifted back to end at Lm
                resx[k] = fbuf_load_x4(Rb, (resid && 4 * w4 < Lm && c < C) ? (c * HW + pixM + w0) * 4 : kOobOffset, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < (NCHW ? 0 : NSI); ++k) {
            const int i = st_pos(k), c = cg * GM_CG + st_c;
            const bool ok = i < Lm && c < C;
            const int pix = pixM + i * pstep;
            if (ADD && !X6) {                             // (X6: the addend starts the accumulators, see below)
                if constexpr (ABF) {                      // 8 bf16 channels of the partial -> two fp32 quads
                    const u32x4 pk = __builtin_bit_cast(u32x4, fbuf_load_x4(Db, ok ? (pix * aps + c) * 2 : kOobOffset, 0));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        add0[k][2 * e] = __builtin_bit_cast(float, pk[e] << 16);
                        add0[k][2 * e + 1] = __builtin_bit_cast(float, pk[e] & 0xffff0000u);
                        add1[k][2 * e] = __builtin_bit_cast(float, pk[2 + e] << 16);
                        add1[k][2 * e + 1] = __builtin_bit_cast(float, pk[2 + e] & 0xffff0000u);
                    }
                } else {
                add0[k] = fbuf_load_x4(Db, ok ? (pix * aps + c) * 4 : kOobOffset, 0);
                if (OBF) add1[k] = fbuf_load_x4(Db, ok ? (pix * aps + c + 4) * 4 : kOobOffset, 0);
                }
            }
            if constexpr (RES) res[k] = __builtin_bit_cast(u32x4, fbuf_load_x4(Rb, (ok && resid) ? (pix * rps + c) * (int)sizeof(OT) : kOobOffset, 0));
        }
        // D^T[m = channel][n = strip position] = features^T x attention^T: a lane ends up with 4 consecutive channels of
        // one position (one ds_write_b128 into the pixel-major output image)
        // (three M tiles per wavefront = strips longer than 128: the four N tiles are accumulated two at a time, so that the
        // accumulators -- live together with 96 fragment and up to 60 prefetch registers -- take 24 VGPRs instead of 48)
        // (X6: all four N tiles at once -- the block is split once per k-step -- and the accumulators START from the column partial,
        //  loaded in their own layout: ca_backward has no gamma, so out = partial + products needs no addend registers of its own)
        constexpr int NH = X6 ? 1 : (TPW >= 3 || ONEG) ? 2 : 1, NTH = 4 / NH;
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            f32x4 acc[TPW][NTH];
#pragma unroll
            for (int a = 0; a < TPW; ++a)
#pragma unroll
                for (int n = 0; n < NTH; ++n) {
                    if constexpr (X6 && ADD) {
                        const int i = 16 * (wv + GS_WAVES * a) + ln, c = cg * GM_CG + 16 * (nh * NTH + n) + 4 * lg;
                        acc[a][n] = fbuf_load_x4(Db, (i < Lm && c < C) ? ((pixM + i * pstep) * aps + c) * 4 : kOobOffset, 0);
                    } else {
                        acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
            for (int ks = 0; ks < (X6 ? NKS : 0); ++ks) {
                if (ks < kp.nbf) {
                    BfSplit3 p3[TPW];
#pragma unroll
                    for (int a = 0; a < TPW; ++a) p3[a] = bf16_split8x3(ax[a][ks]);
#pragma unroll
                    for (int n = 0; n < NTH; ++n) {
                        const int nt = nh * NTH + n;
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = CCA_LDS_LD(img + gtile_f32_idx(32 * ks + 8 * lg + e, 16 * nt + ln));
                        const BfSplit3 f3 = bf16_split8x3(x);
#pragma unroll
                        for (int a = 0; a < TPW; ++a) {
                            if ((wv + GS_WAVES * a) * 16 < Lm) {          // smallest terms first
                                acc[a][n] = mfma_bf16_16x16x32(f3.lo, p3[a].hi, acc[a][n]);
                                acc[a][n] = mfma_bf16_16x16x32(f3.mid, p3[a].mid, acc[a][n]);
                                acc[a][n] = mfma_bf16_16x16x32(f3.hi, p3[a].lo, acc[a][n]);
                                acc[a][n] = mfma_bf16_16x16x32(f3.mid, p3[a].hi, acc[a][n]);
                                acc[a][n] = mfma_bf16_16x16x32(f3.hi, p3[a].mid, acc[a][n]);
                                acc[a][n] = mfma_bf16_16x16x32(f3.hi, p3[a].hi, acc[a][n]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < (X6 ? 0 : NKS); ++ks) {
                if (ks < kp.nbf) {
#pragma unroll
                    for (int n = 0; n < NTH; ++n) {
                        const int nt = nh * NTH + n;
                        // feature fragment: positions 32 ks + 8 lg + e of channel 16 nt + ln
                        BfSplit fb;
                        if constexpr (PL) {
                            fb.hi = t16_frag(img, ks, nt, lane);
                            fb.lo = t16_frag(img + TSP, ks, nt, lane);
                        } else if constexpr (BF) {
                            uint32_t x[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] = lds_load_u16(img, gtile_bf_byte(32 * ks + 8 * lg + e, 16 * nt + ln));
                            fb.hi = u32x4{x[0] | (x[1] << 16), x[2] | (x[3] << 16), x[4] | (x[5] << 16), x[6] | (x[7] << 16)};
                            fb.lo = fb.hi;
                        } else {
                            float x[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] = CCA_LDS_LD(img + gtile_f32_idx(32 * ks + 8 * lg + e, 16 * nt + ln));
                            fb = bf16_split8(x);
                        }
#pragma unroll
                        for (int a = 0; a < TPW; ++a) {
                            if ((wv + GS_WAVES * a) * 16 < Lm) {
                                acc[a][n] = mfma_bf16_16x16x32(fb.hi, ah[a][ks], acc[a][n]);
                                if (!BF) acc[a][n] = mfma_bf16_16x16x32(fb.lo, ah[a][ks], acc[a][n]);
                                acc[a][n] = mfma_bf16_16x16x32(fb.hi, al[a][ks], acc[a][n]);
                            }
                        }
                    }
                }
            }
            if (kp.tail) {
                const int pos = 32 * kp.nbf + lg;
#pragma unroll
                for (int n = 0; n < NTH; ++n) {
                    const int nt = nh * NTH + n;
                    float fbv;
                    if constexpr (PL) fbv = __builtin_bit_cast(float, lds_load_u16(img, t16_byte(pos, 16 * nt + ln)) << 16)
                                          + __builtin_bit_cast(float, lds_load_u16(img + TSP, t16_byte(pos, 16 * nt + ln)) << 16);
                    else if constexpr (BF) fbv = __builtin_bit_cast(float, lds_load_u16(img, gtile_bf_byte(pos, 16 * nt + ln)) << 16);
                    else              fbv = CCA_LDS_LD(img + gtile_f32_idx(pos, 16 * nt + ln));
#pragma unroll
                    for (int a = 0; a < TPW; ++a)
                        if ((wv + GS_WAVES * a) * 16 < Lm) acc[a][n] = mfma_16x16x4(fbv, at[a], acc[a][n]);
                }
                mfma_f32_result_fence();
            }
#pragma unroll
            for (int a = 0; a < TPW; ++a) {
                const int i = 16 * (wv + GS_WAVES * a) + ln;
                if (i < Lm) {
#pragma unroll
                    for (int n = 0; n < NTH; ++n) {
                        const int nt = nh * NTH + n;
                        if constexpr (NCHW) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                CCA_LDS_ST(oimg + oimg_nchw_idx<P>(16 * nt + 4 * lg + q, i), alpha * acc[a][n][q] + addp[a][nt][q]);
                        } else {
                            lds_store_x4(oimg + i * OPX + 16 * nt + 4 * lg, X6 ? acc[a][n] : alpha * acc[a][n]);
                        }
                    }
                }
            }
        }
        barrier_lds_only();
        if constexpr (NCHW) {
            // runs of W floats per channel: CPI channels x LPC lanes (4 positions each) per instruction (2 x 32 up to 128
            // positions, 1 x 64 beyond).  Store instructions this wave issues in this group (for the next group's counted
            // barrier): one 16-byte store per valid channel set (rows shorter than 4: Lm single stores)
            {
                const int crem = C - cg * GM_CG, first = CPI * wv;
                const int nk = crem <= first ? 0 : (crem - first + CPI * GS_WAVES - 1) / (CPI * GS_WAVES);
                nstore_nchw = (nk < NSX ? nk : NSX) * (Lm >= 4 ? 1 : Lm);
            }
            const int crem = C - cg * GM_CG;
#pragma unroll
            for (int k = 0; k < NSX; ++k) {
                // every branch around a store is wave-uniform (scalar) and every store issued has an active lane: the
                // counted barrier may only count instructions that really go to memory
                if (CPI * (wv + GS_WAVES * k) < crem) {
                    const int ch = CPI * (wv + GS_WAVES * k) + (lane / LPC), c = cg * GM_CG + ch, w4 = lane & (LPC - 1);
                    const bool whole = 4 * w4 + 3 < Lm;
                    if (Lm >= 4) {
                        // a row of Lm floats = whole 4-float granules + one granule shifted back to end at Lm (it rewrites up to
                        // 3 floats of its neighbour with the same values): 16-byte stores only, one instruction per channel pair
                        const int w0 = whole ? 4 * w4 : Lm - 4;
                        f32x4 u;
                        if (whole) {
                            u = lds_load_x4(oimg + oimg_nchw_idx<P>(ch, 4 * w4));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) u[e] = CCA_LDS_LD(oimg + oimg_nchw_idx<P>(ch, (4 * w4 < Lm ? w0 : 0) + e));
                        }
                        if (4 * w4 < Lm && c < C) fbuf_store_x4(Ob, u + resx[k], (c * HW + pixM + w0) * 4, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            if (e < Lm) {
                                if (w4 == 0 && c < C)
                                    fbuf_store(Ob, CCA_LDS_LD(oimg + oimg_nchw_idx<P>(ch, e)) + resx[k][e], (c * HW + pixM + e) * 4, 0);
                            }
                        }
                    }
                }
            }
            continue;
        }
        // whole pixel rows leave: 256 (fp32) / 128 (bf16) bytes per pixel, + addend (+ residual), rounded once
        f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};                           // P3: this lane's four channels summed over its pixels
#pragma unroll
        for (int k = 0; k < NSI; ++k) {
            if (wv + GS_WAVES * k < nsi_total) {                         // wave-uniform: exactly `nstore` instructions
                const int i = st_pos(k), c = cg * GM_CG + st_c;
                const float *s = oimg + i * OPX + st_c;
                if (i < Lm && c < C) {
                    f32x4 u = lds_load_x4(s);
                    if (ADD && !X6) u += add0[k];
                    if constexpr (OBF) {
                        f32x4 v = lds_load_x4(s + 4);
                        if (ADD) v += add1[k];
                        if constexpr (RES) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {              // + the bf16 residual (x of functions.py:49)
                                const float lo = __builtin_bit_cast(float, res[k][e] << 16), hi = __builtin_bit_cast(float, res[k][e] & 0xffff0000u);
                                if (e < 2) { u[2 * e] += lo; u[2 * e + 1] += hi; } else { v[2 * e - 4] += lo; v[2 * e - 3] += hi; }
                            }
                        }
                        const f32x4 packed = __builtin_bit_cast(f32x4, u32x4{cvt_pk_bf16(u[0], u[1]), cvt_pk_bf16(u[2], u[3]),
                                                                             cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])});
                        fbuf_store_x4(Ob, packed, ((pixM + i * pstep) * ops + c) * 2, 0);
                    } else {
                        if constexpr (RES) u += __builtin_bit_cast(f32x4, res[k]);          // (+ the fp32 residual; zeros when there is none)
                        if constexpr (P3) {
                            csum += u;
                            const uint32_t h0 = cvt_pk_bf16(u[0], u[1]), h1 = cvt_pk_bf16(u[2], u[3]);
                            const uint32_t l0 = cvt_pk_bf16(u[0] - __builtin_bit_cast(float, h0 << 16), u[1] - __builtin_bit_cast(float, h0 & 0xffff0000u));
                            const uint32_t l1 = cvt_pk_bf16(u[2] - __builtin_bit_cast(float, h1 << 16), u[3] - __builtin_bit_cast(float, h1 & 0xffff0000u));
                            const int off = ((pixM + i * pstep) * ops + c) * 2;
                            fbuf_store_x2(Ob, h0, h1, off, 0);
                            fbuf_store_x2(Ob, l0, l1, off + 2 * j1.p3_plane, 0);
                            fbuf_store_x2(Ob, h0, h1, off + 4 * j1.p3_plane, 0);
                        } else
                        fbuf_store_x4(Ob, u, ((pixM + i * pstep) * ops + c) * 4, 0);
                    }
                }
            }
        }
        if constexpr (P3) {
            // column sums of this strip's rows of the group: lanes l, l + 16, l + 32, l + 48 hold the same four channels (different
            // pixels); the four wavefronts meet in the group's feature slot (dead since the multiply), wavefront 0 adds them in wave
            // order and stores the strip's row of partials -- every order fixed
            float *red = const_cast<float *>(img);
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += shfl_xor(csum[e], 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += shfl_xor(csum[e], 32);
            if (lane < 16) lds_store_x4(red + wv * GM_CG + 4 * lane, csum);
            barrier_lds_only();
            if (wv == 0) {                                                 // (wave-uniform: the store below is counted by the next group's barrier)
                float *cs = job1 ? j1.cs1 : j1.cs;
                if (lane < 16 && cg * GM_CG + 4 * lane < C) {
                    f32x4 t = lds_load_x4(red + 4 * lane);
#pragma unroll
                    for (int w = 1; w < GS_WAVES; ++w) t += lds_load_x4(red + w * GM_CG + 4 * lane);
                    const FBuf Cb = make_fbuf(cs, 0x7ffffff0u);
                    fbuf_store_x4(Cb, t, ((id0 * j1.cs_stride) + cg * GM_CG + 4 * lane) * 4, 0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gmap3: the same strip contraction as gmap_kernel on SPLIT-PLANE features with a pixel-major fp32 output, restructured
// for MEMORY-LEVEL PARALLELISM.  Measured (profiles/r03b_family_compare.txt): pre-splitting the features changed the
// gmap launches by 2 % -- they are bound by the ONE feature tile a workgroup has in flight (a group's iteration lasts one
// loaded memory latency, ~4 us, however little it computes).  Here a workgroup keeps a ring of THREE feature tiles (two
// in flight while one is consumed), and the results leave STRAIGHT FROM THE ACCUMULATORS: the swapped MFMA leaves a lane
// with 4 consecutive channels of one position = one 16-byte store, the four N tiles of a pixel row are stored back to
// back by the same wavefront (256 contiguous bytes per pixel), so there is no output image in LDS, no second barrier per
// group, and the LDS that image used is the third ring slot (79,872 B: still two workgroups per CU).  The addend of the
// row pass is prefetched in the accumulator layout one group ahead.  One counted barrier per group: every wait names
// exactly how many of this wavefront's newer vector-memory operations (ring fills, addend prefetch, stores) may stay in
// flight -- they complete in issue order.
// ---------------------------------------------------------------------------------------------------------------
// NBUF ring slots (fills run NBUF - 1 tiles ahead), WPC workgroups per CU: <3, 2> keeps two tiles in flight per workgroup;
// <2, 3> trades one of them for a third workgroup per CU (53,248 B of LDS, <= 168 VGPRs) -- 768 slots for the 776 strips
// of a branch at the headline shape, where 512 slots left the second round half empty (profiles/r03c_family_compare.txt).
// FT = bf16p_t (split planes: hi | lo tiles, three products) or bf16_t (bf16 features, BASELINE configs[4]: one tile, the two
// products with the attention's hi and lo halves); the output is fp32 pixel-major either way (the column partial).
// OT = bf16_t (bf16 features only, round 5): the column PARTIAL leaves as bf16 -- what the reference's own bf16 arithmetic does
// (functions.py:46-47 under bf16: out_H and out_W are each a bf16 bmm result before they are added) -- half the bytes of the
// fp32 partial, which at BASELINE configs[4] was 2.2 GB of the step's 8.85 GB.  Two N tiles of a position are paired through one
// lane exchange (lane ^ 16) so that a lane still stores 16 bytes: 8 consecutive bf16 channels.
template <int P, bool ROW, bool TRANS, bool ADD, int NBUF = 3, int WPC = 2, typename FT = bf16p_t, typename OT = float>
__global__ __launch_bounds__(GS_THREADS, WPC) void gmap3_kernel(const float *__restrict__ T, const FT *__restrict__ F,
                                                               const float *__restrict__ addend, const float *__restrict__ gamma,
                                                               OT *__restrict__ out, int C, int H, int W, long fbs, int fps,
                                                               long abs_, int aps, long obs, int ops, int n_whole, int split) {
    constexpr bool OBF = std::is_same<OT, bf16_t>::value;
    static_assert(!OBF || (std::is_same<FT, bf16_t>::value && !ADD), "gmap3: the bf16 partial belongs to the bf16 family's column passes");
    constexpr int NT = (P + 15) / 16, TPW = (NT + GS_WAVES - 1) / GS_WAVES, NKS = P / 32;
    constexpr bool F32 = std::is_same<FT, float>::value;                // fp32 pixel-major features (F32T tiles, split per fragment)
    constexpr int NPL = std::is_same<FT, bf16p_t>::value ? 2 : 1;       // planes per feature tile
    static_assert(NPL == 2 || F32 || std::is_same<FT, bf16_t>::value, "gmap3: bf16p_t, bf16_t or float features");
    // (fp32 features: the padded 4-position pieces of GTile<float> -- a fragment's eight positions are two runs of four at a
    //  constant 256-byte stride, i.e. two address registers + immediate offsets; the XOR-swizzled F32T tile needs an address per
    //  position and pushed this kernel over the 168 VGPRs of three workgroups per CU)
    constexpr int TSP = t16_size(P), FSZ = F32 ? GTile<float>::size(P) : NPL * TSP, NPF = F32 ? GTile<float>::pieces(P) : NPL * t16_pieces(P), D = NBUF - 1;
    static_assert(P % 4 == 0 && NBUF * FSZ * 4 * WPC <= 163840 && (NBUF == 2 || NBUF == 3), "gmap3: LDS of WPC workgroups per CU");
    // (the ring fills are LDS-DMAs the compiler does not see -- fbuf_load_to_lds_x4_uncounted, cca_platform.hpp: with the
    // builtin form it drained the fills of the next two tiles before every group's first transposing read)
    __shared__ __attribute__((aligned(16))) float lds[NBUF * FSZ];
    CCA_LDS_REGISTER(lds);
    const int HW = H * W, S = H + W;
    const int L = ROW ? W : H, G = ROW ? H : W;
    const int ncg = (C + GM_CG - 1) / GM_CG;
    int id = blockIdx.x, cg0 = 0, cg1 = ncg;
    if (id >= n_whole) {
        const int r = id - n_whole, part = r % split;
        id = n_whole + r / split;
        cg0 = part * ncg / split;
        cg1 = (part + 1) * ncg / split;
    }
    const int b = id / G, g = id - b * G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int pix0 = ROW ? g * W : g, pstep = ROW ? 1 : W;
    const int a_off = ROW ? H : 0;

    const FBuf Tb = make_fbuf(T + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
    const FBuf Fb = make_fbuf(reinterpret_cast<const float *>(F + (size_t)b * fbs),
                              F32 ? ((size_t)(HW - 1) * fps + C) * 4 : ((size_t)(HW - 1) * fps + NPL * C) * 2);
    const FBuf Ob = make_fbuf(reinterpret_cast<const float *>(out + (size_t)b * obs), ((size_t)(HW - 1) * ops + C) * sizeof(OT));
    const FBuf Db = make_fbuf(ADD ? addend + (size_t)b * abs_ : T, ADD ? ((size_t)(HW - 1) * aps + C) * sizeof(float) : 4);
    const float alpha = gamma ? gamma[0] : 1.f;
    const BandK kp = band_ksteps(L);

    // ring fill of group cg: every piece of both planes is issued whatever the strip length (zero fill beyond it), so a
    // wavefront issues exactly npw instructions per tile
    const int npw = (NPF - wv + GS_WAVES - 1) / GS_WAVES;
    auto issue_feat = [&](int cg, float *dst) {
#pragma unroll
        for (int k = 0; k < (NPF + GS_WAVES - 1) / GS_WAVES; ++k) {
            const int it = wv + GS_WAVES * k;
            if (it < NPF) {                                                  // (wave-uniform)
                if constexpr (F32) {
                    gtile_dma_piece<float>(Fb, dst, it, lane, pix0, pstep, L, fps, cg * GM_CG, C);
                } else {
                    const int plane = it >= NPF / NPL;
                    t16_dma_piece(Fb, dst + plane * TSP, it - plane * (NPF / NPL), lane, pix0, pstep, L, fps, cg * GM_CG, C, plane ? C : 0);
                }
            }
        }
    };
    issue_feat(cg0, lds);
    if (D > 1 && cg0 + 1 < cg1) issue_feat(cg0 + 1, lds + FSZ);

    // the strip's attention block -> MFMA fragments in registers (as gmap_kernel)
    u32x4 ah[TPW][NKS], al[TPW][NKS];
    float at[TPW];
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        const int t = wv + GS_WAVES * a, m = 16 * t + ln;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float x[8];
            if (ks < kp.nbf && 16 * t < L) {                          // wave-uniform
                const int k0 = 32 * ks + 8 * lg;
                if (!TRANS) {
                    const int base = ((pix0 + m * pstep) * S + a_off + k0) * 4;
                    const f32x4 u = fbuf_load_x4(Tb, (m < L && k0 < L) ? base : kOobOffset, 0);
                    const f32x4 v = fbuf_load_x4(Tb, (m < L && k0 + 4 < L) ? base + 16 : kOobOffset, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = k0 + e < L ? u[e] : 0.f; x[4 + e] = k0 + 4 + e < L ? v[e] : 0.f; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = fbuf_load(Tb, (m < L && k0 + e < L) ? ((pix0 + (k0 + e) * pstep) * S + a_off + m) * 4 : kOobOffset, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = 0.f;
            }
            const BfSplit sp = bf16_split8(x);
            ah[a][ks] = sp.hi;
            al[a][ks] = sp.lo;
        }
        const int kt = 32 * kp.nbf + lg;
        at[a] = fbuf_load(Tb, (kp.tail && m < L && kt < L) ? (TRANS ? ((pix0 + kt * pstep) * S + a_off + m) * 4
                                                                     : ((pix0 + m * pstep) * S + a_off + kt) * 4) : kOobOffset, 0);
    }

    // stores / addend loads of a group: one 16-byte access per owned M tile and N tile whose channels exist (both
    // conditions wave-uniform, every instruction issued has valid lanes: only such instructions may be counted)
    int ntile_w = 0;
#pragma unroll
    for (int a = 0; a < TPW; ++a) ntile_w += (wv + GS_WAVES * a) * 16 < L ? 1 : 0;
    auto nnt = [&](int cg) { const int rem = C - cg * GM_CG; return rem >= GM_CG ? 4 : (rem + 15) / 16; };
    // per-wave accesses of a group (bf16 output: one 16-byte store per PAIR of N tiles)
    auto nacc = [&](int cg) { return (cg >= cg0 && cg < cg1) ? ntile_w * (OBF ? (nnt(cg) + 1) / 2 : nnt(cg)) : 0; };
    f32x4 addp[ADD ? 2 : 1][ADD ? TPW : 1][4];
    auto load_addend = [&](int cg, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (ADD) {
#pragma unroll
            for (int a = 0; a < TPW; ++a) {
                const int i = 16 * (wv + GS_WAVES * a) + ln;
                if ((wv + GS_WAVES * a) * 16 < L) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int c = cg * GM_CG + 16 * nt + 4 * lg;
                        if (cg * GM_CG + 16 * nt < C)
                            addp[slot][a][nt] = fbuf_load_x4(Db, (i < L && c < C) ? ((pix0 + i * pstep) * aps + c) * 4 : kOobOffset, 0);
                    }
                }
            }
        }
    };
    load_addend(cg0, std::integral_constant<int, 0>{});

    // one group; SL = which half of the addend registers holds its addend (compile time: a dynamically indexed register
    // array would live in scratch memory)
    auto group = [&](int cg, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        const float *img = lds + ((cg - cg0) % NBUF) * FSZ;
        float *nxt = lds + ((cg + D - cg0) % NBUF) * FSZ;                     // ring slot of tile cg + D (= the slot of tile cg - 1)
        // tile cg landed and every wavefront is done with tile cg - 1.  This wavefront's vector-memory operations issued
        // AFTER the fill of tile cg may stay in flight; oldest first (an iteration issues addend(c + 1), fill(c + D), stores(c)):
        //   D = 2: stores(cg - 2), addend(cg), fill(cg + 1), stores(cg - 1)        D = 1: stores(cg - 1)
        // (First group: the prologue's loads are drained anyway.)
        if (cg == cg0) barrier_dma_keep<0>();
        else if (D == 2) barrier_dma_keep_n(nacc(cg - 2) + (ADD ? nacc(cg) : 0) + (cg + 1 < cg1 ? npw : 0) + nacc(cg - 1));
        else             barrier_dma_keep_n(nacc(cg - 1));
        if (ADD && cg + 1 < cg1) load_addend(cg + 1, std::integral_constant<int, SL ^ 1>{});
        if (cg + D < cg1) issue_feat(cg + D, nxt);
        f32x4 acc[TPW][4];
#pragma unroll
        for (int a = 0; a < TPW; ++a)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < kp.nbf) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    u32x4 fh, fl;
                    if constexpr (F32) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = CCA_LDS_LD(img + gtile_f32_idx(32 * ks + 8 * lg + e, 16 * nt + ln));
                        const BfSplit sp = bf16_split8(x);
                        fh = sp.hi;
                        fl = sp.lo;
                    } else {
                        fh = t16_frag(img, ks, nt, lane);
                        fl = fh;
                        if constexpr (NPL == 2) fl = t16_frag(img + TSP, ks, nt, lane);
                    }
#pragma unroll
                    for (int a = 0; a < TPW; ++a) {
                        if ((wv + GS_WAVES * a) * 16 < L) {
                            acc[a][nt] = mfma_bf16_16x16x32(fh, ah[a][ks], acc[a][nt]);
                            if constexpr (NPL == 2 || F32) acc[a][nt] = mfma_bf16_16x16x32(fl, ah[a][ks], acc[a][nt]);
                            acc[a][nt] = mfma_bf16_16x16x32(fh, al[a][ks], acc[a][nt]);
                        }
                    }
                }
            }
        }
        if (kp.tail) {
            const int pos = 32 * kp.nbf + lg;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float fbv;
                if constexpr (F32) {
                    fbv = CCA_LDS_LD(img + gtile_f32_idx(pos, 16 * nt + ln));
                } else {
                    fbv = __builtin_bit_cast(float, lds_load_u16(img, t16_byte(pos, 16 * nt + ln)) << 16);
                    if constexpr (NPL == 2) fbv += __builtin_bit_cast(float, lds_load_u16(img + TSP, t16_byte(pos, 16 * nt + ln)) << 16);
                }
#pragma unroll
                for (int a = 0; a < TPW; ++a)
                    if ((wv + GS_WAVES * a) * 16 < L) acc[a][nt] = mfma_16x16x4(fbv, at[a], acc[a][nt]);
            }
            mfma_f32_result_fence();
        }
        // D^T[m = channel][n = position]: lane (ln, lg) holds channels 16 nt + 4 lg .. + 3 of position 16 t + ln
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            const int i = 16 * (wv + GS_WAVES * a) + ln;
            if ((wv + GS_WAVES * a) * 16 < L) {
                if constexpr (OBF) {
                    // N tiles (2 p, 2 p + 1): lanes with an even lg end up with 8 consecutive channels of tile 2 p (their own 4 and
                    // the 4 of lane ^ 16), lanes with an odd lg with 8 of tile 2 p + 1 -- one 16-byte store of 8 bf16 each
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        if (cg * GM_CG + 32 * pr < C) {                      // (wave-uniform: the pair's first tile has channels)
                            const bool odd = (lg & 1) != 0;
                            const f32x4 t0 = alpha * acc[a][2 * pr], t1 = alpha * acc[a][2 * pr + 1];
                            f32x4 mine, got;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                mine[e] = odd ? t1[e] : t0[e];
                                got[e] = shfl_xor(odd ? t0[e] : t1[e], 16);
                            }
                            const f32x4 lo4 = odd ? got : mine, hi4 = odd ? mine : got;
                            const int c = cg * GM_CG + 16 * (2 * pr + (odd ? 1 : 0)) + 8 * (lg >> 1);
                            const f32x4 packed = __builtin_bit_cast(f32x4, u32x4{cvt_pk_bf16(lo4[0], lo4[1]), cvt_pk_bf16(lo4[2], lo4[3]),
                                                                                 cvt_pk_bf16(hi4[0], hi4[1]), cvt_pk_bf16(hi4[2], hi4[3])});
                            fbuf_store_x4(Ob, packed, (i < L && c < C) ? ((pix0 + i * pstep) * ops + c) * 2 : kOobOffset, 0);
                        }
                    }
                } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int c = cg * GM_CG + 16 * nt + 4 * lg;
                    if (cg * GM_CG + 16 * nt < C) {
                        f32x4 u = alpha * acc[a][nt];
                        if constexpr (ADD) u += addp[SL][a][nt];
                        fbuf_store_x4(Ob, u, (i < L && c < C) ? ((pix0 + i * pstep) * ops + c) * 4 : kOobOffset, 0);
                    }
                }
                }
            }
        }
    };
    for (int cg = cg0; cg < cg1; cg += 2) {
        group(cg, std::integral_constant<int, 0>{});
        if (cg + 1 < cg1) group(cg + 1, std::integral_constant<int, 1>{});
    }
}

// NCHW fp32 -> pixel-major fp32 (the gradient dy of an NCHW module output, ca_map_backward's features): 64 pixels x 64
// channels per workgroup through a padded LDS tile; reads runs of 64 pixels per channel, writes 256-byte pixel rows.
__global__ __launch_bounds__(256) void nchw_to_pm_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int HW,
                                                         long sbs, long dbs, int dps) {
    __shared__ float tile[64 * 65];
    const int ntp = (HW + 63) / 64;
    // (XCD-aware tile order: see nchw_to_planes_kernel)
    const int lid = xcd_logical_id((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int ncg = (int)gridDim.y, tix = lid % ntp, rest = lid / ntp;
    const int b = rest / ncg, p0 = tix * 64, c0 = (rest - b * ncg) * 64;
    const FBuf Sb = make_fbuf(src + (size_t)b * sbs, (size_t)C * HW * sizeof(float));
    const FBuf Db = make_fbuf(dst + (size_t)b * dbs, ((size_t)(HW - 1) * dps + C) * sizeof(float));
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ch = (tid >> 4) + 16 * it, p4 = 4 * (tid & 15);
        const bool ok = c0 + ch < C && p0 + p4 < HW;
        const f32x4 v = fbuf_load_x4(Sb, ok ? ((c0 + ch) * HW + p0 + p4) * 4 : kOobOffset, 0);   // (dwords beyond the tensor read 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ch * 65 + p4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int px = (tid >> 4) + 16 * it, c4 = 4 * (tid & 15);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[(c4 + e) * 65 + px];
        if (p0 + px < HW && c0 + c4 < C) fbuf_store_x4(Db, v, ((p0 + px) * dps + c0 + c4) * 4, 0);
    }
}

// ---- producers of SPLIT-PLANE tensors (see bf16p_t above) ----
// PlaneLayout: where the halves of a pixel's C values go inside its dps-element row -- hi at 0, lo at ``lo`` elements, and
// (three-plane rows for K-concatenated split-bf16 GEMMs, include/ccnet_cca.h CCNET_PLANES_*) a second copy of hi at ``hi2``
struct PlaneLayout { int lo, hi2, width; };
__device__ __forceinline__ void planes_store8(const FBuf &Db, int off_bytes, const PlaneLayout &pl, const float (&x)[8], bool ok) {
    const BfSplit sp = bf16_split8(x);
    fbuf_store_x4(Db, __builtin_bit_cast(f32x4, sp.hi), ok ? off_bytes : kOobOffset, 0);
    fbuf_store_x4(Db, __builtin_bit_cast(f32x4, sp.lo), ok ? off_bytes + 2 * pl.lo : kOobOffset, 0);
    if (pl.hi2 > 0) fbuf_store_x4(Db, __builtin_bit_cast(f32x4, sp.hi), ok ? off_bytes + 2 * pl.hi2 : kOobOffset, 0);
}

// fp32 pixel-major (B, H*W, sps) channels [0, C) -> planes (B, H*W, 2, C) (pixel stride dps >= 2 C elements): the value
// slice of the packed projection, split ONCE by its producer (+ an optional per-channel bias: the projection's, so that the
// GEMM's output needs no pass of its own for it).  One thread = 8 channels of a pixel (32 B in, 16 + 16 B out).
__global__ __launch_bounds__(256) void pm_split_kernel(const float *__restrict__ src, bf16p_t *__restrict__ dst, int C, int HW,
                                                       long sbs, int sps, long dbs, int dps, PlaneLayout pl,
                                                       const float *__restrict__ bias) {
    const int cpp = C >> 3;                                   // 8-channel chunks per pixel
    const int b = blockIdx.y;
    const FBuf Sb = make_fbuf(src + (size_t)b * sbs, ((size_t)(HW - 1) * sps + C) * sizeof(float));
    const FBuf Db = make_fbuf(reinterpret_cast<const float *>(dst + (size_t)b * dbs), ((size_t)(HW - 1) * dps + pl.width) * 2);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < HW * cpp; e += gridDim.x * 256) {
        const int px = e / cpp, c = 8 * (e - px * cpp);
        const f32x4 u = fbuf_load_x4(Sb, (px * sps + c) * 4, 0), v = fbuf_load_x4(Sb, (px * sps + c + 4) * 4, 0);
        float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        if (bias) {                                           // (the projection's bias, added where its output is split)
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] += bias[c + k];
        }
        planes_store8(Db, (px * dps + c) * 2, pl, x, true);
    }
}

// pm_split_kernel + the COLUMN SUMS of the source (round 5): the module's backward needs dqkv as three-plane rows for its two GEMMs
// AND the bias gradients db = sum over all pixels of dqkv -- two passes over the same 192 MB (72 + 47 us at (8,512,97,97)) become one.
// The grid's x extent times 256 is a multiple of the chunks per pixel, so a thread meets the SAME 8 channels in every iteration and
// keeps their sums in registers; a workgroup adds its threads' sums per chunk in thread order and writes one row of ``partials``
// (gridDim.x * gridDim.y rows of C floats); colsum_reduce_kernel adds the rows in row order: fixed order throughout, deterministic.
__global__ __launch_bounds__(256) void pm_split_colsum_kernel(const float *__restrict__ src, bf16p_t *__restrict__ dst, int C, int HW,
                                                              long sbs, int sps, long dbs, int dps, PlaneLayout pl,
                                                              float *__restrict__ partials) {
    __shared__ float red[256 * 8];
    CCA_LDS_REGISTER(red);
    const int cpp = C >> 3;                                   // 8-channel chunks per pixel
    const int b = blockIdx.y, tid = threadIdx.x;
    const FBuf Sb = make_fbuf(src + (size_t)b * sbs, ((size_t)(HW - 1) * sps + C) * sizeof(float));
    const FBuf Db = make_fbuf(reinterpret_cast<const float *>(dst + (size_t)b * dbs), ((size_t)(HW - 1) * dps + pl.width) * 2);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = blockIdx.x * 256 + tid; e < HW * cpp; e += gridDim.x * 256) {
        const int px = e / cpp, c = 8 * (e - px * cpp);
        const f32x4 u = fbuf_load_x4(Sb, (px * sps + c) * 4, 0), v = fbuf_load_x4(Sb, (px * sps + c + 4) * 4, 0);
        const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += x[k];
        planes_store8(Db, (px * dps + c) * 2, pl, x, true);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) CCA_LDS_ST(red + tid * 8 + k, acc[k]);
    __syncthreads();
    // thread t of this workgroup worked on chunk (blockIdx.x * 256 + t) % cpp (constant over its iterations: gridDim.x * 256 % cpp == 0)
    float *row = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C;
    const int first = (int)((blockIdx.x * 256u) % (unsigned)cpp);
    for (int idx = tid; idx < C; idx += 256) {
        const int c = idx >> 3, k = idx & 7;
        float t = 0.f;
        for (int t0 = (c - first + cpp) % cpp; t0 < 256; t0 += cpp) t += CCA_LDS_LD(red + t0 * 8 + k);
        row[idx] = t;
    }
}
// out[c] = sum over ``rows`` rows of partials[row][c] in a FIXED association: a workgroup of 1024 threads owns 16 channels; thread
// (channel c, lane rl < 64) adds the rows rl, rl + 64, ... in order, then the 64 lanes of a channel are added in lane order.  (One
// thread per channel -- 1280 dependent loads on a dozen wavefronts -- took longer than the split pass itself: +190 us per module
// step; 16 lanes per channel: 27 us; profiles/r05z_module_small_batches.txt, r05j_split_colsum_probe.txt.)
constexpr int CS_LANES = 64;
__global__ __launch_bounds__(16 * CS_LANES) void colsum_reduce_kernel(const float *__restrict__ partials, int rows, int C, float *__restrict__ out) {
    __shared__ float red[16 * CS_LANES];
    CCA_LDS_REGISTER(red);
    const int tid = threadIdx.x, c = blockIdx.x * 16 + (tid & 15), rl = tid >> 4;
    float t = 0.f;
    if (c < C)
        for (int r = rl; r < rows; r += CS_LANES) t += partials[(size_t)r * C + c];
    CCA_LDS_ST(red + tid, t);
    __syncthreads();
    if (rl == 0 && c < C) {
        float u = 0.f;
#pragma unroll 8
        for (int k = 0; k < CS_LANES; ++k) u += CCA_LDS_LD(red + 16 * k + (tid & 15));
        out[c] = u;
    }
}

// NCHW fp32 -> planes (the gradient dy of an NCHW module output): 64 pixels x 64 channels per workgroup through a padded
// LDS tile; reads runs of 64 pixels per channel, writes 128-byte plane rows.
__global__ __launch_bounds__(256) void nchw_to_planes_kernel(const float *__restrict__ src, bf16p_t *__restrict__ dst, int C, int HW,
                                                             long sbs, long dbs, int dps, PlaneLayout pl) {
    __shared__ float tile[64 * 65];
    const int ntp = (HW + 63) / 64;
    // a channel's run of 64 pixels is 256 B at arbitrary alignment (HW is odd at 97 x 97): three 128-byte lines, the outer two
    // shared with the neighbouring pixel tiles.  Workgroup ids go round-robin over the 8 XCDs (private L2s): with neighbouring
    // tiles on neighbouring ids every boundary line was fetched from HBM twice (PMC: 384 MB for 308).  Every XCD now takes a
    // contiguous range of (image, channel group, pixel tile) ids, tile fastest: the second touch of a line is an L2 hit.
    const int lid = xcd_logical_id((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int ncg = (int)gridDim.y, tix = lid % ntp, rest = lid / ntp;
    const int b = rest / ncg, p0 = tix * 64, c0 = (rest - b * ncg) * 64;
    const FBuf Sb = make_fbuf(src + (size_t)b * sbs, (size_t)C * HW * sizeof(float));
    const FBuf Db = make_fbuf(reinterpret_cast<const float *>(dst + (size_t)b * dbs), ((size_t)(HW - 1) * dps + pl.width) * 2);
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ch = (tid >> 4) + 16 * it, p4 = 4 * (tid & 15);
        const bool ok = c0 + ch < C && p0 + p4 < HW;
        const f32x4 v = fbuf_load_x4(Sb, ok ? ((c0 + ch) * HW + p0 + p4) * 4 : kOobOffset, 0);   // (dwords beyond the tensor read 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ch * 65 + p4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int px = (tid >> 3) + 32 * it, c8 = 8 * (tid & 7);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = tile[(c8 + e) * 65 + px];
        planes_store8(Db, ((p0 + px) * dps + c0 + c8) * 2, pl, x, p0 + px < HW && c0 + c8 < C);
    }
}

// The module's three 1x1 projections (functions.py:29,32,35) as the operands of ONE stacked GEMM, packed by ONE launch:
//   w   (N, C) fp32, N = 2 Cq + C rows: query | key | value weights;   b (N) fp32: their biases
//   w3  (N, 3 C) bf16: row n = [wh | wl | wh]   (the K-concatenated operand of the split-bf16 x3 projection x^T W^T; may be null)
//   w3t (C, 3 N) bf16: row c = [wh^T | wh^T | wl^T]   (of its adjoint dx = W^T dqkv^T; null with w3)
// with wh = bf16_rne(w), wl = bf16_rne(w - wh) -- the split of bf16_split8 / torch's .to(bfloat16).  One thread per element; the
// host side calls it on EVERY forward (3 us for 1.3 MB) instead of caching torch.cat / split results across calls: a cache
// keyed on tensor versions goes stale under ``p.data`` updates, a launch per call cannot.
__global__ __launch_bounds__(256) void pack_projection_kernel(const float *__restrict__ wq, const float *__restrict__ wk,
                                                              const float *__restrict__ wv, const float *__restrict__ bq,
                                                              const float *__restrict__ bk, const float *__restrict__ bv,
                                                              float *__restrict__ w, float *__restrict__ b, uint16_t *__restrict__ w3,
                                                              uint16_t *__restrict__ w3t, int C, int Cq) {
    const int N = 2 * Cq + C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)N * C; e += (long)gridDim.x * 256) {
        const int n = (int)(e / C), c = (int)(e - (long)n * C);
        const float x = n < Cq ? wq[(long)n * C + c] : n < 2 * Cq ? wk[(long)(n - Cq) * C + c] : wv[(long)(n - 2 * Cq) * C + c];
        w[e] = x;
        if (c == 0) b[n] = n < Cq ? bq[n] : n < 2 * Cq ? bk[n - Cq] : bv[n - 2 * Cq];
        if (w3) {
            const uint32_t h = cvt_pk_bf16(x, 0.f) & 0xffffu;
            const uint32_t l = cvt_pk_bf16(x - __builtin_bit_cast(float, h << 16), 0.f) & 0xffffu;
            uint16_t *r = w3 + (long)n * 3 * C + c;
            r[0] = (uint16_t)h; r[C] = (uint16_t)l; r[2 * C] = (uint16_t)h;
            uint16_t *t = w3t + (long)c * 3 * N + n;
            t[0] = (uint16_t)h; t[N] = (uint16_t)h; t[2 * N] = (uint16_t)l;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gweight: T[b, pixel(i, g), a_off + j] = sum_c X[pixel(i, g), c] * Y[pixel(j, g), c], bf16 pixel-major X / Y
//   ca_forward (X = q, Y = k, K = C/8, MASK: the column self slot is -inf) and the dA half of ca_map_backward
//   (X = dy, Y = v, K = C).  Workgroup = one strip; the channel axis is contiguous, so both MFMA operands are single
//   ds_read_b128 fragments of the swizzled bf16 tiles and the products are exact: no split, one
//   v_mfma_f32_16x16x32_bf16 per tile and 32 channels.  Wavefront w owns the tile rows ti = w, w + 8, ...
// ---------------------------------------------------------------------------------------------------------------
// SINGLE: the contraction is one 64-channel chunk (the energies, K = C/8 <= 64): no double buffering, half the LDS, and
// two (fp32) or more workgroups per CU overlap their latency chains.
// LONG (strips of 133 .. 4 x 132 positions: row strips in ``nb`` blocks each, column strips in ``nbc`` -- see long_block; a branch
// whose strips fit P has one block): a workgroup computes the block T[query block I][key block J] of a strip from the X tile of
// block I and the Y tile of block J.
template <int P, bool MASK, typename FT, bool SINGLE, bool LONG = false>
__global__ __launch_bounds__(GM_THREADS, SINGLE ? 2 : 1) void gweight_kernel(const FT *__restrict__ X, const FT *__restrict__ Y,
                                                                              float *__restrict__ T, int Cx, int H, int W,
                                                                              long xbs, int xps, long ybs, int yps, int nb = 1, int nbc = 1,
                                                                              int n_whole = 0) {
    constexpr bool BF = GTile<FT>::BF;
    constexpr bool PL = std::is_same<FT, bf16p_t>::value;     // split planes: an operand tile = hi image | lo image (bf16 tile geometry)
    static_assert(!(PL && MASK), "gweight: the energies are computed from fp32 q, k (exact products)");
    constexpr bool EXACT = !BF && !PL && MASK;          // the energies feed exp(): exact fp32 products
    constexpr bool EXACT_PART = EXACT && SINGLE && !LONG;   // (the form that takes tail parts, see n_whole)
    constexpr bool PRESPLIT = !BF && !PL && !MASK;      // fp32 dA: tiles are split into bf16 hi / lo images once per chunk
    // split planes: T16 tile geometry (unpadded 1 KiB pieces, cca_gmap's plane tiles) so that THREE stages of X hi | X lo |
    // Y hi | Y lo fit the 160 KB (159,744 B at P = 100): two stages in flight while one is multiplied (with two stages the
    // launch ran at 4.2 TB/s, bound by the one stage a CU had in flight: profiles/r03e_bench.json)
    // bf16 features use the same unpadded row-read T16 geometry (the padded 8-pixel pieces it had put positions p and p + 8 two
    // bank quads apart: 48 % LDS conflict cycles in the configs[4] dA / energies launches, profiles/r03v_bf16_config5_pmc_summary.json)
    constexpr bool T16 = PL || BF;
    constexpr int TSB = T16 ? t16_size(P) : GTile<bf16_t>::size(P);   // dwords per bf16 image
    constexpr int NPB = T16 ? t16_pieces(P) : GTile<bf16_t>::pieces(P);
    constexpr int NT = (P + 15) / 16, NTR = (NT + GM_WAVES - 1) / GM_WAVES;
    constexpr int TSZ = PL ? 2 * TSB : BF ? TSB : GTile<FT>::size(P), NPF = PL ? 2 * NPB : BF ? NPB : GTile<FT>::pieces(P);
    constexpr int NBUF = SINGLE ? 1 : (PL && 3 * 2 * 2 * TSB * 4 <= 163840) ? 3 : 2;
    constexpr int D = NBUF > 1 ? NBUF - 1 : 1;                        // fills run D stages ahead
    constexpr int LDS = 2 * NBUF * TSZ + (PRESPLIT ? 4 * TSB : 0);
    static_assert(LDS * 4 <= 163840, "gweight: LDS");
    __shared__ __attribute__((aligned(16))) float lds[LDS];
    CCA_LDS_REGISTER(lds);
    const int HW = H * W, S = H + W;
    // TAIL PARTS (the fp32 energies, n_whole > 0): the launch's workgroups are latency chains of equal length, so 1552 strips on 768
    // slots take THREE rounds for 2.02 rounds of work.  The strips beyond the whole rounds are cut by query tile row: workgroup
    // n_whole + NT s + r computes tile row r of strip n_whole + s, its key tiles spread over the wavefronts -- the same fill, a
    // seventh of the multiply and of the stores: a short last round (dispatched last: the cut is made on the LINEAR id).
    constexpr int NTP = (P + 15) / 16;
    const bool part = EXACT_PART && n_whole > 0 && (int)blockIdx.x >= n_whole;
    const int trow = part ? ((int)blockIdx.x - n_whole) % NTP : 0;
    const int id = part ? n_whole + ((int)blockIdx.x - n_whole) / NTP
                        : xcd_logical_id(blockIdx.x, (EXACT_PART && n_whole > 0) ? n_whole : (int)gridDim.x);
    const int nbr2 = LONG ? nb * nb : 1, nbc2 = LONG ? nbc * nbc : 1;
    const int per_image = W * nbc2 + H * nbr2;                // W column strips, then H row strips (LONG: blocks x blocks tiles each)
    const int b = id / per_image, r = id - b * per_image;
    const bool row = r >= W * nbc2;
    const int nbs = LONG ? (row ? nb : nbc) : 1, nbb = nbs * nbs;           // blocks of this strip
    const int rs = row ? r - W * nbc2 : r;
    const int blk = LONG ? rs % nbb : 0;
    const int g = rs / nbb;
    const int Ls = row ? W : H;                               // the strip; this workgroup's query / key ranges:
    const int lb = LONG ? long_block(Ls, nbs) : 0;
    const int i0 = LONG ? (blk / nbs) * lb : 0, j0 = LONG ? (blk % nbs) * lb : 0;
    const int L = LONG ? (Ls - i0 < lb ? Ls - i0 : lb) : Ls;               // query positions (rows of T)
    const int Lk = LONG ? (Ls - j0 < lb ? Ls - j0 : lb) : Ls;              // key positions (slots)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int pstep = row ? 1 : W;
    const int pix0 = (row ? g * W : g) + i0 * pstep, pixY = (row ? g * W : g) + j0 * pstep, a_off = (row ? H : 0) + j0;
    const FBuf Xb = make_fbuf(reinterpret_cast<const float *>(X + (size_t)b * xbs), ((size_t)(HW - 1) * xps + (PL ? 2 : 1) * Cx) * sizeof(FT));
    const FBuf Yb = make_fbuf(reinterpret_cast<const float *>(Y + (size_t)b * ybs), ((size_t)(HW - 1) * yps + (PL ? 2 : 1) * Cx) * sizeof(FT));
    const int nch = (Cx + GM_CG - 1) / GM_CG;                 // (SINGLE: the host launches this form only when nch == 1)

    auto issue = [&](int ch) {
        float *xb = lds + (ch % NBUF) * 2 * TSZ, *yb = xb + TSZ;
        for (int it = wv; it < 2 * NPF; it += GM_WAVES) {
            if constexpr (PL) {
                const int op = it >= NPF, r = it - op * NPF, plane = r >= NPB;
                t16_dma_piece<false>(op ? Yb : Xb, (op ? yb : xb) + plane * TSB, r - plane * NPB, lane, op ? pixY : pix0, pstep,
                              op ? Lk : L, op ? yps : xps, ch * GM_CG, Cx, plane ? Cx : 0);
            } else if constexpr (BF) {
                const int op = it >= NPF;
                t16_dma_piece<false>(op ? Yb : Xb, op ? yb : xb, it - op * NPF, lane, op ? pixY : pix0, pstep, op ? Lk : L,
                                     op ? yps : xps, ch * GM_CG, Cx, 0);
            } else {
                if (it < NPF) gtile_dma_piece<FT>(Xb, xb, it, lane, pix0, pstep, L, xps, ch * GM_CG, Cx);
                else          gtile_dma_piece<FT>(Yb, yb, it - NPF, lane, pixY, pstep, Lk, yps, ch * GM_CG, Cx);
            }
        }
    };
    f32x4 acc[NTR][NT];
#pragma unroll
    for (int a = 0; a < NTR; ++a)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bf16 fragment = the 8 consecutive channels 32 kk + 8 lg .. + 7 of one pixel = one 16-byte chunk (chunk q of pixel p
    // at chunk position q ^ (p & 7)) of a bf16 image
    auto frag = [&](const float *tile, int pixel_, int kk) {
        const int pixel = pixel_ < 8 * NPB ? pixel_ : 0;                            // (tile rows beyond the strip: results unused)
        const int chunk = 4 * kk + lg;
        if constexpr (T16) return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(tile) + t16_byte<false>(pixel, 8 * chunk));
        const int off = (pixel >> 3) * GM_PB + (pixel & 7) * 32 + ((chunk ^ (pixel & 7)) << 2);
        return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint32_t *>(tile) + off);
    };
    float *const img = lds + 2 * NBUF * TSZ;     // PRESPLIT: X_hi | X_lo | Y_hi | Y_lo
    // this wavefront's pieces per stage (every piece is issued whatever the strip length: zero fill beyond it)
    const int npw = (2 * NPF - wv + GM_WAVES - 1) / GM_WAVES;
    issue(0);
    if (D > 1 && nch > 1) issue(1);
    for (int ch = 0; ch < nch; ++ch) {
        const float *xb = lds + (ch % NBUF) * 2 * TSZ, *yb = xb + TSZ;
        // stage ch landed, every wavefront is done with stage ch - 1; the D - 1 stages issued after it may stay in flight
        if (D > 1 && ch + 1 < nch) barrier_dma_keep_n(npw);
        else                       barrier_dma_keep<0>();
        if (ch + D < nch) issue(ch + D);
        if constexpr (EXACT) {
            // v_mfma_f32_16x16x4_f32 (bit-identical to an fmaf chain), 16 k-steps of 4 channels; lane (ln, lg) holds
            // channel 4 ks + lg of pixel ln of its tile
#pragma unroll 4
            for (int ks = 0; ks < GM_CG / 4; ++ks) {
                float af[NTR];
#pragma unroll
                for (int a = 0; a < NTR; ++a) {
                    const int px = 16 * (part ? trow : wv + GM_WAVES * a) + ln;
                    af[a] = CCA_LDS_LD(xb + gtile_f32_idx(px < NPF * 4 ? px : 0, 4 * ks + lg));
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t * 16 < Lk && (!part || t == wv)) {             // (a tail part: wavefront w multiplies key tile w of ITS row)
                        const int px = 16 * t + ln;
                        const float bv = CCA_LDS_LD(yb + gtile_f32_idx(px < NPF * 4 ? px : 0, 4 * ks + lg));
#pragma unroll
                        for (int a = 0; a < NTR; ++a)
                            // (operands swapped -- D[key][query], the same k-ordered fmaf chain per element: a lane ends with four
                            //  CONSECUTIVE slots of one query = one 16-byte store instead of four 4-byte ones)
                            if ((part ? a == 0 : (wv + GM_WAVES * a) * 16 < L)) acc[a][t] = mfma_16x16x4(bv, af[a], acc[a][t]);
                    }
                }
            }
            continue;
        }
        const float *xh = xb, *xl = xb, *yh = yb, *yl = yb;
        if constexpr (PL) { xl = xb + TSB; yl = yb + TSB; }
        if constexpr (PRESPLIT) {
            // every element is split ONCE (not once per wavefront that needs it): thread -> (tensor, pixel, 8-channel chunk)
            for (int e = tid; e < 2 * P * 8; e += GM_THREADS) {
                const int ten = e >= P * 8, e2 = ten ? e - P * 8 : e, px = e2 >> 3, q = e2 & 7;
                const float *src = ten ? yb : xb;
                const f32x4 u = lds_load_x4(src + gtile_f32_idx(px, 8 * q)), v = lds_load_x4(src + gtile_f32_idx(px, 8 * q + 4));
                const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                const BfSplit sp = bf16_split8(x);
                const int off = (px >> 3) * GM_PB + (px & 7) * 32 + ((q ^ (px & 7)) << 2);
                uint32_t *d = reinterpret_cast<uint32_t *>(img + ten * 2 * TSB) + off;
                *reinterpret_cast<u32x4 *>(d) = sp.hi;
                *reinterpret_cast<u32x4 *>(d + TSB) = sp.lo;
            }
            barrier_lds_only();
            xh = img; xl = img + TSB; yh = img + 2 * TSB; yl = img + 3 * TSB;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                       // two k-steps of 32 channels
            u32x4 ah[NTR], al[NTR];
#pragma unroll
            for (int a = 0; a < NTR; ++a) {
                ah[a] = frag(xh, 16 * (wv + GM_WAVES * a) + ln, kk);
                if (!BF) al[a] = frag(xl, 16 * (wv + GM_WAVES * a) + ln, kk);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t * 16 < Lk) {
                    const u32x4 bh = frag(yh, 16 * t + ln, kk);
                    u32x4 bl = bh;
                    if (!BF) bl = frag(yl, 16 * t + ln, kk);
#pragma unroll
                    for (int a = 0; a < NTR; ++a)
                        if ((wv + GM_WAVES * a) * 16 < L) {
                            acc[a][t] = mfma_bf16_16x16x32(ah[a], bh, acc[a][t]);
                            if (!BF) {
                                acc[a][t] = mfma_bf16_16x16x32(ah[a], bl, acc[a][t]);
                                acc[a][t] = mfma_bf16_16x16x32(al[a], bh, acc[a][t]);
                            }
                        }
                }
            }
        }
    }
    if constexpr (!BF && MASK) mfma_f32_result_fence();
    // D[m = query position 16 ti + 4 lg + q][n = key position 16 t + ln] -> T rows (64-byte runs per query)
    float *Tg = T + (size_t)b * HW * S;
    if constexpr (EXACT) {
        // D[m = key position 16 t + 4 lg + q][n = query position 16 ti + ln]: 16-byte runs of four slots per query (a run that would
        // cross the end of the branch's slots leaves as single stores: the next slots belong to another workgroup)
        const FBuf Tb = make_fbuf(Tg, (size_t)HW * S * sizeof(float));
#pragma unroll
        for (int a = 0; a < NTR; ++a)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int i = 16 * (part ? trow : wv + GM_WAVES * a) + ln, j = 16 * t + 4 * lg;
                if (t * 16 < Lk && (part ? (a == 0 && t == wv) : (wv + GM_WAVES * a) * 16 < L)) {          // (wave-uniform)
                    f32x4 val = acc[a][t];
                    if (!row) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (i0 + i == j0 + j + q) val[q] = -INFINITY;     // functions.py:11-12 (column self slot)
                    }
                    const int off = ((pix0 + i * pstep) * S + a_off + j) * 4;
                    if (j + 3 < Lk) {
                        fbuf_store_x4(Tb, val, i < L ? off : kOobOffset, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            if (j + q < Lk) fbuf_store(Tb, val[q], i < L ? off + 4 * q : kOobOffset, 0);
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < NTR; ++a)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (part ? trow : wv + GM_WAVES * a) + 4 * lg + q, j = 16 * t + ln;
                if (i < L && j < Lk && (!part || (a == 0 && t == wv))) {
                    float val = acc[a][t][q];
                    if (MASK && !row && i0 + i == j0 + j) val = -INFINITY;  // functions.py:11-12 (column self slot)
                    Tg[(size_t)(pix0 + i * pstep) * S + a_off + j] = val;
                }
            }
}

// ---------------------------------------------------------------------------------------------------------------
// gweight_stream: the dA contraction (T = X_g Y_g^T per strip, K = all channels) on SPLIT-PLANE operands as a PERSISTENT
// kernel.  gweight_kernel runs one workgroup per strip and one workgroup per CU (three stages of X hi | X lo | Y hi | Y lo
// fill the LDS): every strip then pays its first fill (nothing to multiply meanwhile) and its 37 KB of result stores (nothing
// in flight behind them) alone on its CU -- 169 us at the headline shape of which ~117 us is streaming at the CU's share of
// HBM (profiles/r03g_bench.json, r03g_pmc_step_summary.json).  Here a workgroup walks strips first, first + grid, ... and its
// ring of stages runs ACROSS strip boundaries: while the last chunks of one strip are multiplied the first stages of the next
// are landing, and the result stores of a strip drain behind the next strip's fills.  Counted barriers as everywhere: the
// result stores are buffer stores issued only under wave-uniform conditions that leave at least one lane in range, so they
// can be counted (a fully out-of-range store retires early and would break the in-order count).
// ---------------------------------------------------------------------------------------------------------------
// FT = bf16p_t (split planes, three products per term) or bf16_t (bf16 features of BASELINE configs[4]: exact single products)
// (launch bound "4 waves per SIMD": a REGISTER bound of 128 -- the LDS ring admits one workgroup per CU anyway; without it the
// bf16 / P = 100 instantiation came out with 256 VGPRs and 13 spilled)
// YT = float: the Y operand (v) is read as the fp32 pixel-major tensor it is (F32T tiles, hi | lo split per fragment) -- X (dy)
// stays planes: it has to be transposed out of NCHW anyway.
// NB: ring stages (0 = as many as fit the LDS: three at P = 100).  NB = 2 leaves 53 KB of the CU's LDS free: a column-pass workgroup
// of the dv launch (gmap3_kernel<..., 2, 3>: 53,248 B) then runs NEXT TO the persistent workgroup instead of waiting for it to exit.
template <int P, typename FT = bf16p_t, typename YT = FT, int NB = 0>
__global__ __launch_bounds__(GM_THREADS, 4) void gweight_stream_kernel(const FT *__restrict__ X, const YT *__restrict__ Y,
                                                                        float *__restrict__ T, int Cx, int B, int H, int W,
                                                                        long xbs, int xps, long ybs, int yps) {
    constexpr int NPL = std::is_same<FT, bf16p_t>::value ? 2 : 1;        // planes per operand
    static_assert(NPL == 2 || std::is_same<FT, bf16_t>::value, "gweight_stream: bf16p_t or bf16_t operands");
    constexpr int TSB = t16_size(P), NPB = t16_pieces(P);                 // one plane tile: dwords, 1 KiB pieces
    constexpr int NT = (P + 15) / 16, NTR = (NT + GM_WAVES - 1) / GM_WAVES;
    constexpr bool YF = std::is_same<YT, float>::value;
    static_assert(!YF || (NPL == 2 && f32t_size(P) <= 2 * TSB), "gweight_stream: an fp32 Y tile takes the place of its two planes");
    constexpr int NPY = YF ? f32t_pieces(P) : NPL * NPB;                  // DMA pieces of a Y tile
    constexpr int STG = 2 * NPL * TSB, NPS = NPL * NPB + NPY;             // stage = X (hi | lo) | Y (hi | lo, or fp32)
    constexpr int NBUF = NB ? NB : 4 * STG * 4 <= 163840 ? 4 : 3, D = NBUF - 1;     // as many stages as fit: D of them in flight
    static_assert(NBUF >= 2 && NBUF * STG * 4 <= 163840, "gweight_stream: the stages must fit the LDS");
    __shared__ __attribute__((aligned(16))) float lds[NBUF * STG];
    CCA_LDS_REGISTER(lds);
    const int HW = H * W, S = H + W;
    const int nstrips = B * S, first = blockIdx.x, step = gridDim.x;
    const int nloc = first < nstrips ? (nstrips - first + step - 1) / step : 0;
    const int nch = (Cx + GM_CG - 1) / GM_CG, total = nloc * nch;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;

    struct Strip { int b, L, pix0, pstep, a_off; };
    auto strip_of = [&](int n) {                                            // the strip stage n belongs to
        const int sidx = first + (n / nch) * step;
        const int b = sidx / S, r = sidx - b * S;
        const bool row = r >= W;
        const int g = row ? r - W : r;
        return Strip{b, row ? W : H, row ? g * W : g, row ? 1 : W, row ? H : 0};
    };
    const int npw = (NPS - wv + GM_WAVES - 1) / GM_WAVES;                   // this wavefront's pieces per stage (all are issued)
    auto issue = [&](int n) {
        const Strip st = strip_of(n);
        const int ch = n % nch;
        const FBuf Xb = make_fbuf(reinterpret_cast<const float *>(X + (size_t)st.b * xbs), ((size_t)(HW - 1) * xps + NPL * Cx) * 2);
        const FBuf Yb = make_fbuf(reinterpret_cast<const float *>(Y + (size_t)st.b * ybs),
                                  YF ? ((size_t)(HW - 1) * yps + Cx) * 4 : ((size_t)(HW - 1) * yps + NPL * Cx) * 2);
        float *dst = lds + (n % NBUF) * STG;
#pragma unroll
        for (int k = 0; k < (NPS + GM_WAVES - 1) / GM_WAVES; ++k) {
            const int it = wv + GM_WAVES * k;
            if (it < NPS) {                                                 // (wave-uniform)
                const int op = it >= NPL * NPB, r = it - op * NPL * NPB, plane = r >= NPB;
                if (YF && op)
                    f32t_dma_piece(Yb, dst + NPL * TSB, r, lane, st.pix0, st.pstep, st.L, yps, ch * GM_CG, Cx);
                else
                    t16_dma_piece<false>(op ? Yb : Xb, dst + (NPL * op + plane) * TSB, r - plane * NPB, lane, st.pix0, st.pstep, st.L,
                                  op ? yps : xps, ch * GM_CG, Cx, plane ? Cx : 0);
            }
        }
    };
    // result stores this wavefront issues for a strip of length L: one per (owned tile row, q, key tile) that has a lane in range
    auto nstores = [&](int L) {
        if constexpr (YF) {                                                 // (fp32 Y: a wavefront owns a COLUMN of tiles, see the multiply)
            int rows = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int q = 0; q < 4; ++q) rows += 16 * ti + q < L ? 1 : 0;
            return 16 * wv < L ? rows : 0;
        }
        int rows = 0;
#pragma unroll
        for (int a = 0; a < NTR; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) rows += 16 * (wv + GM_WAVES * a) + q < L ? 1 : 0;
        return rows * ((L + 15) / 16);
    };
    auto stores_after = [&](int m) {                                        // stores issued at the end of iteration m
        return (m >= 0 && m % nch == nch - 1) ? nstores(strip_of(m).L) : 0;
    };
    auto frag = [&](const float *tile, int pixel_, int kk) {               // 8 consecutive channels of one position: 16 bytes
        const int pixel = pixel_ < 8 * NPB ? pixel_ : 0;
        return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(tile) + t16_byte<false>(pixel, 8 * (4 * kk + lg)));
    };

    if (total == 0) return;
#pragma unroll
    for (int n = 0; n < D; ++n)
        if (n < total) issue(n);
    f32x4 acc[NTR][NT];
    Strip cur = strip_of(0);
    for (int n = 0; n < total; ++n) {
        const int ch = n % nch;
        // stage n landed, every wavefront is done with stage n - 1.  Issued after the fill of stage n, oldest first (iteration m
        // issues fill(m + D), then stores(m)): stores(n - D), fill(n + 1), stores(n - D + 1), ..., fill(n + D - 1), stores(n - 1)
        // -- they may stay in flight.
        if (n == 0) {
            barrier_dma_keep<0>();
        } else {
            int keep = 0;
#pragma unroll
            for (int d = 1; d <= D; ++d) keep += stores_after(n - d) + (d < D && n + d < total ? npw : 0);
            barrier_dma_keep_n(keep);
        }
        if (n + D < total) issue(n + D);
        if (ch == 0) {
            cur = strip_of(n);
#pragma unroll
            for (int a = 0; a < NTR; ++a)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int L = cur.L;
        const float *xh = lds + (n % NBUF) * STG, *xl = xh + (NPL - 1) * TSB, *yh = xh + NPL * TSB, *yl = yh + (NPL - 1) * TSB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                                    // two k-steps of 32 channels
            if constexpr (YF) {
                // fp32 Y: the wavefront owns the COLUMN of tiles over key tile wv -- acc[0][ti] = (query tile ti, key tile wv) -- so it
                // splits ONE Y fragment per k-step in registers (its own key tile: 2 x 16 bytes out of LDS, v_cvt_pk + subtract) and
                // reads the seven X fragments as plane chunks.  (Owning a ROW of tiles, as the plane form does, every wavefront
                // splits all seven Y fragments: 203 us against 155 us; splitting the tile once in LDS in place: 179 us.)
                static_assert(!YF || NT <= GM_WAVES, "gweight_stream: one key tile per wavefront");
                const int pos = 16 * wv + ln < 4 * NPY ? 16 * wv + ln : 0, c = 32 * kk + 8 * lg;
                const f32x4 u = lds_load_x4(yh + f32t_idx(pos, c)), v = lds_load_x4(yh + f32t_idx(pos, c + 4));
                const float yx[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                const BfSplit sp = bf16_split8(yx);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    if (ti * 16 < L && wv * 16 < L) {
                        const u32x4 xa = frag(xh, 16 * ti + ln, kk), xb = frag(xl, 16 * ti + ln, kk);
                        acc[0][ti] = mfma_bf16_16x16x32(xa, sp.hi, acc[0][ti]);
                        acc[0][ti] = mfma_bf16_16x16x32(xa, sp.lo, acc[0][ti]);
                        acc[0][ti] = mfma_bf16_16x16x32(xb, sp.hi, acc[0][ti]);
                    }
                }
                continue;
            }
            u32x4 ah[NTR], al[NTR];
#pragma unroll
            for (int a = 0; a < NTR; ++a) {
                ah[a] = frag(xh, 16 * (wv + GM_WAVES * a) + ln, kk);
                if constexpr (NPL == 2) al[a] = frag(xl, 16 * (wv + GM_WAVES * a) + ln, kk);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t * 16 < L) {
                    u32x4 bh, bl;
                    {
                        bh = frag(yh, 16 * t + ln, kk);
                        bl = bh;
                        if constexpr (NPL == 2) bl = frag(yl, 16 * t + ln, kk);
                    }
#pragma unroll
                    for (int a = 0; a < NTR; ++a)
                        if ((wv + GM_WAVES * a) * 16 < L) {
                            acc[a][t] = mfma_bf16_16x16x32(ah[a], bh, acc[a][t]);
                            if constexpr (NPL == 2) {
                                acc[a][t] = mfma_bf16_16x16x32(ah[a], bl, acc[a][t]);
                                acc[a][t] = mfma_bf16_16x16x32(al[a], bh, acc[a][t]);
                            }
                        }
                }
            }
        }
        if (ch == nch - 1) {
            // D[m = query position 16 ti + 4 lg + q][n = key position 16 t + ln] -> T rows (64-byte runs per query)
            const FBuf Tb = make_fbuf(T + (size_t)cur.b * HW * S, (size_t)HW * S * sizeof(float));
            if constexpr (YF) {                 // acc[0][ti]: query tile ti x key tile wv
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (16 * wv < L && 16 * ti + q < L) {               // (wave-uniform: lane (ln, lg) = (0, 0) is in range)
                            const int i = 16 * ti + 4 * lg + q, j = 16 * wv + ln;
                            fbuf_store(Tb, acc[0][ti][q], (i < L && j < L) ? ((cur.pix0 + i * cur.pstep) * S + cur.a_off + j) * 4 : kOobOffset, 0);
                        }
                    }
            }
#pragma unroll
            for (int a = 0; a < (YF ? 0 : NTR); ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (16 * (wv + GM_WAVES * a) + q < L) {                 // (wave-uniform: lanes lg = 0 are in range)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            if (t * 16 < L) {                               // (wave-uniform: lanes ln = 0 are in range)
                                const int i = 16 * (wv + GM_WAVES * a) + 4 * lg + q, j = 16 * t + ln;
                                fbuf_store(Tb, acc[a][t][q], (i < L && j < L) ? ((cur.pix0 + i * cur.pstep) * S + cur.a_off + j) * 4 : kOobOffset, 0);
                            }
                        }
                    }
                }
        }
    }
}


}  // namespace cca

// cca_weight.hpp -- "weight-type" strip kernel: channel contraction onto the attention axis.
//
//   T[b, pixel(i, g), a_off + j] = sum_c X[b, c, pos(i, g)] * Y[b, c, pos(j, g)]        i, j in [0, L)
//
// which is, per strip g, the L x L GEMM  X_g^T Y_g  with K = channels.  Used for
//   ca_forward       X = q,  Y = k,  K = C/8   (/root/reference/cc_attention/functions.py:38-39)
//   ca_map_backward  X = dy, Y = v,  K = C     (the dA part of autograd of functions.py:46-47)
//
// Work decomposition (MI355X): one workgroup = NS adjacent strips, one wavefront per strip; column and
// row strips are workgroups of the SAME launch.  Each wavefront keeps its whole L x L output stationary
// in registers as up to 7x7 tiles of the exact-fp32 v_mfma_f32_16x16x4_f32 (bit-identical to an fmaf
// chain).  Both operands stream through a TRIPLE-BUFFERED LDS image in chunks of 8 channels, filled by
// LDS-DMA (buffer_load_dword ... lds: no staging registers, no ds_write pass): while chunk n is multiplied,
// chunk n+1 is landing and the pieces of chunk n+2 are issued between the MFMAs; the one barrier per chunk
// waits with a COUNTED vmcnt that leaves the newest chunk's pieces in flight.
//
// LDS image of one channel of one operand: see "Strip-tile geometry" in cca_common.hpp (16-byte DMA pieces of
// 256 floats; one DMA instruction moves 1 KiB -- with 4-byte pieces the DMA *issue* alone cost a third of the
// kernel).  Lanes outside the strip tile are masked; channels >= Cx are fetched clamped and zeroed at
// fragment-read time.
#pragma once
#include "cca_common.hpp"

#include <type_traits>

namespace cca {

constexpr int W_KC = 8;                                     // channels per chunk = 2 MFMA k-steps
__host__ __device__ constexpr int w_pieces(int ns) { return strip_pieces_c(ns); }             // image length in 64-float units
__host__ __device__ constexpr int w_cp(int ns, bool row) { return w_pieces(ns) * 64 + (row ? 16 : 1); }
__host__ __device__ constexpr int w_cpmax(int ns) { return w_pieces(ns) * 64 + 16; }
__host__ __device__ constexpr int w_op(int ns) { return W_KC * w_cpmax(ns); }   // floats per operand per buffer
__host__ __device__ constexpr int w_lds_floats(int ns) { return 6 * w_op(ns); }  // 2 operands x 3 buffers (162,816 B at NS = 8)

// FULL: compile-time shape -- all 7x7 tiles, every DMA piece issued (strips at least kFullMinStrip long, see
//       cca_common.hpp) -> counted-vmcnt pipeline, no per-tile guards in the hot loop
// BF: always false since round 4 (the packed split-bf16 variant spilled 126 VGPRs and was retired; the parameter stays so
//     that the launch code keeps its shape)
template <int NS, bool ROW, bool MASK, bool FULL, bool BF>
__device__ __forceinline__ void weight_strip_body(float *lds, int b, int tile, const float *__restrict__ X,
                                                  const float *__restrict__ Y, float *__restrict__ T,
                                                  int Ctot, int c_begin, int Cx, int H, int W, long xbs, long ybs) {
    // channels [c_begin, Cx) of the Ctot-channel operands (a K split: the caller sums the partial results)
    constexpr int CP = w_cp(NS, ROW), OP = w_op(NS);
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    const int g0 = tile * NS;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int g = g0 + wv;
    const bool active = g < br.G;
    const int nt = FULL ? kMaxTiles : (L + kTile - 1) / kTile;
    const int npieces = FULL ? strip_pieces4_c(NS) : (NS * L + 255) / 256;  // 16-byte DMA pieces; FULL: L in 97..100
    const int gvalid = (br.G - g0 < NS) ? br.G - g0 : NS;          // strips of this tile inside the image
#define CCA_TILE_ON(t) (FULL || (t) < nt)

    StripLanes4<NS, ROW> sl;
    sl.init(lane, L, W, g0, gvalid);

    const FBuf Xb = make_fbuf(X + (size_t)b * xbs, (size_t)Ctot * HW * sizeof(float));   // xbs: batch stride (elements)
    const FBuf Yb = make_fbuf(Y + (size_t)b * ybs, (size_t)Ctot * HW * sizeof(float));

    // DMA of chunk c0 into buffer `buf`: (operand, channel) pairs are dealt round-robin to the NS waves
    auto issue = [&](int c0, int buf) {
#pragma unroll
        for (int pr = 0; pr < 2 * W_KC / NS; ++pr) {
            const int pair = wv + pr * NS;                           // 0 .. 2*W_KC-1, wave-uniform
            const int op = pair / W_KC, cc = pair % W_KC;
            float *dst = lds + (buf * 2 + op) * OP + cc * CP;
            const int c = (c0 + cc < Cx) ? c0 + cc : Cx - 1;          // clamped: the K padding is zeroed at fragment read
            strip_dma_channel<NS, ROW, FULL>(op ? Yb : Xb, dst, c * HW * 4, npieces, W, sl);
        }
    };

    f32x4 acc[kMaxTiles][kMaxTiles];
#pragma unroll
    for (int rm = 0; rm < kMaxTiles; ++rm)
#pragma unroll
        for (int rn = 0; rn < kMaxTiles; ++rn) acc[rm][rn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment offset: k = l >> 4 (channel), position = 16 t + (l & 15)
    const int ln = lane & 15, lk = lane >> 4;
    const int fr = lk * CP + strip_lds_index<NS, ROW>(ln, wv, L);
    const int tstep = ROW ? kTile : kTile * NS;                    // LDS distance between position tiles

    // DMA of the next chunk cut into single pieces, interleaved with the MFMAs of the current one (a piece costs
    // ~60-100 issue cycles: 26 of them in front of 98 MFMAs would add a third to every chunk)
    constexpr int PIECES = strip_pieces4_c(NS), PW = 2 * W_KC / NS, QT = PW * PIECES, SLOTS = (W_KC / 4) * kMaxTiles;
    auto dma_piece = [&](int q, int c0, int buf) {
        const int pr = q / PIECES, m = q % PIECES;
        if (!(FULL || m < npieces)) return;
        const int pair = wv + pr * NS;
        const int op = pair / W_KC, cc = pair % W_KC;
        const int c = (c0 + cc < Cx) ? c0 + cc : Cx - 1;
        if (FULL) {            // always issued (counted vmcnt); lanes beyond the tile deposit zeros
            if (sl.in_image(m, lane))
                fbuf_load_to_lds_x4(op ? Yb : Xb, lds + (buf * 2 + op) * OP + cc * CP + m * 256,
                                    sl.full_offset(m), c * HW * 4 + sl.piece_soff(m, W));
        } else if (sl.valid(m))
            fbuf_load_to_lds_x4(op ? Yb : Xb, lds + (buf * 2 + op) * OP + cc * CP + m * 256, sl.vb,
                                c * HW * 4 + sl.piece_soff(m, W));
    };

    const int nchunks = (Cx - c_begin + W_KC - 1) / W_KC;
    const int lastc = c_begin + (nchunks - 1) * W_KC;
    issue(c_begin, 0);
    issue(nchunks > 1 ? c_begin + W_KC : c_begin, 1);
    for (int n = 0; n < nchunks; ++n) {
        // chunk n landed; chunk n+1 (the newest QT pieces of every wave) may still be in flight
        // (the counted wait needs every wave to have issued exactly QT pieces per chunk: full tiles of full strips)
        if (FULL && gvalid == NS) barrier_dma_keep<QT>();
        else                      __syncthreads();
        // chunk n+2 -> buffer (n+2) % 3, last read in iteration n-1 (past the end: re-fetch the last chunk, harmless)
        const int c1 = (c_begin + (n + 2) * W_KC < lastc) ? c_begin + (n + 2) * W_KC : lastc;
        const int bnext = (n + 2) % 3, bcur = n % 3;
        if (active) {
            const float *xs = lds + (bcur * 2 + 0) * OP + fr;
            const float *ys = lds + (bcur * 2 + 1) * OP + fr;
#pragma unroll
            for (int ks = 0; ks < W_KC / 4; ++ks) {
                const bool kin = c_begin + n * W_KC + ks * 4 + lk < Cx;   // channels beyond Cx hold clamped data: zero A
                float a[kMaxTiles];
#pragma unroll
                for (int t = 0; t < kMaxTiles; ++t)
                    if (CCA_TILE_ON(t)) {
                        const float v = CCA_LDS_LD(xs + ks * 4 * CP + t * tstep);
                        a[t] = kin ? v : 0.f;
                    }
#pragma unroll
                for (int rn = 0; rn < kMaxTiles; ++rn) {
                    const int sidx = ks * kMaxTiles + rn;
#pragma unroll
                    for (int q = sidx * QT / SLOTS; q < (sidx + 1) * QT / SLOTS; ++q) dma_piece(q, c1, bnext);
                    if (CCA_TILE_ON(rn)) {
                        const float bb = CCA_LDS_LD(ys + ks * 4 * CP + rn * tstep);
#pragma unroll
                        for (int rm = 0; rm < kMaxTiles; ++rm)
                            if (CCA_TILE_ON(rm)) acc[rm][rn] = mfma_16x16x4(a[rm], bb, acc[rm][rn]);
                    }
                }
            }
        } else {
            issue(c1, bnext);                 // strips outside the image still own DMA channels
        }
    }

    if (!active) return;
    mfma_f32_result_fence();
    float *Tg = T + (size_t)b * HW * S + (size_t)g * br.as_g + br.a_off;
    const int iq4 = 4 * lk;
#pragma unroll
    for (int rm = 0; rm < kMaxTiles; ++rm)
#pragma unroll
        for (int rn = 0; rn < kMaxTiles; ++rn)
            if (CCA_TILE_ON(rm) && CCA_TILE_ON(rn)) {
                const int j = rn * kTile + ln;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int iq = rm * kTile + iq4 + r;
                    if (iq < L && j < L) {
                        float val = acc[rm][rn][r];
                        if (MASK && !ROW && iq == j) val = -INFINITY;   // functions.py:11-12 (column self slot)
                        Tg[iq * br.as_q + j] = val;
                    }
                }
            }
#undef CCA_TILE_ON
}

// One launch covers BOTH branches and all images: 1-D grid of B * ksplit * (tiles_col + tiles_row) workgroups in
// XCD-aware order, image-major, then K split, then column tiles, then row tiles.  K split (small batches: 26 strip
// tiles per image cannot fill 256 CUs): split s contracts channels [s * cps * 8, (s + 1) * cps * 8) and writes the
// partial result to slab s -- slab 0 is T, slab s >= 1 is extra + (s - 1) * slab_stride -- and the consumer (the
// softmax kernels) adds the slabs in a fixed order.
// KS: the K split is in use (otherwise ksplit == 1 and the channel range is the compile-time [0, Cx))
template <int NS, bool MASK, bool BF, bool KS = false>
__global__ __launch_bounds__(kWave * NS, 2) void weight_strip_kernel(const float *__restrict__ X,
                                                                      const float *__restrict__ Y,
                                                                      float *__restrict__ T, int Cx, int H, int W,
                                                                      int tiles_col, int tiles_row, long xbs, long ybs,
                                                                      int ksplit, int cps, float *__restrict__ extra,
                                                                      long slab_stride) {
    __shared__ float lds[w_lds_floats(NS)];
    CCA_LDS_REGISTER(lds);
    const int per_image = tiles_col + tiles_row;
    const int id = xcd_logical_id(blockIdx.x, gridDim.x);
    const int bs = id / per_image, t = id - bs * per_image;
    const int b = KS ? bs / ksplit : bs, split = KS ? bs - b * ksplit : 0;
    const int c_begin = KS ? split * cps * W_KC : 0;
    const int c_end = KS ? ((c_begin + cps * W_KC < Cx) ? c_begin + cps * W_KC : Cx) : Cx;
    float *Tout = (!KS || split == 0) ? T : extra + (size_t)(split - 1) * slab_stride;
    const bool row = t >= tiles_col;
    const int tile = row ? t - tiles_col : t;
    const int L = row ? W : H;
    const bool full = L >= kFullMinStrip;
    if (row) {
        if (full) weight_strip_body<NS, true, MASK, true, BF>(lds, b, tile, X, Y, Tout, Cx, c_begin, c_end, H, W, xbs, ybs);
        else      weight_strip_body<NS, true, MASK, false, BF>(lds, b, tile, X, Y, Tout, Cx, c_begin, c_end, H, W, xbs, ybs);
    } else {
        if (full) weight_strip_body<NS, false, MASK, true, BF>(lds, b, tile, X, Y, Tout, Cx, c_begin, c_end, H, W, xbs, ybs);
        else      weight_strip_body<NS, false, MASK, false, BF>(lds, b, tile, X, Y, Tout, Cx, c_begin, c_end, H, W, xbs, ybs);
    }
}

}  // namespace cca

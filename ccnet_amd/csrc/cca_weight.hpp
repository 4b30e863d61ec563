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
// chain).  Both operands stream through a DOUBLE-BUFFERED LDS image in chunks of 8 channels, filled by
// LDS-DMA (buffer_load_dword ... lds: no staging registers, no ds_write pass); chunk n+1 is in flight
// while chunk n is multiplied, and the single __syncthreads() per chunk is the hand-over (the compiler
// drains vmcnt in front of it).
//
// LDS image of one channel of one operand: the NS strips' L positions as ONE lane-linear array of
// ceil(NS*L/64) DMA pieces (64 dwords each):
//   column branch  p = i * NS + (gg ^ swz(i))   8 (4) consecutive w per position -> 32 B (16 B) segments;
//                  the XOR swizzle (2*((i>>2)&3) for NS=8, 2*((i>>3)&1) for NS=4) + channel pitch == 1 mod 32
//                  makes the stride-NS fragment reads bank-conflict-free
//   row branch     p = gg * L + i               the NS rows are contiguous in memory: fully coalesced 256 B
//                  pieces; channel pitch == 16 mod 32 keeps the unit-stride fragment reads conflict-free
// Out-of-range lanes (position >= L, strip outside the image) fetch a clamped, always-valid address and
// land in padding or in strips whose results are never stored; channels >= Cx are zero-filled.
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int W_KC = 8;                                     // channels per chunk = 2 MFMA k-steps
__host__ __device__ constexpr int w_pieces(int ns) { return strip_pieces_c(ns); }             // 13 / 7
__host__ __device__ constexpr int w_cp(int ns, bool row) { return w_pieces(ns) * 64 + (row ? 16 : 1); }
__host__ __device__ constexpr int w_cpmax(int ns) { return w_pieces(ns) * 64 + 16; }
__host__ __device__ constexpr int w_op(int ns) { return W_KC * w_cpmax(ns); }   // floats per operand per buffer
__host__ __device__ constexpr int w_lds_floats(int ns) { return 4 * w_op(ns); }  // 2 operands x 2 buffers

// FULL: the strip needs all 7x7 tiles (97..100 long) -> no per-tile guards in the hot loop
template <int NS, bool ROW, bool MASK, bool FULL>
__device__ __forceinline__ void weight_strip_body(float *lds, int b, int tile, const float *__restrict__ X,
                                                  const float *__restrict__ Y, float *__restrict__ T,
                                                  int Cx, int H, int W) {
    constexpr int CP = w_cp(NS, ROW), OP = w_op(NS);
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    const int g0 = tile * NS;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int g = g0 + wv;
    const bool active = g < br.G;
    const int nt = FULL ? kMaxTiles : (L + kTile - 1) / kTile;
    const int npieces = FULL ? strip_pieces_c(NS) : (NS * L + 63) / 64;     // FULL: L in 97..100
    const int gvalid = (br.G - g0 < NS) ? br.G - g0 : NS;          // strips of this tile inside the image
#define CCA_TILE_ON(t) (FULL || (t) < nt)

    StripLanes<NS, ROW> sl;
    sl.init(lane, L, W, g0, gvalid);

    const FBuf Xb = make_fbuf(X + (size_t)b * Cx * HW, (size_t)Cx * HW * sizeof(float));
    const FBuf Yb = make_fbuf(Y + (size_t)b * Cx * HW, (size_t)Cx * HW * sizeof(float));

    // DMA of chunk c0 into buffer `buf`: (operand, channel) pairs are dealt round-robin to the NS waves
    auto issue = [&](int c0, int buf) {
#pragma unroll
        for (int pr = 0; pr < 2 * W_KC / NS; ++pr) {
            const int pair = wv + pr * NS;                           // 0 .. 2*W_KC-1, wave-uniform
            const int op = pair / W_KC, cc = pair % W_KC;
            float *dst = lds + (buf * 2 + op) * OP + cc * CP;
            const int c = c0 + cc;
            if (c < Cx) {
                const FBuf &src = op ? Yb : Xb;
                const int soff = c * HW * 4;
                strip_dma_channel<NS, ROW, FULL>(src, dst, soff, npieces, W, sl);
            } else {
                for (int m = 0; m < npieces; ++m) CCA_LDS_ST(&dst[m * 64 + lane], 0.f);   // K padding must be 0
            }
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

    const int nchunks = (Cx + W_KC - 1) / W_KC;
    issue(0, 0);
    for (int n = 0; n < nchunks; ++n) {
        __syncthreads();                      // chunk n landed (vmcnt drained) and buffer (n+1)&1 is free again
        if (n + 1 < nchunks) issue((n + 1) * W_KC, (n + 1) & 1);
        if (active) {
            const float *xs = lds + ((n & 1) * 2 + 0) * OP + fr;
            const float *ys = lds + ((n & 1) * 2 + 1) * OP + fr;
#pragma unroll
            for (int ks = 0; ks < W_KC / 4; ++ks) {
                float a[kMaxTiles];
#pragma unroll
                for (int t = 0; t < kMaxTiles; ++t)
                    if (CCA_TILE_ON(t)) a[t] = CCA_LDS_LD(xs + ks * 4 * CP + t * tstep);
#pragma unroll
                for (int rn = 0; rn < kMaxTiles; ++rn)
                    if (CCA_TILE_ON(rn)) {
                        const float bb = CCA_LDS_LD(ys + ks * 4 * CP + rn * tstep);
#pragma unroll
                        for (int rm = 0; rm < kMaxTiles; ++rm)
                            if (CCA_TILE_ON(rm)) acc[rm][rn] = mfma_16x16x4(a[rm], bb, acc[rm][rn]);
                    }
            }
        }
    }

    if (!active) return;
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

// One launch covers BOTH branches and all images: 1-D grid of B * (tiles_col + tiles_row) workgroups in
// XCD-aware order, image-major, then column tiles, then row tiles.
template <int NS, bool MASK>
__global__ __launch_bounds__(kWave * NS, 2) void weight_strip_kernel(const float *__restrict__ X,
                                                                      const float *__restrict__ Y,
                                                                      float *__restrict__ T, int Cx, int H, int W,
                                                                      int tiles_col, int tiles_row) {
    __shared__ float lds[w_lds_floats(NS)];
    CCA_LDS_REGISTER(lds);
    const int per_image = tiles_col + tiles_row;
    const int id = xcd_logical_id(blockIdx.x, gridDim.x);
    const int b = id / per_image, t = id - b * per_image;
    const bool row = t >= tiles_col;
    const int tile = row ? t - tiles_col : t;
    const int L = row ? W : H;
    const bool full = L > (kMaxTiles - 1) * kTile;
    if (row) {
        if (full) weight_strip_body<NS, true, MASK, true>(lds, b, tile, X, Y, T, Cx, H, W);
        else      weight_strip_body<NS, true, MASK, false>(lds, b, tile, X, Y, T, Cx, H, W);
    } else {
        if (full) weight_strip_body<NS, false, MASK, true>(lds, b, tile, X, Y, T, Cx, H, W);
        else      weight_strip_body<NS, false, MASK, false>(lds, b, tile, X, Y, T, Cx, H, W);
    }
}

}  // namespace cca

// cca_weight.hpp -- "weight-type" strip kernel: channel contraction onto the attention axis.
//
//   T[b, pixel(i, g), a_off + j] = sum_c X[b, c, pos(i, g)] * Y[b, c, pos(j, g)]        i, j in [0, L)
//
// which is, per strip g, the L x L GEMM  X_g^T Y_g  with K = channels.  Used for
//   ca_forward       X = q,  Y = k,  K = C/8   (/root/reference/cc_attention/functions.py:38-39)
//   ca_map_backward  X = dy, Y = v,  K = C     (the dA part of autograd of functions.py:46-47)
//
// Work decomposition (MI355X): one workgroup = 8 adjacent strips (8 wavefronts, one strip each), so the
// column branch reads 8 consecutive w per (c, h) and the row branch reads 8 full rows per c.  Each
// wavefront keeps its whole L x L output stationary in registers as up to 7x7 tiles of the exact-fp32
// v_mfma_f32_16x16x4_f32 (bit-identical to an fmaf chain) and streams both operands through LDS in
// chunks of 8 channels; X of the next chunk is prefetched into registers under the MFMAs.
//
// LDS image of one operand chunk: [strip 8][channel 8][position i, pitch 112] + 4 floats per strip:
//   fragment reads  (lane -> i = l & 15 unit stride, k = l >> 4 stride 112 == 16 mod 32)  conflict-free
//   column-loader writes (lane -> strip fastest, stride 900 == 4 mod 32)                  conflict-free
//   row-loader writes    (lane -> i fastest)                                               conflict-free
//
// Loads are unconditional from clamped (always valid) addresses and zeroed by a select afterwards:
// a branch around each load would serialise them (cdna_hip_programming.md section 5, trap (c)).
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int W_KC = 8;
constexpr int W_LDI = 112;
constexpr int W_GS = W_KC * W_LDI + 4;
constexpr int W_OP = kStripsPerBlock * W_GS;      // floats per operand chunk (7200)
constexpr int W_SLOTS = 2;                        // ceil(8 * 112 / 512) loader slots per thread

// FULL: the strip needs all 7x7 tiles (97..112 long) -> no per-tile guards in the hot loop
template <bool ROW, bool MASK, bool FULL>
__device__ __forceinline__ void weight_strip_body(float *lds, const float *__restrict__ X,
                                                  const float *__restrict__ Y, float *__restrict__ T,
                                                  int Cx, int H, int W) {
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    const int b = blockIdx.y, g0 = blockIdx.x * kStripsPerBlock;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int g = g0 + wv;
    const bool active = g < br.G;
    const int nt = FULL ? kMaxTiles : (L + kTile - 1) / kTile;
#define CCA_TILE_ON(t) (FULL || (t) < nt)

    const FBuf Xb = make_fbuf(X + (size_t)b * Cx * HW, (size_t)Cx * HW * sizeof(float));
    const FBuf Yb = make_fbuf(Y + (size_t)b * Cx * HW, (size_t)Cx * HW * sizeof(float));

    // loader slots: each thread owns up to two (position, strip) pairs of the 8-strip tile
    int goff[W_SLOTS], loff[W_SLOTS];
    bool lin[W_SLOTS], lok[W_SLOTS];                     // slot inside the tile / strip inside the image
#pragma unroll
    for (int n = 0; n < W_SLOTS; ++n) {
        const int r = tid + n * kBlock;
        int i, gg;
        if (ROW) { i = r % L; gg = r / L; }
        else     { gg = r & (kStripsPerBlock - 1); i = r >> 3; }
        lin[n] = r < kStripsPerBlock * L;
        lok[n] = lin[n] && (g0 + gg < br.G);
        goff[n] = lok[n] ? 4 * (i * br.fs_i + (g0 + gg) * br.fs_g) : 0;   // bytes; clamped: always valid
        loff[n] = gg * W_GS + i;
    }

    f32x4 acc[kMaxTiles][kMaxTiles];
#pragma unroll
    for (int rm = 0; rm < kMaxTiles; ++rm)
#pragma unroll
        for (int rn = 0; rn < kMaxTiles; ++rn) acc[rm][rn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = (lane >> 4) * W_LDI + (lane & 15);      // fragment offset inside a chunk: k = l>>4, i = l&15
    const float *xs = lds + wv * W_GS + fr;
    const float *ys = lds + W_OP + wv * W_GS + fr;

    // One operand chunk (8 channels x 2 slots) is staged through 16 registers.
    float rs[W_KC][W_SLOTS];
    auto fetch = [&](const FBuf &base, int c0) {
#pragma unroll
        for (int cc = 0; cc < W_KC; ++cc) {
            const bool cin = c0 + cc < Cx;                                  // scalar
            const int soff = (cin ? c0 + cc : Cx - 1) * HW * 4;             // scalar byte offset of the channel
#pragma unroll
            for (int n = 0; n < W_SLOTS; ++n) {
                const float t = fbuf_load(base, goff[n], soff);
                rs[cc][n] = (cin && lok[n]) ? t : 0.f;
            }
        }
    };
    auto stash = [&](float *dst) {
#pragma unroll
        for (int n = 0; n < W_SLOTS; ++n)
            if (lin[n]) {
#pragma unroll
                for (int cc = 0; cc < W_KC; ++cc) CCA_LDS_ST(&dst[loff[n] + cc * W_LDI], rs[cc][n]);
            }
    };

    fetch(Xb, 0);
    for (int c0 = 0; c0 < Cx; c0 += W_KC) {
        __syncthreads();                                   // previous chunk fully consumed
        stash(lds);
        fetch(Yb, c0);
        stash(lds + W_OP);
        __syncthreads();
        if (c0 + W_KC < Cx) fetch(Xb, c0 + W_KC);          // in flight during the MFMAs below
        if (active) {
#pragma unroll
            for (int ks = 0; ks < W_KC / 4; ++ks) {
                float a[kMaxTiles];
#pragma unroll
                for (int t = 0; t < kMaxTiles; ++t)
                    if (CCA_TILE_ON(t)) a[t] = CCA_LDS_LD(xs + ks * 4 * W_LDI + t * kTile);
#pragma unroll
                for (int rn = 0; rn < kMaxTiles; ++rn)
                    if (CCA_TILE_ON(rn)) {
                        const float bb = CCA_LDS_LD(ys + ks * 4 * W_LDI + rn * kTile);
#pragma unroll
                        for (int rm = 0; rm < kMaxTiles; ++rm)
                            if (CCA_TILE_ON(rm)) acc[rm][rn] = mfma_16x16x4(a[rm], bb, acc[rm][rn]);
                    }
            }
        }
    }

    if (!active) return;
    float *Tg = T + (size_t)b * HW * S + (size_t)g * br.as_g + br.a_off;
    const int jn = lane & 15, iq4 = 4 * (lane >> 4);
#pragma unroll
    for (int rm = 0; rm < kMaxTiles; ++rm)
#pragma unroll
        for (int rn = 0; rn < kMaxTiles; ++rn)
            if (CCA_TILE_ON(rm) && CCA_TILE_ON(rn)) {
                const int j = rn * kTile + jn;
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

template <bool ROW, bool MASK>
__global__ __launch_bounds__(kBlock) void weight_strip_kernel(const float *__restrict__ X,
                                                              const float *__restrict__ Y,
                                                              float *__restrict__ T, int Cx, int H, int W) {
    __shared__ float lds[2 * W_OP];
    CCA_LDS_REGISTER(lds);
    const int L = ROW ? W : H;
    if (L > (kMaxTiles - 1) * kTile) weight_strip_body<ROW, MASK, true>(lds, X, Y, T, Cx, H, W);
    else                             weight_strip_body<ROW, MASK, false>(lds, X, Y, T, Cx, H, W);
}

}  // namespace cca

// cca_map.hpp -- "map-type" strip kernels: attention-weighted sums along a strip.
//
// Per strip g (one column (b,:,w) or one row (b,h,:)) with its L x L attention block
// P_g[iq][j] = T[b, pixel(iq, g), a_off + j]:
//
//   TRANS = false   out[b, c, pos(iq, g)] (+)= sum_j  P_g[iq][j] * F[b, c, pos(j, g)]
//        ca_map_forward   T = A,  F = v   (/root/reference/cc_attention/functions.py:42,45-47)
//        ca_backward (dq) T = dE, F = k   (autograd of functions.py:38-39)
//   TRANS = true    out[b, c, pos(j, g)]  (+)= sum_iq P_g[iq][j] * F[b, c, pos(iq, g)]
//        ca_map_backward (dv) T = A,  F = dy
//        ca_backward (dk)     T = dE, F = q
//
// i.e. per strip the (channels x L) * (L x L) GEMM  F_g P_g^T  resp.  F_g P_g.
//
// Work decomposition (MI355X): one workgroup = 8 adjacent strips x a range of 16-channel chunks.  Each
// wavefront owns one strip and keeps that strip's whole attention block STATIONARY in registers as
// the B operands of v_mfma_f32_16x16x4_f32 (25 k-steps x 7 n-tiles = 175 VGPRs for L <= 100), so the
// attention tensor is read once per workgroup; feature chunks stream through LDS (coalesced along w
// for the column branch, along the row for the row branch) and every MFMA needs a single
// ds_read_b32.  Exact fp32: the MFMA is bit-identical to an fmaf chain.
//
// The two branches of one output are combined without atomics: the column launch stores its partial
// sum into ``out`` (EPI_STORE); the row launch then computes, per element and in the same thread that
// re-reads it,  out = alpha * (row_sum + out) + resid   (EPI_FINAL), alpha = *gamma or 1.
//
// LDS image of a feature chunk: [strip 8][channel 16][position k, pitch 102] + 4 floats per strip:
//   fragment reads (lane -> channel = l & 15 stride 102 (== 6 mod 32, even/2 odd), k = l >> 4 unit) conflict-free
//   column-loader writes (lane -> strip fastest, stride 1636 == 4 mod 32)                           conflict-free
//   row-loader writes    (lane -> k fastest)                                                         conflict-free
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int M_MC = 16;                          // channels per chunk = one MFMA M tile
constexpr int M_LDK = 102;                        // >= kMaxStrip, == 2 mod 4
constexpr int M_GS = M_MC * M_LDK + 4;            // 1636
constexpr int M_KS = kMaxStrip / 4;               // 25 k-steps
constexpr int M_SLOTS = 2;                        // ceil(8 * 100 / 512)
constexpr int EPI_STORE = 0, EPI_FINAL = 1;

// FULL: the strip needs all 25 k-steps and 7 n-tiles (97..100 long) -> no guards in the hot loop
template <bool ROW, bool TRANS, int EPI, bool FULL>
__device__ __forceinline__ void map_strip_body(float *lds, const float *__restrict__ T,
                                               const float *__restrict__ F, const float *__restrict__ resid,
                                               const float *__restrict__ gamma, float *out,
                                               int C, int H, int W, int chunks_per_block) {
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    const int b = blockIdx.z, g0 = blockIdx.x * kStripsPerBlock;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int g = g0 + wv;
    const bool active = g < br.G;
    const int nt = FULL ? kMaxTiles : (L + kTile - 1) / kTile;
    const int nks = FULL ? M_KS : (L + 3) / 4;
    const int nchunks = (C + M_MC - 1) / M_MC;
    const int ch_begin = blockIdx.y * chunks_per_block;
    const int ch_end = (ch_begin + chunks_per_block < nchunks) ? ch_begin + chunks_per_block : nchunks;
    const int ln = lane & 15, lk = lane >> 4;

    // K-padding of the feature image must be a true zero (0 * garbage could be NaN): clear [L, pitch)
    {
        const int padw = M_LDK - L;
        for (int idx = tid; idx < kStripsPerBlock * M_MC * padw; idx += kBlock) {
            const int row = idx / padw, col = L + idx - row * padw;
            CCA_LDS_ST(&lds[(row / M_MC) * M_GS + (row % M_MC) * M_LDK + col], 0.f);
        }
    }

    // stationary attention block of this wavefront's strip, as MFMA B fragments:
    //   B[k][n] with k = contraction index, n = output position.  Unconditional loads from clamped
    //   addresses, zeroed by a select (no branch per load).
    float bf[M_KS][kMaxTiles];
    {
        const int gc = active ? g : br.G - 1;
        const FBuf Tb = make_fbuf(T + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
        const int tsoff = 4 * (gc * br.as_g + br.a_off);                 // scalar
#pragma unroll
        for (int ks = 0; ks < M_KS; ++ks)
#pragma unroll
            for (int t = 0; t < kMaxTiles; ++t) {
                const int kidx = ks * 4 + lk, nidx = t * kTile + ln;
                const int iq = TRANS ? kidx : nidx, j = TRANS ? nidx : kidx;
                const bool ok = iq < L && j < L;
                const float v = fbuf_load(Tb, ok ? 4 * (iq * br.as_q + j) : 0, tsoff);
                bf[ks][t] = ok ? v : 0.f;
            }
    }

    // loader slots (position k along the strip, strip gg) as in the weight kernel
    int goff[M_SLOTS], loff[M_SLOTS];
    bool lin[M_SLOTS], lok[M_SLOTS];
#pragma unroll
    for (int n = 0; n < M_SLOTS; ++n) {
        const int r = tid + n * kBlock;
        int k, gg;
        if (ROW) { k = r % L; gg = r / L; }
        else     { gg = r & (kStripsPerBlock - 1); k = r >> 3; }
        lin[n] = r < kStripsPerBlock * L;
        lok[n] = lin[n] && (g0 + gg < br.G);
        goff[n] = lok[n] ? 4 * (k * br.fs_i + (g0 + gg) * br.fs_g) : 0;     // bytes; clamped: always valid
        loff[n] = gg * M_GS + k;
    }

    const FBuf Fb = make_fbuf(F + (size_t)b * C * HW, (size_t)C * HW * sizeof(float));
    float *Ob = out + (size_t)b * C * HW;
    const float *Rb = resid ? resid + (size_t)b * C * HW : nullptr;
    const float alpha = (EPI == EPI_FINAL && gamma) ? gamma[0] : 1.f;
    const float *as = lds + wv * M_GS + ln * M_LDK + lk;   // A fragment: channel = l & 15, k = l >> 4
    const int opos = g * br.fs_g;

    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int c0 = ch * M_MC;
        __syncthreads();                                   // previous chunk consumed (and pad cleared)
        // stage the 16-channel chunk in two halves of 8 channels (16 staging registers)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float rf[M_MC / 2][M_SLOTS];
#pragma unroll
            for (int cc = 0; cc < M_MC / 2; ++cc) {
                const int c = c0 + half * (M_MC / 2) + cc;
                const bool cin = c < C;                                       // scalar
                const int soff = (cin ? c : C - 1) * HW * 4;                  // scalar byte offset
#pragma unroll
                for (int n = 0; n < M_SLOTS; ++n) {
                    const float t = fbuf_load(Fb, goff[n], soff);
                    rf[cc][n] = (cin && lok[n]) ? t : 0.f;
                }
            }
#pragma unroll
            for (int n = 0; n < M_SLOTS; ++n)
                if (lin[n]) {
#pragma unroll
                    for (int cc = 0; cc < M_MC / 2; ++cc)
                        CCA_LDS_ST(&lds[loff[n] + (half * (M_MC / 2) + cc) * M_LDK], rf[cc][n]);
                }
        }
        __syncthreads();
        if (!active) continue;

        f32x4 acc[kMaxTiles];
#pragma unroll
        for (int t = 0; t < kMaxTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < M_KS; ++ks)
            if (FULL || ks < nks) {
                const float a = CCA_LDS_LD(as + ks * 4);
#pragma unroll
                for (int t = 0; t < kMaxTiles; ++t)
                    if (FULL || t < nt) acc[t] = mfma_16x16x4(a, bf[ks][t], acc[t]);
            }

        // D[m = channel 4*(l>>4)+r][n = position t*16 + (l&15)]
#pragma unroll
        for (int t = 0; t < kMaxTiles; ++t)
            if (FULL || t < nt) {
                const int pos = t * kTile + ln;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + 4 * lk + r;
                    if (pos < L && c < C) {
                        const int o = c * HW + pos * br.fs_i + opos;
                        float val = acc[t][r];
                        if (EPI == EPI_FINAL) {
                            val = alpha * (val + Ob[o]);
                            if (Rb) val += Rb[o];
                        }
                        Ob[o] = val;
                    }
                }
            }
    }
}

template <bool ROW, bool TRANS, int EPI>
__global__ __launch_bounds__(kBlock) void map_strip_kernel(const float *__restrict__ T,
                                                           const float *__restrict__ F,
                                                           const float *__restrict__ resid,
                                                           const float *__restrict__ gamma,
                                                           float *out, int C, int H, int W,
                                                           int chunks_per_block) {
    __shared__ float lds[kStripsPerBlock * M_GS];
    CCA_LDS_REGISTER(lds);
    const int L = ROW ? W : H;
    if (L > (M_KS - 1) * 4)
        map_strip_body<ROW, TRANS, EPI, true>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block);
    else
        map_strip_body<ROW, TRANS, EPI, false>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block);
}

}  // namespace cca

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
// Work decomposition (MI355X): one workgroup = NS adjacent strips x a range of 16-channel chunks, one
// wavefront per strip.  Each wavefront keeps its strip's whole attention block STATIONARY in registers as
// the B operands of v_mfma_f32_16x16x4_f32 (25 k-steps x 7 n-tiles = 175 VGPRs for L <= 100), so the
// attention tensor is read once per workgroup.  Feature chunks stream through a DOUBLE-BUFFERED LDS image
// filled by LDS-DMA (chunk n+1 in flight while chunk n is multiplied); every MFMA needs one ds_read_b32.
// Results go back through the SAME LDS image (each wavefront overwrites only its own strip's slots) and
// leave the workgroup as coalesced tile stores mirroring the DMA pattern.  Exact fp32 throughout.
//
// The two branches of one output are combined without atomics: the column launch stores its partial
// sum into ``out`` (EPI_STORE); the row launch then computes, per element and in the same thread that
// re-reads it,  out = alpha * (row_sum + out) + resid   (EPI_FINAL), alpha = *gamma or 1.
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int M_MC = 16;                          // channels per chunk = one MFMA M tile
constexpr int M_KS = kMaxStrip / 4;               // 25 k-steps
constexpr int EPI_STORE = 0, EPI_FINAL = 1;
// channel pitch of the LDS image: pieces * 64 + 17 (odd: see strip geometry in cca_common.hpp; >= NS*L + 3)
__host__ __device__ constexpr int m_cp(int ns) { return strip_pieces_c(ns) * 64 + 17; }
__host__ __device__ constexpr int m_lds_floats(int ns) { return 2 * M_MC * m_cp(ns); }    // two buffers

// FULL: the strip needs all 25 k-steps and 7 n-tiles (97..100 long) -> no guards in the hot loop
template <int NS, bool ROW, bool TRANS, int EPI, bool FULL>
__device__ __forceinline__ void map_strip_body(float *lds, const float *__restrict__ T,
                                               const float *__restrict__ F, const float *__restrict__ resid,
                                               const float *__restrict__ gamma, float *out,
                                               int C, int H, int W, int chunks_per_block) {
    constexpr int CP = m_cp(NS), BUF = M_MC * CP, kBlock = kWave * NS;
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    const int b = blockIdx.z, g0 = blockIdx.x * NS;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int g = g0 + wv;
    const bool active = g < br.G;
    const int nt = FULL ? kMaxTiles : (L + kTile - 1) / kTile;
    const int nks = FULL ? M_KS : (L + 3) / 4;
    const int npieces = FULL ? strip_pieces_c(NS) : (NS * L + 63) / 64;     // FULL: L in 97..100
    const int gvalid = (br.G - g0 < NS) ? br.G - g0 : NS;
    const int nchunks = (C + M_MC - 1) / M_MC;
    const int ch_begin = blockIdx.y * chunks_per_block;
    const int ch_end = (ch_begin + chunks_per_block < nchunks) ? ch_begin + chunks_per_block : nchunks;
    const int ln = lane & 15, lk = lane >> 4;

    const FBuf Fb = make_fbuf(F + (size_t)b * C * HW, (size_t)C * HW * sizeof(float));
    const FBuf Ob = make_fbuf(out + (size_t)b * C * HW, (size_t)C * HW * sizeof(float));
    const FBuf Rb = make_fbuf(resid ? resid + (size_t)b * C * HW : out, (size_t)C * HW * sizeof(float));
    const bool has_resid = resid != nullptr;
    const float alpha = (EPI == EPI_FINAL && gamma) ? gamma[0] : 1.f;

    // channels of a chunk are dealt round-robin to the NS waves, for the DMA and for the tile stores
    auto issue = [&](int ch, int buf) {
#pragma unroll
        for (int pr = 0; pr < M_MC / NS; ++pr) {
            const int cc = wv + pr * NS, c = ch * M_MC + cc;
            float *dst = lds + buf * BUF + cc * CP;
            if (c < C) strip_dma_channel<NS, ROW, FULL>(Fb, dst, c * HW * 4, lane, npieces, L, W, g0, gvalid);
            else for (int m = 0; m < npieces; ++m) CCA_LDS_ST(&dst[m * 64 + lane], 0.f);
        }
    };

    // the few slots between the last DMA piece and the pitch are read as K padding: make them true zeros
    for (int idx = tid; idx < 2 * M_MC * (CP - npieces * 64); idx += kBlock) {
        const int rowi = idx / (CP - npieces * 64), col = npieces * 64 + idx % (CP - npieces * 64);
        CCA_LDS_ST(&lds[rowi * CP + col], 0.f);
    }
    if (ch_begin < ch_end) issue(ch_begin, 0);

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

    // A fragment: channel = l & 15 (pitch CP), contraction position k = 4 ks + (l >> 4)
    //   column image: k * NS + (strip ^ swz(k)), swz constant inside a k-step;  row image: strip * L + k
    const float *as = lds + ln * CP + (ROW ? wv * L + lk : lk * NS);

    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int buf = (ch - ch_begin) & 1;
        __syncthreads();                 // chunk ch landed (vmcnt drained); the other buffer's tile stores are done
        if (ch + 1 < ch_end) issue(ch + 1, buf ^ 1);
        float *img = lds + buf * BUF;
        if (active) {
            f32x4 acc[kMaxTiles];
#pragma unroll
            for (int t = 0; t < kMaxTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *ab = as + buf * BUF;
#pragma unroll
            for (int ks = 0; ks < M_KS; ++ks)
                if (FULL || ks < nks) {
                    const int koff = ROW ? ks * 4 : ks * 4 * NS + (wv ^ col_swizzle<NS>(ks * 4));
                    const float a = CCA_LDS_LD(ab + koff);
#pragma unroll
                    for (int t = 0; t < kMaxTiles; ++t)
                        if (FULL || t < nt) acc[t] = mfma_16x16x4(a, bf[ks][t], acc[t]);
                }
            // D[m = channel 4*(l>>4)+r][n = position t*16 + (l&15)] -> this strip's slots of the image
#pragma unroll
            for (int t = 0; t < kMaxTiles; ++t)
                if (FULL || t < nt) {
                    const int pos = t * kTile + ln;
                    if (pos < L) {
                        float *d = img + (4 * lk) * CP + strip_lds_index<NS, ROW>(pos, wv, L);
#pragma unroll
                        for (int r = 0; r < 4; ++r) CCA_LDS_ST(&d[r * CP], acc[t][r]);
                    }
                }
        }
        __syncthreads();                 // the whole 16-channel output tile is in the image
#pragma unroll
        for (int pr = 0; pr < M_MC / NS; ++pr) {
            const int cc = wv + pr * NS, c = ch * M_MC + cc;
            if (c < C) {
                const int soff = c * HW * 4;
                const float *src = img + cc * CP;
#pragma unroll
                for (int m = 0; m < strip_pieces_c(NS); ++m)
                    if (FULL || m < npieces) {
                        const int off = strip_elem_offset<NS, ROW>(m, lane, L, W, g0, gvalid);
                        const int voff = off < 0 ? kOobOffset : off;    // out-of-range lanes: load 0, store dropped
                        float val = CCA_LDS_LD(&src[m * 64 + lane]);
                        if (EPI == EPI_FINAL) {
                            val = alpha * (val + fbuf_load(Ob, voff, soff));
                            if (has_resid) val += fbuf_load(Rb, voff, soff);
                        }
                        fbuf_store(Ob, val, voff, soff);
                    }
            }
        }
    }
}

template <int NS, bool ROW, bool TRANS, int EPI>
__global__ __launch_bounds__(kWave * NS, 2) void map_strip_kernel(const float *__restrict__ T,
                                                                   const float *__restrict__ F,
                                                                   const float *__restrict__ resid,
                                                                   const float *__restrict__ gamma,
                                                                   float *out, int C, int H, int W,
                                                                   int chunks_per_block) {
    __shared__ float lds[m_lds_floats(NS)];
    CCA_LDS_REGISTER(lds);
    const int L = ROW ? W : H;
    if (L > (M_KS - 1) * 4)
        map_strip_body<NS, ROW, TRANS, EPI, true>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block);
    else
        map_strip_body<NS, ROW, TRANS, EPI, false>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block);
}

}  // namespace cca

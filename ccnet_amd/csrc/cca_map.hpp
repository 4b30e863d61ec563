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
// The two branches of one output are combined without atomics, alpha = *gamma or 1:
//   column launch (EPI_COL / EPI_COL_RESID)   out = alpha * col_sum [+ resid]
//   row launch    (EPI_ROW)   out = alpha * row_sum + out        (re-read by the thread that rewrites it)
// The tile of ``resid`` / ``out`` a workgroup needs at the end of a chunk is fetched by LDS-DMA into a third
// LDS image BEFORE the chunk's MFMAs, so its latency hides under them and costs no registers
// (3 images x 16 channels x 852 floats = 163,584 B of the 163,840 B LDS at NS = 8).
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int M_MC = 16;                          // channels per chunk = one MFMA M tile
constexpr int M_KS = kMaxStrip / 4;               // 25 k-steps
constexpr int M_BKS = 3;                          // bf16 path: 3 k-steps of 32 cover k < 96
constexpr int EPI_COL = 0, EPI_ROW = 1, EPI_COL_RESID = 2;    // column launch without / row launch / column launch with residual
// channel pitch of the LDS image: pieces * 64 + 20 (a multiple of 4 so that 16-byte LDS reads of the tile-store
// phase stay aligned; >= NS*L + 3 for the K-padding reads)
__host__ __device__ constexpr int m_cp(int ns) { return strip_pieces_c(ns) * 64 + 20; }
// prologue images of the attention blocks: 4 strips x 100 rows x pitch 100 = 25 16-byte DMA lanes per row, rows back
// to back, so one LDS-DMA instruction moves 2.56 rows (pitch 102 made the fragment reads conflict-free but needed
// 4-byte DMA lanes: 5x the instructions, ~4.5 us of every launch)
constexpr int M_PP = kMaxStrip, M_SIMG = kMaxStrip * M_PP, M_SPP = 4;
// LDS: max(2 feature buffers + addend image, prologue images) = 40,896 floats = 163,584 B of 163,840 B
__host__ __device__ constexpr int m_lds_floats(int ns) {
    return 3 * M_MC * m_cp(ns) > M_SPP * M_SIMG ? 3 * M_MC * m_cp(ns) : M_SPP * M_SIMG;
}

// FULL: compile-time shape -- all 25 k-steps and 7 n-tiles, every DMA piece and tile store issued (strips at least
//       kFullMinStrip long, see cca_common.hpp) -> counted-vmcnt pipeline, no shape guards in the hot loop
// BF: split-bf16 x3 on v_mfma_f32_16x16x32_bf16 for k < 96 (+ one exact f32 k-step for k = 96..99);
//     only instantiated together with FULL and only used for strips 96..100 long
// EXACT (with FULL): strips 97..100 long -- every DMA piece and store piece of a full tile holds lanes of the tile, every
//       k-step but the last lies inside the strip: plain lane masks, compile-time store count, unguarded k-steps (the
//       headline geometry pays nothing for the generality of the FULL bodies)
template <int NS, bool ROW, bool TRANS, int EPI, bool FULL, bool BF, bool EXACT = false>
__device__ __forceinline__ void map_strip_body(float *lds, const float *__restrict__ T,
                                               const float *__restrict__ F, const float *__restrict__ resid,
                                               const float *__restrict__ gamma, float *out,
                                               int C, int H, int W, int chunks_per_block, int tiles, int nsplit,
                                               int wg_linear, int wg_count, long fbs, long rbs, long obs) {
    constexpr int CP = m_cp(NS), BUF = M_MC * CP, kBlock = kWave * NS, PIECES = strip_pieces_c(NS);
    constexpr int CPW = M_MC / NS;                // channels per wave in the DMA / tile-store phases
    constexpr int PIECES4 = strip_pieces4_c(NS);
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    // logical id -> (image, channel split, tile), tile fastest: neighbouring tiles share an XCD's L2
    const int id = xcd_logical_id(wg_linear, wg_count);
    const int b = id / (tiles * nsplit), rem = id - b * (tiles * nsplit);
    const int split = rem / tiles, g0 = (rem - split * tiles) * NS;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), lane_ = lane, wv = uniform(tid >> 6);
    const int g = g0 + wv;
    const bool active = g < br.G;
    const int nt = FULL ? kMaxTiles : (L + kTile - 1) / kTile;
    const int nks = FULL ? M_KS : (L + 3) / 4;
    const int npieces = FULL ? strip_pieces_c(NS) : (NS * L + 63) / 64;     // FULL: L in 97..100
    const int gvalid = (br.G - g0 < NS) ? br.G - g0 : NS;
    const int nchunks = (C + M_MC - 1) / M_MC;
    const int ch_begin = split * chunks_per_block;
    const int ch_end = (ch_begin + chunks_per_block < nchunks) ? ch_begin + chunks_per_block : nchunks;
    const int ln = lane & 15, lk = lane >> 4;

    StripLanes4<NS, ROW> sl4;                     // 16-byte pieces: LDS-DMA
    sl4.init(lane, L, W, g0, gvalid);
    const int npieces4 = FULL ? strip_pieces4_c(NS) : (NS * L + 255) / 256;

    const FBuf Fb = make_fbuf(F + (size_t)b * fbs, (size_t)C * HW * sizeof(float));        // fbs/rbs/obs: batch strides (elements)
    const FBuf Ob = make_fbuf(out + (size_t)b * obs, (size_t)C * HW * sizeof(float));
    // the tensor added in the epilogue: the residual (column launch, optional) or the column partial (row launch)
    constexpr bool has_add = EPI != EPI_COL;
    const FBuf Ab = make_fbuf(EPI == EPI_COL_RESID ? resid + (size_t)b * rbs : out + (size_t)b * obs, (size_t)C * HW * sizeof(float));
    const float alpha = gamma ? gamma[0] : 1.f;

    // channels of a chunk are dealt round-robin to the NS waves, for the DMA and for the tile stores
    auto issue = [&](int ch, int buf) {
#pragma unroll
        for (int pr = 0; pr < M_MC / NS; ++pr) {
            const int cc = wv + pr * NS, c = ch * M_MC + cc;
            float *dst = lds + buf * BUF + cc * CP;
            strip_dma_channel<NS, ROW, FULL, EXACT>(Fb, dst, (c < C ? c : C - 1) * HW * 4, npieces4, W, sl4);
        }
    };

    // ---- prologue: this wavefront's stationary attention block, as MFMA B fragments B[k][n] (k = contraction
    // index, n = output position).  Reading the fragments straight from global memory is a 16-segment gather
    // per instruction; instead the rows of the block (L contiguous floats each) are brought into LDS by
    // coalesced LDS-DMA, four strips at a time (4 images of 100 rows x pitch 102 fill the LDS), and the fragments are read
    // from there -- the same image serves both orientations.
    float bf[BF ? 1 : M_KS][kMaxTiles];          // f32 path: 25 k-steps of 4
    u32x4 bh[BF ? M_BKS : 1][kMaxTiles], bl[BF ? M_BKS : 1][kMaxTiles];   // bf16 path: 3 k-steps of 32, hi / lo
    float btail[kMaxTiles];                      // bf16 path: exact f32 fragments of k = 96..99
    {
        constexpr int PP = M_PP, SIMG = M_SIMG, SPP = M_SPP;
        const FBuf Tb = make_fbuf(T + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
#pragma unroll
        for (int ph = 0; ph < NS / SPP; ++ph) {
            if (ph) __syncthreads();                      // previous phase's fragments are in registers
            // rows are dealt round-robin to the NS waves
            for (int s = 0; s < SPP; ++s) {
                const int gs = g0 + ph * SPP + s;
                if (gs >= br.G) continue;
                float *img = lds + s * SIMG;
                // lane idx of the image = (row idx / 25, 16-byte chunk idx % 25); the last chunk of a row reads up to 3
                // slots past the strip's L (the pixel's other branch / the next pixel): never used as operands
                const int soff = 4 * (gs * br.as_g + br.a_off), nit = (L * (PP / 4) + 63) / 64;
                for (int it = wv; it < nit; it += NS) {
                    const int idx = 64 * it + lane, iq = idx / (PP / 4), chk = idx - iq * (PP / 4);
                    if (iq < L && 4 * chk < L) fbuf_load_to_lds_x4(Tb, img + 256 * it, 4 * iq * br.as_q + 16 * chk, soff);
                }
            }
            __syncthreads();                              // images landed (vmcnt drained by the barrier)
            if (wv / SPP == ph) {
                const float *img = lds + (wv % SPP) * SIMG;
                if constexpr (BF) {
#pragma unroll
                    for (int ks = 0; ks < M_BKS; ++ks)
#pragma unroll
                        for (int t = 0; t < kMaxTiles; ++t) {
                            const int nidx = t * kTile + ln;
                            float x[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int kidx = ks * 32 + lk * 8 + e;                     // < 96 <= L
                                const bool ok = active && nidx < L;
                                const float v = CCA_LDS_LD(&img[ok ? (TRANS ? kidx * PP + nidx : nidx * PP + kidx) : 0]);
                                x[e] = ok ? v : 0.f;
                            }
                            const BfSplit sp = bf16_split8(x);
                            bh[ks][t] = sp.hi;
                            bl[ks][t] = sp.lo;
                        }
#pragma unroll
                    for (int t = 0; t < kMaxTiles; ++t) {                                   // exact f32 tail k = 96..99
                        const int kidx = 96 + lk, nidx = t * kTile + ln;
                        const bool ok = active && kidx < L && nidx < L;
                        const float v = CCA_LDS_LD(&img[ok ? (TRANS ? kidx * PP + nidx : nidx * PP + kidx) : 0]);
                        btail[t] = ok ? v : 0.f;
                    }
                } else {
#pragma unroll
                    for (int ks = 0; ks < M_KS; ++ks)
#pragma unroll
                        for (int t = 0; t < kMaxTiles; ++t) {
                            const int kidx = ks * 4 + lk, nidx = t * kTile + ln;
                            const int iq = TRANS ? kidx : nidx, j = TRANS ? nidx : kidx;
                            const bool ok = active && iq < L && j < L;
                            const float v = CCA_LDS_LD(&img[ok ? iq * PP + j : 0]);
                            bf[ks][t] = ok ? v : 0.f;
                        }
                }
            }
        }
        __syncthreads();                                  // images consumed: the LDS becomes the chunk buffers
    }

    // masked DMA lanes leave their LDS slots untouched, and slots just past a strip are read as K padding (and then
    // discarded): start from an all-zero LDS so that what is read there is at least defined
    for (int idx = 4 * tid; idx < 3 * BUF; idx += 4 * kBlock) lds_store_x4(&lds[idx], f32x4{0.f, 0.f, 0.f, 0.f});
    __syncthreads();
    // Chunk order.  The column launch of a pair walks its channel chunks upwards, the row launch (which runs right
    // after it) DOWNWARDS: the column launch's last chunks -- features, and the partial sums it has just written --
    // are what the 256 MB Infinity Cache still holds, so the row launch starts on cache-resident data instead of
    // evicting it before it gets there.  CCA_ROW_ASCENDING restores the ascending order (A/B builds).
#ifdef CCA_ROW_ASCENDING
    constexpr bool kDescending = false;
#else
    constexpr bool kDescending = ROW;
#endif
    const int nmine = ch_end - ch_begin;
    const int ch_first = kDescending ? ch_end - 1 : ch_begin, ch_step = kDescending ? -1 : 1;
    if (nmine > 0) issue(ch_first, 0);
    __syncthreads();                     // first chunk landed (the compiler drains vmcnt before the barrier)

    // A fragment: channel = l & 15 (pitch CP), contraction position k = 4 ks + (l >> 4)
    //   column image: k * NS + (strip ^ swz(k)), swz constant inside a k-step;  row image: strip * L + k
    const float *as = lds + ln * CP + (ROW ? wv * L + lk : lk * NS);

    // DMA work of one chunk iteration, cut into single pieces so that it can be interleaved with the MFMAs
    // (a piece costs ~60-100 issue cycles; 52 of them in front of the MFMAs would add ~40 % to a chunk):
    //   q in [0, QA)        addend tile of chunk ch -> third image        (channel pr of this wave, piece m)
    //   q in [QA, QA + QF)  feature chunk ch+1 -> the other buffer
    // The addend pieces come FIRST: the barrier in front of the tile stores then only has to wait for them
    // (counted vmcnt) and the feature pieces of the next chunk keep flying through the store phase.
    constexpr int QA = (EPI != EPI_COL) ? CPW * PIECES4 : 0, QF = CPW * PIECES4, QT = QA + QF;
#ifndef CCA_DMA_SPAN
#define CCA_DMA_SPAN 100         // per cent of the MFMA phase over which a chunk's DMA pieces are issued
#endif
    // piece schedule: slot s of n issues pieces [qlo(s, n), qlo(s + 1, n))
    auto qlo = [](int s, int n) { const int v = s * QT * 100 / (n * CCA_DMA_SPAN); return v < QT ? v : QT; };
    // Branch-free in the FULL path: channels beyond C are clamped (they are output rows that are never
    // stored), and when there is no next chunk the current one is simply fetched again into the idle buffer.
    auto dma_piece = [&](int q, int ch, int chn, int buf) {
        const bool feat = q >= QA;
        const int rem = feat ? q - QA : q;
        const int pr = rem / PIECES4, m = rem % PIECES4;
        if (!(FULL || m < npieces4)) return;
        const int cc = wv + pr * NS;
        const int c = (feat ? chn : ch) * M_MC + cc;
        float *dst = lds + (feat ? (buf ^ 1) : 2) * BUF + cc * CP + m * 256;
        if constexpr (FULL && !EXACT) {
            // every piece is issued (the counted waits rely on it): lanes inside the image but beyond the tile fetch from an
            // out-of-range offset (zeros land), so no piece ever loses all its lanes
            if (sl4.in_image(m, lane_))
                fbuf_load_to_lds_x4(feat ? Fb : Ab, dst, sl4.full_offset(m), (c < C ? c : C - 1) * HW * 4 + sl4.piece_soff(m, W));
        } else if (sl4.valid(m))
            fbuf_load_to_lds_x4(feat ? Fb : Ab, dst, sl4.vb, (c < C ? c : C - 1) * HW * 4 + sl4.piece_soff(m, W));
    };
    // counted waits need every wave to issue exactly QA + QF pieces per chunk (FULL bodies do, whatever the strip length)
    const bool fast = FULL && gvalid == NS;       // full tile of strips >= kFullMinStrip long: 16-byte granule stores
    const bool kfull = EXACT || L >= 4 * (M_KS - 1);       // every k-step but the last lies inside the strip
    const bool counted = fast && (C % M_MC == 0);
    // 16-byte tile stores a wave issues per chunk on the fast path: per channel, the column launch stores one piece per
    // four bands of NS rows, the row launch one piece per 64 granules of its NS rows (+ one instruction for a ragged tail)
    const int nstore = CPW * (ROW ? ((L >> 2) * NS + 63) / 64 + ((L & 3) ? 1 : 0) : ((L + NS - 1) / NS + 3) / 4);
    // EXACT: at least this many, known at compile time (97..100: 24 full granules per row; 13 bands of rows)
    constexpr int NSTORE_MIN = ROW ? CPW * ((NS * 24 + 63) / 64) : CPW * PIECES4;

    for (int it = 0; it < nmine; ++it) {
        const int ch = ch_first + it * ch_step;
        const int buf = it & 1;
        const int chn = (it + 1 < nmine) ? ch + ch_step : ch;
        float *img = lds + buf * BUF;
        f32x4 acc[kMaxTiles];
        if (active) {
#pragma unroll
            for (int t = 0; t < kMaxTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *ab = as + buf * BUF;
            if constexpr (BF) {
                // per-lane strip slot of the column image for the two halves of an 8-element k group
                const int sw0 = wv ^ col_swizzle<NS>(8 * lk), sw1 = wv ^ col_swizzle<NS>(8 * lk + 4);
                const float *abf = lds + buf * BUF + ln * CP + (ROW ? wv * L + 8 * lk : 8 * lk * NS);
                constexpr int SLOTS = M_BKS * kMaxTiles + kMaxTiles;        // 28 interleave slots
#pragma unroll
                for (int ks = 0; ks < M_BKS; ++ks) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = CCA_LDS_LD(abf + (ROW ? ks * 32 + e : (ks * 32 + e) * NS + (e < 4 ? sw0 : sw1)));
                    const BfSplit sp = bf16_split8(x);
#pragma unroll
                    for (int t = 0; t < kMaxTiles; ++t) {
                        const int sidx = ks * kMaxTiles + t;
#pragma unroll
                        for (int q = qlo(sidx, SLOTS); q < qlo(sidx + 1, SLOTS); ++q) dma_piece(q, ch, chn, buf);
                        acc[t] = mfma_bf16_16x16x32(sp.hi, bh[ks][t], acc[t]);
                        acc[t] = mfma_bf16_16x16x32(sp.hi, bl[ks][t], acc[t]);
                        acc[t] = mfma_bf16_16x16x32(sp.lo, bh[ks][t], acc[t]);
                    }
                }
                {
                    const int koff = ROW ? 96 : 96 * NS + (wv ^ col_swizzle<NS>(96));
                    const float araw = CCA_LDS_LD(ab + koff);
                    // K padding (positions L .. 99): in the row image those slots hold the NEXT row's first values; their
                    // B fragments are zero, but 0 * inf would still poison this row, so the A operand is zeroed too
                    const float a = (96 + lk < L) ? araw : 0.f;
#pragma unroll
                    for (int t = 0; t < kMaxTiles; ++t) {
                        const int sidx = M_BKS * kMaxTiles + t;
#pragma unroll
                        for (int q = qlo(sidx, SLOTS); q < qlo(sidx + 1, SLOTS); ++q) dma_piece(q, ch, chn, buf);
                        acc[t] = mfma_16x16x4(a, btail[t], acc[t]);
                    }
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < M_KS; ++ks) {
                // this k-step's share of the DMA pieces: next feature chunk + this chunk's addend tile
#pragma unroll
                for (int q = qlo(ks, M_KS); q < qlo(ks + 1, M_KS); ++q) dma_piece(q, ch, chn, buf);
                if (FULL || ks < nks) {
                    const int koff = ROW ? ks * 4 : ks * 4 * NS + (wv ^ col_swizzle<NS>(ks * 4));
                    const float araw = CCA_LDS_LD(ab + koff);
                    // K padding: zero the A operand as well as the B fragment (see the bf16 path); only the last
                    // k-step of a full strip can hold padding
                    const float a = (FULL && kfull && ks < M_KS - 1) ? araw : ((ks * 4 + lk < L) ? araw : 0.f);
#pragma unroll
                    for (int t = 0; t < kMaxTiles; ++t)
                        if (FULL || t < nt) acc[t] = mfma_16x16x4(a, bf[ks][t], acc[t]);
                }
            }
            }
        } else {
#pragma unroll
            for (int q = 0; q < QT; ++q) dma_piece(q, ch, chn, buf);   // strips outside the image still own channels
        }
        mfma_f32_result_fence();             // the last exact-f32 MFMA may sit at the end of a conditional block
        // D[m = channel 4*(l>>4)+r][n = position t*16 + (l&15)] -> LDS -> tile stores
        if (fast && !ROW) {
            // Column launch, full tile: the results go to LDS in the band-permuted layout of the partial sums
            // (blocked_offset), alpha applied, so that the store phase is a linear copy of 16-byte granules.
            // They land in the idle third image; with a residual that image holds the residual tile, which is
            // added here, and the results overwrite this chunk's feature image instead -- after one more
            // barrier, because other strips' slots of it are still being read as MFMA operands.
            float *dimg = (EPI == EPI_COL_RESID) ? img : lds + 2 * BUF;
            if constexpr (EPI == EPI_COL_RESID) {
                if (counted) barrier_dma_keep<QF>();      // residual tile landed, feature image dead
                else         __syncthreads();
            }
            const int ln = recompute_here(lane_) & 15;    // LDS addresses below are rebuilt per chunk
#pragma unroll
            for (int t = 0; t < kMaxTiles; ++t) {
                const int pos = t * kTile + ln;
                if (pos < L) {
                    const int k = pos / NS, hh = pos - k * NS;
                    const int nr = (L - k * NS < NS) ? L - k * NS : NS;
                    float *d = dimg + (4 * lk) * CP + k * NS * NS + (wv >> 2) * 4 * nr + hh * 4 + (wv & 3);
                    const float *xr = lds + 2 * BUF + (4 * lk) * CP + strip_lds_index<NS, ROW>(pos, wv, L);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float val = alpha * acc[t][r];
                        if constexpr (EPI == EPI_COL_RESID) val += CCA_LDS_LD(&xr[r * CP]);
                        CCA_LDS_ST(&d[r * CP], val);
                    }
                }
            }
            if (EPI == EPI_COL_RESID || !counted) { if (counted) barrier_lds_only(); else __syncthreads(); }
            else                                  barrier_dma_keep<QF>();
        } else {
            if (active) {
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
            // output tile complete and addend tile landed; the QF feature pieces of chunk ch+1 may still be in flight
            if (counted) barrier_dma_keep<QF>();
            else         __syncthreads();
        }
        if (fast) {
            const int lane = recompute_here(lane_);       // store-phase addresses are rebuilt per chunk, not kept live
#pragma unroll
            for (int pr = 0; pr < M_MC / NS; ++pr) {
                const int cc = wv + pr * NS, c = ch * M_MC + cc;
                if (c < C) {
                    const int soff = c * HW * 4;
                    if constexpr (!ROW) {
                        // column launch: linear copy of the band-permuted image (alpha / residual already applied);
                        // granule e0 = 256 m + 4 lane lies in band k = e0 / NS^2 at r = e0 % NS^2, valid while r < NS * nr
                        const float *src = ((EPI == EPI_COL_RESID) ? img : lds + 2 * BUF) + cc * CP;
#pragma unroll
                        for (int m = 0; m < PIECES4; ++m) {
                            const int e0 = m * 256 + 4 * lane;
                            const int k = e0 / (NS * NS), r = e0 - k * NS * NS;
                            const int rem = L - k * NS, nr = rem < NS ? rem : NS;      // rem <= 0 beyond the last band
                            if (r < NS * nr) {
                                const f32x4 val = lds_load_x4(&src[e0]);
                                fbuf_store_x4(Ob, val, 4 * (k * NS * W + g0 * nr + r), soff);
                            }
                        }
                    } else {
                        // row launch: granule p = 64 m + lane is (row p % NS, columns 4 (p / NS) .. + 3); the column
                        // partial sums of the same granule sit at 4 p of the addend image (band-permuted layout)
                        const float *res = img + cc * CP;
                        const float *add = lds + 2 * BUF + cc * CP;
                        const int ngr = L >> 2;                    // full granules per row
#pragma unroll
                        for (int m = 0; m < PIECES4; ++m) {
                            const int pidx = m * 64 + lane;
                            const int s4 = pidx % NS, w4 = pidx / NS;
                            if (w4 < ngr) {
                                f32x4 val;
#pragma unroll
                                for (int i = 0; i < 4; ++i) val[i] = alpha * CCA_LDS_LD(&res[s4 * L + 4 * w4 + i]);
                                if (has_add) val += lds_load_x4(&add[4 * pidx]);
                                fbuf_store_x4(Ob, val, 4 * ((g0 + s4) * W + 4 * w4), soff);
                            }
                        }
                        const int nwt = L & 3;                     // columns of the last, partial granule
                        if (nwt) {
                            const int mul = nwt == 1 ? 32 : nwt == 2 ? 16 : 11;     // lane / nwt for lane < 24
                            const int s4 = (lane * mul) >> 5, wi = lane - s4 * nwt;
                            if (lane < NS * nwt) {
                                const int w = 4 * ngr + wi;
                                float val = alpha * CCA_LDS_LD(&res[s4 * L + w]);
                                if (has_add) val += CCA_LDS_LD(&add[4 * ngr * NS + lane]);
                                fbuf_store(Ob, val, 4 * ((g0 + s4) * W + w), soff);
                            }
                        }
                    }
                }
            }
        } else {
        const int lane = recompute_here(lane_);
#pragma unroll
        for (int pr = 0; pr < M_MC / NS; ++pr) {
            const int cc = wv + pr * NS, c = ch * M_MC + cc;
            if (c < C) {
                const int soff = c * HW * 4;
                const float *src = img + cc * CP;
#pragma unroll
                for (int m = 0; m < strip_pieces_c(NS); ++m)
                    if (FULL || m < npieces) {
                        // natural image order; the column launch scatters into the band-permuted layout, the row
                        // launch gathers its addend from it (slow path: per-element index arithmetic)
                        const int e = m * 64 + lane;
                        float val = alpha * CCA_LDS_LD(&src[e]);
                        if constexpr (!ROW) {
                            const int pos = e / NS, gg = (e % NS) ^ col_swizzle<NS>(pos);
                            const bool ok = pos < L && gg < gvalid;
                            if (has_add) val += CCA_LDS_LD(&lds[2 * BUF + cc * CP + e]);
                            fbuf_store(Ob, val, ok ? 4 * blocked_offset<NS>(pos, g0 + gg, H, W) : kOobOffset, soff);
                        } else {
                            const int s4 = e / L, w = e - s4 * L;
                            const bool ok = s4 < gvalid;
                            if (has_add && ok)
                                val += CCA_LDS_LD(&lds[2 * BUF + cc * CP + blocked_offset<NS>(g0 + s4, w, H, W) - g0 * W]);
                            fbuf_store(Ob, val, ok ? 4 * ((g0 + s4) * W + w) : kOobOffset, soff);
                        }
                    }
            }
        }
        }
        // images free for the next iteration and chunk ch+1 landed; the tile stores -- exactly nstore per wave, the
        // youngest operations -- stay in flight (vector-memory operations retire in issue order: stress-tested under
        // concurrent HBM load, tools/stress_stage.py; an out-of-range, i.e. dropped, store does NOT: never issue one here)
        if constexpr (EXACT) {
            if (counted) barrier_dma_keep<NSTORE_MIN>();
            else         barrier_lds_only();
        } else {
            if (counted) barrier_dma_keep_n(nstore);
            else         barrier_lds_only();
        }
    }
}

template <int NS, bool ROW, bool TRANS, int EPI, bool BF>
__global__ __launch_bounds__(kWave * NS, 2) void map_strip_kernel(const float *__restrict__ T,
                                                                   const float *__restrict__ F,
                                                                   const float *__restrict__ resid,
                                                                   const float *__restrict__ gamma,
                                                                   float *out, int C, int H, int W,
                                                                   int chunks_per_block, int tiles, int nsplit,
                                                                   long fbs, long rbs, long obs) {
    __shared__ __attribute__((aligned(16))) float lds[m_lds_floats(NS)];
    CCA_LDS_REGISTER(lds);
    const int L = ROW ? W : H;
    const int wg_linear = blockIdx.x, wg_count = gridDim.x;
    const bool exact = L > 4 * (M_KS - 1);
    if constexpr (BF) {           // the host only selects BF for strips 96..100 long
        if (exact) map_strip_body<NS, ROW, TRANS, EPI, true, true, true>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs, rbs, obs);
        else       map_strip_body<NS, ROW, TRANS, EPI, true, true, false>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs, rbs, obs);
    } else {
        if (exact)
            map_strip_body<NS, ROW, TRANS, EPI, true, false, true>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs, rbs, obs);
        else if (L >= kFullMinStrip)
            map_strip_body<NS, ROW, TRANS, EPI, true, false>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs, rbs, obs);
        else
            map_strip_body<NS, ROW, TRANS, EPI, false, false>(lds, T, F, resid, gamma, out, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs, rbs, obs);
    }
}

// Two independent map problems that share the attention-shaped operand T in ONE launch: workgroups
// [0, n) run the non-transposed problem (F0 -> out0), workgroups [n, 2n) the transposed one (F1 -> out1).
// Used for ca_backward (dq from k, dk from q): each half alone would leave most CUs idle.
template <int NS, bool ROW, int EPI, bool BF>
__global__ __launch_bounds__(kWave * NS, 2) void map_strip_dual_kernel(const float *__restrict__ T,
                                                                        const float *__restrict__ F0, float *out0,
                                                                        const float *__restrict__ F1, float *out1,
                                                                        const float *__restrict__ gamma,
                                                                        int C, int H, int W,
                                                                        int chunks_per_block, int tiles, int nsplit,
                                                                        long fbs0, long obs0, long fbs1, long obs1) {
    __shared__ __attribute__((aligned(16))) float lds[m_lds_floats(NS)];
    CCA_LDS_REGISTER(lds);
    const int L = ROW ? W : H;
    const int half = gridDim.x / 2;
    const bool second = (int)blockIdx.x >= half;
    const int wg_linear = second ? blockIdx.x - half : blockIdx.x, wg_count = half;
    if constexpr (BF) {           // the host only selects BF for strips 97..100 long
        if (!second) map_strip_body<NS, ROW, false, EPI, true, true>(lds, T, F0, nullptr, gamma, out0, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs0, 0, obs0);
        else         map_strip_body<NS, ROW, true, EPI, true, true>(lds, T, F1, nullptr, gamma, out1, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs1, 0, obs1);
    } else {
        const bool full = L >= kFullMinStrip, exact = L > 4 * (M_KS - 1);
        if (exact) {
            if (!second) map_strip_body<NS, ROW, false, EPI, true, false, true>(lds, T, F0, nullptr, gamma, out0, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs0, 0, obs0);
            else         map_strip_body<NS, ROW, true, EPI, true, false, true>(lds, T, F1, nullptr, gamma, out1, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs1, 0, obs1);
        } else if (!second) {
            if (full) map_strip_body<NS, ROW, false, EPI, true, false>(lds, T, F0, nullptr, gamma, out0, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs0, 0, obs0);
            else      map_strip_body<NS, ROW, false, EPI, false, false>(lds, T, F0, nullptr, gamma, out0, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs0, 0, obs0);
        } else {
            if (full) map_strip_body<NS, ROW, true, EPI, true, false>(lds, T, F1, nullptr, gamma, out1, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs1, 0, obs1);
            else      map_strip_body<NS, ROW, true, EPI, false, false>(lds, T, F1, nullptr, gamma, out1, C, H, W, chunks_per_block, tiles, nsplit, wg_linear, wg_count, fbs1, 0, obs1);
        }
    }
}

}  // namespace cca

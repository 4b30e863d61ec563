// cca_band.hpp -- "row-band" aggregation kernel: BOTH branches of ca_map_forward in one launch, no column->row
// partial sum through HBM.
//
//   out[b, c, h, w] = alpha * ( sum_j A[b,h,w,j] * v[b, j, w, c]  +  sum_j A[b,h,w,H+j] * v[b, h, j, c] ) + resid[b,c,h,w]
//                                \-------- column part --------/     \---------- row part ----------/
//   (/root/reference/cc_attention/functions.py:42-49; v in PIXEL-MAJOR layout (B, H*W, pixel stride), the layout the
//   stacked 1x1 projection GEMM emits for free; resid / out stay NCHW like the module's x and y.)
//
// Work decomposition (MI355X).  One workgroup = (image b, band of <= 16 rows, group of 32 channels), 4 wavefronts, one
// per SIMD, each with the whole 512-register file:
//   * the output tile (16 rows x W x 32 channels) stays in ACCUMULATOR REGISTERS for the whole kernel, in the
//     "row layout" of v_mfma_f32_16x16x32_bf16 D tiles with M = 16 consecutive w, N = 16 channels
//     (wave (mh, ct): w-tiles [4 mh, 4 mh + 4), channels [16 ct, 16 ct + 16), all 16 rows: 16 x 4 x 4 = 256 VGPRs);
//   * column part: for every column w the 16 x H slab A[b, band rows, w, 0:H] times the H x 32 feature column
//     v[b, :, w, cg] (M = band row, K = j, N = channel): one (column, channel tile) job per wave and stage, two columns per
//     stage, 3 stages in flight.  The band workgroups of one (image, channel group) stream the same feature slice
//     at the same time and are neighbours in the XCD-aware launch order, so the slice is fetched from HBM once and
//     served 7x from L2 (tools/probes/band_share_probe.hip: ~20 TB/s L2 -> LDS).  The job's 16 x 16 result has
//     M = band row, so it goes through a small LDS exchange buffer and is added into the row-layout accumulators
//     once per 16 columns;
//   * row part: per band row the W x W block A[b, h, :, H:H+W] times the W x 32 feature row v[b, h, :, cg]
//     (M = w, K = j, N = channel), accumulated on top; then y = resid + alpha * acc leaves through an LDS image
//     [channel][w] as whole NCHW rows.
// All global -> LDS traffic is LDS-DMA in 16-byte lanes; feature lines are 128-byte segments (32 channels of one
// pixel), attention rows 388-byte runs.  Arithmetic: split-bf16 x3 on v_mfma_f32_16x16x32_bf16 for k < 32*floor,
// an exact f32 v_mfma_f32_16x16x4_f32 step for a remainder of <= 4 (97 = 3 x 32 + 1), fp32 accumulation.
#pragma once
#include "cca_common.hpp"

#ifndef BAND_ABL
#define BAND_ABL 0        // development ablations (tools/probes/band_abl.hip): 1 no column MFMA, 2 no column DMA,
#endif                    // 4 no row MFMA, 8 no row DMA, 16 skip the column part, 32 skip the row part

namespace cca {

constexpr int BD_R = 16;                 // rows per band = MFMA M of the column part
constexpr int BD_CC = 32;                // channels per workgroup = two MFMA N tiles
constexpr int BD_THREADS = 256;
constexpr int BD_PP = 8 * BD_CC + 16;    // floats per 8-pixel DMA piece of a feature line (+16: the two k-groups of a
                                         // 32-lane ds_read_b32 group land on different banks)

__host__ __device__ constexpr int bd_max(int a, int b) { return a > b ? a : b; }

// P = padded strip length (multiple of 4, >= max(H, W)): every LDS pitch and DMA instruction count derives from it
template <int P>
struct BandCfg {
    static constexpr int P4 = P / 4;                           // 16-byte chunks per attention row
    static constexpr int NT = (P + 15) / 16;                   // w tiles
    static constexpr int T0 = (NT + 1) / 2;                    // tiles owned by the mh = 0 waves
    static constexpr int NPV = (P + 7) / 8;                    // 8-pixel pieces per feature line
    static constexpr int VSZ = NPV * BD_PP;                    // floats per feature-line image
    static constexpr int ASZ_C = BD_R * P;                     // column part: attention slab of one column
    static constexpr int NAC = (BD_R * P4 + 63) / 64;          // DMA instructions per slab
    static constexpr int STG = 2 * VSZ + 2 * ASZ_C;            // one stage = two columns
    static constexpr int NSTG = 3;
    static constexpr int NITEM_C = 2 * (NPV + NAC);
    static constexpr int XSZ = BD_R * 16 * BD_CC;              // exchange buffer [band row][column in block][channel]
    static constexpr int ASZ_R = P * P;                        // row part: attention block of one row
    static constexpr int NAR = (P * P4 + 63) / 64;
    static constexpr int XRSZ = BD_CC * P;                     // residual / result image [channel][w]
    static constexpr int NXR = (BD_CC * P4 + 63) / 64;
    static constexpr int NITEM_R = NAR + NPV + NXR;
    static constexpr int LDS_C = NSTG * STG + XSZ;
    static constexpr int LDS_R = 2 * ASZ_R + 2 * VSZ + 2 * XRSZ;
    static constexpr int LDS = bd_max(LDS_C, LDS_R);           // floats (P = 100: 39,008 = 156,032 B)
};

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

// B fragment (K x N = 32 line positions x 16 channels) of a feature-line image: lane (n = l & 15, kg = l >> 4) holds
// positions 32 ks + 8 kg + e, e < 8, of channel 16 ct + n
__device__ __forceinline__ BfSplit band_feat_frag(const float *img, int ks, int ct, int lane) {
    const float *p = img + (4 * ks + (lane >> 4)) * BD_PP + 16 * ct + (lane & 15);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = CCA_LDS_LD(p + BD_CC * e);
    return bf16_split8(x);
}
__device__ __forceinline__ float band_feat_one(const float *img, int pos, int ct, int lane) {
    return CCA_LDS_LD(img + (pos >> 3) * BD_PP + (pos & 7) * BD_CC + 16 * ct + (lane & 15));
}

// A fragment (M x K = 16 rows x 32 positions) from attention rows of pitch `pitch`: lane (m = l & 15, kg = l >> 4)
// holds row m, positions 32 ks + 8 kg + e; positions >= L are zero (the slots hold whatever followed the row)
__device__ __forceinline__ BfSplit band_att_frag(const float *rows, int pitch, int ks, int L, int lane) {
    const int k0 = 32 * ks + 8 * (lane >> 4);
    const float *p = rows + (lane & 15) * pitch + k0;
    const f32x4 u = lds_load_x4(p), v = lds_load_x4(p + 4);
    float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (k0 + e < L) ? x[e] : 0.f;
    return bf16_split8(x);
}

__device__ __forceinline__ f32x4 band_mfma3(const BfSplit &a, const BfSplit &b, f32x4 acc) {
    acc = mfma_bf16_16x16x32(a.hi, b.hi, acc);
    acc = mfma_bf16_16x16x32(a.hi, b.lo, acc);
    acc = mfma_bf16_16x16x32(a.lo, b.hi, acc);
    return acc;
}

// forward aggregation + residual, NCHW output
template <int P>
__global__ __launch_bounds__(BD_THREADS, 1) void map_band_fwd_kernel(const float *__restrict__ A, const float *__restrict__ F,
                                                                      const float *__restrict__ resid,
                                                                      const float *__restrict__ gamma, float *out,
                                                                      int C, int H, int W, int nb, int rpb, int ncg,
                                                                      long fbs, int fps) {
    using Cfg = BandCfg<P>;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS];
    CCA_LDS_REGISTER(lds);
    constexpr int P4 = Cfg::P4, NT = Cfg::NT, T0 = Cfg::T0, NPV = Cfg::NPV, VSZ = Cfg::VSZ;
    const int HW = H * W, S = H + W;
    const int id = xcd_logical_id(blockIdx.x, gridDim.x);
    const int band = id % nb, cg = (id / nb) % ncg, b = id / (nb * ncg);
    const int h0 = band * rpb;
    const int nrows = (H - h0 < rpb) ? H - h0 : rpb;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int mh = wv >> 1, ct = wv & 1;                  // accumulator ownership: w tiles [mh * T0, ...), channel tile ct
    const int ln = lane & 15, lg = lane >> 4;
    const int ch0 = cg * BD_CC;

    const FBuf Ab = make_fbuf(A + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
    const FBuf Fb = make_fbuf(F + (size_t)b * fbs, ((size_t)(HW - 1) * fps + C) * sizeof(float));
    const FBuf Rb = make_fbuf(resid + (size_t)b * C * HW, (size_t)C * HW * sizeof(float));
    const FBuf Ob = make_fbuf(out + (size_t)b * C * HW, (size_t)C * HW * sizeof(float));
    const float alpha = gamma ? gamma[0] : 1.f;

    // one 8-pixel piece of a feature line (pixels p0 + i * pstep, i < n) -> img
    auto dma_line_piece = [&](float *img, int piece, int p0, int pstep, int n) {
        const int i = 8 * piece + (lane >> 3), c = ch0 + 4 * (lane & 7);
        if (i < n && c < C) fbuf_load_to_lds_x4(Fb, img + piece * BD_PP, ((p0 + i * pstep) * fps + c) * 4, 0);
    };

    f32x4 acc[BD_R][T0];
#pragma unroll
    for (int r = 0; r < BD_R; ++r)
#pragma unroll
        for (int t = 0; t < T0; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // masked DMA lanes leave their slots untouched and the k padding of a feature image is multiplied by zeroed
    // attention operands: it has to be FINITE, so everything starts from zero
    for (int i = tid * 4; i < Cfg::LDS; i += BD_THREADS * 4) lds_store_x4(&lds[i], f32x4{0.f, 0.f, 0.f, 0.f});
    __syncthreads();

    // ------------------------------------------------------------------------------------------------------------
    // column part
    // ------------------------------------------------------------------------------------------------------------
    if (!(BAND_ABL & 16)) {
        const BandK kc = band_ksteps(H);
        const int nstage = (W + 1) >> 1;
        float *const X = lds + Cfg::NSTG * Cfg::STG;
        auto issue = [&](int gs) {
            if (BAND_ABL & 2) return;
            float *stg = lds + (gs % Cfg::NSTG) * Cfg::STG;
            for (int it = wv; it < Cfg::NITEM_C; it += 4) {
                const int col = it / (NPV + Cfg::NAC), r = it - col * (NPV + Cfg::NAC);
                const int w = 2 * gs + col;
                if (w >= W) continue;
                if (r < NPV) {
                    dma_line_piece(stg + col * VSZ, r, w, W, H);
                } else {
                    const int q = r - NPV, idx = 64 * q + lane;
                    const int rr = idx / P4, chk = idx - rr * P4;
                    if (rr < nrows && 4 * chk < H)
                        fbuf_load_to_lds_x4(Ab, stg + 2 * VSZ + col * Cfg::ASZ_C + 256 * q,
                                            (((h0 + rr) * W + w) * S + 4 * chk) * 4, 0);
                }
            }
        };
        issue(0);
        if (nstage > 1) issue(1);
#pragma unroll
        for (int blk = 0; blk < NT; ++blk) {
            if (blk * 16 < W) {
                for (int s = 0; s < 8; ++s) {
                    const int gs = blk * 8 + s;
                    if (gs >= nstage) break;
                    barrier_dma_keep<0>();           // stages gs (and gs + 1) landed; every wave is done with stage gs - 1
                    if (gs + 2 < nstage) issue(gs + 2);
                    const int col = wv >> 1, w = 2 * gs + col;
                    if (w < W && !(BAND_ABL & 1)) {
                        const float *stg = lds + (gs % Cfg::NSTG) * Cfg::STG;
                        const float *img = stg + col * VSZ, *slab = stg + 2 * VSZ + col * Cfg::ASZ_C;
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                        for (int ks = 0; ks < kc.nbf; ++ks)
                            d = band_mfma3(band_att_frag(slab, P, ks, H, lane), band_feat_frag(img, ks, ct, lane), d);
                        if (kc.tail) {
                            const int pos = 32 * kc.nbf + lg;
                            const float a = pos < H ? CCA_LDS_LD(slab + ln * P + pos) : 0.f;
                            d = mfma_16x16x4(a, band_feat_one(img, pos, ct, lane), d);
                            mfma_f32_result_fence();
                        }
                        // D[m = band row 4 lg + q][n = channel 16 ct + ln] of column w
                        float *xp = X + ((4 * lg) * 16 + (w & 15)) * BD_CC + 16 * ct + ln;
#pragma unroll
                        for (int q = 0; q < 4; ++q) CCA_LDS_ST(xp + q * 16 * BD_CC, d[q]);
                    }
                }
                barrier_lds_only();                  // the block's exchange buffer is complete
                if ((blk < T0) == (mh == 0)) {
                    const float *xp = X + (4 * lg) * BD_CC + 16 * ct + ln;
#pragma unroll
                    for (int r = 0; r < BD_R; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[r][blk < T0 ? blk : blk - T0][q] += CCA_LDS_LD(xp + (r * 16 + q) * BD_CC);
                }
                barrier_lds_only();                  // ... and consumed before the next block overwrites it
            }
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // row part + epilogue, one band row per step
    // ------------------------------------------------------------------------------------------------------------
    if (!(BAND_ABL & 32)) {
        const BandK kr = band_ksteps(W);
        float *const AR = lds, *const VR = lds + 2 * Cfg::ASZ_R, *const XR = VR + 2 * VSZ;
        __syncthreads();
        for (int i = tid * 4; i < 2 * VSZ; i += BD_THREADS * 4) lds_store_x4(&VR[i], f32x4{0.f, 0.f, 0.f, 0.f});
        __syncthreads();
        auto issue = [&](int r) {
            if (BAND_ABL & 8) return;
            const int h = h0 + r;
            float *ar = AR + (r & 1) * Cfg::ASZ_R, *vr = VR + (r & 1) * VSZ, *xr = XR + (r & 1) * Cfg::XRSZ;
            for (int it = wv; it < Cfg::NITEM_R; it += 4) {
                if (it < Cfg::NAR) {
                    const int idx = 64 * it + lane, wq = idx / P4, chk = idx - wq * P4;
                    if (wq < W && 4 * chk < W)
                        fbuf_load_to_lds_x4(Ab, ar + 256 * it, ((h * W + wq) * S + H + 4 * chk) * 4, 0);
                } else if (it < Cfg::NAR + NPV) {
                    dma_line_piece(vr, it - Cfg::NAR, h * W, 1, W);
                } else {
                    const int q = it - Cfg::NAR - NPV, idx = 64 * q + lane, cc = idx / P4, chk = idx - cc * P4;
                    if (cc < BD_CC && ch0 + cc < C && 4 * chk < W)
                        fbuf_load_to_lds_x4(Rb, xr + 256 * q, ((ch0 + cc) * HW + h * W + 4 * chk) * 4, 0);
                }
            }
        };
        issue(0);
        const int tbase = mh ? T0 : 0;
#pragma unroll
        for (int r = 0; r < BD_R; ++r) {
            if (r < nrows) {
                const float *ar = AR + (r & 1) * Cfg::ASZ_R, *vr = VR + (r & 1) * VSZ;
                float *xr = XR + (r & 1) * Cfg::XRSZ;
                const int h = h0 + r;
                barrier_dma_keep<0>();               // row r landed; every wave is done with row r - 1
                if (r + 1 < nrows) issue(r + 1);
                for (int ks = 0; ks < ((BAND_ABL & 4) ? 0 : kr.nbf); ++ks) {
                    const BfSplit fb = band_feat_frag(vr, ks, ct, lane);
#pragma unroll
                    for (int tl = 0; tl < T0; ++tl) {
                        const int t = tbase + tl;
                        if (t < NT && t * 16 < W)
                            acc[r][tl] = band_mfma3(band_att_frag(ar + 16 * t * P, P, ks, W, lane), fb, acc[r][tl]);
                    }
                }
                if (kr.tail) {
                    const int pos = 32 * kr.nbf + lg;
                    const float fbv = band_feat_one(vr, pos, ct, lane);
#pragma unroll
                    for (int tl = 0; tl < T0; ++tl) {
                        const int t = tbase + tl;
                        if (t < NT && t * 16 < W) {
                            const float a = pos < W ? CCA_LDS_LD(ar + (16 * t + ln) * P + pos) : 0.f;
                            acc[r][tl] = mfma_16x16x4(a, fbv, acc[r][tl]);
                        }
                    }
                    mfma_f32_result_fence();
                }
                // y = resid + alpha * acc, in place in the residual image: lane holds w = 16 t + 4 lg .. + 3 of channel 16 ct + ln
#pragma unroll
                for (int tl = 0; tl < T0; ++tl) {
                    const int t = tbase + tl;
                    if (t < NT && 16 * t + 4 * lg < P) {
                        float *p = xr + (16 * ct + ln) * P + 16 * t + 4 * lg;
                        lds_store_x4(p, lds_load_x4(p) + alpha * acc[r][tl]);
                    }
                }
                barrier_lds_only();
                for (int it = wv; it < Cfg::NXR; it += 4) {
                    const int idx = 64 * it + lane, cc = idx / P4, chk = idx - cc * P4;
                    if (cc < BD_CC && ch0 + cc < C && 4 * chk < W) {
                        const f32x4 val = lds_load_x4(xr + cc * P + 4 * chk);
                        const int off = ((ch0 + cc) * HW + h * W + 4 * chk) * 4;
                        if (4 * chk + 3 < W) {
                            fbuf_store_x4(Ob, val, off, 0);
                        } else {
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                if (4 * chk + e < W) fbuf_store(Ob, val[e], off + 4 * e, 0);
                        }
                    }
                }
            }
        }
    }
}

}  // namespace cca

// cca_gmap.hpp -- strip aggregation on PIXEL-MAJOR features: one strip per workgroup, the strip's attention block
// stationary in LDS as pre-split bf16, channel groups of 64 streaming through.
//
//   TRANS = false   out[pixel(i, g), c] (+)= alpha * sum_j P_g[i][j] * F[pixel(j, g), c]
//   TRANS = true    out[pixel(j, g), c] (+)= alpha * sum_i P_g[i][j] * F[pixel(i, g), c]
//   with P_g[i][j] = T[b, pixel(i, g), a_off + j]  (cca_map.hpp has the same contractions on NCHW features)
//
// Why another kernel family: in NCHW the column branch of a strip tile is a stream of 32-byte segments (8 strips x
// 4 B), which the L1 / TA path serves at a third of the row rate -- the column launches of cca_map.hpp are the
// slowest kernels of a step.  With the features PIXEL-MAJOR (B, H*W, pixel stride) -- the layout the value projection
// has when it is computed as x^T W^T -- a pixel's 64 channels are one 256-byte segment, for column strips and row
// strips alike.
//
// Work decomposition (MI355X): workgroup = one strip g of one image, 4 wavefronts = the four 16-channel N tiles of a
// 64-channel group.  Prologue: the L rows of P_g (contiguous in T) arrive by LDS-DMA and are rewritten ONCE as two
// bf16 images (hi, lo = the split of cca_common.hpp; transposed for TRANS; zero beyond the strip), row pitch 272 B so
// that a 16 x 32 MFMA A fragment is one ds_read_b128 per image and no VALU.  Then, per channel group: the L x 64
// feature tile arrives by LDS-DMA (double-buffered, 4 pixels per instruction), each wavefront gathers its B fragments
// (8 ds_read_b32 + split per k-step, reused by all 7 M tiles: 21 MFMAs per gather), accumulates 7 tiles of
// v_mfma_f32_16x16x32_bf16 (+ one exact f32 step for a k remainder <= 4), adds the addend tile that the DMA dropped
// into the output image, and the image leaves as 256-byte pixel rows.
#pragma once
#include "cca_band.hpp"
#include "cca_common.hpp"

namespace cca {

constexpr int GM_CG = 64;                       // channels per group = four MFMA N tiles
constexpr int GM_THREADS = 256;
constexpr int GM_PP = 4 * GM_CG + 8;            // floats per 4-pixel DMA piece of a feature tile (+8: bank spread)
constexpr int GM_BP = 68;                       // dwords per row of a bf16 attention image (136 bf16: 128 + pad)
constexpr int GM_EPI_PM = 0, GM_EPI_PM_ADD = 1; // output pixel-major, without / with a pixel-major addend

template <int P>
struct GmapCfg {
    static constexpr int NT = (P + 15) / 16;                // M tiles
    static constexpr int NPF = (P + 3) / 4;                 // 4-pixel pieces per feature tile
    static constexpr int FSZ = NPF * GM_PP;                 // floats per feature / output tile
    static constexpr int ASZ = P * GM_BP;                   // dwords per bf16 image
    static constexpr int TSZ = P * 4;                       // exact f32 k tail
    static constexpr int OFF_PH = 0, OFF_PL = ASZ, OFF_PT = 2 * ASZ, OFF_F = 2 * ASZ + TSZ, OFF_O = OFF_F + 2 * FSZ;
    static constexpr int LDS = OFF_O + 2 * FSZ;             // floats (P = 100: 40,400 = 161,600 B)
    static constexpr int NPA = (P * (P / 4) + 63) / 64;     // DMA instructions of the raw attention block
    static_assert(P % 4 == 0 && P <= 128, "GmapCfg: padded strip length");
    static_assert(P * P <= 4 * FSZ, "the raw attention block is staged in the feature / output buffers");
};

template <int P, bool ROW, bool TRANS, int EPI>
__global__ __launch_bounds__(GM_THREADS, 1) void gmap_kernel(const float *__restrict__ T, const float *__restrict__ F,
                                                              const float *__restrict__ addend,
                                                              const float *__restrict__ gamma, float *out,
                                                              int C, int H, int W, long fbs, int fps, long obs, int ops) {
    using Cfg = GmapCfg<P>;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS];
    CCA_LDS_REGISTER(lds);
    constexpr int NT = Cfg::NT, NPF = Cfg::NPF, FSZ = Cfg::FSZ, P4 = P / 4;
    const int HW = H * W, S = H + W;
    const int L = ROW ? W : H, G = ROW ? H : W;
    const int id = xcd_logical_id(blockIdx.x, gridDim.x);
    const int b = id / G, g = id - b * G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), nt = uniform(tid >> 6);     // wave = N tile
    const int ln = lane & 15, lg = lane >> 4;
    const int pix0 = ROW ? g * W : g, pstep = ROW ? 1 : W;                             // pixel(i) = pix0 + i * pstep
    const int a_off = ROW ? H : 0;

    const FBuf Tb = make_fbuf(T + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
    const FBuf Fb = make_fbuf(F + (size_t)b * fbs, ((size_t)(HW - 1) * fps + C) * sizeof(float));
    const FBuf Ob = make_fbuf(out + (size_t)b * obs, ((size_t)(HW - 1) * ops + C) * sizeof(float));
    const FBuf Db = make_fbuf((EPI == GM_EPI_PM_ADD ? addend : out) + (size_t)b * obs, ((size_t)(HW - 1) * ops + C) * sizeof(float));
    const float alpha = gamma ? gamma[0] : 1.f;
    const int ncg = (C + GM_CG - 1) / GM_CG;
    const BandK kp = band_ksteps(L);

    uint32_t *const PH = reinterpret_cast<uint32_t *>(lds + Cfg::OFF_PH), *const PL = reinterpret_cast<uint32_t *>(lds + Cfg::OFF_PL);
    float *const PT = lds + Cfg::OFF_PT, *const FB = lds + Cfg::OFF_F, *const OB = lds + Cfg::OFF_O;

    // ---- prologue: raw rows of P_g -> staging (the tile buffers) -> bf16 hi / lo images (+ exact k tail) ----------
    {
        float *stage = FB;                                   // [i][P], P * P floats
        for (int it = nt; it < Cfg::NPA; it += 4) {
            const int idx = 64 * it + lane, i = idx / P4, chk = idx - i * P4;
            if (i < L && 4 * chk < L)
                fbuf_load_to_lds_x4(Tb, stage + 256 * it, ((pix0 + i * pstep) * S + a_off + 4 * chk) * 4, 0);
        }
        __syncthreads();                                     // (drains the DMA)
        // destination element [m][k]: m = output position, k = contraction position
        for (int e = tid; e < P * (P / 2); e += GM_THREADS) {
            const int m = e / (P / 2), k = 2 * (e - m * (P / 2));
            float v0 = 0.f, v1 = 0.f;
            if (m < L) {
                if (k < L)     v0 = CCA_LDS_LD(stage + (TRANS ? k * P + m : m * P + k));
                if (k + 1 < L) v1 = CCA_LDS_LD(stage + (TRANS ? (k + 1) * P + m : m * P + k + 1));
            }
            const uint32_t h = cvt_pk_bf16(v0, v1);
            const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
            PH[m * GM_BP + (k >> 1)] = h;
            PL[m * GM_BP + (k >> 1)] = cvt_pk_bf16(v0 - h0, v1 - h1);
        }
        for (int e = tid; e < P * 4; e += GM_THREADS) {
            const int m = e >> 2, k = 32 * kp.nbf + (e & 3);
            PT[e] = (kp.tail && m < L && k < L) ? CCA_LDS_LD(stage + (TRANS ? k * P + m : m * P + k)) : 0.f;
        }
        __syncthreads();
        // the staging area becomes tile buffers: masked DMA lanes leave their slots alone and the k padding of a
        // feature tile meets zero attention operands -- it has to be finite
        for (int i = tid * 4; i < 4 * FSZ; i += GM_THREADS * 4) lds_store_x4(&FB[i], f32x4{0.f, 0.f, 0.f, 0.f});
        __syncthreads();
    }

    // one 4-pixel piece of a pixel-major tile (pixels pix0 + i * pstep, channels cg*64 ..) -> img
    auto dma_piece = [&](const FBuf &src, float *img, int piece, int cg, int ps) {
        const int i = 4 * piece + (lane >> 4), c = cg * GM_CG + 4 * (lane & 15);
        if (i < L && c < C) fbuf_load_to_lds_x4(src, img + piece * GM_PP, ((pix0 + i * pstep) * ps + c) * 4, 0);
    };
    auto issue = [&](int cg) {
        for (int it = nt; it < NPF; it += 4) dma_piece(Fb, FB + (cg & 1) * FSZ, it, cg, fps);
        if (EPI == GM_EPI_PM_ADD)
            for (int it = nt; it < NPF; it += 4) dma_piece(Db, OB + (cg & 1) * FSZ, it, cg, ops);
    };

    issue(0);
    for (int cg = 0; cg < ncg; ++cg) {
        const float *img = FB + (cg & 1) * FSZ;
        float *oimg = OB + (cg & 1) * FSZ;
        barrier_dma_keep<0>();                   // tile cg (and its addend) landed; every wave is done with group cg - 1
        if (cg + 1 < ncg) issue(cg + 1);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < kp.nbf; ++ks) {
            // B fragment: positions 32 ks + 8 lg + e of channel 16 nt + ln
            const float *p = img + (8 * ks + 2 * lg) * GM_PP + 16 * nt + ln;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = CCA_LDS_LD(p + (e >> 2) * GM_PP + (e & 3) * GM_CG);
            const BfSplit fb = bf16_split8(x);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t * 16 < L) {
                    const int ao = (16 * t + ln) * GM_BP + 16 * ks + 4 * lg;
                    const u32x4 ah = *reinterpret_cast<const u32x4 *>(PH + ao), al = *reinterpret_cast<const u32x4 *>(PL + ao);
                    acc[t] = mfma_bf16_16x16x32(ah, fb.hi, acc[t]);
                    acc[t] = mfma_bf16_16x16x32(ah, fb.lo, acc[t]);
                    acc[t] = mfma_bf16_16x16x32(al, fb.hi, acc[t]);
                }
            }
        }
        if (kp.tail) {
            const int pos = 32 * kp.nbf + lg;
            const float fbv = CCA_LDS_LD(img + (pos >> 2) * GM_PP + (pos & 3) * GM_CG + 16 * nt + ln);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t * 16 < L) acc[t] = mfma_16x16x4(CCA_LDS_LD(PT + (16 * t + ln) * 4 + lg), fbv, acc[t]);
            mfma_f32_result_fence();
        }
        // D[m = position 16 t + 4 lg + q][n = channel 16 nt + ln] -> output image (pixel-major pieces), + addend
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * t + 4 * lg + q;
                if (i < L) {
                    float *d = oimg + (i >> 2) * GM_PP + (i & 3) * GM_CG + 16 * nt + ln;
                    float val = alpha * acc[t][q];
                    if (EPI == GM_EPI_PM_ADD) val += CCA_LDS_LD(d);
                    CCA_LDS_ST(d, val);
                }
            }
        barrier_lds_only();
        for (int it = nt; it < NPF; it += 4) {
            const int i = 4 * it + (lane >> 4), c = cg * GM_CG + 4 * (lane & 15);
            if (i < L && c < C)
                fbuf_store_x4(Ob, lds_load_x4(oimg + it * GM_PP + 4 * lane), ((pix0 + i * pstep) * ops + c) * 4, 0);
        }
    }
}

}  // namespace cca

// cca_long.hpp -- strip kernels for LONG strips (101 .. 320 positions): the same six contractions as
// cca_weight.hpp / cca_map.hpp on shapes whose attention block no longer fits a wavefront's registers
// (129 x 129 of BASELINE configs[4], the 129 x 257 whole-image evaluation map of evaluate.py --whole).
//
// Same decomposition -- one workgroup = NS adjacent strips, one wavefront per strip, exact-fp32
// v_mfma_f32_16x16x4_f32, feature chunks through LDS by LDS-DMA -- with the output positions of a strip cut into
// WINDOWS that become an extra grid dimension:
//   map kernels     a workgroup produces the outputs of a window of NTW position tiles; its wavefronts keep only the
//                   L x (16 NTW) slice of the attention block they need as MFMA B fragments (<= 160 VGPRs)
//   weight kernel   a workgroup produces the rows of a window of NTI query tiles of the L x L result (<= 40 tiles
//                   of accumulators); the channel contraction streams through LDS
// Every window re-streams the features of its strips (from L2 / Infinity Cache: the tensors of these shapes are
// small or re-used at once), which is what the register file costs at these lengths.
//
//   configuration      strips / workgroup   longest strip   map window   weight window
//   NS = 4, 2 waves    4 (16-byte column segments)   144    two 5-tile windows INSIDE the workgroup (map kernel only)
//   NS = 4             4 (16-byte column segments)   160    4 tiles      4 query tiles
//   NS = 2             2                             320    2 tiles      2 query tiles
//
// The column -> row partial sums of a map launch pair stay in the NATURAL layout here (windows of one row band are
// different workgroups, so the band permutation of cca_map.hpp would race); the addend is read from global memory
// in the store phase.  These kernels are deliberately simple (no counted waits, no split-bf16): they exist so that
// no shape up to 320 x 320 falls back to the one-thread-per-output kernels.
#pragma once
#include "cca_common.hpp"
#include "cca_map.hpp"      // EPI_COL / EPI_ROW / EPI_COL_RESID

#include <type_traits>

namespace cca {

constexpr int kLongMaxStrip = 320;
__host__ __device__ constexpr int long_maxl(int ns) { return ns == 4 ? 160 : 320; }
__host__ __device__ constexpr int long_window_tiles(int ns) { return ns == 4 ? 4 : 2; }
__host__ __device__ constexpr int long_cp(int ns) { return ns * long_maxl(ns) + 20; }      // channel pitch of an image
constexpr int LG_MC = 16;                         // channels per chunk of the map kernel = one MFMA M tile
// Epilogue of the long family's ROW launch when the op has a residual: out = alpha * row + partial + resid.  Here the
// residual belongs to the row launch (its addend loads are whole rows; in the column launch they would be 4-byte
// gathers) -- the stationary family keeps it in the column launch because its row launch has no LDS image to spare.
constexpr int EPI_ROW_RESID = 3;
constexpr int LG_KC = 8;                          // channels per chunk of the weight kernel = 2 MFMA k-steps

// global element offset (inside one channel plane) of position `pos` of strip `strip`
template <bool ROW>
__device__ __forceinline__ int long_plane_offset(int pos, int strip, int W) {
    return ROW ? strip * W + pos : pos * W + strip;
}

// LDS-DMA of the NS strips x L positions of one channel plane into an image: column image index pos * NS + s,
// row image index s * L + pos (a linear copy of the NS rows).  Lanes outside the tile are masked; their LDS slots
// keep the zeros the image was initialised with.
template <int NS, bool ROW>
__device__ __forceinline__ void long_dma_plane(const FBuf &src, float *dst, int soff, int lane, int L, int W,
                                               int g0, int gvalid) {
    if (ROW || (NS == 4 && gvalid == NS)) {
        // 16-byte pieces (1 KiB per wave instruction): the row image is a linear copy; with 4 strips per workgroup a
        // column-image position (4 strips) is exactly one 16-byte piece.  A ragged tail of < 4 floats goes by dwords.
        const int total = ROW ? gvalid * L : NS * L, body = total & ~3;
        for (int e0 = 0; e0 < body; e0 += 4 * kWave) {        // wave-uniform trip count
            const int e = e0 + 4 * lane;
            if (e < body) fbuf_load_to_lds_x4(src, dst + e0, ROW ? 4 * (g0 * W + e) : 4 * ((e / NS) * W + g0), soff);
        }
        if (ROW && body < total && lane < total - body) fbuf_load_to_lds(src, dst + body, 4 * (g0 * W + body + lane), soff);
        return;
    }
    const int total = NS * L;
    for (int e0 = 0; e0 < total; e0 += kWave) {               // wave-uniform trip count
        const int e = e0 + lane;
        const int pos = e / NS, s = e % NS;
        if (pos < L && s < gvalid) fbuf_load_to_lds(src, dst + e0, 4 * (pos * W + g0 + s), soff);
    }
}

// ---------------------------------------------------------------------------------------------
// map type:  TRANS = false  out[c, pos(i)] (+)= sum_j P[i][j] F[c, pos(j)]     TRANS = true  sum_i P[i][j] F[c, pos(i)]
// ---------------------------------------------------------------------------------------------
// Workgroup shape of the map kernel: NS strips x WPS wavefronts per strip.  With WPS = 2 the two wavefronts of a
// strip own the two windows of its output positions (strips up to 144 long = 9 tiles = 5 + 4), so the features of
// a strip are streamed ONCE per workgroup instead of once per window.
template <int NS, int WPS>
struct LongMapCfg {
    static constexpr int MAXL = WPS == 2 ? 144 : long_maxl(NS);
    static constexpr int NTW = WPS == 2 ? 5 : long_window_tiles(NS);          // position tiles per wavefront
    static constexpr int CP = NS * MAXL + 20;                                  // channel pitch of a feature image
    static constexpr int NW = NS * WPS;                                        // wavefronts per workgroup
};

template <int NS, int WPS, bool ROW, bool TRANS, int EPI>
__global__ __launch_bounds__(kWave * NS * WPS) void map_long_kernel(const float *__restrict__ T, const float *__restrict__ F,
                                                                    const float *__restrict__ resid,
                                                                    const float *__restrict__ gamma, float *out,
                                                                    int C, int H, int W, int chunks_per_block, int tiles,
                                                                    int nsplit, int nwin, int wtiles, long fbs, long rbs, long obs) {
    using Cfg = LongMapCfg<NS, WPS>;
    constexpr int MAXKS = Cfg::MAXL / 4, NTW = Cfg::NTW, CP = Cfg::CP, NW = Cfg::NW;
    constexpr int WIN = NTW * kTile, RP = WIN * NW;           // window positions; floats per channel of the result image
    // feature images: double-buffered with >= 4 wavefronts per workgroup; with 2 the LDS is better spent on a third
    // resident workgroup (measured: 129x257 fwd+bwd 6.1 ms single-buffered vs 7.2 ms double-buffered)
    constexpr int NB = NW >= 4 ? 2 : 1;
    __shared__ float lds[NB * LG_MC * CP + LG_MC * RP];      // feature image(s) + the result windows
    CCA_LDS_REGISTER(lds);
    float *res = lds + NB * LG_MC * CP;
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    // logical id -> (image, channel split, window, tile), tile fastest (WPS = 2: nwin == 1, windows are wavefronts)
    const int id = xcd_logical_id(blockIdx.x, gridDim.x);
    const int per_image = nsplit * nwin * tiles;
    const int b = id / per_image, r0 = id - b * per_image;
    const int split = r0 / (nwin * tiles), r1 = r0 - split * nwin * tiles;
    const int gwin = r1 / tiles, g0 = (r1 - gwin * tiles) * NS;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4;
    const int sw = wv % NS, part = wv / NS;                   // this wavefront's strip and (WPS = 2) window
    const int g = g0 + sw;
    const bool active = g < br.G;
    const int gvalid = (br.G - g0 < NS) ? br.G - g0 : NS;
    // windows are evened out by the host: wtiles <= NTW position tiles each (129 -> 3 + 3 + 3 tiles, or 5 + 4)
    const int wspan = wtiles * kTile;
    const int p0 = (WPS == 2 ? part : gwin) * wspan;          // first output position of this wavefront's window
    const int nks = (L + 3) / 4;
    const int nchunks = (C + LG_MC - 1) / LG_MC;
    const int ch_begin = split * chunks_per_block;
    const int ch_end = (ch_begin + chunks_per_block < nchunks) ? ch_begin + chunks_per_block : nchunks;

    const FBuf Tb = make_fbuf(T + (size_t)b * HW * S, (size_t)HW * S * sizeof(float));
    const FBuf Fb = make_fbuf(F + (size_t)b * fbs, (size_t)C * HW * sizeof(float));
    const FBuf Ob = make_fbuf(out + (size_t)b * obs, (size_t)C * HW * sizeof(float));
    const FBuf Rb = make_fbuf(EPI == EPI_COL_RESID ? resid + (size_t)b * rbs : out + (size_t)b * obs,
                              (size_t)C * HW * sizeof(float));
    const FBuf Xb = make_fbuf(EPI == EPI_ROW_RESID ? resid + (size_t)b * rbs : out + (size_t)b * obs,
                              (size_t)C * HW * sizeof(float));
    const float alpha = gamma ? gamma[0] : 1.f;

    // stationary slice of the attention block as MFMA B fragments B[k][n]: k = contraction position 4 ks + (l >> 4),
    // n = output position p0 + 16 t + (l & 15); gathered straight from global memory (prologue only)
    float bf[MAXKS][NTW];
#pragma unroll
    for (int ks = 0; ks < MAXKS; ++ks)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int kidx = ks * 4 + lk, nidx = p0 + t * kTile + ln;
            const int iq = TRANS ? kidx : nidx, j = TRANS ? nidx : kidx;
            const bool ok = active && t < wtiles && iq < L && j < L;
            const float v = fbuf_load(Tb, ok ? 4 * (iq * br.as_q + g * br.as_g + br.a_off + j) : 0, 0);
            bf[ks][t] = ok ? v : 0.f;
        }

    for (int idx = tid; idx < NB * LG_MC * CP; idx += kWave * NW) CCA_LDS_ST(&lds[idx], 0.f);
    __syncthreads();

    // feature chunk -> LDS image `buf` (channels dealt round-robin to the wavefronts; clamped channels are never stored)
    auto issue = [&](int ch, int buf) {
#pragma unroll
        for (int pr = 0; pr < LG_MC / NW; ++pr) {
            const int cc = wv + pr * NW, c = ch * LG_MC + cc;
            long_dma_plane<NS, ROW>(Fb, lds + buf * LG_MC * CP + cc * CP, (c < C ? c : C - 1) * HW * 4, lane, L, W, g0, gvalid);
        }
    };
    // element e of a channel's result image -> (window position, strip, window): virtual strip vw = strip + NS * window
    auto decode = [&](int e, int &pos, int &s, bool &ok) {
        const int wp = ROW ? e % WIN : e / NW, vw = ROW ? e / WIN : e % NW;
        s = vw % NS;
        const int base = (WPS == 2 ? vw / NS : gwin) * wspan;
        pos = base + wp;
        ok = wp < wspan && pos < L && s < gvalid;
    };
    if (NB == 2 && ch_begin < ch_end) issue(ch_begin, 0);
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int buf = NB == 2 ? (ch - ch_begin) & 1 : 0;
        const float *img = lds + buf * LG_MC * CP;
        if (NB == 1) issue(ch, 0);                            // (every wavefront is past the previous chunk's MFMAs)
        __syncthreads();                                      // chunk ch landed; the other image is free again
        // The addend of this chunk's output windows (the other branch's partial sums / the residual) does not depend
        // on the MFMAs: request it first, then the next feature chunk -- vector-memory operations complete in order,
        // so the store phase can consume the addend while the younger DMA is still in flight.
        constexpr int NPC = RP / kWave, NPW = (LG_MC / NW) * NPC;      // pieces per channel / per wavefront
        float addend[EPI != EPI_COL ? NPW : 1];
        auto load_addend = [&]() {
#pragma unroll
            for (int q = 0; q < NPW; ++q) {
                const int cc = wv + (q / NPC) * NW, c = ch * LG_MC + cc;
                int pos, s;
                bool ok;
                decode((q % NPC) * kWave + recompute_here(lane), pos, s, ok);
                ok = ok && c < C;
                const int voff = ok ? 4 * long_plane_offset<ROW>(pos, g0 + s, W) : kOobOffset, soff = (c < C ? c : 0) * HW * 4;
                addend[q] = fbuf_load(Rb, voff, soff);
                if (EPI == EPI_ROW_RESID) addend[q] += fbuf_load(Xb, voff, soff);
            }
        };
        // (with two wavefronts per strip the registers are needed for the attention fragments during the MFMAs: there
        // the addend is requested after them, at the price of waiting for the feature DMA as well)
        constexpr bool kEarlyAddend = EPI != EPI_COL && WPS == 1;
        if (kEarlyAddend) load_addend();
        if (NB == 2 && ch + 1 < ch_end) issue(ch + 1, buf ^ 1);   // lands while chunk ch is multiplied and stored
        if (active) {
            // the number of position tiles of this window is wave-uniform but only known at run time: dispatch once
            // per chunk to a body with a compile-time tile count (a guard around every MFMA would serialise them)
            auto compute = [&](auto nt_tag) {
                constexpr int NT = decltype(nt_tag)::value;
                f32x4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float *ab = img + ln * CP;              // A fragment: channel = l & 15
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks)
                    if (ks < nks) {
                        const int k = ks * 4 + lk;
                        const float av = CCA_LDS_LD(&ab[k < L ? (ROW ? sw * L + k : k * NS + sw) : 0]);
                        const float a = k < L ? av : 0.f;     // never multiply a zero fragment by foreign data
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = mfma_16x16x4(a, bf[ks][t], acc[t]);
                    }
                mfma_f32_result_fence();
                // D[m = channel 4 (l >> 4) + r][n = window position 16 t + (l & 15)] -> result image
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int wp = t * kTile + ln;
                    float *d = res + (4 * lk) * RP + (ROW ? wv * WIN + wp : wp * NW + wv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) CCA_LDS_ST(&d[r * RP], acc[t][r]);
                }
            };
            if (WPS == 2 || wtiles >= NTW)      compute(std::integral_constant<int, NTW>{});   // (WPS = 2: one body --
            else if (NTW > 4 && wtiles == 4)    compute(std::integral_constant<int, (NTW > 4 ? 4 : 1)>{});   //  several spill)
            else if (NTW > 2 && wtiles == 3)    compute(std::integral_constant<int, (NTW > 2 ? 3 : 1)>{});
            else if (wtiles == 2)               compute(std::integral_constant<int, 2>{});
            else                                compute(std::integral_constant<int, 1>{});
        }
        if (EPI != EPI_COL && !kEarlyAddend) load_addend();
        barrier_lds_only();                                   // result windows complete (the DMA stays in flight)
        // windows of the output tile -> global memory
#pragma unroll
        for (int q = 0; q < NPW; ++q) {
            const int cc = wv + (q / NPC) * NW, c = ch * LG_MC + cc, e = (q % NPC) * kWave + recompute_here(lane);
            int pos, s;
            bool ok;
            decode(e, pos, s, ok);
            if (ok && c < C) {
                float val = alpha * CCA_LDS_LD(&res[cc * RP + e]);
                if (EPI != EPI_COL) val += addend[q];
                fbuf_store(Ob, val, 4 * long_plane_offset<ROW>(pos, g0 + s, W), c * HW * 4);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight type:  T[b, pixel(i, g), a_off + j] = sum_c X[b, c, pos(i, g)] * Y[b, c, pos(j, g)]
// ---------------------------------------------------------------------------------------------
// Workgroup shape of the weight kernel, as for the map kernel: with WPS = 2 the two wavefronts of a strip own the two
// halves of its query tiles (5 + 4 of 9 tiles x 9 key tiles = 45 accumulator tiles), and X, Y stream once.
template <int NS, int WPS>
struct LongWeightCfg {
    static constexpr int MAXL = WPS == 2 ? 144 : long_maxl(NS);
    static constexpr int NTI = WPS == 2 ? 5 : long_window_tiles(NS);          // query tiles per wavefront
    static constexpr int NTJ = MAXL / kTile;
    static constexpr int CP = NS * MAXL + 20;
    static constexpr int NW = NS * WPS;
};

template <int NS, int WPS, bool ROW, bool MASK>
__device__ __forceinline__ void weight_long_body(float *lds, int b, int tile, int gwin, int wtiles, const float *__restrict__ X,
                                                 const float *__restrict__ Y, float *__restrict__ T,
                                                 int Cx, int H, int W, long xbs, long ybs) {
    using Cfg = LongWeightCfg<NS, WPS>;
    constexpr int NTI = Cfg::NTI, NTJ = Cfg::NTJ, CP = Cfg::CP, NW = Cfg::NW;
    float *xi = lds, *yi = lds + LG_KC * CP;
    const Branch br = make_branch(ROW, H, W);
    const int L = br.L, HW = H * W, S = H + W;
    const int g0 = tile * NS;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4;
    const int sw = wv % NS, part = wv / NS;                   // this wavefront's strip and (WPS = 2) query window
    const int g = g0 + sw;
    const bool active = g < br.G;
    const int gvalid = (br.G - g0 < NS) ? br.G - g0 : NS;
    const int i0 = (WPS == 2 ? part : gwin) * wtiles * kTile; // first query position of this (evened-out) window
    const int ntj = (L + kTile - 1) / kTile;

    const FBuf Xb = make_fbuf(X + (size_t)b * xbs, (size_t)Cx * HW * sizeof(float));
    const FBuf Yb = make_fbuf(Y + (size_t)b * ybs, (size_t)Cx * HW * sizeof(float));

    f32x4 acc[NTI][NTJ];
#pragma unroll
    for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int idx = tid; idx < 2 * LG_KC * CP; idx += kWave * NW) CCA_LDS_ST(&lds[idx], 0.f);
    __syncthreads();

    const int nchunks = (Cx + LG_KC - 1) / LG_KC;
    for (int n = 0; n < nchunks; ++n) {
        // 2 operands x 8 channels dealt round-robin to the wavefronts
#pragma unroll
        for (int pr = 0; pr < 2 * LG_KC / NW; ++pr) {
            const int pair = wv + pr * NW, op = pair / LG_KC, cc = pair % LG_KC;
            const int c = (n * LG_KC + cc < Cx) ? n * LG_KC + cc : Cx - 1;      // K padding is zeroed at fragment read
            long_dma_plane<NS, ROW>(op ? Yb : Xb, (op ? yi : xi) + cc * CP, c * HW * 4, lane, L, W, g0, gvalid);
        }
        __syncthreads();
        if (active) {
            auto compute = [&](auto nt_tag) {
                constexpr int NT = decltype(nt_tag)::value;   // query tiles of this window (compile time, see above)
#pragma unroll
                for (int ks = 0; ks < LG_KC / 4; ++ks) {
                    const bool kin = n * LG_KC + ks * 4 + lk < Cx;
                    const float *xs = xi + (ks * 4 + lk) * CP, *ys = yi + (ks * 4 + lk) * CP;
                    float a[NT];
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti) {
                        const int p = i0 + ti * kTile + ln;
                        const float v = CCA_LDS_LD(&xs[p < L ? (ROW ? sw * L + p : p * NS + sw) : 0]);
                        a[ti] = (kin && p < L) ? v : 0.f;
                    }
#pragma unroll
                    for (int tj = 0; tj < NTJ; ++tj)
                        if (tj < ntj) {
                            const int p = tj * kTile + ln;
                            const float v = CCA_LDS_LD(&ys[p < L ? (ROW ? sw * L + p : p * NS + sw) : 0]);
                            const float bb = p < L ? v : 0.f;
#pragma unroll
                            for (int ti = 0; ti < NT; ++ti) acc[ti][tj] = mfma_16x16x4(a[ti], bb, acc[ti][tj]);
                        }
                }
            };
            if (WPS == 2 || wtiles >= NTI)      compute(std::integral_constant<int, NTI>{});   // (WPS = 2: one body)
            else if (NTI > 4 && wtiles == 4)    compute(std::integral_constant<int, (NTI > 4 ? 4 : 1)>{});
            else if (NTI > 2 && wtiles == 3)    compute(std::integral_constant<int, (NTI > 2 ? 3 : 1)>{});
            else if (wtiles == 2)               compute(std::integral_constant<int, 2>{});
            else                                compute(std::integral_constant<int, 1>{});
        }
        __syncthreads();
    }

    if (!active) return;
    float *Tg = T + (size_t)b * HW * S + (size_t)g * br.as_g + br.a_off;
#pragma unroll
    for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj)
            if (tj < ntj) {
                const int j = tj * kTile + ln;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int iq = i0 + ti * kTile + 4 * lk + r;
                    if (ti < wtiles && iq < L && j < L) {
                        float val = acc[ti][tj][r];
                        if (MASK && !ROW && iq == j) val = -INFINITY;      // functions.py:11-12 (column self slot)
                        Tg[(size_t)iq * br.as_q + j] = val;
                    }
                }
            }
}

// one launch covers both branches: per image, column (tile, window) workgroups first, then the row ones
template <int NS, int WPS, bool MASK>
__global__ __launch_bounds__(kWave * NS * WPS) void weight_long_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                                       float *__restrict__ T, int Cx, int H, int W,
                                                                       int tiles_col, int win_col, int wt_col, int tiles_row,
                                                                       int win_row, int wt_row, long xbs, long ybs) {
    __shared__ float lds[2 * LG_KC * LongWeightCfg<NS, WPS>::CP];
    CCA_LDS_REGISTER(lds);
    const int ncol = tiles_col * win_col, per_image = ncol + tiles_row * win_row;
    const int id = xcd_logical_id(blockIdx.x, gridDim.x);
    const int b = id / per_image, t = id - b * per_image;
    if (t < ncol) weight_long_body<NS, WPS, false, MASK>(lds, b, t % tiles_col, t / tiles_col, wt_col, X, Y, T, Cx, H, W, xbs, ybs);
    else          weight_long_body<NS, WPS, true, MASK>(lds, b, (t - ncol) % tiles_row, (t - ncol) / tiles_row, wt_row, X, Y, T, Cx, H, W, xbs, ybs);
}

}  // namespace cca

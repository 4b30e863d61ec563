// cca_gemm.hpp -- the three GEMMs of the module's stacked 1x1 projections (functions.py:29,32,35 and their adjoints) as hand-written
// MFMA kernels on the three-plane bf16 operands of the split-bf16 x3 scheme: proj_gemm_kernel (forward, and -- batched, NCHW, with an
// addend -- the adjoint with respect to the input) and proj_wgrad_kernel (the adjoint with respect to the weight, below).
//
//     out[m][n] = sum_k A[m][k] * Wt[n][k] + bias[n]        A  (M, K) bf16, row stride lda   (x as three planes per pixel: K = 3 C)
//                                                           Wt (N, K) bf16, row stride ldw   (the packed weight rows [wh | wl | wh])
//                                                           out (M, ldo) fp32                (the pixel-major q | k | v the core reads)
//
// VERDICT r5 item 5c.  The stock bf16 -> fp32 GEMM runs this tall-skinny product (M = 75 272, N = 640, K = 1536) at 199-213 us
// and its bias epilogue costs more than a pass of its own (285-330 us with it, 260-269 us as product + in-place add,
// profiles/r06h_fwd_gemm_bias_ab.txt, r06n_fwd_gemm_ab.txt).  Here: both operands are K-contiguous, i.e. exactly the plane tiles of
// cca_gmap.hpp (T16 geometry: 1 KiB LDS-DMA pieces of 8 rows x 64 k, fragments = one ds_read_b128 per lane) and the contraction is
// gweight_kernel's with a large tile: workgroup = 256 x 128 outputs, 8 wavefronts of 64 x 64 (4 x 4 MFMA tiles of 16 x 16,
// v_mfma_f32_16x16x32_bf16: 32 MFMAs per wavefront and 64-wide k step against 16 fragment reads; two wavefronts per SIMD, so one's
// address / fill / wait instructions issue under the other's MFMAs), three LDS stages of 48 KiB.  The MFMA operands are SWAPPED
// (D^T = Wt . A^T) so that a lane ends with four consecutive n of one m: bias and addend start the accumulators and the results
// leave as 16-byte stores straight from them, 256 contiguous bytes per row and wavefront.  Workgroup ids are decoded XCD-contiguously
// with the tiles that share a tile of the streamed operand adjacent: their second to fifth read of it is an L2 hit.
//
// The loop is a register ping-pong: a k step is two half steps of 16 MFMAs; the 8 fragment reads of the NEXT half step are
// requested before the MFMAs of the current one (reads and waits outside the compiler's bookkeeping, cca_platform.hpp: across
// the back edge hipcc waits for ALL tracked reads), and the stage barrier sits between the two halves -- by then every wavefront
// holds its whole stage in registers, so the stage's slot is refilled at once and two stages (96 KiB) are in flight while one is
// multiplied.  The fill is six instructions per wavefront whose per-lane offsets are loop invariants (rows clamped to the matrix:
// tail rows are computed on a valid row's data and never stored), issued in the shadow of the second half's first MFMAs.
//
// Measured: 192-206 us with the bias (the stock product's time without it), dx 209-219 us with dy.  What bounds the k step
// (~1 us against 0.44-0.48 us of MFMA; DESIGN.md 3.6, profiles/r06n_gemm_elimination.txt, r06r_gemm_residency_and_pmc.txt) is
// inside the CU, not HBM: with both operands L2-resident the launch is only 10-15 % faster.  The suspect is the LDS path -- per
// step and CU 128 ds_read_b128 wave-instructions (512 LDS cycles) plus 48 KiB of LDS-DMA writes (alone: 0.456 us, ~1000 cycles)
// against 1024 MFMA cycles per SIMD -- but the model is incomplete: the variant that cut the DMA bytes by a third (declared plane
// structure, two tile rings) was no faster.  Also measured and dropped: an L2 prefetch wavefront, a register-staged
// fill (ds_write_b128), padded row strides, four wavefronts of 128 x 64, the ping-pong in the dW kernel.
#pragma once
#include "cca_gmap.hpp"

namespace cca {

constexpr int PG_BM = 256, PG_BN = 128, PG_BK = 64, PG_THREADS = 512, PG_WAVES = PG_THREADS / kWave;
constexpr int PG_APIECES = PG_BM / 8, PG_BPIECES = PG_BN / 8;                  // 1 KiB pieces of 8 rows x 64 k (bf16)
constexpr int PG_STAGE = (PG_APIECES + PG_BPIECES) * T16_PIECE;                // dwords per stage (A tile | B tile)
constexpr int PG_NBUF = 3;                                                     // 144 KiB
constexpr int PG_NPA = PG_APIECES / PG_WAVES, PG_NPB = PG_BPIECES / PG_WAVES, PG_NPW = PG_NPA + PG_NPB;   // fill instructions per wavefront and stage

// one launch: ``batches`` products out_b = A_b . Wt_b^T (+ bias) (+ add_b); a batch stride of 0 shares the operand
struct ProjGemmJob {
    const bf16_t *A, *Wt;
    const float *bias, *add;      // (N) or null; (batches, M, ldo) in the layout of ``out`` or null
    float *out;
    int M, N, K, lda, ldw, ldo, batches;
    long sa, sw, so;              // batch strides in elements (A, Wt, out / add)
};

template <bool KTAIL>
__global__ __launch_bounds__(PG_THREADS, 1) void proj_gemm_kernel(const ProjGemmJob job) {
    __shared__ __attribute__((aligned(16))) float lds[PG_NBUF * PG_STAGE];
    CCA_LDS_REGISTER(lds);
    const int M = job.M, N = job.N, K = job.K, lda = job.lda, ldw = job.ldw, ldo = job.ldo;
    const int ntm = (M + PG_BM - 1) / PG_BM, ntn = (N + PG_BN - 1) / PG_BN;
    const int lid = xcd_logical_id((int)blockIdx.x, (int)gridDim.x);
    const int bt = lid / (ntm * ntn), tl = lid - bt * (ntm * ntn);
    // the axis with FEWER tiles runs fastest: the workgroups that share a tile of the other (streamed) operand are neighbours
    const int m0 = (ntn <= ntm ? tl / ntn : tl % ntm) * PG_BM, n0 = (ntn <= ntm ? tl % ntn : tl / ntm) * PG_BN;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;                                       // this wavefront: rows wm * 64 .., columns wn * 64 ..
    const float *bias = job.bias;
    const FBuf Ab = make_fbuf(reinterpret_cast<const float *>(job.A + bt * job.sa), ((size_t)(M - 1) * lda + K) * 2);
    const FBuf Wb = make_fbuf(reinterpret_cast<const float *>(job.Wt + bt * job.sw), ((size_t)(N - 1) * ldw + K) * 2);
    const FBuf Ob = make_fbuf(job.out + bt * job.so, ((size_t)(M - 1) * ldo + N) * sizeof(float));
    const FBuf Cb = make_fbuf(job.add ? job.add + bt * job.so : job.out, ((size_t)(M - 1) * ldo + N) * sizeof(float));
    const int nk = (K + PG_BK - 1) / PG_BK;

    // fill: wavefront wv moves A pieces wv, wv + 8, .. and Wt pieces wv, wv + 8; lane = (row lane >> 3 of the piece, LDS chunk
    // slot lane & 7), which holds the row's 16-byte k chunk slot ^ row (t16_byte<false>)
    int offa[PG_NPA], offb[PG_NPB];
    {
        const int pr = lane >> 3, q = (lane & 7) ^ pr;
#pragma unroll
        for (int p = 0; p < PG_NPA; ++p) {
            const int r = m0 + 8 * (wv + PG_WAVES * p) + pr;
            offa[p] = ((r < M ? r : M - 1) * lda + 8 * q) * 2;
        }
#pragma unroll
        for (int p = 0; p < PG_NPB; ++p) {
            const int r = n0 + 8 * (wv + PG_WAVES * p) + pr;
            offb[p] = ((r < N ? r : N - 1) * ldw + 8 * q) * 2;
        }
    }
    const int kq = 8 * ((lane & 7) ^ (lane >> 3));                             // first k of this lane's chunk within a stage
    auto piece = [&](int it, int slot, int i) {                                // fill instruction i of PG_NPW (A pieces first)
        float *as = lds + slot * PG_STAGE + wv * T16_PIECE;
        // (KTAIL, a K that is no multiple of 64: chunks past the end of a row are fetched out of range = zeros)
        const bool dead = KTAIL && it * PG_BK + kq >= K;
        const int koff = it * PG_BK * 2;
        if (i < PG_NPA) fbuf_load_to_lds_x4_uncounted(Ab, as + PG_WAVES * i * T16_PIECE, dead ? kOobOffset : offa[i] + koff);
        else            fbuf_load_to_lds_x4_uncounted(Wb, as + (PG_APIECES + PG_WAVES * (i - PG_NPA)) * T16_PIECE,
                                                      dead ? kOobOffset : offb[i - PG_NPA] + koff);
    };
    auto issue = [&](int it, int slot) {
#pragma unroll
        for (int i = 0; i < PG_NPW; ++i) piece(it, slot, i);
    };
    // fragments of one half step kk (32 k): 8 consecutive k of one row = one 16-byte read; chunk 4 kk + lg of a row lies 64 kk bytes
    // from chunk lg, XORed (t16_byte<false>), the next 16 rows 2 KiB further
    const int fra = t16_byte<false>(wm * 64 + ln, 8 * lg), frb = t16_byte<false>(wn * 64 + ln, 8 * lg) + PG_APIECES * T16_PIECE * 4;
    auto read = [&](u32x4 (&bf)[4], u32x4 (&af)[4], int slot, int kk) {
        const char *st = reinterpret_cast<const char *>(lds + slot * PG_STAGE);
        const char *pb = st + (frb ^ (64 * kk)), *pa = st + (fra ^ (64 * kk));
        bf[0] = lds_read_x4_uncounted<0>(pb);     bf[1] = lds_read_x4_uncounted<2048>(pb);
        bf[2] = lds_read_x4_uncounted<4096>(pb);  bf[3] = lds_read_x4_uncounted<6144>(pb);
        af[0] = lds_read_x4_uncounted<0>(pa);     af[1] = lds_read_x4_uncounted<2048>(pa);
        af[2] = lds_read_x4_uncounted<4096>(pa);  af[3] = lds_read_x4_uncounted<6144>(pa);
    };

    // D^T[n][m]: lane (ln, lg) of tile (t, j) holds columns n0 + wn * 64 + 16 j + 4 lg .. + 3 of row m0 + wm * 64 + 16 t + ln.
    // The accumulators START from bias + addend (the stock GEMM's bias / beta = 1 epilogues cost more than a pass of their own):
    // the addend's 16-byte loads are in flight next to the first three fills and are consumed (added to the bias) before the loop.
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * lg;
        f32x4 b4;
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = (bias && n + q < N) ? bias[n + q] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][j] = b4;
    }
    f32x4 addv[4][4];
    if (job.add) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = m0 + wm * 64 + 16 * t + ln;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + 16 * j + 4 * lg;
                if (n + 3 < N) {
                    addv[t][j] = fbuf_load_x4(Cb, m < M ? (m * ldo + n) * 4 : kOobOffset, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) addv[t][j][q] = fbuf_load(Cb, (m < M && n + q < N) ? (m * ldo + n + q) * 4 : kOobOffset, 0);
                }
            }
        }
    }

    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 2) issue(2, 2);
    if (nk > 2)       barrier_dma_keep<2 * PG_NPW>();                         // stage 0 landed
    else if (nk > 1)  barrier_dma_keep<PG_NPW>();
    else              barrier_dma_keep<0>();
    if (job.add) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] += addv[t][j];
    }
    u32x4 b0[4], a0[4], b1[4], a1[4];
    read(b0, a0, 0, 0);
    int slot = 0;
    for (int it = 0; it < nk; ++it) {
        const int next = slot == PG_NBUF - 1 ? 0 : slot + 1;
        read(b1, a1, slot, 1);
        lds_wait_keep<8>(b0, a0);                  // the first half's fragments are here, the second half's on their way
        sched_fence();                              // (the order is the pipeline: hipcc would otherwise regroup reads and MFMAs)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = mfma_bf16_16x16x32(b0[j], a0[t], acc[t][j]);
        sched_fence();
        const bool fill = it + 3 < nk;
        if (it + 1 < nk) {
            // stage it + 1 landed and every wavefront holds stage `it` in registers: its slot takes stage it + 3 at once; the
            // fill of stage it + 2 (this wavefront's newest vector-memory operations) stays in flight
            if (it + 2 < nk) barrier_dma_keep<PG_NPW>();
            else             barrier_dma_keep<0>();
        }
        read(b0, a0, next, 0);                      // (after the last stage: eight reads of a dead slot that nobody uses)
        lds_wait_keep<8>(b1, a1);
        sched_fence();
        // second half; the fill instructions of stage it + 3 ride in the shadow of its first MFMAs
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[t][j] = mfma_bf16_16x16x32(b1[j], a1[t], acc[t][j]);
                if (4 * t + j < PG_NPW) {
                    if (fill) piece(it + 3, slot, 4 * t + j);
                    sched_fence();
                }
            }
        sched_fence();
        slot = next;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int m = m0 + wm * 64 + 16 * t + ln;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + 4 * lg;
            if (n + 3 < N) {
                fbuf_store_x4(Ob, acc[t][j], m < M ? (m * ldo + n) * 4 : kOobOffset, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < N) fbuf_store(Ob, acc[t][j][q], m < M ? (m * ldo + n + q) * 4 : kOobOffset, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The weight gradient of the stacked projection (the backward-weight of functions.py:29,32,35):
//
//     dW[n][c] = sum_r D[r][n] * X[r][c]        D (R, N) bf16 rows: dq | dk | dv planes [dh | dl | dh] of every pixel (row stride ldd)
//                                               X (R, C) bf16 rows: x planes [xh | xh | xl] of the same pixels     (row stride ldx)
//
// -- a contraction over ROWS (R = 3 B H W = 225 816 at (8,512,97,97)) into a 640 x 512 output: ten 128 x 256 output tiles, so the
// rows are cut into S slabs (S x 10 ~ one workgroup per CU) and workgroup (slab, tile) writes its partial sum to ``part`` (S, N, C);
// the host adds the S partials in a fixed order (deterministic, no atomics).  Both operands are contiguous ACROSS the contraction
// axis, i.e. they are cca_gmap.hpp's plane tiles as they stand: a stage = 64 rows = two T16 tiles of D (64 n each) and four of X
// (64 c each), 48 KiB moved by 6 LDS-DMA instructions per wavefront whose offsets advance by a constant; fragments = t16_frag
// (two transposing reads, K down the tile's positions).  Geometry and pipeline as proj_gemm_kernel: 8 wavefronts of 64 n x 64 c,
// three stages, the stage barrier between the two half steps.  (The stock route ran this as 24 batched GEMMs + a sum over the
// batch: 264-268 us, profiles/r06o_dx_gemm_ab.txt.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int PW_BN = 128, PW_BC = 256, PW_TILE = t16_size(PG_BK);                               // dwords per T16 tile of 64 rows
constexpr int PW_NT = PW_BN / 64 + PW_BC / 64;                                                  // T16 tiles per stage (= fills per wavefront)
static_assert(PW_NT * PW_TILE == PG_STAGE && PG_BK / 8 == PG_WAVES, "wgrad: one piece row of every tile per wavefront");

struct ProjWgradJob {
    const bf16_t *D, *X;
    float *part;                  // (S, N, C)
    int R, N, C, ldd, ldx, S, slab;   // slab = rows per slab, a multiple of 64
};

__global__ __launch_bounds__(PG_THREADS, 1) void proj_wgrad_kernel(const ProjWgradJob job) {
    __shared__ __attribute__((aligned(16))) float lds[PG_NBUF * PG_STAGE];
    CCA_LDS_REGISTER(lds);
    const int N = job.N, C = job.C, ldd = job.ldd, ldx = job.ldx;
    const int ntc = (C + PW_BC - 1) / PW_BC, ntn = (N + PW_BN - 1) / PW_BN;
    const int lid = xcd_logical_id((int)blockIdx.x, (int)gridDim.x);                 // the tiles of one slab are neighbours (one XCD: shared rows)
    const int sl = lid / (ntn * ntc), tl = lid - sl * (ntn * ntc);
    const int n0 = (tl / ntc) * PW_BN, c0 = (tl % ntc) * PW_BC;
    const int r0 = sl * job.slab, r1 = r0 + job.slab < job.R ? r0 + job.slab : job.R;
    const int nk = r1 > r0 ? (r1 - r0 + PG_BK - 1) / PG_BK : 0;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = uniform(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int wn = wv >> 2, wc = wv & 3;                                             // this wavefront: n columns 64 wn .., c columns 64 wc ..
    const FBuf Db = make_fbuf(reinterpret_cast<const float *>(job.D), ((size_t)(job.R - 1) * ldd + N) * 2);
    const FBuf Xb = make_fbuf(reinterpret_cast<const float *>(job.X), ((size_t)(job.R - 1) * ldx + C) * 2);
    const FBuf Ob = make_fbuf(job.part + (size_t)sl * N * C, (size_t)N * C * sizeof(float));

    // fill: wavefront wv moves piece wv (rows 8 wv .. + 7 of the stage) of all six tiles; lane = (row lane >> 3, LDS chunk slot
    // lane & 7), which holds the row's 16-byte channel chunk q (t16_dma_piece, FLIP layout)
    const int pr = lane >> 3, q = (lane & 7) ^ pr ^ ((wv & 1) << 2), rs = 8 * wv + pr;       // row of this lane within a stage
    int off[PW_NT];
#pragma unroll
    for (int t = 0; t < PW_NT; ++t) {
        const bool isd = t < PW_BN / 64;
        const int ch = (isd ? n0 + 64 * t : c0 + 64 * (t - PW_BN / 64)) + 8 * q;
        off[t] = ch < (isd ? N : C) ? ((r0 + rs) * (isd ? ldd : ldx) + ch) * 2 : kOobOffset;      // (columns past the matrix: zeros)
    }
    auto issue = [&](int it, int slot) {
        float *st = lds + slot * PG_STAGE + wv * T16_PIECE;
        const bool dead = r0 + it * PG_BK + rs >= r1;                                            // rows past the slab: zeros
#pragma unroll
        for (int t = 0; t < PW_NT; ++t) {
            const bool isd = t < PW_BN / 64;
            const int o = (dead || off[t] == kOobOffset) ? kOobOffset : off[t] + it * PG_BK * (isd ? ldd : ldx) * 2;
            fbuf_load_to_lds_x4_uncounted(isd ? Db : Xb, st + t * PW_TILE, o);
        }
    };

    // lane (ln, lg) of tile (a, b) holds c = c0 + 64 wc + 16 b + 4 lg .. + 3 of n = n0 + 64 wn + 16 a + ln
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 2) issue(2, 2);
    if (nk > 2)       barrier_dma_keep<2 * PW_NT>();                                 // stage 0 landed
    else if (nk > 1)  barrier_dma_keep<PW_NT>();
    else              barrier_dma_keep<0>();
    int slot = 0;
    for (int it = 0; it < nk; ++it) {
        const float *dt = lds + slot * PG_STAGE + wn * PW_TILE, *xt = lds + slot * PG_STAGE + (PW_BN / 64 + wc) * PW_TILE;
        u32x4 df[4], xf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { df[a] = t16_frag(dt, 0, a, lane); xf[a] = t16_frag(xt, 0, a, lane); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = mfma_bf16_16x16x32(xf[b], df[a], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) { df[a] = t16_frag(dt, 1, a, lane); xf[a] = t16_frag(xt, 1, a, lane); }
        if (it + 1 < nk) {
            // stage it + 1 landed and every wavefront holds the rest of stage `it` in registers (the barrier drains its LDS reads):
            // the slot takes stage it + 3 at once, the fill of stage it + 2 stays in flight
            if (it + 2 < nk) barrier_dma_keep<PW_NT>();
            else             barrier_dma_keep<0>();
            if (it + 3 < nk) issue(it + 3, slot);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = mfma_bf16_16x16x32(xf[b], df[a], acc[a][b]);
        slot = slot == PG_NBUF - 1 ? 0 : slot + 1;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = n0 + 64 * wn + 16 * a + ln;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = c0 + 64 * wc + 16 * b + 4 * lg;
            if (c + 3 < C) {
                fbuf_store_x4(Ob, acc[a][b], n < N ? (n * C + c) * 4 : kOobOffset, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) fbuf_store(Ob, acc[a][b][e], n < N ? (n * C + c + e) * 4 : kOobOffset, 0);
            }
        }
    }
}

}  // namespace cca

// cca_api.hip -- extern "C" entry points of libccnet_cca.so (see include/ccnet_cca.h).
//
// Built for the device with   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -I.   (__graft_entry__.build()).
// The CPU test-suite compiles this same file with the host compiler against tests/emu/cca_platform.hpp (first on
// its include path) to execute the kernels in a SIMT emulator; that build is test-only.
#include "../../include/ccnet_cca.h"

#include "cca_common.hpp"
#include "cca_direct.hpp"
#include "cca_gmap.hpp"
#include "cca_gemm.hpp"
#include "cca_map.hpp"
#include "cca_long.hpp"
#include "cca_softmax.hpp"
#include "cca_weight.hpp"
#include "cca_probe.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <string>

namespace {

// Process-wide MODE words (implementation family, arithmetic, profiling branch mask).  They are the library's only
// state: plain atomics, read once at the top of an entry point, so concurrent callers never see a torn value; a
// setter racing with a call in flight on another thread affects either that call or the next, never half of one.
thread_local std::string g_last_error = "";
std::atomic<int> g_impl{CCNET_IMPL_AUTO};
std::atomic<int> g_branch_mask{CCNET_BRANCH_BOTH};     // profiling aid: which branch launches are issued
// arithmetic of the map kernels: 0 = exact f32 MFMA everywhere, 1 = split-bf16 x3 in the ROW launches only (default),
// 2 = split-bf16 x3 in both launches.  Measured on MI355X (profiles/): the row launches gain ~20 % (their traffic is
// fully coalesced, so the 5x cheaper MFMA phase shows), the column launches gain nothing (they are bound by the L2
// request rate of their 32-byte segments).  Only strips 97..100 long have a split-bf16 kernel.
std::atomic<int> g_map_bf16{1};
// The weight kernels of this family (ca_forward, and ca_map_backward's dA) run exact f32.  (Rounds 1-3 shipped a packed split-bf16
// dA variant -- 135 us against 250 us at the headline shape -- whose 196 stationary accumulators + packing temporaries spilled 126
// VGPRs; round 4 retired it: the module's default fp32 routes run the split-plane kernels of cca_gmap.hpp, this family is the
// reference-shaped exact one: VERDICT r3 item 8.)

// development / A-B options (ccnet_cca_set_option): "planes_ring" 0 = gmap_kernel (two tiles, output image in LDS) for every
// split-plane pass, 1 = the passes with a pixel-major output run gmap3_kernel (three-tile ring, stores from the accumulators,
// two workgroups per CU), 2 = as 1 with the column passes on the two-tile / three-workgroups-per-CU form
std::atomic<int> g_planes_ring{2};
// "planes_stream" 1 (default) = the split-plane dA contraction runs the persistent gweight_stream_kernel, 0 = gweight_kernel
std::atomic<int> g_planes_stream{1};
// "planes_overlap": the dv passes of the pixel-major / split-plane backwards run on the library's side stream; 2 = from before
// dA on, 1 = next to softmax-backward and dq | dk only, 0 = everything on the caller's stream; -1 (default) = what measured best
// per family: 2 on split planes (0.821 -> 0.776 -> 0.760 ms at the headline shape), 1 on the bf16 / fp32 pixel-major entries
// (configs[4] bf16: 1.917 -> 1.903, but 1.959 with 2: its one-workgroup-per-strip dA shares the CUs badly)
std::atomic<int> g_planes_overlap{-1};
// "planes_xcd" 1 (default): the final NCHW row pass of the split-plane forward decodes its strips from an XCD-aware id (consecutive
// rows of an image on ONE XCD): the 388-byte NCHW rows of x / y share every boundary line with their neighbour row
std::atomic<int> g_planes_xcd{1};
// "da_stages": ring stages of the persistent dA kernel (plane-free form).  Three fill the CU's LDS (159,744 B): the dv column pass on
// the side stream could not place a single workgroup next to it and in fact WAITED for the dA workgroups to exit (kernel-trace
// timeline: its 69 us of work spanned 223 us).  With two stages (106,496 B) one 53,248-byte column workgroup fits per CU and the two
// launches really run together: backward 0.451 -> 0.404 ms, step 0.727-0.749 -> 0.683-0.692 ms in the same run
// (profiles/r04lg_ab_dA_two_stages.txt); alone the two-stage kernel is no slower (151.6 vs 159.8 us).
std::atomic<int> g_da_stages{2};
// "bf16_partial" 1 (default): the column -> row partial of the bf16 pixel-major family (aggregation and dv) is bf16, not fp32
std::atomic<int> g_bf16_partial{1};
std::atomic<int> g_dqdk_wpc3{1};            // "dqdk_wpc3": ca_backward of the fp32 routes at C/8 <= 64 on the three-workgroups-per-CU form of gmap_kernel
std::atomic<int> g_energy_tail{1};          // the fp32 energies launch cuts the strips beyond its whole rounds into tile-row parts
// "dqdk_exact" 1 (default): ca_backward of every fp32 pixel-major / split-plane route multiplies as SIX bf16 terms of a three-way
// split (cca::bf16_split8x3; gmap_kernel, SIX): fp32-equivalent products at any logit scale, +3..4 us per launch at the headline
// shape.  0 = three terms (split-bf16 x3: ~1.2e-5 of the gradient's magnitude; leaves the absolute 1e-3 bar at ~2 x the default
// logit scale).  (Rounds 4-5: 1 = exact-f32 MFMA, +25 us per launch; 2 = x3 + a statistic + two gated exact launches, 12 us of
// every step for launches that exit at once and a step time that depended on the data.  Both removed in round 6.)
std::atomic<int> g_dqdk_exact{1};

int fail(int code, const char *what) {
    char buf[256];
    snprintf(buf, sizeof(buf), "ccnet_cca: %s (code %d)", what, code);
    g_last_error = buf;
    return code;
}

// CCA_LAUNCH clears the sticky error of earlier, unrelated HIP calls before launching, so what is read here
// belongs to the launch just issued
int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "ccnet_cca: launch of %s failed: %s", what, hipGetErrorString(e));
        g_last_error = buf;
        return (int)e;
    }
    return 0;
}

int check_shape(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(CCNET_E_BADSHAPE, "non-positive dimension");
    if (B > 65535) return fail(CCNET_E_BADSHAPE, "batch exceeds the grid limit 65535");
    const double per_image = (double)H * W * (H + W);
    if (per_image >= 536870912.0 || (double)C * H * W >= 536870912.0)
        return fail(CCNET_E_BADSHAPE, "per-image tensor exceeds 2^29 elements (32-bit byte offsets)");
    return 0;
}

// the fused entry points always compute both branches: a profiling mask left behind by a tool must not turn them
// into silently wrong results
int require_both_branches(const char *what) {
    if (g_branch_mask.load() != CCNET_BRANCH_BOTH) {
        static thread_local std::string msg;
        msg = std::string(what) + ": a profiling branch mask is set (option branch_mask); restore 3 first";
        return fail(CCNET_E_BADFLAGS, msg.c_str());
    }
    return 0;
}

// 1 = stationary MFMA strip kernels (strips <= 100), 2 = windowed MFMA strip kernels for long strips (<= 320),
// 0 = direct kernels, <0 = error
int pick_impl(int H, int W) {
    const int longest = H > W ? H : W, impl = g_impl.load();
    if (impl == CCNET_IMPL_DIRECT) return 0;
    if (longest <= cca::kMaxStrip) return 1;
    if (longest <= cca::kLongMaxStrip) return 2;
    if (impl == CCNET_IMPL_MFMA) return fail(CCNET_E_BADSHAPE, "CCNET_IMPL_MFMA forced but max(H,W) > 320");
    return 0;
}

unsigned direct_grid(size_t total) {
    size_t blocks = (total + cca::D_BLOCK - 1) / cca::D_BLOCK;
    const size_t cap = 256 * 32;
    return (unsigned)(blocks < cap ? blocks : cap);
}

bool map_bf16(int H, int W, bool row) {
    const int lo = H < W ? H : W, hi = H < W ? W : H;
    const int mode = g_map_bf16.load();
    return (mode == 2 || (mode == 1 && row)) && lo >= 96 && hi <= cca::kMaxStrip;
}

// number of CUs of the CURRENT device (the channel splits are balanced for it; MI355X: 256), cached per device
int num_cus() {
    static std::atomic<int> cache[64];
    const int dev = cca_current_device();
    if (dev < 0 || dev >= 64) return 256;
    int v = cache[dev].load();
    if (v <= 0) {
        v = cca_current_device_cus();
        if (v <= 0) v = 256;
        cache[dev].store(v);
    }
    return v;
}

// Channel split of the map kernels.  A workgroup (one per CU: its LDS images fill the CU) pays a fixed
// prologue (loading the stationary attention blocks, worth about kPrologueChunks chunks) and then
// chunks_per_block chunks; the grid runs in ceil(workgroups / CUs) waves.  Pick the split that minimises
// waves * (chunks_per_block + prologue).
void map_grid(int ns, int B, int C, int G, dim3 &grid, int &chunks_per_block, int &tiles, int &cs, int problems = 1) {
    tiles = (G + ns - 1) / ns;
    const int nchunks = (C + cca::M_MC - 1) / cca::M_MC;
    const int base = B * tiles, cus = num_cus();
    const double prologue = 3.0;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= nchunks; ++s) {
        const int cpb = (nchunks + s - 1) / s;
        const int real_s = (nchunks + cpb - 1) / cpb;
        const int waves = (base * real_s * problems + cus - 1) / cus;
        const double cost = waves * (cpb + prologue);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = real_s; }
    }
    cs = best;
    chunks_per_block = (nchunks + cs - 1) / cs;
    cs = (nchunks + chunks_per_block - 1) / chunks_per_block;
    grid = dim3((unsigned)(tiles * cs * B * problems));   // 1-D: the kernel decodes an XCD-aware logical id
}

// out = alpha * (column sums + row sums) + resid, both branches, strip kernels
template <int NS, bool TRANS, bool BFC, bool BFR>
int launch_map_pair_ns(const float *T, const float *F, const float *resid, const float *gamma, float *out,
                       int B, int C, int H, int W, ccnet_stream_t stream, const char *what,
                       long fbs, long rbs, long obs) {
    dim3 grid;
    int cpb, tiles, cs;
    const int mask = g_branch_mask.load();
    if (mask & CCNET_BRANCH_COL) {
        map_grid(NS, B, C, /*G=*/W, grid, cpb, tiles, cs);
        if (resid) {
            if (TRANS) return fail(CCNET_E_BADFLAGS, "residual epilogue only exists for the forward aggregation");
            CCA_LAUNCH((cca::map_strip_kernel<NS, false, false, cca::EPI_COL_RESID, BFC>), grid, dim3(cca::kWave * NS),
                       stream, T, F, resid, gamma, out, C, H, W, cpb, tiles, cs, fbs, rbs, obs);
        } else {
            CCA_LAUNCH((cca::map_strip_kernel<NS, false, TRANS, cca::EPI_COL, BFC>), grid, dim3(cca::kWave * NS), stream,
                       T, F, resid, gamma, out, C, H, W, cpb, tiles, cs, fbs, rbs, obs);
        }
        if (int e = launch_status(what)) return e;
    }
    if (mask & CCNET_BRANCH_ROW) {
        map_grid(NS, B, C, /*G=*/H, grid, cpb, tiles, cs);
        CCA_LAUNCH((cca::map_strip_kernel<NS, true, TRANS, cca::EPI_ROW, BFR>), grid, dim3(cca::kWave * NS), stream,
                   T, F, (const float *)nullptr, gamma, out, C, H, W, cpb, tiles, cs, fbs, rbs, obs);
        return launch_status(what);
    }
    return 0;
}

// dq (non-transposed, F0 = k) and dk (transposed, F1 = q) in one launch per branch.  These launches have only
// C/8 channels (4 chunks per workgroup at the headline shape) and are bound by the prologue + the matrix pipe; BF
// selects the split-bf16 x3 arithmetic for both branches (isolated stage 94 -> 76 us; opt-in, see ca_backward_impl).
template <int NS, bool BF>
int launch_map_dual_ns(const float *T, const float *F0, float *out0, const float *F1, float *out1,
                       int B, int C, int H, int W, ccnet_stream_t stream, const char *what,
                       long fbs0, long obs0, long fbs1, long obs1) {
    dim3 grid;
    int cpb, tiles, cs;
    const int mask = g_branch_mask.load();
    if (mask & CCNET_BRANCH_COL) {
        map_grid(NS, B, C, /*G=*/W, grid, cpb, tiles, cs, 2);
        CCA_LAUNCH((cca::map_strip_dual_kernel<NS, false, cca::EPI_COL, BF>), grid, dim3(cca::kWave * NS), stream,
                   T, F0, out0, F1, out1, (const float *)nullptr, C, H, W, cpb, tiles, cs, fbs0, obs0, fbs1, obs1);
        if (int e = launch_status(what)) return e;
    }
    if (mask & CCNET_BRANCH_ROW) {
        map_grid(NS, B, C, /*G=*/H, grid, cpb, tiles, cs, 2);
        CCA_LAUNCH((cca::map_strip_dual_kernel<NS, true, cca::EPI_ROW, BF>), grid, dim3(cca::kWave * NS), stream,
                   T, F0, out0, F1, out1, (const float *)nullptr, C, H, W, cpb, tiles, cs, fbs0, obs0, fbs1, obs1);
        return launch_status(what);
    }
    return 0;
}

template <bool TRANS>
int launch_map_pair(const float *T, const float *F, const float *resid, const float *gamma, float *out,
                    int B, int C, int H, int W, ccnet_stream_t stream, const char *what,
                    long fbs, long rbs, long obs) {
    const bool bc = map_bf16(H, W, false), br = map_bf16(H, W, true);
    if (bc && br) return launch_map_pair_ns<8, TRANS, true, true>(T, F, resid, gamma, out, B, C, H, W, stream, what, fbs, rbs, obs);
    if (br)       return launch_map_pair_ns<8, TRANS, false, true>(T, F, resid, gamma, out, B, C, H, W, stream, what, fbs, rbs, obs);
    return launch_map_pair_ns<8, TRANS, false, false>(T, F, resid, gamma, out, B, C, H, W, stream, what, fbs, rbs, obs);
}

// K split of the weight-type contractions for SMALL BATCHES (reference recipe: 1-2 images per GPU, README.md:97,
// engine.py:88).  One image has 26 strip tiles, so at B = 1 the weight kernels would run on 26 of 256 CUs, each
// walking all channels: instead the channels are cut into n ranges, every (tile, range) is a workgroup, range s
// writes partial slab s (slab 0 = the output tensor, slabs 1.. in the caller's workspace) and the softmax kernel that
// consumes the tensor adds the slabs in a fixed order -- deterministic, no atomics.
struct KSplit {
    int n = 1;              // number of channel ranges (1 = no split)
    int cps = 0;            // 8-channel chunks per range
    float *extra = nullptr; // slabs 1 .. n-1
    long stride = 0;        // elements between slabs
};
// how many ranges a K-channel contraction is cut into at this shape (strip-stationary kernels only)
int ksplit_count(int B, int K, int H, int W) {
    const int longest = H > W ? H : W;
    if (g_impl.load() == CCNET_IMPL_DIRECT || longest > cca::kMaxStrip) return 1;
    const int per_image = (W + 7) / 8 + (H + 7) / 8, nchunks = (K + cca::W_KC - 1) / cca::W_KC;
    int n = num_cus() / (B * per_image);
    if (n > nchunks / 2) n = nchunks / 2;               // at least two chunks per range
    return n < 1 ? 1 : n;
}
size_t ksplit_bytes(int B, int K, int H, int W) {
    return (size_t)(ksplit_count(B, K, H, W) - 1) * B * H * W * (H + W) * sizeof(float);
}
// plan for a contraction whose extra slabs may live at ws[0 .. bytes): falls back to fewer ranges when they do not fit
KSplit ksplit_plan(int B, int K, int H, int W, void *ws, size_t bytes) {
    KSplit ks;
    const size_t slab = (size_t)B * H * W * (H + W) * sizeof(float);
    int n = ksplit_count(B, K, H, W);
    if (!ws) n = 1;
    while (n > 1 && (size_t)(n - 1) * slab > bytes) --n;
    const int nchunks = (K + cca::W_KC - 1) / cca::W_KC;
    ks.cps = (nchunks + n - 1) / n;
    ks.n = (nchunks + ks.cps - 1) / ks.cps;
    ks.extra = static_cast<float *>(ws);
    ks.stride = (long)(slab / sizeof(float));
    return ks;
}

// both branches in ONE launch (column workgroups first, then row workgroups)
template <int NS, bool MASK, bool BF>
int launch_weight_ns(const float *X, const float *Y, float *T, int B, int Cx, int H, int W,
                     ccnet_stream_t stream, const char *what, long xbs, long ybs, const KSplit &ks) {
    const int mask = g_branch_mask.load();
    const int tc = (mask & CCNET_BRANCH_COL) ? (W + NS - 1) / NS : 0;
    const int tr = (mask & CCNET_BRANCH_ROW) ? (H + NS - 1) / NS : 0;
    const int nchunks = (Cx + cca::W_KC - 1) / cca::W_KC;
    if (ks.n > 1)
        CCA_LAUNCH((cca::weight_strip_kernel<NS, MASK, BF, true>), dim3((tc + tr) * B * ks.n), dim3(cca::kWave * NS), stream,
                   X, Y, T, Cx, H, W, tc, tr, xbs, ybs, ks.n, ks.cps, ks.extra, ks.stride);
    else
        CCA_LAUNCH((cca::weight_strip_kernel<NS, MASK, BF, false>), dim3((tc + tr) * B), dim3(cca::kWave * NS), stream,
                   X, Y, T, Cx, H, W, tc, tr, xbs, ybs, 1, nchunks, (float *)nullptr, 0L);
    return launch_status(what);
}

template <bool MASK>
int launch_weight_pair(const float *X, const float *Y, float *T, int B, int Cx, int H, int W,
                       ccnet_stream_t stream, const char *what, long xbs, long ybs, const KSplit &ks = KSplit()) {
    return launch_weight_ns<8, MASK, false>(X, Y, T, B, Cx, H, W, stream, what, xbs, ybs, ks);
}

int softmax_forward(const float *E, float *A, int B, int H, int W, ccnet_stream_t stream, const KSplit &ks = KSplit()) {
    const int npix = B * H * W, S = H + W;
    const dim3 grid((npix + cca::SM_WAVES - 1) / cca::SM_WAVES), block(cca::SM_BLOCK);
    const float *extra = ks.extra;
    if (S <= 256)      CCA_LAUNCH((cca::softmax_fwd_kernel<4>), grid, block, stream, E, A, npix, S, ks.n, extra, ks.stride);
    else if (S <= 512) CCA_LAUNCH((cca::softmax_fwd_kernel<8>), grid, block, stream, E, A, npix, S, ks.n, extra, ks.stride);
    else               CCA_LAUNCH(cca::softmax_fwd_generic_kernel, grid, block, stream, E, A, npix, S, ks.n, extra, ks.stride);
    return launch_status("softmax_fwd");
}


// ---- long strips (101 .. 320): windowed MFMA strip kernels of cca_long.hpp --------------------------------------
// grid = images x channel splits x windows x strip tiles; the channel split only has to top the grid up to a few
// workgroups per CU (windows and the 4- or 2-strip tiles already make many)
// windows of a strip: as few as the register file allows, evened out (129 positions = 9 tiles -> 3 + 3 + 3)
template <int NS>
void long_windows(int L, int &nwin, int &wtiles) {
    const int ntiles = (L + cca::kTile - 1) / cca::kTile, cap = cca::long_window_tiles(NS);
    nwin = (ntiles + cap - 1) / cap;
    wtiles = (ntiles + nwin - 1) / nwin;
}

template <int NS, int WPS>
void long_map_grid(int B, int C, int L, int G, dim3 &grid, int &cpb, int &tiles, int &cs, int &nwin, int &wtiles) {
    tiles = (G + NS - 1) / NS;
    if (WPS == 2) {                 // the two windows of a strip are the two wavefronts that own it
        nwin = 1;
        wtiles = ((L + cca::kTile - 1) / cca::kTile + 1) / 2;
    } else {
        long_windows<NS>(L, nwin, wtiles);
    }
    const int nchunks = (C + cca::LG_MC - 1) / cca::LG_MC;
    const long base = (long)B * tiles * nwin, want = (WPS == 2 ? 2L : 4L) * num_cus();
    int s = (int)((want + base - 1) / base);
    if (s < 1) s = 1;
    if (s > nchunks) s = nchunks;
    cpb = (nchunks + s - 1) / s;
    cs = (nchunks + cpb - 1) / cpb;
    grid = dim3((unsigned)(base * cs));
}

// one launch of a pair: the column launch (row == false) or the row launch of configuration <NS, WPS>
template <int NS, int WPS, bool TRANS>
int launch_long_map_one(bool row, const float *T, const float *F, const float *resid, const float *gamma, float *out,
                        int B, int C, int H, int W, ccnet_stream_t stream, const char *what,
                        long fbs, long rbs, long obs) {
    dim3 grid;
    int cpb, tiles, cs, nwin, wt;
    const dim3 block(cca::kWave * NS * WPS);
    if (resid && TRANS) return fail(CCNET_E_BADFLAGS, "residual epilogue only exists for the forward aggregation");
    if (!row) {
        // the residual is added by the ROW launch of this family (whole-row addend loads); a column-only run (profiling
        // mask) keeps it here
        const bool resid_here = resid && !(g_branch_mask.load() & CCNET_BRANCH_ROW);
        long_map_grid<NS, WPS>(B, C, /*L=*/H, /*G=*/W, grid, cpb, tiles, cs, nwin, wt);
        if (resid_here) {
            CCA_LAUNCH((cca::map_long_kernel<NS, WPS, false, false, cca::EPI_COL_RESID>), grid, block, stream,
                       T, F, resid, gamma, out, C, H, W, cpb, tiles, cs, nwin, wt, fbs, rbs, obs);
        } else {
            CCA_LAUNCH((cca::map_long_kernel<NS, WPS, false, TRANS, cca::EPI_COL>), grid, block, stream,
                       T, F, (const float *)nullptr, gamma, out, C, H, W, cpb, tiles, cs, nwin, wt, fbs, rbs, obs);
        }
    } else {
        long_map_grid<NS, WPS>(B, C, /*L=*/W, /*G=*/H, grid, cpb, tiles, cs, nwin, wt);
        if (resid && !TRANS) {
            CCA_LAUNCH((cca::map_long_kernel<NS, WPS, true, false, cca::EPI_ROW_RESID>), grid, block, stream,
                       T, F, resid, gamma, out, C, H, W, cpb, tiles, cs, nwin, wt, fbs, rbs, obs);
        } else {
            CCA_LAUNCH((cca::map_long_kernel<NS, WPS, true, TRANS, cca::EPI_ROW>), grid, block, stream,
                       T, F, (const float *)nullptr, gamma, out, C, H, W, cpb, tiles, cs, nwin, wt, fbs, rbs, obs);
        }
    }
    return launch_status(what);
}

// the two launches of a pair choose their configuration independently from the length of THEIR strips (H = 129,
// W = 257: column strips 129 long -> 4 strips x 2 wavefronts, row strips 257 long -> 2 strips): the partial sums are
// in the natural layout, so they combine freely
template <bool TRANS>
int launch_long_map_pair(const float *T, const float *F, const float *resid, const float *gamma, float *out,
                         int B, int C, int H, int W, ccnet_stream_t stream, const char *what,
                         long fbs, long rbs, long obs) {
    const bool two_waves = true;
    const int mask = g_branch_mask.load();
    for (int row = 0; row < 2; ++row) {
        if (!(mask & (row ? CCNET_BRANCH_ROW : CCNET_BRANCH_COL))) continue;
        const int L = row ? W : H;
        int e;
        if (two_waves && L <= cca::LongMapCfg<4, 2>::MAXL)
            e = launch_long_map_one<4, 2, TRANS>(row, T, F, resid, gamma, out, B, C, H, W, stream, what, fbs, rbs, obs);
        else if (L <= cca::long_maxl(4))
            e = launch_long_map_one<4, 1, TRANS>(row, T, F, resid, gamma, out, B, C, H, W, stream, what, fbs, rbs, obs);
        else
            e = launch_long_map_one<2, 1, TRANS>(row, T, F, resid, gamma, out, B, C, H, W, stream, what, fbs, rbs, obs);
        if (e) return e;
    }
    return 0;
}

template <int NS, int WPS, bool MASK>
int launch_long_weight_ns(const float *X, const float *Y, float *T, int B, int Cx, int H, int W,
                          ccnet_stream_t stream, const char *what, long xbs, long ybs, bool do_col, bool do_row) {
    int wc, wtc, wr, wtr;
    if (WPS == 2) {                 // the two query windows of a strip are the two wavefronts that own it
        wc = wr = 1;
        wtc = ((H + cca::kTile - 1) / cca::kTile + 1) / 2;
        wtr = ((W + cca::kTile - 1) / cca::kTile + 1) / 2;
    } else {
        long_windows<NS>(H, wc, wtc);
        long_windows<NS>(W, wr, wtr);
    }
    const int tc = do_col ? (W + NS - 1) / NS : 0, tr = do_row ? (H + NS - 1) / NS : 0;
    if (tc * wc + tr * wr == 0) return 0;
    CCA_LAUNCH((cca::weight_long_kernel<NS, WPS, MASK>), dim3((unsigned)((tc * wc + tr * wr) * B)),
               dim3(cca::kWave * NS * WPS), stream, X, Y, T, Cx, H, W, tc, wc, wtc, tr, wr, wtr, xbs, ybs);
    return launch_status(what);
}

// per branch: strips up to 144 -> 4 strips x 2 wavefronts (one pass over the channels), up to 160 -> 4 strips with
// query windows, up to 320 -> 2 strips with query windows
template <bool MASK>
int launch_long_weight_pair(const float *X, const float *Y, float *T, int B, int Cx, int H, int W,
                            ccnet_stream_t stream, const char *what, long xbs, long ybs) {
    const bool two_waves = true;
    const int mask = g_branch_mask.load();
    const bool col = mask & CCNET_BRANCH_COL, row = mask & CCNET_BRANCH_ROW;
    const int lim2 = two_waves ? cca::LongWeightCfg<4, 2>::MAXL : 0, lim4 = cca::long_maxl(4);
    const bool c2 = H <= lim2, r2 = W <= lim2, c4 = !c2 && H <= lim4, r4 = !r2 && W <= lim4;
    if (int e = launch_long_weight_ns<4, 2, MASK>(X, Y, T, B, Cx, H, W, stream, what, xbs, ybs, col && c2, row && r2)) return e;
    if (int e = launch_long_weight_ns<4, 1, MASK>(X, Y, T, B, Cx, H, W, stream, what, xbs, ybs, col && c4, row && r4)) return e;
    return launch_long_weight_ns<2, 1, MASK>(X, Y, T, B, Cx, H, W, stream, what, xbs, ybs, col && !c2 && !c4, row && !r2 && !r4);
}

// ---- strided internals: every feature tensor is (B, C, H, W) with a dense (C, H, W) image per batch and a
// ---- caller-given batch stride in elements (dense = C*H*W), so q/k/v may be channel slices of one projection.
int ca_forward_impl(const float *q, const float *k, float *out, int B, int Cq, int H, int W, int flags,
                    ccnet_stream_t stream, long qbs, long kbs, KSplit ks = KSplit()) {
    if (int e = check_shape(B, Cq, H, W)) return e;
    if (!q || !k || !out) return fail(CCNET_E_NULLPTR, "ca_forward: null tensor");
    if (flags != CCNET_CA_ENERGY && flags != CCNET_CA_SOFTMAX) return fail(CCNET_E_BADFLAGS, "ca_forward: bad flags");
    const int impl = pick_impl(H, W);
    if (impl < 0) return impl;
    if (impl != 1 || flags != CCNET_CA_SOFTMAX) ks = KSplit();      // slabs are summed by the softmax kernel only
    if (impl == 1) {
        if (int e = launch_weight_pair<true>(q, k, out, B, Cq, H, W, stream, "ca_forward", qbs, kbs, ks)) return e;
    } else if (impl == 2) {
        if (int e = launch_long_weight_pair<true>(q, k, out, B, Cq, H, W, stream, "ca_forward(long)", qbs, kbs)) return e;
    } else {
        const size_t total = (size_t)B * H * W * (H + W);
        CCA_LAUNCH((cca::direct_weight_kernel<true>), dim3(direct_grid(total)), dim3(cca::D_BLOCK), stream,
                   q, k, out, Cq, H, W, total, qbs, kbs);
        if (int e = launch_status("ca_forward(direct)")) return e;
    }
    if (flags == CCNET_CA_SOFTMAX) return softmax_forward(out, out, B, H, W, stream, ks);
    return 0;
}

int ca_backward_impl(const float *dE, const float *q, const float *k, float *dq, float *dk,
                     int B, int Cq, int H, int W, ccnet_stream_t stream, long qbs, long kbs, long dqbs, long dkbs) {
    if (int e = check_shape(B, Cq, H, W)) return e;
    if (!dE || !q || !k || !dq || !dk) return fail(CCNET_E_NULLPTR, "ca_backward: null tensor");
    const int impl = pick_impl(H, W);
    if (impl < 0) return impl;
    if (impl == 1) {
        // dq (non-transposed) and dk (transposed) share one launch per branch; exact f32 (a split-bf16 variant cost accuracy
        // where it is scarcest -- 6-8e-4 max-abs on dq / dk -- bought nothing inside the step, and was retired in round 4)
        return launch_map_dual_ns<8, false>(dE, k, dq, q, dk, B, Cq, H, W, stream, "ca_backward(dq,dk)", kbs, dqbs, qbs, dkbs);
    }
    if (impl == 2) {
        if (int e = launch_long_map_pair<false>(dE, k, nullptr, nullptr, dq, B, Cq, H, W, stream, "ca_backward(dq,long)", kbs, 0, dqbs))
            return e;
        return launch_long_map_pair<true>(dE, q, nullptr, nullptr, dk, B, Cq, H, W, stream, "ca_backward(dk,long)", qbs, 0, dkbs);
    }
    const size_t total = (size_t)B * Cq * H * W;
    CCA_LAUNCH((cca::direct_map_kernel<float>), dim3(direct_grid(total)), dim3(cca::D_BLOCK), stream,
               dE, k, (const float *)nullptr, (const float *)nullptr, dq, Cq, H, W, total, kbs, 0L, dqbs);
    if (int e = launch_status("ca_backward(dq,direct)")) return e;
    CCA_LAUNCH((cca::direct_mapT_kernel<float>), dim3(direct_grid(total)), dim3(cca::D_BLOCK), stream,
               dE, q, (const float *)nullptr, dk, Cq, H, W, total, qbs, dkbs);
    return launch_status("ca_backward(dk,direct)");
}

int ca_map_forward_impl(const float *A, const float *v, const float *x, const float *gamma, float *out,
                        int B, int C, int H, int W, ccnet_stream_t stream, long vbs, long xbs, long obs) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (!A || !v || !out) return fail(CCNET_E_NULLPTR, "ca_map_forward: null tensor");
    const int impl = pick_impl(H, W);
    if (impl < 0) return impl;
    if (impl == 1) return launch_map_pair<false>(A, v, x, gamma, out, B, C, H, W, stream, "ca_map_forward", vbs, xbs, obs);
    if (impl == 2) return launch_long_map_pair<false>(A, v, x, gamma, out, B, C, H, W, stream, "ca_map_forward(long)", vbs, xbs, obs);
    const size_t total = (size_t)B * C * H * W;
    CCA_LAUNCH((cca::direct_map_kernel<float>), dim3(direct_grid(total)), dim3(cca::D_BLOCK), stream,
               A, v, x, gamma, out, C, H, W, total, vbs, xbs, obs);
    return launch_status("ca_map_forward(direct)");
}

// ks: K split of the dA contraction; the caller must hand the SAME plan to the softmax backward that consumes dA
int ca_map_backward_impl(const float *dout, const float *A, const float *v, const float *gamma,
                         float *dA, float *dv, int B, int C, int H, int W, ccnet_stream_t stream,
                         long dobs, long vbs, long dvbs, const KSplit &ks = KSplit()) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (!dout || !A || !v) return fail(CCNET_E_NULLPTR, "ca_map_backward: null tensor");
    const int impl = pick_impl(H, W);
    if (impl < 0) return impl;
    // dv before dA: the two are independent, and this order leaves more of dy in the Infinity Cache for the dA
    // kernel's 32-byte-segment column reads (measured: backward 533 -> 524 us at the headline shape)
    if (dv && impl == 1) {
        if (int e = launch_map_pair<true>(A, dout, nullptr, gamma, dv, B, C, H, W, stream, "ca_map_backward(dv)", dobs, 0, dvbs)) return e;
        dv = nullptr;
    }
    if (dA) {
        if (impl == 1) {
            if (int e = launch_weight_pair<false>(dout, v, dA, B, C, H, W, stream, "ca_map_backward(dA)", dobs, vbs, ks)) return e;
        } else if (impl == 2) {
            if (int e = launch_long_weight_pair<false>(dout, v, dA, B, C, H, W, stream, "ca_map_backward(dA,long)", dobs, vbs)) return e;
        } else {
            const size_t total = (size_t)B * H * W * (H + W);
            CCA_LAUNCH((cca::direct_weight_kernel<false>), dim3(direct_grid(total)), dim3(cca::D_BLOCK), stream,
                       dout, v, dA, C, H, W, total, dobs, vbs);
            if (int e = launch_status("ca_map_backward(dA,direct)")) return e;
        }
    }
    if (dv) {
        if (impl == 1)
            return launch_map_pair<true>(A, dout, nullptr, gamma, dv, B, C, H, W, stream, "ca_map_backward(dv)", dobs, 0, dvbs);
        if (impl == 2)
            return launch_long_map_pair<true>(A, dout, nullptr, gamma, dv, B, C, H, W, stream, "ca_map_backward(dv,long)", dobs, 0, dvbs);
        const size_t total = (size_t)B * C * H * W;
        CCA_LAUNCH((cca::direct_mapT_kernel<float>), dim3(direct_grid(total)), dim3(cca::D_BLOCK), stream,
                   A, dout, gamma, dv, C, H, W, total, dobs, dvbs);
        return launch_status("ca_map_backward(dv,direct)");
    }
    return 0;
}

// a batch stride must hold one dense image and keep 4-byte-element addressing inside 32-bit buffer offsets
int check_stride(long bs, int C, int H, int W, const char *what) {
    if (bs < (long)C * H * W) {
        static thread_local std::string msg;
        msg = std::string(what) + ": batch stride smaller than C*H*W";
        return fail(CCNET_E_BADSHAPE, msg.c_str());
    }
    return 0;
}

}  // namespace

extern "C" {

int ccnet_cca_version(void) { return CCNET_CCA_VERSION; }
const char *ccnet_cca_arch(void) { return "gfx950"; }
const char *ccnet_cca_last_error_string(void) { return g_last_error.c_str(); }
static int set_impl(int impl) {
    if (impl == CCNET_IMPL_AUTO || impl == CCNET_IMPL_DIRECT || impl == CCNET_IMPL_MFMA) return g_impl.exchange(impl);
    return g_impl.load();
}
static int get_impl(void) { return g_impl.load(); }
static int set_precision(int precision) {
    const int mb = g_map_bf16.load();
    const int prev = mb == 2 ? CCNET_PRECISION_BF16X3 : (mb == 1 ? CCNET_PRECISION_DEFAULT : CCNET_PRECISION_F32);
    if (precision == CCNET_PRECISION_F32)     g_map_bf16.store(0);
    if (precision == CCNET_PRECISION_DEFAULT) g_map_bf16.store(1);
    if (precision == CCNET_PRECISION_BF16X3)  g_map_bf16.store(2);
    return prev;
}
static int get_precision(void) {
    const int mb = g_map_bf16.load();
    return mb == 2 ? CCNET_PRECISION_BF16X3 : (mb == 1 ? CCNET_PRECISION_DEFAULT : CCNET_PRECISION_F32);
}
static int set_branch_mask(int mask) {
    if (mask >= 1 && mask <= 3) return g_branch_mask.exchange(mask);
    return g_branch_mask.load();
}

int ccnet_cca_shape_uses_mfma(int B, int C, int H, int W) {
    (void)B; (void)C;
    if (g_impl.load() == CCNET_IMPL_DIRECT || H <= 0 || W <= 0) return 0;
    const int longest = H > W ? H : W;
    return longest <= cca::kMaxStrip ? 1 : longest <= cca::kLongMaxStrip ? 2 : 0;
}

int ccnet_ca_softmax_forward_f32(const float *energy, float *out, int B, int H, int W, ccnet_stream_t stream) {
    if (int e = check_shape(B, 1, H, W)) return e;
    if (!energy || !out) return fail(CCNET_E_NULLPTR, "softmax_forward: null tensor");
    return softmax_forward(energy, out, B, H, W, stream);
}

int ccnet_ca_forward_f32(const float *q, const float *k, float *out, int B, int Cq, int H, int W, int flags,
                         ccnet_stream_t stream) {
    const long d = (long)Cq * H * W;
    return ca_forward_impl(q, k, out, B, Cq, H, W, flags, stream, d, d);
}

int ccnet_ca_backward_f32(const float *dE, const float *q, const float *k, float *dq, float *dk,
                          int B, int Cq, int H, int W, ccnet_stream_t stream) {
    const long d = (long)Cq * H * W;
    return ca_backward_impl(dE, q, k, dq, dk, B, Cq, H, W, stream, d, d, d, d);
}

static size_t ws_softmax_backward_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t npix = (size_t)B * H * W;
    return ((npix + cca::SM_WAVES - 1) / cca::SM_WAVES) * sizeof(float);
}

namespace {
// ``defer_reduce`` (may be null): do NOT launch the fixed-order reduction of the dgamma partials -- *defer_reduce receives their
// count and the caller folds the reduction into a later launch of the same stream (the dq | dk column pass, GmapJob::red_*).
int softmax_backward_impl(const float *A, const float *dA, const float *gamma, float *dE, float *dgamma,
                          void *workspace, size_t workspace_bytes, int B, int H, int W, ccnet_stream_t stream,
                          const KSplit &ks, int *defer_reduce = nullptr) {
    if (int e = check_shape(B, 1, H, W)) return e;
    if (!A || !dA || !dE) return fail(CCNET_E_NULLPTR, "softmax_backward: null tensor");
    const int npix = B * H * W, S = H + W;
    const int want = (npix + cca::SM_WAVES - 1) / cca::SM_WAVES;
    const int nblocks = want < cca::SM_MAX_BLOCKS ? want : cca::SM_MAX_BLOCKS;
    float *partials = nullptr;
    if (dgamma) {
        if (!workspace || workspace_bytes < (size_t)nblocks * sizeof(float))
            return fail(CCNET_E_WORKSPACE, "softmax_backward: workspace missing or too small");
        partials = static_cast<float *>(workspace);
    }
    const dim3 grid(nblocks), block(cca::SM_BLOCK);
    const float *extra = ks.extra;
    if (S <= 256)      CCA_LAUNCH((cca::softmax_bwd_kernel<4>), grid, block, stream, A, dA, gamma, dE, partials, npix, S, ks.n, extra, ks.stride);
    else if (S <= 512) CCA_LAUNCH((cca::softmax_bwd_kernel<8>), grid, block, stream, A, dA, gamma, dE, partials, npix, S, ks.n, extra, ks.stride);
    else               CCA_LAUNCH(cca::softmax_bwd_generic_kernel, grid, block, stream, A, dA, gamma, dE, partials, npix, S, ks.n, extra, ks.stride);
    if (int e = launch_status("softmax_bwd")) return e;
    if (dgamma && defer_reduce) {
        *defer_reduce = nblocks;
        return 0;
    }
    if (dgamma) {
        CCA_LAUNCH(cca::reduce_partials_kernel, dim3(1), block, stream, (const float *)partials, nblocks, dgamma);
        return launch_status("reduce_partials");
    }
    return 0;
}
size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
}  // namespace

int ccnet_ca_softmax_backward_f32(const float *A, const float *dA, const float *gamma, float *dE, float *dgamma,
                                  void *workspace, size_t workspace_bytes, int B, int H, int W,
                                  ccnet_stream_t stream) {
    return softmax_backward_impl(A, dA, gamma, dE, dgamma, workspace, workspace_bytes, B, H, W, stream, KSplit());
}

static size_t ws_forward_bytes(int B, int C, int Cq, int H, int W) {
    (void)C;
    if (B <= 0 || Cq <= 0 || H <= 0 || W <= 0) return 0;
    return ksplit_bytes(B, Cq, H, W);
}

static size_t ws_backward_bytes(int B, int C, int Cq, int H, int W) {
    (void)Cq;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return align256(ws_softmax_backward_bytes(B, H, W)) + ksplit_bytes(B, C, H, W);
}

int ccnet_ca_map_forward_f32(const float *A, const float *v, const float *x, const float *gamma, float *out,
                             int B, int C, int H, int W, ccnet_stream_t stream) {
    const long d = (long)C * H * W;
    return ca_map_forward_impl(A, v, x, gamma, out, B, C, H, W, stream, d, d, d);
}

extern "C++" {
namespace {
// one strip per workgroup and two workgroups per CU: whole rounds of strips first, then the remainder cut into channel
// ranges (a short last round instead of a full-length one that keeps a few CUs busy)
struct GmapPlan {
    int grid, n_whole, split;
};
// ``per_cu`` workgroups fit a CU.  The remainder strips are cut into the number of channel ranges that minimises the
// estimated length of the last phase: ceil(parts / slots) sub-rounds of (channel groups per part + prologue) -- the
// prologue (a strip's attention block) costs about one channel group of traffic.  (Round 2 used floor(slots / rem), which
// left 264 remainder strips of the headline shape uncut: a second round at half occupancy.)
GmapPlan gmap_plan(int strips, int C, int per_cu = 2) {
    const int slots = per_cu * num_cus(), ncg = (C + cca::GM_CG - 1) / cca::GM_CG;
    GmapPlan p;
    p.n_whole = strips / slots * slots;
    const int rem = strips - p.n_whole;
    p.split = 1;
    if (rem) {
        double best = 1e30;
        for (int s = 1; s <= ncg; ++s) {
            const int parts = rem * s, rounds = (parts + slots - 1) / slots;
            const double cost = rounds * ((ncg + s - 1) / s + 1.0);
            if (cost < best - 1e-9) { best = cost; p.split = s; }
        }
    }
    p.grid = p.n_whole + rem * p.split;
    return p;
}
}  // namespace
}  // extern "C++"

int ccnet_ca_map_backward_f32(const float *dout, const float *A, const float *v, const float *gamma,
                              float *dA, float *dv, int B, int C, int H, int W, ccnet_stream_t stream) {
    const long d = (long)C * H * W;
    return ca_map_backward_impl(dout, A, v, gamma, dA, dv, B, C, H, W, stream, d, d, d);
}

int ccnet_cca_forward_ws_f32(const float *q, const float *k, const float *v, const float *x, const float *gamma,
                             float *y, float *A, int B, int C, int Cq, int H, int W,
                             long q_bs, long k_bs, long v_bs, void *workspace, size_t workspace_bytes,
                             ccnet_stream_t stream) {
    if (int e = require_both_branches("cca_forward")) return e;
    if (!q || !k || !v || !x || !gamma || !y || !A) return fail(CCNET_E_NULLPTR, "cca_forward: null tensor");
    if (int e = check_shape(B, C, H, W)) return e;
    if (int e = check_shape(B, Cq, H, W)) return e;
    if (int e = check_stride(q_bs, Cq, H, W, "cca_forward(q)")) return e;
    if (int e = check_stride(k_bs, Cq, H, W, "cca_forward(k)")) return e;
    if (int e = check_stride(v_bs, C, H, W, "cca_forward(v)")) return e;
    if (int e = ca_forward_impl(q, k, A, B, Cq, H, W, CCNET_CA_SOFTMAX, stream, q_bs, k_bs,
                                ksplit_plan(B, Cq, H, W, workspace, workspace_bytes))) return e;
    const long d = (long)C * H * W;
    return ca_map_forward_impl(A, v, x, gamma, y, B, C, H, W, stream, v_bs, d, d);
}

int ccnet_cca_attention_strided_f32(const float *q, const float *k, float *A, int B, int Cq, int H, int W,
                                    long q_bs, long k_bs, ccnet_stream_t stream) {
    if (int e = require_both_branches("cca_attention")) return e;
    if (int e = check_shape(B, Cq, H, W)) return e;
    if (int e = check_stride(q_bs, Cq, H, W, "cca_attention(q)")) return e;
    if (int e = check_stride(k_bs, Cq, H, W, "cca_attention(k)")) return e;
    return ca_forward_impl(q, k, A, B, Cq, H, W, CCNET_CA_SOFTMAX, stream, q_bs, k_bs);
}

int ccnet_cca_forward_f32(const float *q, const float *k, const float *v, const float *x, const float *gamma,
                          float *y, float *A, int B, int C, int Cq, int H, int W, ccnet_stream_t stream) {
    return ccnet_cca_forward_ws_f32(q, k, v, x, gamma, y, A, B, C, Cq, H, W,
                                    (long)Cq * H * W, (long)Cq * H * W, (long)C * H * W, nullptr, 0, stream);
}

int ccnet_cca_backward_strided_f32(const float *dy, const float *q, const float *k, const float *v, const float *A,
                                   const float *gamma, float *dq, float *dk, float *dv, float *dgamma, float *scratch,
                                   void *workspace, size_t workspace_bytes, int B, int C, int Cq, int H, int W,
                                   long q_bs, long k_bs, long v_bs, long dq_bs, long dk_bs, long dv_bs,
                                   ccnet_stream_t stream) {
    if (int e = require_both_branches("cca_backward")) return e;
    if (!dy || !q || !k || !v || !A || !gamma || !dq || !dk || !dv || !dgamma || !scratch)
        return fail(CCNET_E_NULLPTR, "cca_backward: null tensor");
    if (int e = check_shape(B, C, H, W)) return e;
    if (int e = check_shape(B, Cq, H, W)) return e;
    if (int e = check_stride(q_bs, Cq, H, W, "cca_backward(q)")) return e;
    if (int e = check_stride(k_bs, Cq, H, W, "cca_backward(k)")) return e;
    if (int e = check_stride(v_bs, C, H, W, "cca_backward(v)")) return e;
    if (int e = check_stride(dq_bs, Cq, H, W, "cca_backward(dq)")) return e;
    if (int e = check_stride(dk_bs, Cq, H, W, "cca_backward(dk)")) return e;
    if (int e = check_stride(dv_bs, C, H, W, "cca_backward(dv)")) return e;
    // workspace = [softmax-backward partial sums | pad to 256 B | K-split slabs of dA (small batches, optional)]
    const size_t part = align256(ws_softmax_backward_bytes(B, H, W));
    KSplit ks;
    if (workspace && workspace_bytes > part)
        ks = ksplit_plan(B, C, H, W, static_cast<char *>(workspace) + part, workspace_bytes - part);
    // t = un-scaled dA into scratch (+ slabs), dv = gamma * (A^T-weighted sums of dy)
    if (int e = ca_map_backward_impl(dy, A, v, gamma, scratch, dv, B, C, H, W, stream, (long)C * H * W, v_bs, dv_bs, ks))
        return e;
    // dgamma = sum A*t ; dE = gamma * A * (t - sum_s A t), in place (t = the sum of the slabs)
    if (int e = softmax_backward_impl(A, scratch, gamma, scratch, dgamma, workspace, workspace_bytes,
                                      B, H, W, stream, ks)) return e;
    return ca_backward_impl(scratch, q, k, dq, dk, B, Cq, H, W, stream, q_bs, k_bs, dq_bs, dk_bs);
}

int ccnet_cca_backward_f32(const float *dy, const float *q, const float *k, const float *v, const float *A,
                           const float *gamma, float *dq, float *dk, float *dv, float *dgamma, float *scratch,
                           void *workspace, size_t workspace_bytes, int B, int C, int Cq, int H, int W,
                           ccnet_stream_t stream) {
    const long dq_ = (long)Cq * H * W, dc = (long)C * H * W;
    return ccnet_cca_backward_strided_f32(dy, q, k, v, A, gamma, dq, dk, dv, dgamma, scratch, workspace,
                                          workspace_bytes, B, C, Cq, H, W, dq_, dq_, dc, dq_, dq_, dc, stream);
}

/* ---- pixel-major paths (cca_gmap.hpp): features as (B, H*W, pixel stride) views, fp32 attention ---- */
extern "C++" {
namespace {
using cca::bf16_t;
template <typename FT> struct PmTraits;
template <> struct PmTraits<bf16_t> { static constexpr int kAlign = 8, kMaxStrip = 132; };
template <> struct PmTraits<float>  { static constexpr int kAlign = 4, kMaxStrip = 100; };

// a fixed-order sum of ``n`` floats at ``src`` into ``dst[0]`` that rides on another launch (GmapJob::red_*)
struct DeferredSum {
    const float *src = nullptr;
    int n = 0;
    float *dst = nullptr;
};

template <typename FT>
int check_pm_view(const char *what, long bs, int ps, int C, int H, int W) {
    constexpr int al = PmTraits<FT>::kAlign;
    if (ps < C || ps % al || bs < (long)(H * W - 1) * ps + C || bs % al) return fail(CCNET_E_BADSHAPE, what);
    if ((double)H * W * ps >= 536870912.0) return fail(CCNET_E_BADSHAPE, what);
    return 0;
}
// out = FT(alpha * contraction + resid): column strips into the fp32 partial, row strips add it and round once.
// NCHW (fp32, TRANS = false): resid / out are NCHW tensors with batch strides rbs / obs (rps / ops unused).
template <int P, bool TRANS, typename FT, bool NCHW = false>
int launch_gmap_pm(const float *T, const FT *F, const FT *resid, const float *gamma, FT *out, float *partial, int B, int C,
                   int H, int W, long fbs, int fps, long rbs, int rps, long obs, int ops, ccnet_stream_t stream) {
    const long pbs = (long)H * W * C;
    const GmapPlan gc = gmap_plan(B * W, C), gr = gmap_plan(B * H, C);
    if (std::is_same<FT, bf16_t>::value && g_planes_ring.load() != 0) {
        // bf16 features: the column pass on the ring kernel (three feature tiles, stores from the accumulators)
        if constexpr (std::is_same<FT, bf16_t>::value && !NCHW) {
            if (g_bf16_partial.load()) {
                // ... and its partial as bf16 (what the reference's bf16 arithmetic rounds out_H / the column half of dv to anyway):
                // half the bytes of the fp32 partial, written once and read once per pass
                CCA_LAUNCH((cca::gmap3_kernel<P, false, TRANS, false, 3, 2, bf16_t, bf16_t>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS), stream,
                           T, F, (const float *)nullptr, gamma, reinterpret_cast<bf16_t *>(partial), C, H, W, fbs, fps, 0L, 0, pbs, C, gc.n_whole, gc.split);
                if (int e = launch_status("gmap_pm(column, bf16 partial)")) return e;
                CCA_LAUNCH((cca::gmap_kernel<P, true, TRANS, true, FT, FT, false, false, 2, false, false, true>), dim3((unsigned)gr.grid), dim3(cca::GS_THREADS),
                           stream, T, F, (const float *)partial, resid, gamma, out, C, H, W, fbs, fps, pbs, C, rbs, rps, obs, ops,
                           gr.n_whole, gr.split, cca::GmapJob<FT, FT>{});
                return launch_status("gmap_pm(row, bf16 partial)");
            }
        }
        if constexpr (std::is_same<FT, bf16_t>::value)
            CCA_LAUNCH((cca::gmap3_kernel<P, false, TRANS, false, 3, 2, bf16_t>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS), stream,
                       T, F, (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0, pbs, C, gc.n_whole, gc.split);
    } else {
        CCA_LAUNCH((cca::gmap_kernel<P, false, TRANS, false, FT, float>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS),
                   stream, T, F, (const float *)nullptr, (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0,
                   0L, 0, pbs, C, gc.n_whole, gc.split, cca::GmapJob<FT, float>{});
    }
    if (int e = launch_status("gmap_pm(column)")) return e;
    CCA_LAUNCH((cca::gmap_kernel<P, true, TRANS, true, FT, FT, NCHW>), dim3((unsigned)gr.grid), dim3(cca::GS_THREADS),
               stream, T, F, (const float *)partial, resid, gamma, out, C, H, W, fbs, fps, pbs, C, rbs, rps, obs, ops,
               gr.n_whole, gr.split, cca::GmapJob<FT, FT>{});
    return launch_status("gmap_pm(row)");
}
template <bool TRANS, typename FT>
int gmap_pm(const float *T, const FT *F, const FT *resid, const float *gamma, FT *out, float *partial,
            int B, int C, int H, int W, long fbs, int fps, long rbs, int rps, long obs, int ops, ccnet_stream_t stream) {
    if ((H > W ? H : W) <= 100)
        return launch_gmap_pm<100, TRANS, FT>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
    if constexpr (PmTraits<FT>::kMaxStrip >= 132)
        return launch_gmap_pm<132, TRANS, FT>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
    return fail(CCNET_E_BADSHAPE, "gmap_pm: strip too long for this element type");
}
// Three-plane output of a backward (gmap_kernel, P3): dq | dk | dv leave as bf16 hi | lo | hi planes inside rows of ``ps`` bf16
// elements (plane stride ``plane`` = 2 Cq + C: dq at channel 0, dk at Cq, dv at 2 Cq of every plane) + one row of column-sum
// partials per row strip (``cs``: B * H rows of ``plane`` floats)
struct P3Out {
    uint16_t *d3 = nullptr;
    long bs = 0;
    int ps = 0, plane = 0;
    float *cs = nullptr;
};
// dq (features k) and dk (features q) from the same dE: one launch per branch, the second half of the workgroups runs the dk job
template <int P, typename FT, bool SIX, int WPC, bool P3 = false>
int launch_gmap_dual_pair(const float *dE, const FT *k, const FT *q, FT *dq, FT *dk, float *pq, float *pk, long pbs, int B, int Cq,
                          int H, int W, long kbs, int kps, long qbs, int qps, long dqbs, int dqps, long dkbs, int dkps,
                          ccnet_stream_t stream, const DeferredSum &red, const P3Out *p3 = nullptr) {
    const GmapPlan gc = gmap_plan(B * W, Cq, WPC), gr = gmap_plan(B * H, Cq, WPC);
    cca::GmapJob<FT, float> jc{q, nullptr, pk, qbs, pbs, qps, Cq, gc.grid};
    jc.red_src = red.src; jc.red_n = red.n; jc.red_dst = red.dst;
    CCA_LAUNCH((cca::gmap_kernel<P, false, false, false, FT, float, false, true, WPC, false, SIX>), dim3(cca::gmap_dual_grid(gc.grid)),
               dim3(cca::GS_THREADS), stream, dE, k, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, pq, Cq, H, W,
               kbs, kps, 0L, 0, 0L, 0, pbs, Cq, gc.n_whole, gc.split, jc);
    if (int e = launch_status("gmap_dual_pm(column)")) return e;
    cca::GmapJob<FT, FT> jr{q, pk, dk, qbs, dkbs, qps, dkps, gr.grid};
    if constexpr (P3) {
        // dq -> channels [0, Cq), dk -> [Cq, 2 Cq) of every plane of the three-plane rows (strides in bf16 elements)
        jr.out = reinterpret_cast<FT *>(p3->d3 + Cq);
        jr.obs = p3->bs; jr.ops = p3->ps; jr.p3_plane = p3->plane;
        jr.cs = p3->cs; jr.cs1 = p3->cs + Cq; jr.cs_stride = p3->plane;
        CCA_LAUNCH((cca::gmap_kernel<P, true, false, true, FT, FT, false, true, WPC, false, SIX, false, true>), dim3(cca::gmap_dual_grid(gr.grid)),
                   dim3(cca::GS_THREADS), stream, dE, k, (const float *)pq, (const FT *)nullptr, (const float *)nullptr,
                   reinterpret_cast<FT *>(p3->d3), Cq, H, W, kbs, kps, pbs, Cq, 0L, 0, p3->bs, p3->ps, gr.n_whole, gr.split, jr);
        return launch_status("gmap_dual_pm(row, three-plane output)");
    }
    CCA_LAUNCH((cca::gmap_kernel<P, true, false, true, FT, FT, false, true, WPC, false, SIX>), dim3(cca::gmap_dual_grid(gr.grid)),
               dim3(cca::GS_THREADS), stream, dE, k, (const float *)pq, (const FT *)nullptr, (const float *)nullptr, dq, Cq, H, W,
               kbs, kps, pbs, Cq, 0L, 0, dqbs, dqps, gr.n_whole, gr.split, jr);
    return launch_status("gmap_dual_pm(row)");
}
template <int P, typename FT>
int launch_gmap_dual_pm(const float *dE, const FT *k, const FT *q, FT *dq, FT *dk, float *partial, int B, int Cq, int H, int W,
                        long kbs, int kps, long qbs, int qps, long dqbs, int dqps, long dkbs, int dkps, ccnet_stream_t stream,
                        const DeferredSum &red) {
    const long pbs = (long)H * W * Cq;
    float *pq = partial, *pk = partial + (size_t)B * pbs;
#define CCA_DUAL_PAIR(SIX_, WPC_)                                                                                             \
    launch_gmap_dual_pair<P, FT, SIX_, WPC_>(dE, k, q, dq, dk, pq, pk, pbs, B, Cq, H, W, kbs, kps, qbs, qps, dqbs, dqps, dkbs, dkps, stream, red)
    if constexpr (std::is_same<FT, float>::value) {
        const bool six = g_dqdk_exact.load() != 0;
        if constexpr (P <= 100) {
            // one channel group per strip (C/8 <= 64, the reference's geometry): the three-workgroups-per-CU form -- these launches are
            // latency chains (attention block -> one tile -> one multiply -> stores), more of them per CU is what shortens them
            if (Cq <= cca::GM_CG && g_dqdk_wpc3.load()) return six ? CCA_DUAL_PAIR(true, 3) : CCA_DUAL_PAIR(false, 3);
        }
        return six ? CCA_DUAL_PAIR(true, 2) : CCA_DUAL_PAIR(false, 2);
    } else {
        // (bf16 q | k on the one-group / three-workgroups-per-CU form, its row pass's accumulators starting from the partial -- 162 / 165
        //  VGPRs at 132 positions, no spills -- measured SLOWER at configs[4]: dq | dk 130 + 120 -> 140 + 143 us, step 1.732 -> 1.743-1.753 ms,
        //  profiles/r06g_bf16_dqdk_three_per_cu_ab.txt; not kept)
        return CCA_DUAL_PAIR(false, 2);
    }
#undef CCA_DUAL_PAIR
}
template <typename FT>
int gmap_dual_pm(const float *dE, const FT *k, const FT *q, FT *dq, FT *dk, float *partial, int B, int Cq, int H, int W,
                 long kbs, int kps, long qbs, int qps, long dqbs, int dqps, long dkbs, int dkps, ccnet_stream_t stream,
                 const DeferredSum &red = DeferredSum()) {
    if ((H > W ? H : W) <= 100)
        return launch_gmap_dual_pm<100, FT>(dE, k, q, dq, dk, partial, B, Cq, H, W, kbs, kps, qbs, qps, dqbs, dqps, dkbs, dkps, stream, red);
    if constexpr (PmTraits<FT>::kMaxStrip >= 132)
        return launch_gmap_dual_pm<132, FT>(dE, k, q, dq, dk, partial, B, Cq, H, W, kbs, kps, qbs, qps, dqbs, dqps, dkbs, dkps, stream, red);
    return fail(CCNET_E_BADSHAPE, "gmap_dual_pm: strip too long for this element type");
}
template <bool MASK, typename FT>
int gweight_pm(const FT *X, const FT *Y, float *T, int B, int Cx, int H, int W, long xbs, int xps, long ybs, int yps,
               ccnet_stream_t stream) {
    // (bf16 dA stays on gweight_kernel: the persistent gweight_stream_kernel<132, bf16_t> measured 350 us against 301 us at
    // configs[4] -- profiles/r03j_bf16_compare.txt -- wavefront 0 owns two of the nine tile rows and every barrier waits for it)
    const dim3 grid((unsigned)(B * (H + W))), block(cca::GM_THREADS);
    const bool single = Cx <= cca::GM_CG;           // one chunk: the single-buffered form (more workgroups per CU)
    if constexpr (MASK && std::is_same<FT, float>::value) {
        // the fp32 energies at the headline geometry: three 52.8 KB workgroups per CU, every one the same latency chain -- the strips
        // beyond the whole rounds are cut into tile-row parts (gweight_kernel, n_whole) so that the last round is a short one
        if (single && (H > W ? H : W) <= 100 && g_energy_tail.load()) {
            const int strips = B * (H + W), slots = 3 * num_cus(), nt = (100 + 15) / 16;       // (tile rows of the 100-position kernel)
            const int n_whole = strips / slots * slots, rem = strips - n_whole;
            if (n_whole > 0 && rem > 0 && rem * nt <= slots / 2) {
                CCA_LAUNCH((cca::gweight_kernel<100, MASK, FT, true>), dim3((unsigned)(n_whole + rem * nt)), block, stream, X, Y, T, Cx, H, W,
                           xbs, xps, ybs, yps, 1, 1, n_whole);
                return launch_status("gweight_pm(energies, tail parts)");
            }
        }
    }
#define CCA_GWEIGHT(P_)                                                                                                   \
    do {                                                                                                                  \
        if (single) CCA_LAUNCH((cca::gweight_kernel<P_, MASK, FT, true>), grid, block, stream, X, Y, T, Cx, H, W, xbs, xps, ybs, yps); \
        else        CCA_LAUNCH((cca::gweight_kernel<P_, MASK, FT, false>), grid, block, stream, X, Y, T, Cx, H, W, xbs, xps, ybs, yps); \
    } while (0)
    if ((H > W ? H : W) <= 100) {
        CCA_GWEIGHT(100);
    } else {
        if constexpr (PmTraits<FT>::kMaxStrip >= 132)
            CCA_GWEIGHT(132);
        else
            return fail(CCNET_E_BADSHAPE, "gweight_pm: strip too long for this element type");
    }
#undef CCA_GWEIGHT
    return launch_status("gweight_pm");
}
template <typename FT>
int check_pm_problem(const char *what, int B, int C, int Cq, int H, int W) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (int e = check_shape(B, Cq, H, W)) return e;
    constexpr int al = PmTraits<FT>::kAlign;
    if ((H > W ? H : W) > PmTraits<FT>::kMaxStrip || C % al || Cq % al) return fail(CCNET_E_BADSHAPE, what);
    return 0;
}
size_t pm_workspace_bytes(int B, int C, int Cq, int H, int W, int backward) {
    if (B <= 0 || C <= 0 || Cq <= 0 || H <= 0 || W <= 0) return 0;
    // forward: the column partial of the aggregation.  backward: softmax-backward's slabs | the column partial of dv | the
    // column partials of dq and dk side by side (a region of their own: those launches may run next to the dv passes)
    const size_t px = (size_t)B * H * W * sizeof(float);
    if (!backward) return px * C;
    return align256(ws_softmax_backward_bytes(B, H, W)) + align256(px * C) + align256(px * 2 * Cq) + 256;   // (+ 256 spare bytes: ABI 210 sized it so)
}
float *partial_qk_of(float *partial, int B, int C, int H, int W) {
    return reinterpret_cast<float *>(reinterpret_cast<char *>(partial) + align256((size_t)B * H * W * C * sizeof(float)));
}
// fork / join of the library's side stream around the launches of a backward that are independent of the caller's chain
// ("planes_overlap": 0 = never fork, 1 = dv next to softmax-backward and dq | dk, 2 = dv next to dA as well)
struct SideFork {
    hipStream_t main, side = nullptr;
    explicit SideFork(ccnet_stream_t m) : main((hipStream_t)m) {}
    void fork() { if (!side) side = cca_side::fork(main); }
    ccnet_stream_t stream() const { return side ? (ccnet_stream_t)side : (ccnet_stream_t)main; }
    // every path out of the backward joins (a capture must not end forked)
    int join(int e) {
        if (side && !cca_side::join(main) && !e) e = fail(1, "cca_backward: joining the side stream failed");
        side = nullptr;
        return e;
    }
};

template <typename FT>
int cca_forward_pm(const char *name, const FT *q, const FT *k, const FT *v, const FT *x, const float *gamma, FT *y, float *A,
                   int B, int C, int Cq, int H, int W, long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps,
                   long x_bs, int x_ps, long y_bs, int y_ps, void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    if (int e = require_both_branches(name)) return e;
    if (!q || !k || !v || !x || !gamma || !y || !A) return fail(CCNET_E_NULLPTR, "cca_forward_pm: null tensor");
    if (int e = check_pm_problem<FT>("cca_forward_pm: strip length / channel divisibility (see ccnet_cca.h)", B, C, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_forward_pm: q view", q_bs, q_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_forward_pm: k view", k_bs, k_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_forward_pm: v view", v_bs, v_ps, C, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_forward_pm: x view", x_bs, x_ps, C, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_forward_pm: y view", y_bs, y_ps, C, H, W)) return e;
    if (!workspace || workspace_bytes < pm_workspace_bytes(B, C, Cq, H, W, 0))
        return fail(CCNET_E_WORKSPACE, "cca_forward_pm: workspace missing or too small");
    if (int e = gweight_pm<true, FT>(q, k, A, B, Cq, H, W, q_bs, q_ps, k_bs, k_ps, stream)) return e;
    if (int e = softmax_forward(A, A, B, H, W, stream)) return e;
    return gmap_pm<false, FT>(A, v, x, gamma, y, (float *)workspace, B, C, H, W, v_bs, v_ps, x_bs, x_ps, y_bs, y_ps, stream);
}

template <typename FT>
int cca_backward_pm(const char *name, const FT *dy, const FT *q, const FT *k, const FT *v, const float *A, const float *gamma,
                    FT *dq, FT *dk, FT *dv, float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                    long dy_bs, int dy_ps, long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps,
                    long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                    void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    if (int e = require_both_branches(name)) return e;
    if (!dy || !q || !k || !v || !A || !gamma || !dq || !dk || !dv || !dgamma || !scratch)
        return fail(CCNET_E_NULLPTR, "cca_backward_pm: null tensor");
    if (int e = check_pm_problem<FT>("cca_backward_pm: strip length / channel divisibility (see ccnet_cca.h)", B, C, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: dy view", dy_bs, dy_ps, C, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: q view", q_bs, q_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: k view", k_bs, k_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: v view", v_bs, v_ps, C, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: dq view", dq_bs, dq_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: dk view", dk_bs, dk_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<FT>("cca_backward_pm: dv view", dv_bs, dv_ps, C, H, W)) return e;
    if (!workspace || workspace_bytes < pm_workspace_bytes(B, C, Cq, H, W, 1))
        return fail(CCNET_E_WORKSPACE, "cca_backward_pm: workspace missing or too small");
    const size_t sm = align256(ws_softmax_backward_bytes(B, H, W));
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + sm);
    // t = un-scaled dA (the adjoint of the aggregation, functions.py:46-47), dv = gamma * A^T-weighted dy
    // dv (two C-sized, HBM-bound passes) is independent of the chain dA -> dE -> dq | dk: it runs on the side stream
    // (-1 = what measured best: 1 for fp32 features and for bf16 with the fp32 partial -- 1.917 -> 1.903 ms at configs[4], 1.959 with 2;
    //  with the bf16 partial the column pass is light enough to share the CUs with dA: 1.64-1.70 -> 1.60-1.61 ms with 2,
    //  profiles/r05b_bf16_partial_ab.txt)
    const bool bf16_light = std::is_same<FT, bf16_t>::value && g_bf16_partial.load() != 0 && g_planes_ring.load() != 0;
    const int overlap = g_planes_overlap.load() < 0 ? (bf16_light ? 2 : 1) : g_planes_overlap.load();
    SideFork sf(stream);
    if (overlap == 2) sf.fork();
    int e = gweight_pm<false, FT>(dy, v, scratch, B, C, H, W, dy_bs, dy_ps, v_bs, v_ps, stream);
    if (overlap == 1) sf.fork();
    if (!e) e = gmap_pm<true, FT>(A, dy, nullptr, gamma, dv, partial, B, C, H, W, dy_bs, dy_ps, 0L, 0, dv_bs, dv_ps, sf.stream());
    // dgamma = sum A t;  dE = gamma * A * (t - sum_s A t), in place
    // (the fixed-order sum of the dgamma partials rides on the dq | dk column launch: one launch less)
    DeferredSum red{static_cast<const float *>(workspace), 0, dgamma};
    if (!e) e = softmax_backward_impl(A, scratch, gamma, scratch, dgamma, workspace, sm, B, H, W, stream, KSplit(), &red.n);
    // the column partials of dq and dk sit side by side in their own region
    if (!e) e = gmap_dual_pm<FT>(scratch, k, q, dq, dk, partial_qk_of(partial, B, C, H, W), B, Cq, H, W, k_bs, k_ps, q_bs, q_ps,
                                 dq_bs, dq_ps, dk_bs, dk_ps, stream, red);
    return sf.join(e);
}
}  // namespace
}  // extern "C++"

static size_t ws_pm_bytes(int B, int C, int Cq, int H, int W, int backward) {
    return pm_workspace_bytes(B, C, Cq, H, W, backward);
}

int ccnet_cca_forward_pm_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *x,
                              const float *gamma, uint16_t *y, float *A, int B, int C, int Cq, int H, int W,
                              long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long x_bs, int x_ps,
                              long y_bs, int y_ps, void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    return cca_forward_pm<bf16_t>("cca_forward_pm_bf16", (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)x,
                                  gamma, (bf16_t *)y, A, B, C, Cq, H, W, q_bs, q_ps, k_bs, k_ps, v_bs, v_ps, x_bs, x_ps, y_bs, y_ps,
                                  workspace, workspace_bytes, stream);
}
int ccnet_cca_forward_pm_f32(const float *q, const float *k, const float *v, const float *x,
                             const float *gamma, float *y, float *A, int B, int C, int Cq, int H, int W,
                             long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long x_bs, int x_ps,
                             long y_bs, int y_ps, void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    return cca_forward_pm<float>("cca_forward_pm_f32", q, k, v, x, gamma, y, A, B, C, Cq, H, W, q_bs, q_ps, k_bs, k_ps, v_bs, v_ps,
                                 x_bs, x_ps, y_bs, y_ps, workspace, workspace_bytes, stream);
}

int ccnet_cca_backward_pm_bf16(const uint16_t *dy, const uint16_t *q, const uint16_t *k, const uint16_t *v,
                               const float *A, const float *gamma, uint16_t *dq, uint16_t *dk, uint16_t *dv,
                               float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                               long dy_bs, int dy_ps, long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps,
                               long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                               void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    return cca_backward_pm<bf16_t>("cca_backward_pm_bf16", (const bf16_t *)dy, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                                   A, gamma, (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv, dgamma, scratch, B, C, Cq, H, W,
                                   dy_bs, dy_ps, q_bs, q_ps, k_bs, k_ps, v_bs, v_ps, dq_bs, dq_ps, dk_bs, dk_ps, dv_bs, dv_ps,
                                   workspace, workspace_bytes, stream);
}
int ccnet_cca_backward_pm_f32(const float *dy, const float *q, const float *k, const float *v,
                              const float *A, const float *gamma, float *dq, float *dk, float *dv,
                              float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                              long dy_bs, int dy_ps, long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps,
                              long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                              void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    return cca_backward_pm<float>("cca_backward_pm_f32", dy, q, k, v, A, gamma, dq, dk, dv, dgamma, scratch, B, C, Cq, H, W,
                                  dy_bs, dy_ps, q_bs, q_ps, k_bs, k_ps, v_bs, v_ps, dq_bs, dq_ps, dk_bs, dk_ps, dv_bs, dv_ps,
                                  workspace, workspace_bytes, stream);
}

/* ---- SPLIT-PLANE path (cca_gmap.hpp, bf16p_t): the fp32 core with its C-sized contraction operands (v, dy) pre-split
 * ---- into bf16 hi | lo planes by their producers.  q, k stay fp32 pixel-major (the energies are exact fp32 products), the
 * ---- module's x, y, dy are NCHW, dq | dk | dv leave as fp32 pixel-major views (one GEMM operand downstream). ---- */
extern "C++" {
namespace {
using cca::bf16p_t;
int check_planes_view(const char *what, long bs, int ps, int C, int H, int W, int nplanes = 2) {
    if (C % 8 || ps < nplanes * C || ps % 8 || bs < (long)(H * W - 1) * ps + nplanes * C || bs % 8) return fail(CCNET_E_BADSHAPE, what);
    if ((double)H * W * ps >= 1073741824.0) return fail(CCNET_E_BADSHAPE, what);            /* 2-byte elements, 31-bit offsets */
    return 0;
}
// ``long_strips``: the entry point also takes strips of 133 .. 528 positions -- rows, columns or both -- as 2 .. 4 blocks
// (cca::long_block)
int check_planes_problem(const char *what, int B, int C, int Cq, int H, int W, bool long_strips = false) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (int e = check_shape(B, Cq, H, W)) return e;
    const int lim = long_strips ? 4 * 132 : 132;
    if (H > lim || W > lim || C % 8 || Cq % 4) return fail(CCNET_E_BADSHAPE, what);
    if ((W > 132 || H > 132) && Cq > cca::GM_CG) return fail(CCNET_E_BADSHAPE, what);   /* (the blocked energies kernel: one 64-channel chunk) */
    return 0;
}
inline int long_blocks(int L) { return (L + 131) / 132; }
// CCNET_PLANES_* -> where the halves go inside a pixel's row (false: unknown layout)
bool plane_layout(int layout, int C, cca::PlaneLayout *pl) {
    if (layout == CCNET_PLANES_HL) *pl = cca::PlaneLayout{C, 0, 2 * C};
    else if (layout == CCNET_PLANES_HLH) *pl = cca::PlaneLayout{C, 2 * C, 3 * C};
    else if (layout == CCNET_PLANES_HHL) *pl = cca::PlaneLayout{2 * C, C, 3 * C};
    else return false;
    return true;
}
// column strips -> fp32 partial, row strips add it (+ the NCHW residual) and write the output
// P = 100: column passes on the ring kernel (two slots, three workgroups per CU by default), row passes on gmap_kernel with two
// workgroups per CU.  P = 132 (strips 101 .. 132): the ring kernel with two slots and two workgroups per CU, the row passes with
// ONE workgroup per CU (two plane tiles + the output image are 104 KB).
template <int P, bool TRANS>
int launch_gmap3_planes(const float *T, const bf16p_t *F, const float *gamma, float *out, float *partial, int B, int C, int H, int W,
                        long fbs, int fps, long obs, int ops, bool row_too, ccnet_stream_t stream) {
    const long pbs = (long)H * W * C;
    const GmapPlan gr = gmap_plan(B * H, C);
    const int ring = g_planes_ring.load();
    if constexpr (P > 100) {
        const GmapPlan gc = gmap_plan(B * W, C);
        CCA_LAUNCH((cca::gmap3_kernel<P, false, TRANS, false, 2, 2>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS), stream, T, F,
                   (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0, pbs, C, gc.n_whole, gc.split);
        return launch_status("gmap3_planes(column)");
    } else {
    if (ring == 2) {                        // three workgroups per CU, two ring slots
        const GmapPlan gc = gmap_plan(B * W, C, 3);
        CCA_LAUNCH((cca::gmap3_kernel<100, false, TRANS, false, 2, 3>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS), stream, T, F,
                   (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0, pbs, C, gc.n_whole, gc.split);
    } else {
        const GmapPlan gc = gmap_plan(B * W, C);
        CCA_LAUNCH((cca::gmap3_kernel<100, false, TRANS, false>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS), stream, T, F,
                   (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0, pbs, C, gc.n_whole, gc.split);
    }
    if (int e = launch_status("gmap3_planes(column)")) return e;
    // (row passes: the accumulator-layout addend prefetch of gmap3 is still slower than gmap_kernel's output image --
    // 121 vs 103 us at the headline shape, profiles/r03e_bench.json -- so only "planes_ring" 1 uses it there)
    if (!row_too || ring != 1) return 0;
    CCA_LAUNCH((cca::gmap3_kernel<100, true, TRANS, true>), dim3((unsigned)gr.grid), dim3(cca::GS_THREADS), stream, T, F,
               (const float *)partial, gamma, out, C, H, W, fbs, fps, pbs, C, obs, ops, gr.n_whole, gr.split);
    return launch_status("gmap3_planes(row)");
    }
}
// the row pass of whole row strips (<= P positions): adds the column partial (+ the NCHW residual) and writes the output
template <int P, bool TRANS, bool NCHW>
int launch_gmap_planes_row_p(const float *T, const bf16p_t *F, const float *resid, const float *gamma, float *out, float *partial,
                             int B, int C, int H, int W, long fbs, int fps, long rbs, int rps, long obs, int ops, ccnet_stream_t stream) {
    constexpr int WPC = P > 100 ? 1 : 2;
    const long pbs = (long)H * W * C;
    const GmapPlan gr = gmap_plan(B * H, C, WPC);
    cca::GmapJob<bf16p_t, float> job{};
    if (NCHW && g_planes_xcd.load() && (B * H) % 8 == 0 && gr.n_whole % 8 == 0) job.xcd = B * H / 8;
    CCA_LAUNCH((cca::gmap_kernel<P, true, TRANS, true, bf16p_t, float, NCHW, false, WPC>), dim3((unsigned)gr.grid), dim3(cca::GS_THREADS),
               stream, T, F, (const float *)partial, resid, gamma, out, C, H, W, fbs, fps, pbs, C, rbs, rps, obs, ops,
               gr.n_whole, gr.split, job);
    return launch_status("gmap_planes(row)");
}
template <int P, bool TRANS, bool NCHW>
int launch_gmap_planes_p(const float *T, const bf16p_t *F, const float *resid, const float *gamma, float *out, float *partial,
                         int B, int C, int H, int W, long fbs, int fps, long rbs, int rps, long obs, int ops, ccnet_stream_t stream) {
    const long pbs = (long)H * W * C;
    const GmapPlan gc = gmap_plan(B * W, C);
    const int ring = P > 100 ? 2 : g_planes_ring.load();
    if (ring) {
        if (int e = launch_gmap3_planes<P, TRANS>(T, F, gamma, out, partial, B, C, H, W, fbs, fps, obs, ops, !NCHW, stream)) return e;
        if (!NCHW && ring == 1) return 0;
    } else {
        if constexpr (P <= 100) {
            CCA_LAUNCH((cca::gmap_kernel<100, false, TRANS, false, bf16p_t, float>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS),
                       stream, T, F, (const float *)nullptr, (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0,
                       0L, 0, pbs, C, gc.n_whole, gc.split, cca::GmapJob<bf16p_t, float>{});
            if (int e = launch_status("gmap_planes(column)")) return e;
        }
    }
    return launch_gmap_planes_row_p<P, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
}
// aggregation (or its dv adjoint, TRANS) with LONG rows (133 .. 528 positions), after the column pass has left its fp32 partial:
// the row pass as nb launches, one per block of the CONTRACTED positions: partial += gamma * (attention
// block) . (feature block) in place, the last one writes the output (NCHW: + the residual).  A workgroup owns one OUTPUT block of
// a row strip.
// P = 100 with blocks of <= 100 positions where four of them cover the row (W <= 400): the 100-position kernels keep their
// attention fragments in 96 registers and run two workgroups per CU; at 132 positions they take 192 and run one (1 wave per SIMD)
template <int P, bool TRANS, bool NCHW>
int launch_gmap_planes_long_rows_p(const float *T, const bf16p_t *F, const float *resid, const float *gamma, float *out, float *partial,
                                   int B, int C, int H, int W, long fbs, int fps, long rbs, int rps, long obs, int ops,
                                   ccnet_stream_t stream) {
    constexpr int WPC = P > 100 ? 1 : 2;
    const int nb = (W + P - 1) / P;
    const long pbs = (long)H * W * C;
    const GmapPlan gr = gmap_plan(B * H * nb, C, WPC);
    for (int j = 0; j < nb; ++j) {
        cca::GmapJob<bf16p_t, float> job{};
        job.nb = nb;
        job.jblk = j;
        if (j + 1 < nb)
            CCA_LAUNCH((cca::gmap_kernel<P, true, TRANS, true, bf16p_t, float, false, false, WPC, true>), dim3((unsigned)gr.grid),
                       dim3(cca::GS_THREADS), stream, T, F, (const float *)partial, (const float *)nullptr, gamma, partial, C, H, W,
                       fbs, fps, pbs, C, 0L, 0, pbs, C, gr.n_whole, gr.split, job);
        else
            CCA_LAUNCH((cca::gmap_kernel<P, true, TRANS, true, bf16p_t, float, NCHW, false, WPC, true>), dim3((unsigned)gr.grid),
                       dim3(cca::GS_THREADS), stream, T, F, (const float *)partial, resid, gamma, out, C, H, W, fbs, fps, pbs, C,
                       rbs, rps, obs, ops, gr.n_whole, gr.split, job);
        if (int e = launch_status("gmap_planes(long rows)")) return e;
    }
    return 0;
}
// LONG columns (133 .. 528 positions; round 4: maps whose BOTH sides exceed 132 -- the reference's multi-scale whole-image
// evaluation, evaluate.py:146-166 at scales > 1 -- no longer fall to the windowed / any-shape strip kernels): the column pass as
// nb launches of gmap_kernel, one per block of the contracted positions; the first one starts the partial, the others update it
// in place.  A workgroup owns one OUTPUT block of a column strip.  Blocks of <= 100 positions (two workgroups per CU) up to 400.
template <int P, bool TRANS>
int launch_gmap_planes_long_cols_p(const float *T, const bf16p_t *F, const float *gamma, float *partial, int B, int C, int H, int W,
                                   long fbs, int fps, ccnet_stream_t stream) {
    constexpr int WPC = P > 100 ? 1 : 2;
    const int nb = (H + P - 1) / P;
    const long pbs = (long)H * W * C;
    const GmapPlan gc = gmap_plan(B * W * nb, C, WPC);
    for (int j = 0; j < nb; ++j) {
        cca::GmapJob<bf16p_t, float> job{};
        job.nb = nb;
        job.jblk = j;
        if (j == 0)
            CCA_LAUNCH((cca::gmap_kernel<P, false, TRANS, false, bf16p_t, float, false, false, WPC, true>), dim3((unsigned)gc.grid),
                       dim3(cca::GS_THREADS), stream, T, F, (const float *)nullptr, (const float *)nullptr, gamma, partial, C, H, W,
                       fbs, fps, 0L, 0, 0L, 0, pbs, C, gc.n_whole, gc.split, job);
        else
            CCA_LAUNCH((cca::gmap_kernel<P, false, TRANS, true, bf16p_t, float, false, false, WPC, true>), dim3((unsigned)gc.grid),
                       dim3(cca::GS_THREADS), stream, T, F, (const float *)partial, (const float *)nullptr, gamma, partial, C, H, W,
                       fbs, fps, pbs, C, 0L, 0, pbs, C, gc.n_whole, gc.split, job);
        if (int e = launch_status("gmap_planes(long columns)")) return e;
    }
    return 0;
}
template <bool TRANS, bool NCHW>
int launch_gmap_planes(const float *T, const bf16p_t *F, const float *resid, const float *gamma, float *out, float *partial,
                       int B, int C, int H, int W, long fbs, int fps, long rbs, int rps, long obs, int ops, ccnet_stream_t stream) {
    // (rows of 101 .. 132 positions stay whole on the 132-position kernels: as two blocks on the 100-position ones the extra pass
    // over the partial costs more than the second workgroup per CU returns -- (16,512,129,129) 3.55 -> 3.70 ms, profiles/r04f_planes_129.txt)
    if (H <= 132 && W <= 132) {
        if ((H > W ? H : W) <= 100)
            return launch_gmap_planes_p<100, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
        return launch_gmap_planes_p<132, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
    }
    // column pass -> fp32 partial
    int e;
    if (H > 400)      e = launch_gmap_planes_long_cols_p<132, TRANS>(T, F, gamma, partial, B, C, H, W, fbs, fps, stream);
    else if (H > 132) e = launch_gmap_planes_long_cols_p<100, TRANS>(T, F, gamma, partial, B, C, H, W, fbs, fps, stream);
    else              e = launch_gmap3_planes<132, TRANS>(T, F, gamma, out, partial, B, C, H, W, fbs, fps, 0L, 0, false, stream);
    if (e) return e;
    // row pass(es): += the partial (+ the NCHW residual) -> the output
    if (W > 400) return launch_gmap_planes_long_rows_p<132, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
    if (W > 132) return launch_gmap_planes_long_rows_p<100, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
    if (W > 100) return launch_gmap_planes_row_p<132, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
    return launch_gmap_planes_row_p<100, TRANS, NCHW>(T, F, resid, gamma, out, partial, B, C, H, W, fbs, fps, rbs, rps, obs, ops, stream);
}
// The forward aggregation on fp32 pixel-major v AS IT IS (strips <= 100): F32T tiles, hi | lo split per fragment -- no planes
// tensor, no split pass.  Column strips -> fp32 partial (ring kernel, three workgroups per CU), row strips add it and the NCHW
// residual and write NCHW y.
int launch_gmap_direct_f32(const float *T, const float *v, const float *resid, const float *gamma, float *out, float *partial,
                           int B, int C, int H, int W, long fbs, int fps, long rbs, long obs, ccnet_stream_t stream) {
    const long pbs = (long)H * W * C;
    const GmapPlan gc = gmap_plan(B * W, C, 3), gr = gmap_plan(B * H, C, 2);
    CCA_LAUNCH((cca::gmap3_kernel<100, false, false, false, 2, 3, float>), dim3((unsigned)gc.grid), dim3(cca::GS_THREADS), stream, T, v,
               (const float *)nullptr, gamma, partial, C, H, W, fbs, fps, 0L, 0, pbs, C, gc.n_whole, gc.split);
    if (int e = launch_status("gmap3_direct(column)")) return e;
    cca::GmapJob<float, float> job{};
    if (g_planes_xcd.load() && (B * H) % 8 == 0 && gr.n_whole % 8 == 0) job.xcd = B * H / 8;
    CCA_LAUNCH((cca::gmap_kernel<100, true, false, true, float, float, true, false, 2>), dim3((unsigned)gr.grid), dim3(cca::GS_THREADS),
               stream, T, v, (const float *)partial, resid, gamma, out, C, H, W, fbs, fps, pbs, C, rbs, 0, obs, 0,
               gr.n_whole, gr.split, job);
    return launch_status("gmap_direct(row)");
}
// strips 101 .. 132 with fp32 q | k: the energies and dq | dk kernels of the pixel-major fp32 family at 132 positions (one
// workgroup per CU for the latter: fp32 tiles)
int gweight_energies_f32(const float *q, const float *k, float *A, int B, int Cq, int H, int W, long qbs, int qps, long kbs, int kps,
                         ccnet_stream_t stream) {
    if ((H > W ? H : W) <= 100) return gweight_pm<true, float>(q, k, A, B, Cq, H, W, qbs, qps, kbs, kps, stream);
    if (W > 132 || H > 132) {          // long strips: blocks x blocks tiles per strip of a branch that exceeds 132 positions
        const int nb = long_blocks(W), nbc = long_blocks(H);
        const dim3 grid((unsigned)(B * (W * nbc * nbc + H * nb * nb))), block(cca::GM_THREADS);
        CCA_LAUNCH((cca::gweight_kernel<132, true, float, true, true>), grid, block, stream, q, k, A, Cq, H, W, qbs, qps, kbs, kps, nb, nbc);
        return launch_status("gweight_energies(long strips)");
    }
    const dim3 grid((unsigned)(B * (H + W))), block(cca::GM_THREADS);
    if (Cq <= cca::GM_CG) CCA_LAUNCH((cca::gweight_kernel<132, true, float, true>), grid, block, stream, q, k, A, Cq, H, W, qbs, qps, kbs, kps);
    else                  CCA_LAUNCH((cca::gweight_kernel<132, true, float, false>), grid, block, stream, q, k, A, Cq, H, W, qbs, qps, kbs, kps);
    return launch_status("gweight_energies(132)");
}
// (strips beyond 100 positions: always the six-term products -- VERDICT r5 item 4b: the device-gated exact redo of round 5 stopped
//  at 100 positions and left the whole-image maps of evaluate.py:102-143 on the three-term form)
int gmap_dual_f32(const float *dE, const float *k, const float *q, float *dq, float *dk, float *partial, int B, int Cq, int H, int W,
                  long kbs, int kps, long qbs, int qps, long dqbs, int dqps, long dkbs, int dkps, ccnet_stream_t stream,
                  const DeferredSum &red) {
    if ((H > W ? H : W) <= 100)
        return gmap_dual_pm<float>(dE, k, q, dq, dk, partial, B, Cq, H, W, kbs, kps, qbs, qps, dqbs, dqps, dkbs, dkps, stream, red);
    const long pbs = (long)H * W * Cq;
    float *pq = partial, *pk = partial + (size_t)B * pbs;
    const GmapPlan gr = gmap_plan(B * H, Cq, 1);
    if (H > 132) {          // long columns: one launch per block of the contracted positions; the first starts the partials
        const bool p100 = H <= 400;
        const int nb = p100 ? (H + 99) / 100 : long_blocks(H);
        const GmapPlan gl = gmap_plan(B * W * nb, Cq, p100 ? 2 : 1);
        for (int j = 0; j < nb; ++j) {
            cca::GmapJob<float, float> jl{q, j ? pk : nullptr, pk, qbs, pbs, qps, Cq, gl.grid, nb, j};
            if (j == 0) { jl.red_src = red.src; jl.red_n = red.n; jl.red_dst = red.dst; }
            const float *add = j ? pq : nullptr;
            const long abs_ = j ? pbs : 0L;
            const int aps = j ? Cq : 0;
#define CCA_DUAL_COL(PP, ADD, WPC)                                                                                                    \
            CCA_LAUNCH((cca::gmap_kernel<PP, false, false, ADD, float, float, false, true, WPC, true, true>), dim3(cca::gmap_dual_grid(gl.grid)), \
                       dim3(cca::GS_THREADS), stream, dE, k, add, (const float *)nullptr, (const float *)nullptr, pq, Cq, H, W, kbs, kps, \
                       abs_, aps, 0L, 0, pbs, Cq, gl.n_whole, gl.split, jl)
            if (p100) { if (j) CCA_DUAL_COL(100, true, 2); else CCA_DUAL_COL(100, false, 2); }
            else      { if (j) CCA_DUAL_COL(132, true, 1); else CCA_DUAL_COL(132, false, 1); }
#undef CCA_DUAL_COL
            if (int e = launch_status("gmap_dual_f32(long columns)")) return e;
        }
    } else {
        // (C/8 <= 64: one channel group per strip -> the one-slot form, two workgroups per CU; see gmap_kernel, ONEG)
        const bool one = Cq <= cca::GM_CG && g_dqdk_wpc3.load() != 0;
        const GmapPlan gc = gmap_plan(B * W, Cq, one ? 2 : 1);
        cca::GmapJob<float, float> jc{q, nullptr, pk, qbs, pbs, qps, Cq, gc.grid};
        jc.red_src = red.src; jc.red_n = red.n; jc.red_dst = red.dst;
        if (one)
            CCA_LAUNCH((cca::gmap_kernel<132, false, false, false, float, float, false, true, 2, false, true>), dim3(cca::gmap_dual_grid(gc.grid)), dim3(cca::GS_THREADS),
                       stream, dE, k, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, pq, Cq, H, W, kbs, kps, 0L, 0,
                       0L, 0, pbs, Cq, gc.n_whole, gc.split, jc);
        else
        CCA_LAUNCH((cca::gmap_kernel<132, false, false, false, float, float, false, true, 1, false, true>), dim3(cca::gmap_dual_grid(gc.grid)), dim3(cca::GS_THREADS),
                   stream, dE, k, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, pq, Cq, H, W, kbs, kps, 0L, 0,
                   0L, 0, pbs, Cq, gc.n_whole, gc.split, jc);
        if (int e = launch_status("gmap_dual_f32(column, 132)")) return e;
    }
    if (W > 132) {          // long rows: one launch per block of the contracted positions, the partials updated in place
        const bool p100 = W <= 400;          // (blocks of <= 100 positions on the two-workgroups-per-CU kernels, see launch_gmap_planes_long_rows)
        const int nb = p100 ? (W + 99) / 100 : long_blocks(W);
        const GmapPlan gl = gmap_plan(B * H * nb, Cq, p100 ? 2 : 1);
        for (int j = 0; j < nb; ++j) {
            const bool last = j + 1 == nb;
            const cca::GmapJob<float, float> jl{q, pk, last ? dk : pk, qbs, last ? dkbs : pbs, qps, last ? dkps : Cq, gl.grid, nb, j};
            if (p100)
                CCA_LAUNCH((cca::gmap_kernel<100, true, false, true, float, float, false, true, 2, true, true>), dim3(cca::gmap_dual_grid(gl.grid)),
                           dim3(cca::GS_THREADS), stream, dE, k, (const float *)pq, (const float *)nullptr, (const float *)nullptr,
                           last ? dq : pq, Cq, H, W, kbs, kps, pbs, Cq, 0L, 0, last ? dqbs : pbs, last ? dqps : Cq, gl.n_whole, gl.split, jl);
            else
                CCA_LAUNCH((cca::gmap_kernel<132, true, false, true, float, float, false, true, 1, true, true>), dim3(cca::gmap_dual_grid(gl.grid)),
                           dim3(cca::GS_THREADS), stream, dE, k, (const float *)pq, (const float *)nullptr, (const float *)nullptr,
                           last ? dq : pq, Cq, H, W, kbs, kps, pbs, Cq, 0L, 0, last ? dqbs : pbs, last ? dqps : Cq, gl.n_whole, gl.split, jl);
            if (int e = launch_status("gmap_dual_f32(long rows)")) return e;
        }
        return 0;
    }
    if (Cq <= cca::GM_CG && g_dqdk_wpc3.load() != 0) {
        const GmapPlan gr2 = gmap_plan(B * H, Cq, 2);
        const cca::GmapJob<float, float> jr2{q, pk, dk, qbs, dkbs, qps, dkps, gr2.grid};
        CCA_LAUNCH((cca::gmap_kernel<132, true, false, true, float, float, false, true, 2, false, true>), dim3(cca::gmap_dual_grid(gr2.grid)), dim3(cca::GS_THREADS),
                   stream, dE, k, (const float *)pq, (const float *)nullptr, (const float *)nullptr, dq, Cq, H, W, kbs, kps, pbs, Cq,
                   0L, 0, dqbs, dqps, gr2.n_whole, gr2.split, jr2);
        return launch_status("gmap_dual_f32(row, 132, one slot)");
    }
    const cca::GmapJob<float, float> jr{q, pk, dk, qbs, dkbs, qps, dkps, gr.grid};
    CCA_LAUNCH((cca::gmap_kernel<132, true, false, true, float, float, false, true, 1, false, true>), dim3(cca::gmap_dual_grid(gr.grid)), dim3(cca::GS_THREADS),
               stream, dE, k, (const float *)pq, (const float *)nullptr, (const float *)nullptr, dq, Cq, H, W, kbs, kps, pbs, Cq,
               0L, 0, dqbs, dqps, gr.n_whole, gr.split, jr);
    return launch_status("gmap_dual_f32(row, 132)");
}
size_t planes_bytes(int B, int C, int H, int W) { return (size_t)B * H * W * 2 * C * 2; }
}  // namespace
}  // extern "C++"

static size_t ws_planes_bytes(int B, int C, int Cq, int H, int W, int backward) {
    const size_t base = pm_workspace_bytes(B, C, Cq, H, W, backward);
    if (!base) return 0;
    return align256(base) + (backward ? planes_bytes(B, C, H, W) : 0);                       /* + dy as planes */
}

static size_t ws_split_colsum_bytes(int B, int C, int H, int W);
static size_t ws_planes3_bytes(int B, int C, int Cq, int H, int W);

size_t ccnet_cca_workspace_bytes(int entry, int B, int C, int Cq, int H, int W) {
    switch (entry) {
    case CCNET_WS_SOFTMAX_BACKWARD: return ws_softmax_backward_bytes(B, H, W);
    case CCNET_WS_FORWARD: return ws_forward_bytes(B, C, Cq, H, W);
    case CCNET_WS_BACKWARD: return ws_backward_bytes(B, C, Cq, H, W);
    case CCNET_WS_PM_FORWARD: return ws_pm_bytes(B, C, Cq, H, W, 0);
    case CCNET_WS_PM_BACKWARD: return ws_pm_bytes(B, C, Cq, H, W, 1);
    case CCNET_WS_PLANES_FORWARD: return ws_planes_bytes(B, C, Cq, H, W, 0);
    case CCNET_WS_PLANES_BACKWARD: return ws_planes_bytes(B, C, Cq, H, W, 1);
    case CCNET_WS_SPLIT_COLSUM: return ws_split_colsum_bytes(B, C, H, W);
    case CCNET_WS_PLANES3_BACKWARD: return ws_planes3_bytes(B, C, Cq, H, W);
    }
    return 0;
}

int ccnet_cca_split_planes_f32(const float *src, uint16_t *dst, int B, int C, int H, int W, long src_bs, int src_ps,
                               long dst_bs, int dst_ps, int layout, const float *bias, ccnet_stream_t stream) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (!src || !dst) return fail(CCNET_E_NULLPTR, "split_planes: null tensor");
    cca::PlaneLayout pl;
    if (!plane_layout(layout, C, &pl)) return fail(CCNET_E_BADFLAGS, "split_planes: layout is one of CCNET_PLANES_*");
    if (int e = check_pm_view<float>("split_planes: source view (fp32 pixel-major)", src_bs, src_ps, C, H, W)) return e;
    if (int e = check_planes_view("split_planes: destination view (C % 8, pixel stride >= planes * C)", dst_bs, dst_ps, C, H, W, pl.width / C)) return e;
    const int hw = H * W;
    const long items = (long)hw * (C / 8);
    const unsigned gx = (unsigned)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
    CCA_LAUNCH(cca::pm_split_kernel, dim3(gx, (unsigned)B), dim3(256), stream, src, (bf16p_t *)dst, C, hw, src_bs, src_ps, dst_bs, dst_ps, pl, bias);
    return launch_status("split_planes");
}

// grid of the split + column-sum launch: x extent a multiple of g = cpp / gcd(cpp, 256) (a thread then keeps its 8 channels), about
// five workgroups per CU in all
static unsigned split_colsum_gx(int B, int C, int H, int W) {
    const int cpp = C / 8;
    int a = cpp, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const int g = cpp / a;
    const long items = (long)H * W * cpp;
    long want = 1280 / (B > 0 ? B : 1);
    if (want > (items + 255) / 256) want = (items + 255) / 256;
    long gx = want / g * g;
    if (gx < g) gx = g;
    return (unsigned)gx;
}
static size_t ws_split_colsum_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || C % 8 || H <= 0 || W <= 0) return 0;
    return (size_t)split_colsum_gx(B, C, H, W) * B * C * sizeof(float);
}

int ccnet_cca_split_planes_colsum_f32(const float *src, uint16_t *dst, float *colsum, void *workspace, size_t workspace_bytes,
                                      int B, int C, int H, int W, long src_bs, int src_ps, long dst_bs, int dst_ps, int layout,
                                      ccnet_stream_t stream) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (!src || !dst || !colsum) return fail(CCNET_E_NULLPTR, "split_planes_colsum: null tensor");
    cca::PlaneLayout pl;
    if (!plane_layout(layout, C, &pl)) return fail(CCNET_E_BADFLAGS, "split_planes_colsum: layout is one of CCNET_PLANES_*");
    if (int e = check_pm_view<float>("split_planes_colsum: source view (fp32 pixel-major)", src_bs, src_ps, C, H, W)) return e;
    if (int e = check_planes_view("split_planes_colsum: destination view (C % 8, pixel stride >= planes * C)", dst_bs, dst_ps, C, H, W, pl.width / C)) return e;
    if (!workspace || workspace_bytes < ws_split_colsum_bytes(B, C, H, W))
        return fail(CCNET_E_WORKSPACE, "split_planes_colsum: workspace missing or too small (CCNET_WS_SPLIT_COLSUM)");
    const unsigned gx = split_colsum_gx(B, C, H, W);
    CCA_LAUNCH(cca::pm_split_colsum_kernel, dim3(gx, (unsigned)B), dim3(256), stream, src, (bf16p_t *)dst, C, H * W, src_bs, src_ps,
               dst_bs, dst_ps, pl, (float *)workspace);
    if (int e = launch_status("split_planes_colsum")) return e;
    CCA_LAUNCH(cca::colsum_reduce_kernel, dim3((unsigned)((C + 15) / 16)), dim3(16 * cca::CS_LANES), stream, (const float *)workspace, (int)(gx * B), C, colsum);
    return launch_status("split_planes_colsum(reduce)");
}

int ccnet_cca_nchw_to_planes_f32(const float *src, uint16_t *dst, int B, int C, int H, int W, long src_bs, long dst_bs, int dst_ps,
                                 int layout, ccnet_stream_t stream) {
    if (int e = check_shape(B, C, H, W)) return e;
    if (!src || !dst) return fail(CCNET_E_NULLPTR, "nchw_to_planes: null tensor");
    cca::PlaneLayout pl;
    if (!plane_layout(layout, C, &pl)) return fail(CCNET_E_BADFLAGS, "nchw_to_planes: layout is one of CCNET_PLANES_*");
    if (src_bs < (long)C * H * W) return fail(CCNET_E_BADSHAPE, "nchw_to_planes: source batch stride");
    if (int e = check_planes_view("nchw_to_planes: destination view (C % 8, pixel stride >= planes * C)", dst_bs, dst_ps, C, H, W, pl.width / C)) return e;
    const int hw = H * W;
    CCA_LAUNCH(cca::nchw_to_planes_kernel, dim3((unsigned)(B * ((hw + 63) / 64)), (unsigned)((C + 63) / 64)), dim3(256), stream,
               src, (bf16p_t *)dst, C, hw, src_bs, dst_bs, dst_ps, pl);
    return launch_status("nchw_to_planes");
}

int ccnet_cca_pack_projection_f32(const float *wq, const float *bq, const float *wk, const float *bk, const float *wv, const float *bv,
                                  float *w, float *b, uint16_t *w3, uint16_t *w3t, int C, int Cq, ccnet_stream_t stream) {
    if (!wq || !bq || !wk || !bk || !wv || !bv || !w || !b) return fail(CCNET_E_NULLPTR, "pack_projection: null tensor");
    if ((w3 == nullptr) != (w3t == nullptr)) return fail(CCNET_E_NULLPTR, "pack_projection: w3 and w3t come together");
    if (C <= 0 || Cq <= 0 || (double)(2 * Cq + C) * C * 3 >= 2147483648.0) return fail(CCNET_E_BADSHAPE, "pack_projection: channel counts");
    const long items = (long)(2 * Cq + C) * C;
    const unsigned gx = (unsigned)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
    CCA_LAUNCH(cca::pack_projection_kernel, dim3(gx), dim3(256), stream, wq, wk, wv, bq, bk, bv, w, b, w3, w3t, C, Cq);
    return launch_status("pack_projection");
}

static int launch_proj_gemm(const cca::ProjGemmJob &job, const char *what, ccnet_stream_t stream) {
    const long tiles = (long)((job.M + cca::PG_BM - 1) / cca::PG_BM) * ((job.N + cca::PG_BN - 1) / cca::PG_BN) * job.batches;
    if (tiles >= 2147483647L) return fail(CCNET_E_BADSHAPE, "projection GEMM: too many tiles");
    if (job.K % cca::PG_BK) CCA_LAUNCH(cca::proj_gemm_kernel<true>, dim3((unsigned)tiles), dim3(cca::PG_THREADS), stream, job);
    else                    CCA_LAUNCH(cca::proj_gemm_kernel<false>, dim3((unsigned)tiles), dim3(cca::PG_THREADS), stream, job);
    return launch_status(what);
}

/* functions.py:29,32,35 as one stacked GEMM on bf16 operands with fp32 accumulation and output (cca_gemm.hpp):
 * out[m][n] = sum_k a[m][k] * wt[n][k] + bias[n].  ``a``: (M, K) bf16, row stride lda; ``wt``: (N, K) bf16, row stride ldw; ``out``:
 * (M, N) fp32, row stride ldo (elements; K, lda, ldw % 8 == 0, ldo % 4 == 0); ``bias`` may be NULL. */
int ccnet_cca_projection_bf16(const uint16_t *a, const uint16_t *wt, const float *bias, float *out, int M, int N, int K,
                              long lda, long ldw, long ldo, ccnet_stream_t stream) {
    if (!a || !wt || !out) return fail(CCNET_E_NULLPTR, "projection_bf16: null tensor");
    if (M <= 0 || N <= 0 || K <= 0 || K % 8 || lda % 8 || ldw % 8 || ldo % 4 || lda < K || ldw < K || ldo < N)
        return fail(CCNET_E_BADSHAPE, "projection_bf16: K, lda, ldw % 8 == 0, ldo % 4 == 0, strides >= extents");
    if ((double)M * lda >= 1073741824.0 || (double)N * ldw >= 1073741824.0 || (double)M * ldo >= 536870912.0)
        return fail(CCNET_E_BADSHAPE, "projection_bf16: 31-bit byte offsets");
    cca::ProjGemmJob job{(const cca::bf16_t *)a, (const cca::bf16_t *)wt, bias, nullptr, out, M, N, K, (int)lda, (int)ldw, (int)ldo, 1, 0, 0, 0};
    return launch_proj_gemm(job, "projection_bf16", stream);
}

/* The adjoint of the stacked projection with respect to its input (the backward-data of functions.py:29,32,35), NCHW:
 * dx[b][c][p] = sum_k w[c][k] * d[b][p][k] + add[b][c][p] -- ``w`` (C, K) bf16 row stride ldw_ (``w3t`` of ccnet_cca_pack_projection_f32,
 * K = 3 (2 Cq + C)), ``d`` (B, P, K) bf16, pixel stride ldd, batch stride d_bs (dq | dk | dv as three planes per pixel,
 * ccnet_cca_backward_planes3_f32), ``add`` (B, C, P) fp32 or NULL (dy: the residual branch), ``dx`` (B, C, P) fp32, both contiguous.
 * The same kernel as ccnet_cca_projection_bf16, one launch over B products; the addend starts the accumulators. */
int ccnet_cca_projection_adjoint_bf16(const uint16_t *w, const uint16_t *d, const float *add, float *dx, int B, int C, int P, int K,
                                      long ldw_, long ldd, long d_bs, ccnet_stream_t stream) {
    if (!w || !d || !dx) return fail(CCNET_E_NULLPTR, "projection_adjoint_bf16: null tensor");
    if (B <= 0 || C <= 0 || P <= 0 || K <= 0 || K % 8 || ldw_ % 8 || ldd % 8 || d_bs % 8 || ldw_ < K || ldd < K || d_bs < (long)(P - 1) * ldd + K)
        return fail(CCNET_E_BADSHAPE, "projection_adjoint_bf16: K, ldw, ldd, d_bs % 8 == 0, strides >= extents");
    if ((double)C * ldw_ >= 1073741824.0 || (double)P * ldd >= 1073741824.0 || (double)C * P >= 536870912.0)
        return fail(CCNET_E_BADSHAPE, "projection_adjoint_bf16: 31-bit byte offsets per image");
    cca::ProjGemmJob job{(const cca::bf16_t *)w, (const cca::bf16_t *)d, nullptr, add, dx, C, P, K, (int)ldw_, (int)ldd, P, B, 0, d_bs, (long)C * P};
    return launch_proj_gemm(job, "projection_adjoint_bf16", stream);
}

/* The adjoint of the stacked projection with respect to its weight (the backward-weight of functions.py:29,32,35):
 * sum_s part[s][n][c] = sum_r d[r][n] * x[r][c] over R rows -- ``d`` (R, N) bf16 row stride ldd: dq | dk | dv as three planes per pixel,
 * ``x`` (R, C) bf16 row stride ldx: x as three planes per pixel (row r of one pairs with row r of the other: [dh | dl | dh] against
 * [xh | xh | xl]) --, cut into S slabs of rows; ``part`` (S, N, C) fp32 receives one partial sum per slab (every element written:
 * a slab without rows writes zeros) and the caller adds them in a fixed order.  N, C, ldd, ldx % 8 == 0. */
int ccnet_cca_projection_wgrad_bf16(const uint16_t *d, const uint16_t *x, float *part, int R, int N, int C, long ldd, long ldx, int S,
                                    ccnet_stream_t stream) {
    if (!d || !x || !part) return fail(CCNET_E_NULLPTR, "projection_wgrad_bf16: null tensor");
    if (R <= 0 || N <= 0 || C <= 0 || S <= 0 || N % 8 || C % 8 || ldd % 8 || ldx % 8 || ldd < N || ldx < C)
        return fail(CCNET_E_BADSHAPE, "projection_wgrad_bf16: N, C, ldd, ldx % 8 == 0, strides >= extents, S >= 1");
    if ((double)R * ldd >= 1073741824.0 || (double)R * ldx >= 1073741824.0 || (double)N * C >= 536870912.0)
        return fail(CCNET_E_BADSHAPE, "projection_wgrad_bf16: 31-bit byte offsets");
    const int slab = (int)(((long)R + (long)S * cca::PG_BK - 1) / ((long)S * cca::PG_BK)) * cca::PG_BK;
    const long tiles = (long)((N + cca::PW_BN - 1) / cca::PW_BN) * ((C + cca::PW_BC - 1) / cca::PW_BC) * S;
    if (tiles >= 2147483647L) return fail(CCNET_E_BADSHAPE, "projection_wgrad_bf16: too many tiles");
    cca::ProjWgradJob job{(const cca::bf16_t *)d, (const cca::bf16_t *)x, part, R, N, C, (int)ldd, (int)ldx, S, slab};
    CCA_LAUNCH(cca::proj_wgrad_kernel, dim3((unsigned)tiles), dim3(cca::PG_THREADS), stream, job);
    return launch_status("projection_wgrad_bf16");
}

int ccnet_cca_forward_planes_f32(const float *q, const float *k, const float *v, const float *v_bias, uint16_t *v_planes,
                                 const float *x, const float *gamma, float *y, float *A,
                                 int B, int C, int Cq, int H, int W,
                                 long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long vp_bs, int vp_ps,
                                 void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    if (int e = require_both_branches("cca_forward_planes_f32")) return e;
    if (!q || !k || (!v_planes && !v) || !x || !gamma || !y || !A) return fail(CCNET_E_NULLPTR, "cca_forward_planes: null tensor");
    if (int e = check_planes_problem("cca_forward_planes: strips <= 528 positions (> 132: C/8 <= 64), C % 8 == 0, Cq % 4 == 0",
                                     B, C, Cq, H, W, true)) return e;
    if (int e = check_pm_view<float>("cca_forward_planes: q view", q_bs, q_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<float>("cca_forward_planes: k view", k_bs, k_ps, Cq, H, W)) return e;
    if (!workspace || workspace_bytes < ws_planes_bytes(B, C, Cq, H, W, 0))
        return fail(CCNET_E_WORKSPACE, "cca_forward_planes: workspace missing or too small");
    // (argument errors of the aggregation half must surface BEFORE the first launch)
    const bool direct = !v_planes;
    if (direct && ((H > W ? H : W) > 100 || v_bias))
        return fail(CCNET_E_BADSHAPE, "cca_forward_planes: the plane-free form (v_planes == NULL) serves strips <= 100 without a bias");
    if (v) if (int e = check_pm_view<float>("cca_forward_planes: v view (fp32 pixel-major)", v_bs, v_ps, C, H, W)) return e;
    if (!direct) if (int e = check_planes_view("cca_forward_planes: v planes view", vp_bs, vp_ps, C, H, W)) return e;
    if ((double)C * H * W >= 536870912.0) return fail(CCNET_E_BADSHAPE, "cca_forward_planes: image exceeds 2^29 elements");
    if (int e = gweight_energies_f32(q, k, A, B, Cq, H, W, q_bs, q_ps, k_bs, k_ps, stream)) return e;
    if (int e = softmax_forward(A, A, B, H, W, stream)) return e;
    // functions.py:42-49: aggregation + epilogue.  Longer strips: v (fp32, the value slice of the projection) -> planes first, inside the
    // entry point (every pass the op needs belongs to the op), on the caller's stream: running it on the side stream NEXT
    // TO the affinity launch was measured and lost (fwd 0.334 -> 0.346 ms: profiles/r04m_ab_two_stage_lds_staging.txt, "split-*" rows).
    if (v && !direct)
        if (int e = ccnet_cca_split_planes_f32(v, v_planes, B, C, H, W, v_bs, v_ps, vp_bs, vp_ps, CCNET_PLANES_HL, v_bias, stream)) return e;
    const long img = (long)C * H * W;
    if (direct) return launch_gmap_direct_f32(A, v, x, gamma, y, (float *)workspace, B, C, H, W, v_bs, v_ps, img, img, stream);
    return launch_gmap_planes<false, true>(A, (const bf16p_t *)v_planes, x, gamma, y, (float *)workspace, B, C, H, W, vp_bs, vp_ps,
                                           img, 0, img, 0, stream);
}

/* The attention tensor alone from pixel-major q, k views (what the pixel-major / split-plane forwards leave in ``A``): the host
 * calls it in the backward pass when it did NOT keep A (recompute instead of save). */
int ccnet_cca_attention_pm(const void *q, const void *k, float *A, int bf16, int B, int Cq, int H, int W,
                           long q_bs, int q_ps, long k_bs, int k_ps, ccnet_stream_t stream) {
    if (int e = require_both_branches("cca_attention_pm")) return e;
    if (!q || !k || !A) return fail(CCNET_E_NULLPTR, "cca_attention_pm: null tensor");
    if (bf16) {
        if (int e = check_pm_problem<bf16_t>("cca_attention_pm: strip length / channel divisibility (see ccnet_cca.h)", B, Cq, Cq, H, W)) return e;
        if (int e = check_pm_view<bf16_t>("cca_attention_pm: q view", q_bs, q_ps, Cq, H, W)) return e;
        if (int e = check_pm_view<bf16_t>("cca_attention_pm: k view", k_bs, k_ps, Cq, H, W)) return e;
        if (int e = gweight_pm<true, bf16_t>((const bf16_t *)q, (const bf16_t *)k, A, B, Cq, H, W, q_bs, q_ps, k_bs, k_ps, stream)) return e;
    } else {
        if (int e = check_planes_problem("cca_attention_pm: strips <= 528 positions (> 132: C/8 <= 64), Cq % 4 == 0", B, 8 * Cq, Cq, H, W, true)) return e;
        if (int e = check_pm_view<float>("cca_attention_pm: q view", q_bs, q_ps, Cq, H, W)) return e;
        if (int e = check_pm_view<float>("cca_attention_pm: k view", k_bs, k_ps, Cq, H, W)) return e;
        if (int e = gweight_energies_f32((const float *)q, (const float *)k, A, B, Cq, H, W, q_bs, q_ps, k_bs, k_ps, stream)) return e;
    }
    return softmax_forward(A, A, B, H, W, stream);
}

static size_t ws_planes3_bytes(int B, int C, int Cq, int H, int W) {
    const size_t base = ws_planes_bytes(B, C, Cq, H, W, 1);
    return base ? align256(base) + align256((size_t)B * H * (2 * Cq + C) * sizeof(float)) : 0;     /* + the column-sum partials */
}

// ``p3`` != nullptr: dq | dk | dv leave as three-plane rows + the bias gradients (ccnet_cca_backward_planes3_f32); dq / dk / dv unused
static int backward_planes_impl(const float *dy, const float *q, const float *k, const float *v, const uint16_t *v_planes,
                                const float *A, const float *gamma, float *dq, float *dk, float *dv, float *dgamma, float *scratch,
                                int B, int C, int Cq, int H, int W, long q_bs, int q_ps, long k_bs, int k_ps,
                                long v_bs, int v_ps, long vp_bs, int vp_ps, long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                                void *workspace, size_t workspace_bytes, ccnet_stream_t stream, const P3Out *p3) {
    if (int e = require_both_branches("cca_backward_planes_f32")) return e;
    if (!dy || !q || !k || (!v_planes && !v) || !A || !gamma || (!p3 && (!dq || !dk || !dv)) || !dgamma || !scratch)
        return fail(CCNET_E_NULLPTR, "cca_backward_planes: null tensor");
    const bool direct = !v_planes;                 // v as the fp32 pixel-major tensor (what a plane-free forward leaves)
    if (direct && (H > W ? H : W) > 100)
        return fail(CCNET_E_BADSHAPE, "cca_backward_planes: the plane-free form (v_planes == NULL) serves strips <= 100");
    if (int e = check_planes_problem("cca_backward_planes: strips <= 528 positions (> 132: C/8 <= 64), C % 8 == 0, Cq % 4 == 0",
                                     B, C, Cq, H, W, true)) return e;
    if (int e = check_pm_view<float>("cca_backward_planes: q view", q_bs, q_ps, Cq, H, W)) return e;
    if (int e = check_pm_view<float>("cca_backward_planes: k view", k_bs, k_ps, Cq, H, W)) return e;
    if (direct) { if (int e = check_pm_view<float>("cca_backward_planes: v view (fp32 pixel-major)", v_bs, v_ps, C, H, W)) return e; }
    else if (int e = check_planes_view("cca_backward_planes: v planes view", vp_bs, vp_ps, C, H, W)) return e;
    if (!p3) {
        if (int e = check_pm_view<float>("cca_backward_planes: dq view", dq_bs, dq_ps, Cq, H, W)) return e;
        if (int e = check_pm_view<float>("cca_backward_planes: dk view", dk_bs, dk_ps, Cq, H, W)) return e;
        if (int e = check_pm_view<float>("cca_backward_planes: dv view", dv_bs, dv_ps, C, H, W)) return e;
    }
    if (!workspace || workspace_bytes < (p3 ? ws_planes3_bytes(B, C, Cq, H, W) : ws_planes_bytes(B, C, Cq, H, W, 1)))
        return fail(CCNET_E_WORKSPACE, "cca_backward_planes: workspace missing or too small");
    const size_t sm = align256(ws_softmax_backward_bytes(B, H, W));
    const size_t base = align256(pm_workspace_bytes(B, C, Cq, H, W, 1));
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + sm);
    uint16_t *dy_pl = reinterpret_cast<uint16_t *>(static_cast<char *>(workspace) + base);
    const long dbs = (long)H * W * 2 * C;
    // dy (NCHW, the module's gradient) -> planes, once: it is a contraction operand of four launches
    if (int e = ccnet_cca_nchw_to_planes_f32(dy, dy_pl, B, C, H, W, (long)C * H * W, dbs, 2 * C, CCNET_PLANES_HL, stream)) return e;
    const bf16p_t *dyp = (const bf16p_t *)dy_pl, *vp = (const bf16p_t *)v_planes;
    // dv (two C-sized, HBM-bound passes) is independent of the chain dA -> dE -> dq | dk: it runs on the library's side stream
    // next to that chain and joins before this function returns
    const int overlap = g_planes_overlap.load() < 0 ? 2 : g_planes_overlap.load();
    SideFork sf(stream);
    if (overlap == 2) sf.fork();
    // t = un-scaled dA (the adjoint of the aggregation, functions.py:46-47), dv = gamma * A^T-weighted dy
    int e = 0;
    if (W > 132 || H > 132) {           // long strips: (query block, key block) tiles per strip
        const int nb = long_blocks(W), nbc = long_blocks(H);
        const dim3 grid((unsigned)(B * (W * nbc * nbc + H * nb * nb))), block(cca::GM_THREADS);
        CCA_LAUNCH((cca::gweight_kernel<132, false, bf16p_t, false, true>), grid, block, stream, dyp, vp, scratch, C, H, W, dbs, 2 * C,
                   vp_bs, vp_ps, nb, nbc);
        e = launch_status("gweight_planes(dA, long strips)");
    } else if ((H > W ? H : W) > 100) {
        const dim3 grid((unsigned)(B * (H + W))), block(cca::GM_THREADS);
        CCA_LAUNCH((cca::gweight_kernel<132, false, bf16p_t, false>), grid, block, stream, dyp, vp, scratch, C, H, W, dbs, 2 * C, vp_bs, vp_ps);
        e = launch_status("gweight_planes(dA, 132)");
    } else if (const int ps = g_planes_stream.load()) {
        // persistent: one workgroup per CU walks the strips of both branches, its ring runs across strip boundaries
        // (option values > 1 cap the number of workgroups: tests make one workgroup walk many strips)
        const int nstrips = B * (H + W), cus = ps > 1 ? ps : num_cus();
        const dim3 grid((unsigned)(nstrips < cus ? nstrips : cus)), block(cca::GM_THREADS);
        if (direct && g_da_stages.load() == 2)
            CCA_LAUNCH((cca::gweight_stream_kernel<100, bf16p_t, float, 2>), grid, block, stream, dyp, v, scratch, C, B, H, W, dbs, 2 * C, v_bs, v_ps);
        else if (direct) CCA_LAUNCH((cca::gweight_stream_kernel<100, bf16p_t, float>), grid, block, stream, dyp, v, scratch, C, B, H, W, dbs, 2 * C, v_bs, v_ps);
        else        CCA_LAUNCH((cca::gweight_stream_kernel<100>), grid, block, stream, dyp, vp, scratch, C, B, H, W, dbs, 2 * C, vp_bs, vp_ps);
        e = launch_status("gweight_stream(dA)");
    } else {
        if (direct) return fail(CCNET_E_BADFLAGS, "cca_backward_planes: the plane-free form runs the persistent dA kernel (option planes_stream != 0)");
        const dim3 grid((unsigned)(B * (H + W))), block(cca::GM_THREADS);
        CCA_LAUNCH((cca::gweight_kernel<100, false, bf16p_t, false>), grid, block, stream, dyp, vp, scratch, C, H, W, dbs, 2 * C, vp_bs, vp_ps);
        e = launch_status("gweight_planes(dA)");
    }
    if (overlap == 1) sf.fork();
    if (!e && p3) {
        // dv: the column pass as ever (ring kernel -> fp32 partial), the row pass writes channels [2 Cq, 2 Cq + C) of the three planes
        e = launch_gmap3_planes<100, true>(A, dyp, gamma, nullptr, partial, B, C, H, W, dbs, 2 * C, 0L, 0, false, sf.stream());
        if (!e) {
            const long pbs = (long)H * W * C;
            const GmapPlan gr = gmap_plan(B * H, C, 2);
            cca::GmapJob<bf16p_t, float> job{};
            job.p3_plane = p3->plane;
            job.cs = p3->cs + 2 * Cq;
            job.cs_stride = p3->plane;
            CCA_LAUNCH((cca::gmap_kernel<100, true, true, true, bf16p_t, float, false, false, 2, false, false, false, true>), dim3((unsigned)gr.grid),
                       dim3(cca::GS_THREADS), sf.stream(), A, dyp, (const float *)partial, (const float *)nullptr, gamma,
                       reinterpret_cast<float *>(p3->d3 + 2 * Cq), C, H, W, dbs, 2 * C, pbs, C, 0L, 0, p3->bs, p3->ps, gr.n_whole, gr.split, job);
            e = launch_status("gmap_planes(row, three-plane output)");
        }
    } else
    if (!e) e = launch_gmap_planes<true, false>(A, dyp, nullptr, gamma, dv, partial, B, C, H, W, dbs, 2 * C, 0L, 0, dv_bs, dv_ps, sf.stream());
    // dgamma = sum A t;  dE = gamma * A * (t - sum_s A t), in place
    // (the fixed-order sum of the dgamma partials rides on the dq | dk column launch: one launch less)
    DeferredSum red{static_cast<const float *>(workspace), 0, dgamma};
    if (!e) e = softmax_backward_impl(A, scratch, gamma, scratch, dgamma, workspace, sm, B, H, W, stream, KSplit(), &red.n);
    if (!e && p3) {
        const long pbs = (long)H * W * Cq;
        float *pq = partial_qk_of(partial, B, C, H, W), *pk = pq + (size_t)B * pbs;
        e = launch_gmap_dual_pair<100, float, true, 3, true>(scratch, k, q, nullptr, nullptr, pq, pk, pbs, B, Cq, H, W, k_bs, k_ps, q_bs, q_ps,
                                                             0L, 0, 0L, 0, stream, red, p3);
    } else
    if (!e) e = gmap_dual_f32(scratch, k, q, dq, dk, partial_qk_of(partial, B, C, H, W), B, Cq, H, W, k_bs, k_ps, q_bs, q_ps,
                              dq_bs, dq_ps, dk_bs, dk_ps, stream, red);
    return sf.join(e);
}

int ccnet_cca_backward_planes_f32(const float *dy, const float *q, const float *k, const float *v, const uint16_t *v_planes,
                                  const float *A, const float *gamma, float *dq, float *dk, float *dv, float *dgamma, float *scratch,
                                  int B, int C, int Cq, int H, int W, long q_bs, int q_ps, long k_bs, int k_ps,
                                  long v_bs, int v_ps, long vp_bs, int vp_ps, long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                                  void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    return backward_planes_impl(dy, q, k, v, v_planes, A, gamma, dq, dk, dv, dgamma, scratch, B, C, Cq, H, W, q_bs, q_ps, k_bs, k_ps, v_bs, v_ps,
                                vp_bs, vp_ps, dq_bs, dq_ps, dk_bs, dk_ps, dv_bs, dv_ps, workspace, workspace_bytes, stream, nullptr);
}

/* The same backward for a caller whose next operation is the split-bf16 projection adjoint (the module: dx = W^T dqkv^T and
 * dW = dqkv^T x as bf16 GEMMs on three-plane operands): dq | dk | dv are WRITTEN as the three-plane rows those GEMMs read --
 * CCNET_PLANES_HLH, (B, HW, 3, 2 Cq + C) bf16 with pixel stride ``d3_ps`` >= 3 (2 Cq + C) and batch stride ``d3_bs`` (elements) --
 * and ``dbias`` (2 Cq + C floats) receives their sums over all images and pixels (the bias gradients), added in a fixed order.
 * Plane-free form only: v fp32 pixel-major, strips <= 100, C/8 <= 64. */
int ccnet_cca_backward_planes3_f32(const float *dy, const float *q, const float *k, const float *v, const float *A, const float *gamma,
                                   uint16_t *d3, float *dbias, float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                                   long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long d3_bs, int d3_ps,
                                   void *workspace, size_t workspace_bytes, ccnet_stream_t stream) {
    if (!d3 || !dbias || !v) return fail(CCNET_E_NULLPTR, "cca_backward_planes3: null tensor");
    const int ct = 2 * Cq + C;
    if ((H > W ? H : W) > 100 || Cq > cca::GM_CG || Cq <= 0 || C % 8 || Cq % 4)
        return fail(CCNET_E_BADSHAPE, "cca_backward_planes3: strips <= 100, C/8 <= 64, C % 8 == 0, Cq % 4 == 0");
    if (d3_ps < 3 * ct || d3_ps % 4 || d3_bs < (long)(H * W - 1) * d3_ps + 3 * ct || d3_bs % 4 || (double)H * W * d3_ps >= 1073741824.0)
        return fail(CCNET_E_BADSHAPE, "cca_backward_planes3: d3 view");
    if (!g_dqdk_wpc3.load() || !g_dqdk_exact.load() || g_planes_ring.load() != 2 || !g_planes_stream.load())
        return fail(CCNET_E_BADFLAGS, "cca_backward_planes3: runs the default launch forms only (options dqdk_wpc3, dqdk_exact, planes_ring, planes_stream)");
    if (!workspace || workspace_bytes < ws_planes3_bytes(B, C, Cq, H, W))
        return fail(CCNET_E_WORKSPACE, "cca_backward_planes3: workspace missing or too small (CCNET_WS_PLANES3_BACKWARD)");
    P3Out p3;
    p3.d3 = d3; p3.bs = d3_bs; p3.ps = d3_ps; p3.plane = ct;
    p3.cs = reinterpret_cast<float *>(static_cast<char *>(workspace) + align256(ws_planes_bytes(B, C, Cq, H, W, 1)));
    if (int e = backward_planes_impl(dy, q, k, v, nullptr, A, gamma, nullptr, nullptr, nullptr, dgamma, scratch, B, C, Cq, H, W, q_bs, q_ps, k_bs, k_ps,
                                     v_bs, v_ps, 0L, 0, 0L, 0, 0L, 0, 0L, 0, workspace, workspace_bytes, stream, &p3)) return e;
    CCA_LAUNCH(cca::colsum_reduce_kernel, dim3((unsigned)((ct + 15) / 16)), dim3(16 * cca::CS_LANES), stream, (const float *)p3.cs, B * H, ct, dbias);
    return launch_status("colsum_reduce(three-plane backward)");
}

// Options: status in the return value, values through out-parameters (a value of -1 -- "planes_overlap" auto -- must
// not look like an error code).  Invalid values are rejected with CCNET_E_BADFLAGS and leave the option unchanged.
namespace {
struct OptionRange { const char *name; std::atomic<int> *word; int lo, hi; };
const OptionRange *find_word_option(const std::string &n) {
    static const OptionRange table[] = {
        {"planes_ring", &g_planes_ring, 0, 2},
        {"planes_stream", &g_planes_stream, 0, 1 << 20},
        {"planes_overlap", &g_planes_overlap, -1, 2},
        {"planes_xcd", &g_planes_xcd, 0, 1},
        {"energy_tail", &g_energy_tail, 0, 1},
        {"dqdk_wpc3", &g_dqdk_wpc3, 0, 1},
        {"da_stages", &g_da_stages, 2, 3},
        {"dqdk_exact", &g_dqdk_exact, 0, 1},
        {"bf16_partial", &g_bf16_partial, 0, 1},
    };
    for (const OptionRange &o : table)
        if (n == o.name) return &o;
    return nullptr;
}
}  // namespace

int ccnet_cca_get_option(const char *name, int *value) {
    if (!name || !value) return fail(CCNET_E_NULLPTR, "get_option: null name / value");
    const std::string n(name);
    if (n == "impl") *value = get_impl();
    else if (n == "precision") *value = get_precision();
    else if (n == "branch_mask") *value = g_branch_mask.load();
    else if (const OptionRange *o = find_word_option(n)) *value = o->word->load();
    else return fail(CCNET_E_BADFLAGS, "get_option: unknown option");
    return 0;
}

int ccnet_cca_set_option(const char *name, int value, int *previous) {
    if (!name) return fail(CCNET_E_NULLPTR, "set_option: null name");
    const std::string n(name);
    int prev = 0;
    if (n == "impl") {
        if (value != CCNET_IMPL_AUTO && value != CCNET_IMPL_DIRECT && value != CCNET_IMPL_MFMA) return fail(CCNET_E_BADFLAGS, "set_option: impl is one of CCNET_IMPL_*");
        prev = set_impl(value);
    } else if (n == "precision") {
        if (value != CCNET_PRECISION_F32 && value != CCNET_PRECISION_BF16X3 && value != CCNET_PRECISION_DEFAULT) return fail(CCNET_E_BADFLAGS, "set_option: precision is one of CCNET_PRECISION_*");
        prev = set_precision(value);
    } else if (n == "branch_mask") {
        if (value < 1 || value > 3) return fail(CCNET_E_BADFLAGS, "set_option: branch_mask is one of CCNET_BRANCH_*");
        prev = set_branch_mask(value);
    } else if (const OptionRange *o = find_word_option(n)) {
        if (value < o->lo || value > o->hi) return fail(CCNET_E_BADFLAGS, "set_option: value out of range for this option");
        prev = o->word->exchange(value);
    } else {
        return fail(CCNET_E_BADFLAGS, "set_option: unknown option");
    }
    if (previous) *previous = prev;
    return 0;
}

/* ---- launch profiler: per-launch HIP-event durations inside a step (see cca_platform.hpp, cca_prof) ---- */
int ccnet_cca_profile_begin(int max_launches) {
    const char *why = cca_prof::begin(max_launches);
    return why ? fail(CCNET_E_BADFLAGS, why) : 0;
}

int ccnet_cca_profile_end(float *ms, char *names, int name_stride, int cap) {
    const char *why = nullptr;
    const int n = cca_prof::end(ms, names, name_stride, cap, &why);
    return why ? fail(CCNET_E_BADFLAGS, why) : n;            /* number of launches recorded (>= 0) */
}

/* ---- device-state probes (measurement aids, see cca_probe.hpp): nothing here is on the product path ---- */
int ccnet_cca_probe_clock(unsigned long long *samples, int nwg, int nsamples, int interval_ticks, ccnet_stream_t stream) {
    if (!samples) return fail(CCNET_E_NULLPTR, "probe_clock: null buffer");
    if (nwg < 1 || nwg > 1024 || nsamples < 2 || nsamples > (1 << 20) || interval_ticks < 1) return fail(CCNET_E_BADFLAGS, "probe_clock: 1..1024 workgroups, >= 2 samples, interval >= 1 tick");
    // the kernel busy-waits nsamples x interval ticks of the 100 MHz reference clock: bounded to two seconds (an exported
    // symbol must not be able to hang the device)
    if ((long long)nsamples * interval_ticks > 200000000LL) return fail(CCNET_E_BADFLAGS, "probe_clock: nsamples x interval_ticks <= 2e8 (two seconds)");
    CCA_LAUNCH(cca::probe_clock_kernel, dim3((unsigned)nwg), dim3(cca::kWave), stream, samples, nsamples, interval_ticks);
    return launch_status("probe_clock");
}

int ccnet_cca_probe_mfma(unsigned long long *clk, float *sink, int nwg, int iters, ccnet_stream_t stream) {
    if (!clk || !sink) return fail(CCNET_E_NULLPTR, "probe_mfma: null buffer");
    if (nwg < 1 || nwg > 65535 || iters < 1) return fail(CCNET_E_BADFLAGS, "probe_mfma: 1..65535 workgroups, iters >= 1");
    CCA_LAUNCH(cca::probe_mfma_kernel, dim3((unsigned)nwg), dim3(256), stream, clk, sink, iters);
    return launch_status("probe_mfma");
}

int ccnet_cca_probe_dma(const float *src, size_t src_bytes, unsigned long long *clk, int nwg, int reps, int row_stride_bytes,
                        ccnet_stream_t stream) {
    if (!src || !clk) return fail(CCNET_E_NULLPTR, "probe_dma: null buffer");
    if (nwg < 1 || nwg > 65535 || reps < 1 || row_stride_bytes < 256 || row_stride_bytes % 4) return fail(CCNET_E_BADFLAGS, "probe_dma: 1..65535 workgroups, reps >= 1, row stride >= 256 B");
    if (src_bytes >= ((size_t)1 << 31)) src_bytes = ((size_t)1 << 31) - 256;                 /* 32-bit byte offsets */
    const long span_rows = (long)(src_bytes / (size_t)row_stride_bytes);
    if (span_rows < 101) return fail(CCNET_E_BADSHAPE, "probe_dma: the source must hold > 100 rows at this stride");
    CCA_LAUNCH(cca::probe_dma_kernel, dim3((unsigned)nwg), dim3(cca::GS_THREADS), stream, src, src_bytes, clk, reps, row_stride_bytes, (int)span_rows);
    return launch_status("probe_dma");
}

int ccnet_cca_mfma_selftest(float *scratch, ccnet_stream_t stream) {
    if (!scratch) return fail(CCNET_E_NULLPTR, "mfma_selftest: null scratch");
    CCA_LAUNCH(cca::mfma_selftest_kernel, dim3(1), dim3(cca::kWave), stream, scratch);
    if (int e = launch_status("mfma_selftest")) return e;
    float bad = -1.f;
    if (hipMemcpyAsync(&bad, scratch, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return fail(1, "mfma_selftest: copy-back failed");
    if (bad != 0.f) return fail(1000 + (int)bad, "mfma_selftest: f32 fragment layout mismatch");
    CCA_LAUNCH(cca::mfma_bf16_selftest_kernel, dim3(1), dim3(cca::kWave), stream, scratch);
    if (int e = launch_status("mfma_bf16_selftest")) return e;
    bad = -1.f;
    if (hipMemcpyAsync(&bad, scratch, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return fail(1, "mfma_selftest: copy-back failed");
    if (bad != 0.f) return fail(2000 + (int)bad, "mfma_selftest: bf16 fragment layout mismatch");
    return 0;
}

}  // extern "C"

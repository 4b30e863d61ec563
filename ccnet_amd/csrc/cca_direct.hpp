// cca_direct.hpp -- one-thread-per-output kernels for ANY (B, C, H, W).
//
// They implement the same six contractions as the strip kernels (cca_weight.hpp / cca_map.hpp)
// straight from the index definitions of /root/reference/cc_attention/functions.py:38-47 and its
// autograd.  They serve shapes the strip kernels do not cover (a strip longer than 100) and are the
// on-device cross-check of the MFMA path (tests/test_gpu_parity.py runs both).  Threads are laid out
// with w fastest so feature reads coalesce; accumulation is a plain fmaf chain in slot order.  The feature element
// type FT is float or bf16_t (bf16 storage, fp32 arithmetic: BASELINE configs[4]); attention-shaped tensors and
// gamma are always fp32.
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int D_BLOCK = 256;

// T[b,h,w,s] = sum_c X[b,c,h,w] * Y[b,c,src(s)];   column self slot -> -inf when MASK
template <bool MASK, typename FT = float>
__global__ __launch_bounds__(D_BLOCK) void direct_weight_kernel(const FT *X, const FT *Y, float *T,
                                                                int Cx, int H, int W, size_t total, long xbs, long ybs) {
    const int S = H + W, HW = H * W;
    for (size_t idx = (size_t)blockIdx.x * D_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * D_BLOCK) {
        // idx enumerates (b, h, s, w) with w fastest so that a wave reads contiguous w
        const int w = int(idx % W);
        size_t rest = idx / W;
        const int s = int(rest % S);
        rest /= S;
        const int h = int(rest % H);
        const int b = int(rest / H);
        const int sh = (s < H) ? s : h, sw = (s < H) ? w : s - H;
        const FT *xp = X + (size_t)b * xbs + (size_t)h * W + w;
        const FT *yp = Y + (size_t)b * ybs + (size_t)sh * W + sw;
        float acc = 0.f;
        for (int c = 0; c < Cx; ++c) acc = fmaf(load_f32(xp + (size_t)c * HW), load_f32(yp + (size_t)c * HW), acc);
        if (MASK && s == h) acc = -INFINITY;
        T[(((size_t)b * H + h) * W + w) * S + s] = acc;
    }
}

// out[b,c,h,w] = alpha * sum_s T[b,h,w,s] * F[b,c,src(s)] + resid
template <typename FT = float>
__global__ __launch_bounds__(D_BLOCK) void direct_map_kernel(const float *T, const FT *F, const FT *resid,
                                                             const float *gamma, FT *out,
                                                             int C, int H, int W, size_t total, long fbs, long rbs, long obs) {
    const int S = H + W, HW = H * W;
    const float alpha = gamma ? gamma[0] : 1.f;
    for (size_t idx = (size_t)blockIdx.x * D_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * D_BLOCK) {
        const int w = int(idx % W);
        size_t rest = idx / W;
        const int h = int(rest % H);
        rest /= H;
        const int c = int(rest % C);
        const int b = int(rest / C);
        const float *t = T + (((size_t)b * H + h) * W + w) * S;
        const FT *f = F + (size_t)b * fbs + (size_t)c * HW;
        float acc = 0.f;
        for (int j = 0; j < H; ++j) acc = fmaf(t[j], load_f32(f + (size_t)j * W + w), acc);
        for (int j = 0; j < W; ++j) acc = fmaf(t[H + j], load_f32(f + (size_t)h * W + j), acc);
        float val = alpha * acc;
        const size_t in_image = (size_t)c * HW + (size_t)h * W + w;
        if (resid) val += load_f32(resid + (size_t)b * rbs + in_image);
        store_f32(out + (size_t)b * obs + in_image, val);
    }
}

// out[b,c,j,w] = alpha * ( sum_h T[b,h,w,j] * F[b,c,h,w]  +  sum_w' T[b,j,w',H+w] * F[b,c,j,w'] )
template <typename FT = float>
__global__ __launch_bounds__(D_BLOCK) void direct_mapT_kernel(const float *T, const FT *F, const float *gamma,
                                                              FT *out, int C, int H, int W, size_t total,
                                                              long fbs, long obs) {
    const int S = H + W, HW = H * W;
    const float alpha = gamma ? gamma[0] : 1.f;
    for (size_t idx = (size_t)blockIdx.x * D_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * D_BLOCK) {
        const int w = int(idx % W);
        size_t rest = idx / W;
        const int j = int(rest % H);
        rest /= H;
        const int c = int(rest % C);
        const int b = int(rest / C);
        const float *tb = T + (size_t)b * HW * S;
        const FT *f = F + (size_t)b * fbs + (size_t)c * HW;
        float acc = 0.f;
        for (int h = 0; h < H; ++h) acc = fmaf(tb[((size_t)h * W + w) * S + j], load_f32(f + (size_t)h * W + w), acc);
        for (int w2 = 0; w2 < W; ++w2)
            acc = fmaf(tb[((size_t)j * W + w2) * S + H + w], load_f32(f + (size_t)j * W + w2), acc);
        store_f32(out + (size_t)b * obs + (size_t)c * HW + (size_t)j * W + w, alpha * acc);
    }
}

// Device self-test of the v_mfma_f32_16x16x4_f32 fragment layout assumed by cca_common.hpp.
// Asymmetric, exactly representable operands; result[0] = number of mismatching accumulator entries.
__global__ __launch_bounds__(kWave) void mfma_selftest_kernel(float *result) {
    const int l = lane_id();
    // A[i][k] = 1 + i/4 + 3k ; B[k][j] = 1/2 + 7k - j/8
    const float a = 1.f + 0.25f * (l & 15) + 3.f * (l >> 4);
    const float b = 0.5f + 7.f * (l >> 4) - 0.125f * (l & 15);
    f32x4 d = mfma_16x16x4(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
    float bad = 0.f;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float ref = 0.f;
        for (int k = 0; k < 4; ++k)
            ref = fmaf(1.f + 0.25f * row + 3.f * k, 0.5f + 7.f * k - 0.125f * col, ref);
        if (d[r] != ref) bad += 1.f;
    }
    bad = wave_sum(bad);
    if (l == 0) result[0] = bad;
}

// Same for v_mfma_f32_16x16x32_bf16 (A[i][k], B[k][j] with k = 8 (l >> 4) + e).  Small integers: exact in bf16.
__global__ __launch_bounds__(kWave) void mfma_bf16_selftest_kernel(float *result) {
    const int l = lane_id();
    float av[8], bv[8];
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (l >> 4) + e;
        av[e] = (float)(1 + (l & 15) + 2 * (k % 5));            // A[i][k]
        bv[e] = (float)(3 + 2 * k - (l & 15));                  // B[k][j]
    }
    const BfSplit a = bf16_split8(av), b = bf16_split8(bv);
    f32x4 d = mfma_bf16_16x16x32(a.hi, b.hi, f32x4{0.f, 0.f, 0.f, 0.f});
    float bad = 0.f;
    for (int e = 0; e < 4; ++e)
        if (a.lo[e] != 0u || b.lo[e] != 0u) bad += 100.f;       // small integers must split with a zero low part
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float ref = 0.f;
        for (int k = 0; k < 32; ++k) ref += (float)(1 + row + 2 * (k % 5)) * (float)(3 + 2 * k - col);
        if (d[r] != ref) bad += 1.f;
    }
    bad = wave_sum(bad);
    if (l == 0) result[0] = bad;
}

}  // namespace cca

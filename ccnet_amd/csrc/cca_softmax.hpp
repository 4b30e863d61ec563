// cca_softmax.hpp -- the (H+W)-wide masked softmax over attention slots and its adjoint.
//
// forward   A = softmax_s(E)                       /root/reference/cc_attention/functions.py:40
//           (the column self slot holds -inf -> exp() == 0 exactly, the reference's structural zero)
// backward  dE = g * A * (dA - sum_s A dA),  dgamma = sum_pixels sum_s A dA
//
// Slabs: the K-split weight kernels (small batches, cca_weight.hpp) leave nslab partial tensors -- slab 0 is the
// tensor itself, slab s >= 1 is extra + (s - 1) * slab_stride -- which these kernels add in slab order while loading
// (deterministic; -inf + -inf = -inf keeps the masked slot).
//
// One wavefront per pixel: the S = H+W slots of a pixel are contiguous (776 B at 97x97), each lane
// keeps ceil(S/64) of them in registers, max / sum are wave-64 butterfly reductions (no LDS, no atomics;
// the dgamma partials are combined in a fixed order so the result is run-to-run deterministic).
#pragma once
#include "cca_common.hpp"

namespace cca {

constexpr int SM_WAVES = 4;                       // pixels per workgroup
constexpr int SM_BLOCK = SM_WAVES * kWave;

template <int NREG>                               // S <= 64 * NREG
__global__ __launch_bounds__(SM_BLOCK) void softmax_fwd_kernel(const float *E, float *A, int npix, int S,
                                                               int nslab, const float *extra, long slab_stride) {
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const int pix = blockIdx.x * SM_WAVES + wv;
    if (pix >= npix) return;                      // wave-uniform
    const float *e = E + (size_t)pix * S;
    float *a = A + (size_t)pix * S;
    float v[NREG];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int s = lane + r * kWave;
        v[r] = (s < S) ? e[s] : -INFINITY;
        for (int sl = 1; sl < nslab; ++sl)
            if (s < S) v[r] += extra[(size_t)(sl - 1) * slab_stride + (size_t)pix * S + s];
        m = fmaxf(m, v[r]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        v[r] = expf(v[r] - m);
        sum += v[r];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int s = lane + r * kWave;
        if (s < S) a[s] = v[r] * inv;
    }
}

// any S: three passes over the pixel's slots (they stay in L1/L2)
__global__ __launch_bounds__(SM_BLOCK) void softmax_fwd_generic_kernel(const float *E, float *A, int npix, int S,
                                                                       int nslab, const float *extra, long slab_stride) {
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const int pix = blockIdx.x * SM_WAVES + wv;
    if (pix >= npix) return;
    const float *e = E + (size_t)pix * S;
    float *a = A + (size_t)pix * S;
    auto val = [&](int s) {
        float x = e[s];
        for (int sl = 1; sl < nslab; ++sl) x += extra[(size_t)(sl - 1) * slab_stride + (size_t)pix * S + s];
        return x;
    };
    float m = -INFINITY;
    for (int s = lane; s < S; s += kWave) m = fmaxf(m, val(s));
    m = wave_max(m);
    float sum = 0.f;
    for (int s = lane; s < S; s += kWave) sum += expf(val(s) - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    // (E may alias A: a lane reads every slot it writes before writing it)
    for (int s = lane; s < S; s += kWave) a[s] = expf(val(s) - m) * inv;
}

// The grid is capped (SM_MAX_BLOCKS) and strides over the pixels, so at most SM_MAX_BLOCKS partial sums of
// dgamma are produced; each wavefront adds its pixels in a fixed order: deterministic.
constexpr int SM_MAX_BLOCKS = 2048;

template <int NREG>
__global__ __launch_bounds__(SM_BLOCK) void softmax_bwd_kernel(const float *A, const float *dA,
                                                               const float *gamma, float *dE,
                                                               float *partials, int npix, int S,
                                                               int nslab, const float *extra, long slab_stride) {
    __shared__ float red[SM_WAVES];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const float g = gamma ? gamma[0] : 1.f;
    float wsum = 0.f;
    for (int pix = blockIdx.x * SM_WAVES + wv; pix < npix; pix += gridDim.x * SM_WAVES) {   // wave-uniform
        const float *a = A + (size_t)pix * S;
        const float *d = dA + (size_t)pix * S;
        float *o = dE + (size_t)pix * S;
        float av[NREG], dv[NREG];
        float rsum = 0.f;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const int s = lane + r * kWave;
            av[r] = (s < S) ? a[s] : 0.f;
            dv[r] = (s < S) ? d[s] : 0.f;
            for (int sl = 1; sl < nslab; ++sl)
                if (s < S) dv[r] += extra[(size_t)(sl - 1) * slab_stride + (size_t)pix * S + s];
            rsum += av[r] * dv[r];
        }
        rsum = wave_sum(rsum);
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const int s = lane + r * kWave;
            if (s < S) o[s] = g * av[r] * (dv[r] - rsum);
        }
        wsum += rsum;
    }
    if (partials) {
        if (lane == 0) red[wv] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < SM_WAVES; ++i) t += red[i];
            partials[blockIdx.x] = t;
        }
    }
}

__global__ __launch_bounds__(SM_BLOCK) void softmax_bwd_generic_kernel(const float *A, const float *dA,
                                                                       const float *gamma, float *dE,
                                                                       float *partials, int npix, int S,
                                                                       int nslab, const float *extra, long slab_stride) {
    __shared__ float red[SM_WAVES];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const float g = gamma ? gamma[0] : 1.f;
    float wsum = 0.f;
    for (int pix = blockIdx.x * SM_WAVES + wv; pix < npix; pix += gridDim.x * SM_WAVES) {   // wave-uniform
        const float *a = A + (size_t)pix * S;
        const float *d = dA + (size_t)pix * S;
        float *o = dE + (size_t)pix * S;
        auto dval = [&](int s) {
            float x = d[s];
            for (int sl = 1; sl < nslab; ++sl) x += extra[(size_t)(sl - 1) * slab_stride + (size_t)pix * S + s];
            return x;
        };
        float rsum = 0.f;
        for (int s = lane; s < S; s += kWave) rsum += a[s] * dval(s);
        rsum = wave_sum(rsum);
        for (int s = lane; s < S; s += kWave) o[s] = g * a[s] * (dval(s) - rsum);
        wsum += rsum;
    }
    if (partials) {
        if (lane == 0) red[wv] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < SM_WAVES; ++i) t += red[i];
            partials[blockIdx.x] = t;
        }
    }
}

// fixed-order reduction of the per-workgroup partial sums -> out[0]  (single workgroup)
__global__ __launch_bounds__(SM_BLOCK) void reduce_partials_kernel(const float *partials, int n, float *out) {
    __shared__ float red[SM_BLOCK];
    float t = 0.f;
    for (int i = threadIdx.x; i < n; i += SM_BLOCK) t += partials[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = SM_BLOCK / 2; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace cca

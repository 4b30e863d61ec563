// Development probe: times map_band_fwd_kernel (csrc/cca_band.hpp) with parts of it compiled out (-DBAND_ABL=mask).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../ccnet_amd/csrc -DBAND_ABL=<mask> band_abl.hip -o band_abl_<mask>
#include "cca_common.hpp"
#include "cca_band.hpp"   // (lives next to this probe since round 3; needs -I ../../ccnet_amd/csrc)
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = 512, H = 97, W = 97, S = H + W;
    const size_t nf = (size_t)B * C * H * W, na = (size_t)B * H * W * S;
    float *A, *v, *x, *y, *gamma;
    hipMalloc(&A, na * 4); hipMalloc(&v, nf * 4); hipMalloc(&x, nf * 4); hipMalloc(&y, nf * 4); hipMalloc(&gamma, 4);
    std::vector<float> h(nf);
    for (size_t i = 0; i < nf; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(v, h.data(), nf * 4, hipMemcpyHostToDevice); hipMemcpy(x, h.data(), nf * 4, hipMemcpyHostToDevice);
    std::vector<float> ha(na);
    for (size_t i = 0; i < na; ++i) ha[i] = (float)((i * 40503u) & 0xff) / (256.f * S);
    hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice);
    const float g = 0.5f; hipMemcpy(gamma, &g, 4, hipMemcpyHostToDevice);
    const int nb = (H + 15) / 16, rpb = (H + nb - 1) / nb, ncg = C / 32, grid = B * ncg * nb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { hipLaunchKernelGGL((cca::map_band_fwd_kernel<100>), dim3(grid), dim3(256), 0, 0, A, v, x, gamma, y, C, H, W, nb, rpb, ncg, (long)H * W * C, C); };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("BAND_ABL=%d B=%d grid=%d: %.1f us (err %d)\n", BAND_ABL, B, grid, ms * 1e3 / iters, (int)hipGetLastError());
    return 0;
}

// Probe: semantics of ds_read_b64_tr_b16 (gfx950) with ARBITRARY per-lane addresses.
// Hypothesis (cdna_hip_programming.md, T10): inside each 16-lane group, lane i receives element (i & 3) of the 8 bytes
// addressed by lane 4 j + (i >> 2), for j = 0..3 -- i.e. the group reads a [4 rows][16 cols] block whose row r is
// supplied as four 8-byte pieces by lanes 4 r .. 4 r + 3, and lane i gets column i.
// build: hipcc --offload-arch=gfx950 -O3 tr16_probe.hip -o tr16_probe ; run: ./tr16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k(const int *addr, uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) s16x4 *)(lds + addr[threadIdx.x]);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

int main() {
    int h_addr[64];
    uint16_t h_out[256];
    int *d_addr;
    uint16_t *d_out;
    hipMalloc(&d_addr, sizeof(h_addr));
    hipMalloc(&d_out, sizeof(h_out));
    int bad_total = 0;
    for (int trial = 0; trial < 3; ++trial) {
        srand(17 + trial);
        for (int l = 0; l < 64; ++l)
            h_addr[l] = trial == 0 ? ((l & 15) >> 2) * 16 + 4 * (l & 3) + (l >> 4) * 64      // canonical [4][16] blocks
                                   : 4 * (rand() % 2000);                                     // anything 8-byte aligned
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int grp = l & ~15, i = l & 15;
                const int want = h_addr[grp + 4 * j + (i >> 2)] + (i & 3);
                if (h_out[l * 4 + j] != want) {
                    if (bad < 8) printf("trial %d lane %d elem %d: got %d want %d\n", trial, l, j, h_out[l * 4 + j], want);
                    ++bad;
                }
            }
        printf("trial %d: %d mismatches\n", trial, bad);
        if (trial == 0) {
            printf("lane 0..19 (canonical):");
            for (int l = 0; l < 20; ++l) printf(" [%d %d %d %d]", h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
            printf("\n");
        }
        bad_total += bad;
    }
    printf(bad_total ? "tr16 probe: HYPOTHESIS WRONG\n" : "tr16 probe: hypothesis holds\n");
    return bad_total != 0;
}

// Cycles per MT19937 regeneration of the two schedule-independent forms (pvol_rng_dev.h), one wave alone on its SIMD and the state in LDS:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Ics348b-pbrt_amd/csrc -Iinclude tools/mt_probe.hip -o tools/mt_probe.out && ./tools/mt_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "pvol_rng_dev.h"
template <bool WG1>
__global__ void probe(uint32_t seed, int n, unsigned long long *cy, uint32_t *out) {
    __shared__ uint32_t mt[MT_N + 8];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        mt_seed<WG1>(mt, seed, lane);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; ++i) mt_regenerate<WG1>(mt, lane);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cy[blockIdx.x] = t1 - t0; out[blockIdx.x] = mt[5]; }
    }
}
int main() {
    unsigned long long *cy; uint32_t *out;
    hipMallocManaged(&cy, 8 * 4096); hipMallocManaged(&out, 4 * 4096);   // one entry per block, at most 4096 blocks below
    const int n = 2000;
    for (int blocks : {1, 512, 4096}) {
        hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(64), 0, 0, 5489u, 10, cy, out); hipDeviceSynchronize();   // warm up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(64), 0, 0, 5489u, n, cy, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        const unsigned long long a = cy[0]; const uint32_t oa = out[0];
        hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(128), 0, 0, 5489u, n, cy, out);
        hipEventRecord(e2, 0); hipEventSynchronize(e2);
        float ms1 = 0, ms2 = 0; hipEventElapsedTime(&ms1, e0, e1); hipEventElapsedTime(&ms2, e1, e2);
        printf("blocks %d: volatile loop %.0f ticks, %.0f ns per regeneration | two-round-trip form %.0f ticks, %.0f ns | same state: %s\n", blocks, (double)a / n, ms1 * 1e6 / n,
               (double)cy[0] / n, ms2 * 1e6 / n, oa == out[0] ? "yes" : "NO");
    }
    return 0;
}

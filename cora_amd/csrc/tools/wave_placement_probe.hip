// wave_placement_probe.hip - which SIMD do the waves of a workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4],
// pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] on gfx9-family parts.)  Prints, for workgroups of 4 and 8
// waves with and without enough LDS to force one workgroup per CU, the SIMD ids of the waves of a few workgroups.
//   hipcc --offload-arch=gfx950 -O2 wave_placement_probe.hip -o wave_placement_probe && ./wave_placement_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(unsigned *out, int spin) {
    extern __shared__ double lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    // keep the workgroup resident for a while so that the grid really co-resides as in the kernels of interest
    double x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1.0000001 + 1e-9;
    if (x == 1.2345) lds[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}

int main() {
    const int nwg = 512;
    unsigned *d;
    hipMalloc(&d, nwg * 16 * sizeof(unsigned));
    for (int waves : {4, 8}) {
        for (size_t shm : {(size_t)0, (size_t)80 * 1024, (size_t)124 * 1024}) {
            hipMemset(d, 0xff, nwg * 16 * sizeof(unsigned));
            hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            probe<<<nwg, 64 * waves, shm>>>(d, 200000);
            hipDeviceSynchronize();
            std::vector<unsigned> h(nwg * 16);
            hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
            int hist[5] = {0, 0, 0, 0, 0};   // workgroups by number of DISTINCT SIMDs their waves use
            for (int b = 0; b < nwg; b++) {
                unsigned mask = 0;
                for (int w = 0; w < waves; w++) mask |= 1u << ((h[b * 16 + w] >> 4) & 3);
                hist[__builtin_popcount(mask)]++;
            }
            printf("waves/WG %d, LDS %3zu KB: workgroups using 1/2/3/4 distinct SIMDs: %d %d %d %d;  WG 0 simd ids:", waves, shm / 1024,
                   hist[1], hist[2], hist[3], hist[4]);
            for (int w = 0; w < waves; w++) printf(" %u", (h[w] >> 4) & 3);
            printf("  cu %u;  WG 1:", (h[0] >> 8) & 15);
            for (int w = 0; w < waves; w++) printf(" %u", (h[16 + w] >> 4) & 3);
            printf("  cu %u\n", (h[16] >> 8) & 15);
        }
    }
    return 0;
}

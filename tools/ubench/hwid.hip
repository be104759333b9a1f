// Developer probe: where do the waves of two co-resident 256-thread workgroups (80 KB of LDS each) land?  Prints, per workgroup, LDS_BASE and
// the SIMD of each of its four waves (HW_REG_HW_ID).   hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip && ./hwid
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all bits
    const unsigned base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6);  // LDS_BASE
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) lds[threadIdx.x] += 1.0;                 // stay resident for a while
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw, out[blockIdx.x * 8 + 4 + (threadIdx.x >> 6)] = base;
}
int main() {
    const int nb = 1024;
    unsigned* d;
    hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80400);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 80400, 0, d, 200000);
    hipDeviceSynchronize();
    static unsigned h[nb * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[2][4][4] = {};  // [lds base != 0][wave][simd]
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) hist[h[b * 8 + 4 + w] != 0][w][(h[b * 8 + w] >> 4) & 3]++;
    for (int s = 0; s < 2; ++s)
        for (int w = 0; w < 4; ++w) printf("LDS_BASE %s wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", s ? "!= 0" : "== 0", w, hist[s][w][0], hist[s][w][1], hist[s][w][2], hist[s][w][3]);
    for (int b = 0; b < 6; ++b) printf("block %d: cu %u se %u  base %u  simd of waves %u %u %u %u\n", b, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7, h[b * 8 + 4], (h[b * 8] >> 4) & 3, (h[b * 8 + 1] >> 4) & 3, (h[b * 8 + 2] >> 4) & 3, (h[b * 8 + 3] >> 4) & 3);
    return 0;
}

// PMC calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern"):
// streams a buffer far larger than L2 + MALL with the access widths the QP kernel uses (8 bytes per lane, coalesced) and,
// for comparison, 16 bytes per lane; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this binary gives the counter-to-byte factors.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double* __restrict__ p, double* out, size_t n) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 1.2345e301) out[0] = s;
}
__global__ void read16(const double2* __restrict__ p, double* out, size_t n) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i].x + p[i].y;
    if (s == 1.2345e301) out[0] = s;
}
__global__ void write8(double* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
// 48-byte pieces (six lanes x 8 B) of scattered 128-byte lines: the pattern of the per-control-point vectors in a sweep
__global__ void read8_pieces(const double* __restrict__ p, double* out, size_t n) {
    double s = 0;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t g = t / 6; g * 16 + 6 <= n; g += (size_t)gridDim.x * blockDim.x / 6) s += p[g * 16 + t % 6];
    if (s == 1.2345e301) out[0] = s;
}
int main() {
    const size_t n = (size_t)1 << 30;  // 8 GiB of doubles
    double *p, *o;
    hipMalloc(&p, n * 8), hipMalloc(&o, 64);
    hipMemset(p, 0, n * 8);
    hipDeviceSynchronize();
    read8<<<4096, 256>>>(p, o, n);
    read16<<<4096, 256>>>((const double2*)p, o, n / 2);
    write8<<<4096, 256>>>(p, n);
    read8_pieces<<<4092, 252>>>(p, o, n);
    hipDeviceSynchronize();
    printf("bytes: read8 %zu read16 %zu write8 %zu read8_pieces(useful) %zu\n", n * 8, n * 8, n * 8, n / 16 * 48);
    return 0;
}

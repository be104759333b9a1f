// LDS broadcast-read throughput of ONE wave (developer micro-benchmark): cycles per instruction for b64 / b128 same-address reads,
// all 64 lanes or 36 lanes active, interleaved with FMAs or not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double ldsd;
typedef __attribute__((address_space(3))) d2 ldsd2;
#define REP 64
template <int MODE>
__global__ void k(double* out, long long* cyc, int nact) {
    __shared__ __attribute__((aligned(16))) double buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = i * 1e-3;
    __syncthreads();
    if ((int)threadIdx.x >= nact) return;
    const ldsd* p = (const ldsd*)buf;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        const ldsd* q = p + (r & 7) * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {  // b128 broadcast + 2 FMAs
                const d2 t = *(const ldsd2*)(q + 2 * i);
                acc[i] = __builtin_fma(acc[i], 0.5, t[0]);
                acc[(i + 8) & 15] = __builtin_fma(acc[(i + 8) & 15], 0.5, t[1]);
            } else if (MODE == 1) {  // b64 broadcast + 1 FMA
                acc[i] = __builtin_fma(acc[i], 0.5, q[i]);
            } else if (MODE == 2) {  // b128 broadcast only (sum at the end)
                const d2 t = *(const ldsd2*)(q + 2 * i);
                acc[i] += t[0] + t[1];
            } else if (MODE == 3) {  // per-lane b128 (own row, stride 38 doubles)
                const d2 t = *(const ldsd2*)(q + (threadIdx.x & 15) * 38 + 2 * i);
                acc[i] = __builtin_fma(acc[i], 0.5, t[0]);
                acc[(i + 8) & 15] = __builtin_fma(acc[(i + 8) & 15], 0.5, t[1]);
            } else {  // FMAs only
                acc[i] = __builtin_fma(acc[i], 0.5, 1.0);
                acc[(i + 8) & 15] = __builtin_fma(acc[(i + 8) & 15], 0.5, 2.0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    const char* names[] = {"b128 bcast + 2 FMA", "b64 bcast + 1 FMA", "b128 bcast + 2 add", "b128 per-lane rows + 2 FMA", "2 FMA only"};
    for (int nact : {64, 36, 32}) for (int waves : {1, 4}) {
        printf("active lanes %d, waves %d:", nact, waves);
        for (int m = 0; m < 5; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) k<0><<<1, 64 * waves>>>(out, cyc, waves > 1 ? 64 * waves : nact);
                if (m == 1) k<1><<<1, 64 * waves>>>(out, cyc, waves > 1 ? 64 * waves : nact);
                if (m == 2) k<2><<<1, 64 * waves>>>(out, cyc, waves > 1 ? 64 * waves : nact);
                if (m == 3) k<3><<<1, 64 * waves>>>(out, cyc, waves > 1 ? 64 * waves : nact);
                if (m == 4) k<4><<<1, 64 * waves>>>(out, cyc, waves > 1 ? 64 * waves : nact);
            }
            hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("  [%s] %.1f cyc/iter", names[m], h / (double)(REP * 16));
        }
        printf("\n");
    }
    return 0;
}

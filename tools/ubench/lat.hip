// Developer micro-benchmark (not product): instruction costs that decide the shape of the knot-block factorisation of qp.hip.
//   hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
// Every kernel runs W waves on ONE workgroup (so W/4 waves per SIMD of one CU) and reports shader cycles (s_memtime) per operation
// of wave 0.  "dep" = one dependent chain per wave, "indN" = N independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 4096

__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int NCH>
__global__ void k_fma(double* out, long long* cyc, double a0) {
    double x[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = a0 + i + threadIdx.x;
    const double m = a0 * 0.999, c = a0 * 1e-3;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) x[i] = __builtin_fma(x[i], m, c);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the column update of ldl_rows: a[k] -= lc * readlane(a[c], k), NK-1 independent updates per "column", columns dependent through lane c's pivot
__global__ void k_ldlcol(double* out, long long* cyc, double a0) {
    double a[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) a[k] = (k == (int)(threadIdx.x & 63) ? 40.0 : 0.01 * a0 * (k + 1));
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < 64; ++r) {
#pragma unroll
        for (int c = 0; c < 36; ++c) {
            const double dcc = rl(a[c], c);
            double inv = __builtin_amdgcn_rcp(dcc);
            inv = __builtin_fma(__builtin_fma(-dcc, inv, 1.0), inv, inv);
            const double lc = a[c] * inv;
#pragma unroll
            for (int k = c + 1; k < 36; ++k) a[k] -= lc * rl(a[c], k);
            a[c] = lc;
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) a[k] = a[k] * 1e-3 + (k == (int)(threadIdx.x & 63) ? 40.0 : 0.01);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / 64;
}

template <int NCH>
__global__ void k_mfma16(double* out, long long* cyc, double a0) {
    d4 acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = d4{a0, a0, a0, a0};
    const double a = a0 * 1e-3 + threadIdx.x * 1e-6, b = a0 * 1e-3;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH>
__global__ void k_mfma4(double* out, long long* cyc, double a0) {
    double acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = a0;
    const double a = a0 * 1e-3 + threadIdx.x * 1e-6, b = a0 * 1e-3;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// layout discovery of v_mfma_f64_4x4x4f64: A = e_p (one lane holds 1.0), B = all lanes hold (lane + 1): which lanes of D see what
__global__ void k_mfma4_layout(double* out, int pa, int pb) {
    const int l = threadIdx.x;
    // mode pa >= 0: A lane pa = 1, B = lane + 1   -> D[l] tells which B lanes pair with A lane pa
    // mode pb >= 0: B lane pb = 1, A = lane + 1
    double a, b;
    if (pa >= 0) a = (l == pa) ? 1.0 : 0.0, b = l + 1;
    else a = l + 1, b = (l == pb) ? 1.0 : 0.0;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

// dependent LDS round trip (f64) and dependent ds_bpermute
__global__ void k_lds(double* out, long long* cyc, double a0) {
    __shared__ double buf[512];
    buf[threadIdx.x] = (double)((threadIdx.x + 1) & 63);
    __syncthreads();
    int idx = threadIdx.x & 63;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) idx = (int)buf[idx];
    long long t1 = __builtin_readcyclecounter();
    int v = idx;
    long long t2 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) << 2, v);
    long long t3 = __builtin_readcyclecounter();
    out[threadIdx.x] = idx + v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = t3 - t2;
}

// x = x * rl(x, c) dependent through a readlane (VALU -> SGPR -> VALU)
__global__ void k_rlchain(double* out, long long* cyc, double a0) {
    double x = a0 + 1e-9 * threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP / 36; ++r) {
#pragma unroll
        for (int c = 0; c < 36; ++c) x = __builtin_fma(x, 0.5, rl(x, c) * 0.5);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// DPP row_bcast / wave-wide broadcast alternative: v_readfirstlane-free dependent step through DPP quad_perm
__global__ void k_dppchain(double* out, long long* cyc, double a0) {
    double x = a0 + 1e-9 * threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x111, 0xf, 0xf, true);
        x = __builtin_fma(x, 0.5, __hiloint2double(hi, lo) * 0.5);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, 8 * 4096), hipMalloc(&cyc, 8 * 64);
    long long h[64];
    auto rep = [&](const char* name, int waves, double per) {
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("%-34s waves/WG %2d : %8.2f cycles per op (total %lld)\n", name, waves, h[0] / per, h[0]);
    };
    for (int w : {1, 4, 8, 16, 32}) {
        if (w > 16) continue;  // 1024 threads max
        k_fma<1><<<1, 64 * w>>>(out, cyc, 1.0), rep("v_fma_f64 dep", w, REP);
        k_fma<4><<<1, 64 * w>>>(out, cyc, 1.0), rep("v_fma_f64 ind4 (per fma)", w, REP * 4.0);
        k_fma<8><<<1, 64 * w>>>(out, cyc, 1.0), rep("v_fma_f64 ind8 (per fma)", w, REP * 8.0);
    }
    for (int w : {1, 4, 8, 16}) k_ldlcol<<<1, 64 * w>>>(out, cyc, 1.0), rep("ldl_rows<36> (cycles per block)", w, 1.0);
    for (int w : {1, 4, 8, 16}) {
        k_mfma16<1><<<1, 64 * w>>>(out, cyc, 1.0), rep("mfma_f64_16x16x4 dep", w, REP);
        k_mfma16<4><<<1, 64 * w>>>(out, cyc, 1.0), rep("mfma_f64_16x16x4 ind4 (per mfma)", w, REP * 4.0);
        k_mfma4<1><<<1, 64 * w>>>(out, cyc, 1.0), rep("mfma_f64_4x4x4 dep", w, REP);
        k_mfma4<4><<<1, 64 * w>>>(out, cyc, 1.0), rep("mfma_f64_4x4x4 ind4 (per mfma)", w, REP * 4.0);
    }
    k_lds<<<1, 64>>>(out, cyc, 1.0);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("LDS f64 dependent read: %.1f cycles; ds_bpermute dependent: %.1f cycles\n", h[0] / (double)REP, h[1] / (double)REP);
    k_rlchain<<<1, 64>>>(out, cyc, 1.0), rep("readlane(2)+mul+fma dependent step", 1, (REP / 36) * 36.0);
    k_dppchain<<<1, 64>>>(out, cyc, 1.0), rep("dpp(2)+mul+fma dependent step", 1, REP);
    // layout of the 4x4x4 (4 blocks) f64 MFMA
    double hv[64];
    printf("mfma_f64_4x4x4 layout: A lane p = 1, B lane l = l+1 -> D per lane\n");
    for (int p : {0, 1, 4, 5, 16, 17, 21, 63}) {
        k_mfma4_layout<<<1, 64>>>(out, p, -1);
        hipDeviceSynchronize();
        hipMemcpy(hv, out, 512, hipMemcpyDeviceToHost);
        printf(" A@%2d:", p);
        for (int l = 0; l < 64; ++l) if (hv[l] != 0) printf(" D[%d]=B%d", l, (int)hv[l] - 1);
        printf("\n");
    }
    for (int p : {0, 1, 4, 5, 16, 17, 21, 63}) {
        k_mfma4_layout<<<1, 64>>>(out, -1, p);
        hipDeviceSynchronize();
        hipMemcpy(hv, out, 512, hipMemcpyDeviceToHost);
        printf(" B@%2d:", p);
        for (int l = 0; l < 64; ++l) if (hv[l] != 0) printf(" D[%d]=A%d", l, (int)hv[l] - 1);
        printf("\n");
    }
    return 0;
}

// Developer test + timing of kernels/knot_lds.inc (one wavefront, a chain of knot steps) against a host restatement.
//   hipcc --offload-arch=gfx950 -O3 -I../../swarm_simulator_amd/csrc/kernels -o knot knot.hip && ./knot
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "knot_lds.inc"

constexpr int NK = 36;
constexpr int STEPS = 6;

// T: [STEPS][NK*NK] element (r,k) at k*NK + r (symmetric); E: [STEPS][9]; out M, X: [STEPS][NK*NK] element (r,k) at k*NK + r; dinv [STEPS][NK]
__global__ __launch_bounds__(512) void chain_kernel(const double* T, const double* E, double* Mo, double* Xo, double* Do, long long* cyc, int* okflag, int waves_active) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const int wave = threadIdx.x >> 6, r = threadIdx.x & 63;
    if (wave >= waves_active) return;
    using A = KlArea<NK>;
    kl_lds* base = (kl_lds*)(lds_raw + wave * A::SIZE);
    kl_lds *C = base + A::C, *MX = base + A::MX, *I = base + A::I, *U = base + A::C;
    for (int i = r; i < A::SIZE; i += 64) base[i] = 0.0;
    kl_sync();
    const bool act = r < NK;
    const int rr = act ? r : 0;
    bool ok = true;
    long long t0 = __builtin_readcyclecounter(), tl = 0, tm = 0, tx = 0, ts = 0, tst = 0;
    for (int i = 0; i < STEPS; ++i) {
        long long c0 = __builtin_readcyclecounter();
#ifndef NO_SYRK
        if (i > 0) kl_syrk<NK>(MX, I, U, r, false);
#endif
        long long c1 = __builtin_readcyclecounter();
        double a[NK];
        const double* Tg = T + (size_t)i * NK * NK;
#pragma unroll
        for (int k = 0; k < NK; ++k) a[k] = Tg[k * NK + rr];
        if (i > 0) {
#pragma unroll
            for (int k = 0; k < NK; ++k) a[k] -= U[rr * KL_LDU + k];
            kl_sync();
        }
        double m[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) m[k] = (k == r) ? 1.0 : 0.0;
        if (!kl_ldl<NK>(a, C, I, r, act)) ok = false;
        long long c2 = __builtin_readcyclecounter();
#ifndef NO_M
        kl_row_times_LinvT<NK>(m, C, I);
        kl_store_rows<NK>(m, MX, r, act);
        long long c2b = __builtin_readcyclecounter();
        if (act && wave == 0) {   // row r of M = L^-T, entries k >= r: pairs, 16 bytes per lane and instruction
            double* Mr = Mo + (size_t)i * NK * NK + (size_t)r * NK;
#pragma unroll
            for (int k = 0; k < NK; k += 2)
                if (k + 1 >= r) *(kl_d2*)(Mr + k) = kl_d2{m[k], m[k + 1]};
            Do[i * NK + r] = I[r];
        }
        tst += __builtin_readcyclecounter() - c2b;
#endif
        long long c3 = __builtin_readcyclecounter();
        const double* Ei = E + 9 * i;
        const double e0 = Ei[rr % 3], e1 = Ei[3 + rr % 3], e2 = Ei[6 + rr % 3];  // T_{j+1,j}[r][3g+q] = E[q][r%3]
#ifndef NO_X
        double x[NK];
        kl_coupling_rows<NK>(x, MX, r, act, e0, e1, e2);
        if (act && wave == 0 && i == STEPS - 1) {  // (the product does not store X: checked here on the last step only)
#pragma unroll
            for (int k = 0; k < NK; ++k) Xo[(size_t)i * NK * NK + k * NK + r] = x[k];
        }
#endif
        long long c4 = __builtin_readcyclecounter();
        ts += c1 - c0, tl += c2 - c1, tm += c3 - c2, tx += c4 - c3;
    }
    long long t1 = __builtin_readcyclecounter();
    if (r == 0) cyc[wave * 8 + 0] = (t1 - t0) / STEPS, cyc[wave * 8 + 1] = ts / STEPS, cyc[wave * 8 + 2] = tl / STEPS, cyc[wave * 8 + 3] = tm / STEPS, cyc[wave * 8 + 4] = tx / STEPS, cyc[wave * 8 + 5] = tst / STEPS;
    if (!ok && r == 0) *okflag = 1;
}

int main() {
    std::vector<double> T(STEPS * NK * NK), E(STEPS * 9);
    srand(7);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    for (int i = 0; i < STEPS; ++i) {
        std::vector<double> B(NK * NK);
        for (auto& v : B) v = rnd();
        for (int r = 0; r < NK; ++r)
            for (int k = 0; k < NK; ++k) {
                double s = 0;
                for (int q = 0; q < NK; ++q) s += B[r * NK + q] * B[k * NK + q];
                T[(size_t)i * NK * NK + k * NK + r] = s + (r == k ? 30.0 + 1e4 * (r % 5 == 0) : 0.0);  // SPD, some large diagonal entries like IPM weights
            }
        for (int e = 0; e < 9; ++e) E[9 * i + e] = 3.0 * rnd();
    }
    // host restatement
    std::vector<double> Mh(STEPS * NK * NK), Xh(STEPS * NK * NK), Dh(STEPS * NK);
    {
        std::vector<double> U(NK * NK, 0.0);
        for (int i = 0; i < STEPS; ++i) {
            std::vector<double> A(NK * NK), L(NK * NK, 0.0), d(NK);
            for (int r = 0; r < NK; ++r)
                for (int k = 0; k < NK; ++k) A[r * NK + k] = T[(size_t)i * NK * NK + k * NK + r] - (i > 0 ? U[r * NK + k] : 0.0);
            for (int c = 0; c < NK; ++c) {
                d[c] = A[c * NK + c];
                L[c * NK + c] = 1;
                for (int r = c + 1; r < NK; ++r) L[r * NK + c] = A[r * NK + c] / d[c];
                for (int r = c + 1; r < NK; ++r)
                    for (int k = c + 1; k < NK; ++k) A[r * NK + k] -= L[r * NK + c] * d[c] * L[k * NK + c];
            }
            // M = L^-T: solve L' M = I  -> M[r][k]: row r of L^-T = column r of L^-1
            std::vector<double> Li(NK * NK, 0.0);
            for (int c = 0; c < NK; ++c) {  // column c of L^-1
                for (int r = 0; r < NK; ++r) {
                    double s = (r == c) ? 1.0 : 0.0;
                    for (int k = 0; k < r; ++k) s -= L[r * NK + k] * Li[k * NK + c];
                    Li[r * NK + c] = s;
                }
            }
            for (int r = 0; r < NK; ++r)
                for (int k = 0; k < NK; ++k) Mh[(size_t)i * NK * NK + r * NK + k] = Li[k * NK + r];  // M[r][k] = Li[k][r], row-major
            for (int r = 0; r < NK; ++r) Dh[i * NK + r] = 1.0 / d[r];
            std::vector<double> X(NK * NK);
            for (int r = 0; r < NK; ++r)
                for (int k = 0; k < NK; ++k) {
                    double s = 0;
                    for (int q = 0; q < 3; ++q) s += E[9 * i + 3 * q + r % 3] * Li[k * NK + 3 * (r / 3) + q];
                    X[r * NK + k] = s;
                    Xh[(size_t)i * NK * NK + k * NK + r] = s;
                }
            for (int r = 0; r < NK; ++r)
                for (int k = 0; k < NK; ++k) {
                    double s = 0;
                    for (int c = 0; c < NK; ++c) s += X[r * NK + c] / d[c] * X[k * NK + c];
                    U[r * NK + k] = s;
                }
        }
    }
    double *dT, *dE, *dM, *dX, *dD;
    long long* dc;
    int* dok;
    hipMalloc(&dT, T.size() * 8), hipMalloc(&dE, E.size() * 8), hipMalloc(&dM, Mh.size() * 8), hipMalloc(&dX, Xh.size() * 8), hipMalloc(&dD, Dh.size() * 8);
    hipMalloc(&dc, 8 * 64 * 8), hipMalloc(&dok, 4);
    hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice), hipMemcpy(dE, E.data(), E.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dok, 0, 4);
    const size_t lds = 5 * KlArea<NK>::SIZE * sizeof(double);  // five areas = 145 KB
    hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("LDS per chain wave: %d doubles = %.1f KB\n", KlArea<NK>::SIZE, KlArea<NK>::SIZE * 8 / 1024.0);
    for (int waves : {1, 2, 4}) {
        // waves chain waves per workgroup; with 512 threads = 8 waves (2 per SIMD), only `waves` of them work
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(512), lds, 0, dT, dE, dM, dX, dD, dc, dok, waves);
        if (hipDeviceSynchronize() != hipSuccess) printf("launch failed\n");
        long long h[64];
        hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        printf("chain waves %d: cycles per step %lld  (syrk %lld, load+ldl %lld, Linv^T rows + LDS + global store %lld of which global store %lld, coupling rows %lld)\n", waves, h[0], h[1], h[2], h[3], h[5], h[4]);
    }
    // 5 waves working: one SIMD carries two chains
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(512), lds, 0, dT, dE, dM, dX, dD, dc, dok, 5);
    if (hipDeviceSynchronize() != hipSuccess) printf("launch failed\n");
    {
        long long h[64];
        hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        printf("chain waves 5 (one SIMD carries two): cycles per step %lld %lld %lld %lld %lld\n", h[0], h[8], h[16], h[24], h[32]);
    }
    std::vector<double> Mg(Mh.size()), Xg(Xh.size()), Dg(Dh.size());
    int okf;
    hipMemcpy(Mg.data(), dM, Mg.size() * 8, hipMemcpyDeviceToHost), hipMemcpy(Xg.data(), dX, Xg.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(Dg.data(), dD, Dg.size() * 8, hipMemcpyDeviceToHost), hipMemcpy(&okf, dok, 4, hipMemcpyDeviceToHost);
    double em = 0, ex = 0, ed = 0, sm = 0, sx = 0;
    for (size_t i = 0; i < Mh.size(); ++i) {
        const int rr = (int)(i % (NK * NK)) / NK, kk = (int)(i % NK);
        if ((kk | 1) >= rr) em = fmax(em, fabs(Mg[i] - Mh[i])), sm = fmax(sm, fabs(Mh[i]));   // only pairs with k + 1 >= r are stored
        if (i >= (size_t)(STEPS - 1) * NK * NK) ex = fmax(ex, fabs(Xg[i] - Xh[i])), sx = fmax(sx, fabs(Xh[i]));
    }
    for (size_t i = 0; i < Dh.size(); ++i) ed = fmax(ed, fabs(Dg[i] - Dh[i]) / fabs(Dh[i]));
    printf("pivots positive: %s   max|M err| %.3g (scale %.3g)  max|X err| %.3g (scale %.3g)  max rel 1/d err %.3g\n", okf ? "NO" : "yes", em, sm, ex, sx, ed);
    const bool pass = !okf && em < 1e-11 * fmax(1.0, sm) && ex < 1e-11 * fmax(1.0, sx) && ed < 1e-12;
    printf("%s\n", pass ? "PASS" : "FAIL");
    return pass ? 0 : 1;
}

// Developer test + timing of kernels/lh_inverse.inc (one wavefront): append rows to W = L^-1 of an SPD matrix S, solve, delete, against a
// host restatement.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I../../swarm_simulator_amd/csrc/kernels -o lhw lhw.hip && ./lhw
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) double kl_lds;
#define WSYNC()                                             \
    do {                                                    \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                    \
    } while (0)
__device__ __forceinline__ double rl_dyn(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}
#include "lh_inverse.inc"

constexpr int NMAX = 112;

__device__ double wsum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// S: [n][n] in global memory.  Appends rows 0..n-1 in order, then deletes `ndel` rows (positions del[]), solving after every step.
__global__ __launch_bounds__(64) void lh_kernel(const double* S, const double* d, int n, const int* del, int ndel, double* Wout, double* yout,
                                                long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Lc = lds;                        // packed W
    double* scr = lds + lhp_size(NMAX);
    double* wv = scr + NMAX;
    kl_lds* Wl = (kl_lds*)Lc;
    double* Wg = nullptr;
    const int lane = threadIdx.x;
    int nP = 0;
    lh_reset(Wl, Wg, NMAX);
    long long t_rows = 0, t_cols = 0, t_del = 0, t_app = 0;
    for (int j = 0; j < n; ++j) {
        const long long a0 = __builtin_readcyclecounter();
        double sr[LH_H], lr[LH_H], tr[LH_H], wr[LH_H];
#pragma unroll
        for (int h = 0; h < LH_H; ++h) sr[h] = lane + 64 * h < nP ? S[(size_t)(lane + 64 * h) * n + j] : 0.0;
        const long long c0 = __builtin_readcyclecounter();
        lh_rows(Wl, Wg, sr, lr, nP, lane);
        const long long c1 = __builtin_readcyclecounter();
        lhv_load(wv, wr, nP, lane);
        double ll = 0, lw = 0;
#pragma unroll
        for (int h = 0; h < LH_H; ++h) ll += lr[h] * lr[h], lw += lr[h] * wr[h];
        ll = wsum(ll), lw = wsum(lw);
        const double djj = S[(size_t)j * n + j] - ll;
        const long long c2 = __builtin_readcyclecounter();
        lh_cols(Wl, Wg, lr, tr, nP, lane);
        const long long c3 = __builtin_readcyclecounter();
        const double ir = fast_rsqrt(djj);
        {
            const int rowo = lhp_row(nP), len = lhp_len(nP >> 3);
#pragma unroll
            for (int h = 0; h < LH_H; ++h) {
                const int cq = lane + 64 * h;
                if (cq < len) Lc[rowo + cq] = cq < nP ? -ir * tr[h] : (cq == nP ? ir : 0.0);
            }
        }
        if (lane == 0) wv[nP] = (d[j] - lw) * ir;
        nP++;
        WSYNC();
        const long long a1 = __builtin_readcyclecounter();
        t_rows += c1 - c0, t_cols += c3 - c2, t_app += a1 - a0;
    }
    if (lane == 0) cyc[0] = t_rows / n, cyc[1] = t_cols / n, cyc[2] = t_app / n;
    // solve at full size, timed
    {
        double wr[LH_H], yr[LH_H];
        const long long c0 = __builtin_readcyclecounter();
        lhv_load(wv, wr, nP, lane);
        lh_cols(Wl, Wg, wr, yr, nP, lane);
        lhv_store(yout, yr, nP, lane);
        WSYNC();
        if (lane == 0) cyc[3] = __builtin_readcyclecounter() - c0;
    }
    for (int r = 0; r < nP; ++r)
        for (int c = lane; c <= r; c += 64) Wout[r * (r + 1) / 2 + c] = Lc[lhp_row(r) + c];
    WSYNC();
    for (int q = 0; q < ndel; ++q) {
        const long long c0 = __builtin_readcyclecounter();
        lh_delete(Wl, Wg, scr, nP, del[q], lane);
        nP--;
        t_del += __builtin_readcyclecounter() - c0;
    }
    if (lane == 0) cyc[4] = ndel ? t_del / ndel : 0;
    for (int r = 0; r < nP; ++r)
        for (int c = lane; c < lhp_len(r >> 3); c += 64)  // with the stored part right of the diagonal, which must be zero
            if (c <= r) Wout[NMAX * (NMAX + 1) / 2 + r * (r + 1) / 2 + c] = Lc[lhp_row(r) + c];
            else if (Lc[lhp_row(r) + c] != 0.0) Wout[NMAX * (NMAX + 1) / 2] = 1e300;
}

static void host_winv(const std::vector<double>& S, int n, const std::vector<int>& idx, std::vector<double>& W) {
    const int m = (int)idx.size();
    std::vector<double> L(m * m, 0.0), Wi(m * m, 0.0);
    for (int c = 0; c < m; ++c) {
        for (int r = c; r < m; ++r) {
            double s = S[(size_t)idx[r] * n + idx[c]];
            for (int k = 0; k < c; ++k) s -= L[r * m + k] * L[c * m + k];
            L[r * m + c] = r == c ? sqrt(s) : s / L[c * m + c];
        }
    }
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < m; ++r) {
            double s = r == c ? 1.0 : 0.0;
            for (int k = 0; k < r; ++k) s -= L[r * m + k] * Wi[k * m + c];
            Wi[r * m + c] = s / L[r * m + r];
        }
    W.assign(m * (m + 1) / 2, 0.0);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c <= r; ++c) W[r * (r + 1) / 2 + c] = Wi[r * m + c];
}

int main() {
    bool all = true;
    for (int n : {35, 64, 70, 110}) {
        std::vector<double> S(n * n), d(n), B(n * n);
        srand(11 + n);
        auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
        for (auto& v : B) v = rnd();
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                double s = 0;
                for (int q = 0; q < n; ++q) s += B[r * n + q] * B[c * n + q];
                S[r * n + c] = s + (r == c ? 0.5 : 0.0);
            }
        for (auto& v : d) v = rnd();
        std::vector<int> del = {n / 3, 0, n / 2, n - 4};  // positions at the time of the deletion
        double *dS, *dd, *dW, *dy;
        int* ddel;
        long long* dc;
        hipMalloc(&dS, S.size() * 8), hipMalloc(&dd, d.size() * 8), hipMalloc(&dW, NMAX * (NMAX + 1) * 8), hipMalloc(&dy, NMAX * 8);
        hipMalloc(&ddel, del.size() * 4), hipMalloc(&dc, 64);
        hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice), hipMemcpy(dd, d.data(), d.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(ddel, del.data(), del.size() * 4, hipMemcpyHostToDevice);
        const size_t lds = (lhp_size(NMAX) + 2 * NMAX + 8) * 8;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(lh_kernel, dim3(1), dim3(64), lds, 0, dS, dd, n, ddel, (int)del.size(), dW, dy, dc);
        if (hipDeviceSynchronize() != hipSuccess) printf("launch failed\n");
        long long cyc[8];
        std::vector<double> Wg(NMAX * (NMAX + 1)), yg(n);
        hipMemcpy(cyc, dc, 64, hipMemcpyDeviceToHost), hipMemcpy(Wg.data(), dW, Wg.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(yg.data(), dy, n * 8, hipMemcpyDeviceToHost);
        // host: W of the full set, y = S^-1 d, W after the deletions
        std::vector<int> idx(n);
        for (int i = 0; i < n; ++i) idx[i] = i;
        std::vector<double> Wh;
        host_winv(S, n, idx, Wh);
        double ew = 0, sw = 0;
        for (size_t i = 0; i < Wh.size(); ++i) ew = fmax(ew, fabs(Wh[i] - Wg[i])), sw = fmax(sw, fabs(Wh[i]));
        // y check through the residual S y - d
        double ey = 0;
        for (int r = 0; r < n; ++r) {
            double s = -d[r];
            for (int c = 0; c < n; ++c) s += S[r * n + c] * yg[c];
            ey = fmax(ey, fabs(s));
        }
        for (int p : del) idx.erase(idx.begin() + p);
        std::vector<double> Wh2;
        host_winv(S, n, idx, Wh2);
        double ed = 0;
        for (size_t i = 0; i < Wh2.size(); ++i) ed = fmax(ed, fabs(Wh2[i] - Wg[NMAX * (NMAX + 1) / 2 + i]));
        const bool ok = ew < 1e-10 * fmax(1.0, sw) && ey < 1e-10 && ed < 1e-10 * fmax(1.0, sw);
        all = all && ok;
        printf("n %3d: cycles per append: rows product %lld, columns product %lld, whole append %lld; solve at n %lld; per deletion %lld   |W err| %.2g (scale %.2g) "
               "|S y - d| %.2g  |W err after deletions| %.2g  %s\n", n, cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], ew, sw, ey, ed, ok ? "ok" : "FAIL");
    }
    printf("%s\n", all ? "PASS" : "FAIL");
    return all ? 0 : 1;
}

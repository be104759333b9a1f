// Developer micro-benchmark (not product): the 64 x 64 pivot-tile inverse of the joint solver's sweep (kernels/jqp.hip inv64_lds), the
// dependent chain of a lone joint mission.  One workgroup, 256 threads; times from the 100 MHz wall clock.
// MI355X, round 4: inv64_lds 19.9 us, of which the four 16 x 16 Gauss-Jordan leaves 4 x 2.0 us (300 cycles per column step: the chain
// pivot -> reciprocal (v_rcp_f64 + two Newton steps) -> multiplier -> update).  A leaf that keeps the matrix in registers (pivot row by DPP
// row_newbcast, pivot column by v_permlane32_swap + v_permlane16_swap, no LDS in the loop) measured the same 1.97 us: the reciprocal
// chain, not the data movement, is the leaf's time.  Fetching the operands of a wave's three trailing products together and interleaving
// their MFMA chains made the tile inverse SLOWER (21.7 us against 19.8 us on the same matrix): not kept either.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../swarm_simulator_amd/csrc/kernels -I../../include -o inv64 inv64.hip && ./inv64
#include "../../swarm_simulator_amd/csrc/kernels/jqp.hip"
#include <cstdio>
#include <vector>
#include <cmath>

namespace {
#define REP 200
__global__ __launch_bounds__(256) void k_inv64(const double* A, double* out, long long* t, int variant) {
    __shared__ double Am[JT * LDA];
    __shared__ InvScratch sc;
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    long long acc = 0;
    for (int rep = 0; rep < REP; ++rep) {
        for (int i = threadIdx.x; i < JTT; i += 256) Am[(i >> 6) * LDA + (i & 63)] = A[i];
        __syncthreads();
        const long long t0 = wall_clock64();
        inv64_lds(Am, &sc, &bad);
        __syncthreads();
        acc += wall_clock64() - t0;
    }
    for (int i = threadIdx.x; i < JTT; i += 256) out[i] = Am[(i >> 6) * LDA + (i & 63)];
    if (threadIdx.x == 0) t[0] = acc, t[1] = bad;
}
__global__ __launch_bounds__(64) void k_gj16(const double* A, double* out, long long* t, int variant) {
    __shared__ double D[16 * 18];
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    long long acc = 0;
    bool ok = true;
    for (int rep = 0; rep < REP; ++rep) {
        for (int q = 0; q < 4; ++q) D[r * 18 + 4 * g + q] = A[r * 64 + 4 * g + q];
        JQ_WSYNC();
        const long long t0 = wall_clock64();
        ok = gj16_lds(D, lane);
        JQ_WSYNC();
        acc += wall_clock64() - t0;
    }
    for (int q = 0; q < 4; ++q) out[r * 16 + 4 * g + q] = D[r * 18 + 4 * g + q];
    if (lane == 0) t[0] = acc, t[1] = ok;
}
}  // namespace

int main() {
    std::vector<double> A(JTT), R(JTT);
    // SPD test matrix: diagonally dominant with decaying off-diagonals, condition ~1e6
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) A[i * 64 + j] = (i == j ? 1.0 + 1e-3 * i : 0.0) + 0.9 * std::exp(-0.05 * std::abs(i - j));  // (exponential kernel: SPD)
    for (int i = 0; i < 64; ++i) A[i * 64 + i] += 3.0;
    double *dA, *dO;
    long long* dT;
    hipMalloc(&dA, JTT * 8), hipMalloc(&dO, JTT * 8), hipMalloc(&dT, 16);
    hipMemcpy(dA, A.data(), JTT * 8, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 1; ++variant) {
        long long T[2];
        for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(k_inv64, dim3(1), dim3(256), 0, 0, dA, dO, dT, variant);
        hipMemcpy(T, dT, 16, hipMemcpyDeviceToHost), hipMemcpy(R.data(), dO, JTT * 8, hipMemcpyDeviceToHost);
        double err = 0;  // || A * (-R) - I ||_max
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s = 0;
                for (int k = 0; k < 64; ++k) s += A[i * 64 + k] * -R[k * 64 + j];
                err = std::fmax(err, std::fabs(s - (i == j)));
            }
        printf("inv64 variant %d: %.2f us per call, bad %lld, |A inv - I|max %.2e\n", variant, T[0] * 10.0 / REP / 1e3, T[1], err);
        for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(k_gj16, dim3(1), dim3(64), 0, 0, dA, dO, dT, variant);
        hipMemcpy(T, dT, 16, hipMemcpyDeviceToHost), hipMemcpy(R.data(), dO, 256 * 8, hipMemcpyDeviceToHost);
        err = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int k = 0; k < 16; ++k) s += A[i * 64 + k] * R[k * 16 + j];
                err = std::fmax(err, std::fabs(s - (i == j)));
            }
        printf("gj16  variant %d: %.2f us per call, ok %lld, |A inv - I|max %.2e\n", variant, T[0] * 10.0 / REP / 1e3, T[1], err);
    }
    return 0;
}

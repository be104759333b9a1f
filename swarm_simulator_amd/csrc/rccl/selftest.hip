// selftest.hip -- lib/rbp_rccl_selftest: the RCCL exchange hook on ONE GPU (a pair of one rank: send to self).  Fills a device buffer with a
// pattern, trades it through rbp_rccl_exchange in three sizes (a vector, a 64-agent inverse, a 256-agent inverse), compares.  Prints
// "rbp_rccl selftest ok" and returns 0.  tests/test_gpu_rccl_pair.py runs it on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "rbp_rccl.h"

__global__ void fill(double* p, size_t n, double seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed + 0.5 * (double)i;
}
__global__ void differ(const double* a, const double* b, size_t n, unsigned long long* cnt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__double_as_longlong(a[i]) != __double_as_longlong(b[i])) atomicAdd(cnt, 1ull);
}

int main() {
    unsigned char id[RBP_RCCL_ID_BYTES];
    if (rbp_rccl_unique_id(id)) return std::printf("unique id: %s\n", rbp_rccl_last_error()), 2;
    rbp_rccl_pair* pair = nullptr;
    if (rbp_rccl_pair_create(&pair, 0, 0, 1, id)) return std::printf("pair: %s\n", rbp_rccl_last_error()), 3;
    const size_t sizes[3] = {2304, (size_t)576 * 576, (size_t)2304 * 2304};
    double *send = nullptr, *recv = nullptr;
    unsigned long long* cnt = nullptr;
    if (hipMalloc((void**)&send, sizes[2] * 8) != hipSuccess || hipMalloc((void**)&recv, sizes[2] * 8) != hipSuccess || hipMalloc((void**)&cnt, 8) != hipSuccess) return 4;
    for (int t = 0; t < 3; ++t) {
        const size_t n = sizes[t];
        (void)hipMemset(recv, 0xFF, n * 8);
        (void)hipMemset(cnt, 0, 8);
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, send, n, 1.0 + t);
        if (hipDeviceSynchronize() != hipSuccess) return 5;
        if (rbp_rccl_exchange(pair, send, recv, n * 8)) return std::printf("exchange: %s\n", rbp_rccl_last_error()), 6;
        hipLaunchKernelGGL(differ, dim3(256), dim3(256), 0, 0, send, recv, n, cnt);
        unsigned long long h = 1;
        if (hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost) != hipSuccess) return 7;
        std::printf("exchange of %zu bytes: %llu words differ\n", n * 8, h);
        if (h) return 8;
    }
    // the stream-ordered form (rbp_exchange_stream_fn): fill, exchange and comparison are ENQUEUED on one stream, nothing waits in between
    {
        hipStream_t st;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 9;
        const size_t n = sizes[1];
        (void)hipMemsetAsync(recv, 0xFF, n * 8, st);
        (void)hipMemsetAsync(cnt, 0, 8, st);
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, send, n, 7.0);
        if (rbp_rccl_exchange_stream(pair, send, recv, n * 8, (void*)st)) return std::printf("exchange_stream: %s\n", rbp_rccl_last_error()), 10;
        hipLaunchKernelGGL(differ, dim3(256), dim3(256), 0, st, send, recv, n, cnt);
        unsigned long long h = 1;
        if (hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 11;
        std::printf("stream-ordered exchange of %zu bytes: %llu words differ\n", n * 8, h);
        if (h) return 12;
        (void)hipStreamDestroy(st);
    }
    rbp_rccl_pair_destroy(pair);
    std::printf("rbp_rccl selftest ok\n");
    return 0;
}

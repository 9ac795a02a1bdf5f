// atomic_microbench.hip -- rates of the building blocks of a scatter-add backward on
// gfx950: LDS fp32 atomic add (conflict-free / 2 lanes per address), LDS read-modify-
// write without atomics, coalesced and strided global fp32 atomics.
// (profiles/archive/r01_atomic_microbench.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(float *out, int iters) {
    __shared__ float s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
    __syncthreads();
    const int t = threadIdx.x;
    float v = 1.0f + t * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int a = (it * 37 + j * 257 + (MODE == 1 ? t / 2 : t)) & 8191;  // MODE 1: 2 lanes share an address
            if (MODE == 2) { s[a] = s[a] + v; }           // plain RMW (racy across waves; rate only)
            else if (MODE == 3) { if ((t & 15) == 0) atomicAdd(&s[a], v); }  // 4 of 64 lanes active
            else if (MODE == 4) { if ((t & 63) < 16) atomicAdd(&s[a], v); }  // first 16 lanes active
            else atomicAdd(&s[a], v);
        }
    }
    __syncthreads();
    if (s[t] == 123.456f) out[0] = s[t];
}
template <typename I>
__global__ __launch_bounds__(256) void lds_int_kernel(float *out, int iters) {
    __shared__ I s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0;
    __syncthreads();
    const int t = threadIdx.x;
    I v = (I)(1 + t);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int a = (it * 37 + j * 257 + t) & 4095;
            atomicAdd(&s[a], v);
        }
    }
    __syncthreads();
    if (s[t] == (I)123456) out[0] = 1.0f;
}
template <int STRIDE>
__global__ __launch_bounds__(256) void glob_kernel(float *buf, size_t n, int iters) {
    size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * STRIDE;
    for (int it = 0; it < iters; ++it) {
        size_t a = (base + (size_t)it * 256 * 4096 * STRIDE) % n;
        atomicAdd(buf + a, 1.0f);
    }
}
int main() {
    float *out; CK(hipMalloc(&out, 1 << 20));
    const size_t n = 256ull << 20;  // 1 GiB of floats
    float *buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    hipEvent_t a, z; CK(hipEventCreate(&a)); CK(hipEventCreate(&z));
    auto timeit = [&](auto f) { f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); f(); CK(hipEventRecord(z)); CK(hipEventSynchronize(z)); float ms; CK(hipEventElapsedTime(&ms, a, z)); return ms; };
    const int blocks = 2048, iters = 2000;
    const double ops = (double)blocks * 256 * iters * 16;
    float ms;
    ms = timeit([&] { lds_kernel<0><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add f32, conflict-free   %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU @2.4GHz)\n", ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    ms = timeit([&] { lds_kernel<1><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add f32, 2 lanes/address %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU)\n", ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    ms = timeit([&] { lds_kernel<2><<<blocks, 256>>>(out, iters); });
    printf("lds plain read+add+write            %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU)\n", ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    ms = timeit([&] { lds_kernel<3><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add f32, 4/64 lanes on   %8.3f ms  %8.1f G wave-instr/s (%.1f clk/instr/CU)\n", ms, ops / 64 / ms / 1e6, 256 * 2.4e9 / (ops / 64 / ms * 1e3));
    ms = timeit([&] { lds_kernel<4><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add f32, 16/64 lanes on  %8.3f ms  %8.1f G wave-instr/s (%.1f clk/instr/CU)\n", ms, ops / 64 / ms / 1e6, 256 * 2.4e9 / (ops / 64 / ms * 1e3));
    ms = timeit([&] { lds_kernel<0><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add f32, 64/64 lanes on  %8.3f ms  %8.1f G wave-instr/s (%.1f clk/instr/CU)\n", ms, ops / 64 / ms / 1e6, 256 * 2.4e9 / (ops / 64 / ms * 1e3));
    ms = timeit([&] { lds_int_kernel<unsigned int><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add u32, conflict-free   %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU)\n", ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    ms = timeit([&] { lds_int_kernel<unsigned long long><<<blocks, 256>>>(out, iters); });
    printf("lds atomic add u64, conflict-free   %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU)\n", ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    const int gi = 64;
    const double gops = 4096.0 * 256 * gi;
    ms = timeit([&] { glob_kernel<1><<<4096, 256>>>(buf, n, gi); });
    printf("global atomic add f32, coalesced    %8.3f ms  %8.1f G ops/s\n", ms, gops / ms / 1e6);
    ms = timeit([&] { glob_kernel<16><<<4096, 256>>>(buf, n, gi); });
    printf("global atomic add f32, 64-B stride  %8.3f ms  %8.1f G ops/s\n", ms, gops / ms / 1e6);
    ms = timeit([&] { glob_kernel<311><<<4096, 256>>>(buf, n, gi); });
    printf("global atomic add f32, scattered    %8.3f ms  %8.1f G ops/s\n", ms, gops / ms / 1e6);
    return 0;
}

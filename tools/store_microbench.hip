// store_microbench.hip -- write-pattern calibration for the plane-sweep volume
// (profiles/archive/r01_store_microbench.txt).  Not part of the product library.
//
// The volume write is >99 % of the path's HBM bytes, so the achievable write
// bandwidth of the candidate store shapes bounds every kernel design:
//   fill16   : 16 B/lane, aligned, fully contiguous (ideal)
//   copy16   : 16 B/lane read + write (the guide's 6.3 TB/s reference point)
//   plane2   : 2 B/lane; each wave-store is a 128-B run in a different channel plane
//              (lane = lattice point, loop over channels) -- gather kernel v0
//   plane4   : 4 B/lane (two points per lane)
//   plane16  : 16 B/lane (eight bf16 points per lane), aligned
//   plane16u : 16 B/lane, run start only 2-byte aligned (W_out = 311 rows)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void fill16(uint4 *p, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(i, 1, 2, 3);
    for (; i < n16; i += stride) p[i] = v;
}
__global__ void copy16(const uint4 *s, uint4 *d, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) d[i] = s[i];
}
// planes: C planes of N elements (2 B each). block handles points [blk*256*V, ...), loops channels.
template <int V>  // elements (2 B) per lane
__global__ void plane_store(unsigned short *out, size_t N, int C, int misalign) {
    size_t n = ((size_t)blockIdx.x * 256 + threadIdx.x) * V + misalign;
    if (n + V > N) return;
    unsigned short *o = out + n;
    for (int c = 0; c < C; ++c) {
        if constexpr (V == 1) { *o = (unsigned short)c; }
        else if constexpr (V == 2) { *(unsigned int *)o = c; }
        else if constexpr (V == 4) { *(uint2 *)o = make_uint2(c, c); }
        else { *(uint4 *)o = make_uint4(c, c, c, c); }
        o += N;
    }
}

template <typename F> float timeit(F f, int iters = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const int C = 512; const size_t N = 112ull * 94 * 311;  // one N* volume, bf16
    const size_t bytes = (size_t)C * N * 2;
    unsigned short *buf, *src;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&src, bytes + 4096));
    CK(hipMemset(src, 1, bytes));
    printf("volume bytes %.3f GB\n", bytes / 1e9);
    auto rep = [&](const char *name, float ms, double b) { printf("%-10s %8.3f ms  %8.1f GB/s\n", name, ms, b / ms / 1e6); };
    rep("fill16", timeit([&] { fill16<<<256 * 8, 256>>>((uint4 *)buf, bytes / 16); }), bytes);
    rep("fill16big", timeit([&] { fill16<<<256 * 32, 256>>>((uint4 *)buf, bytes / 16); }), bytes);
    rep("copy16", timeit([&] { copy16<<<256 * 8, 256>>>((const uint4 *)src, (uint4 *)buf, bytes / 16); }), 2.0 * bytes);
    rep("plane2", timeit([&] { plane_store<1><<<(N + 255) / 256, 256>>>(buf, N, C, 0); }), bytes);
    rep("plane4", timeit([&] { plane_store<2><<<(N / 2 + 255) / 256, 256>>>(buf, N, C, 0); }), bytes);
    rep("plane8", timeit([&] { plane_store<4><<<(N / 4 + 255) / 256, 256>>>(buf, N, C, 0); }), bytes);
    rep("plane16", timeit([&] { plane_store<8><<<(N / 8 + 255) / 256, 256>>>(buf, N, C, 0); }), bytes);
    rep("plane16u", timeit([&] { plane_store<8><<<(N / 8 + 255) / 256, 256>>>(buf, N, C, 1); }), bytes);
    rep("plane4u", timeit([&] { plane_store<2><<<(N / 2 + 255) / 256, 256>>>(buf, N, C, 1); }), bytes);
    return 0;
}

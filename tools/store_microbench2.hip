// store_microbench2.hip -- how close to the HBM write peak can a streaming
// store get on this box?  (profiles/archive/r01_store_microbench2.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define NTSTORE(v, ptr) __builtin_nontemporal_store(*(u32x4 *)&(v), (u32x4 *)(ptr))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>  // 0 plain, 1 nontemporal
__global__ void fill16(uint4 *p, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(i, 1, 2, 3);
    for (; i < n16; i += stride) {
        if (MODE == 0) p[i] = v;
        else NTSTORE(v, &p[i]);
    }
}
// each block owns a contiguous chunk (no grid-stride interleave)
__global__ void fill16_chunk(uint4 *p, size_t n16) {
    size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    size_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    uint4 v = make_uint4(lo, 1, 2, 3);
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) p[i] = v;
}
__global__ void read16(const uint4 *p, size_t n16, uint4 *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n16; i += stride) { uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678) *sink = acc;
}
// plane pattern: lane = 8 bf16 points, loop channels; XCD-aware chunk order option; nt option
template <int MODE, int SWZ>
__global__ void plane16(unsigned short *out, size_t N, int C, int nchunks) {
    int bid = blockIdx.x;
    if (SWZ) { int cpx = nchunks / 8; bid = (bid % 8) * cpx + bid / 8; }
    size_t n = ((size_t)bid * 256 + threadIdx.x) * 8;
    if (n + 8 > N) return;
    uint4 *o = (uint4 *)(out + n);
    const size_t pl = N / 8;
    for (int c = 0; c < C; ++c) {
        uint4 v = make_uint4(c, c, c, c);
        if (MODE == 0) *o = v; else NTSTORE(v, o);
        o += pl;
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
    const int C = 512; const size_t N = 112ull * 94 * 311;
    for (int rep = 0; rep < 2; ++rep) {
        const size_t bytes = (size_t)C * N * 2 * (rep ? 8 : 1);  // 3.35 GB and 26.8 GB (B=8)
        unsigned short *buf; CK(hipMalloc(&buf, bytes + 4096));
        uint4 *sink; CK(hipMalloc(&sink, 64));
        printf("== buffer %.2f GB\n", bytes / 1e9);
        auto rep_ = [&](const char *name, float ms, double b) { printf("%-22s %8.3f ms  %8.1f GB/s\n", name, ms, b / ms / 1e6); fflush(stdout); };
        for (int nb : {256, 512, 1024, 2048, 4096, 16384})
            for (int nt : {256, 512, 1024}) {
                char nm[64]; snprintf(nm, 64, "fill16 g%d t%d", nb, nt);
                rep_(nm, timeit([&] { fill16<0><<<nb, nt>>>((uint4 *)buf, bytes / 16); }), bytes);
            }
        rep_("fill16 nt g2048 t256", timeit([&] { fill16<1><<<2048, 256>>>((uint4 *)buf, bytes / 16); }), bytes);
        rep_("fill16 nt g1024 t512", timeit([&] { fill16<1><<<1024, 512>>>((uint4 *)buf, bytes / 16); }), bytes);
        rep_("fill16_chunk g2048", timeit([&] { fill16_chunk<<<2048, 256>>>((uint4 *)buf, bytes / 16); }), bytes);
        rep_("fill16_chunk g256 t1024", timeit([&] { fill16_chunk<<<256, 1024>>>((uint4 *)buf, bytes / 16); }), bytes);
        rep_("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(buf, 1, bytes, 0)); }), bytes);
        rep_("hipMemsetD32Async", timeit([&] { CK(hipMemsetD32Async((hipDeviceptr_t)buf, 7, bytes / 4, 0)); }), bytes);
        rep_("read16 g2048", timeit([&] { read16<<<2048, 256>>>((const uint4 *)buf, bytes / 16, sink); }), bytes);
        rep_("read16 g8192", timeit([&] { read16<<<8192, 256>>>((const uint4 *)buf, bytes / 16, sink); }), bytes);
        if (rep == 0) {
            int nch = (int)(N / 8 / 256);
            rep_("plane16", timeit([&] { plane16<0, 0><<<nch, 256>>>(buf, N, C, nch); }), (double)nch * 256 * 16 * C);
            rep_("plane16 nt", timeit([&] { plane16<1, 0><<<nch, 256>>>(buf, N, C, nch); }), (double)nch * 256 * 16 * C);
            rep_("plane16 xcdswz", timeit([&] { plane16<0, 1><<<nch / 8 * 8, 256>>>(buf, N, C, nch / 8 * 8); }), (double)(nch / 8 * 8) * 256 * 16 * C);
            rep_("plane16 nt xcdswz", timeit([&] { plane16<1, 1><<<nch / 8 * 8, 256>>>(buf, N, C, nch / 8 * 8); }), (double)(nch / 8 * 8) * 256 * 16 * C);
        }
        CK(hipFree(buf)); CK(hipFree(sink));
    }
    return 0;
}

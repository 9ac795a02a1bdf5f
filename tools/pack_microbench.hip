// pack_microbench.hip -- variants of the NCHW -> [B][C/8][H][W][8] re-blocking pre-pass of the plane
// sweep (bf16, N* maps: 2 x 8 x 256 x 94 x 311), timed with HIP events.  The shipped kernel ran at
// 0.20 ms (2.4 TB/s) per launch pair; this finds out what it should cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pack_microbench.hip -o tools/pack_microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned short T;
constexpr int CB = 8;

// v0: the shipped kernel (PPL pixels per lane, 2-byte loads)
template <int PPL>
__global__ __launch_bounds__(256) void pack_v0(const T *__restrict__ src, uint4 *__restrict__ dst, int C, int HW, int nblk)
{
    const int pix0 = blockIdx.x * (256 * PPL) + threadIdx.x;
    const int blk = blockIdx.y, b = blockIdx.z;
    const T *s = src + ((size_t)b * C + (size_t)blk * CB) * HW;
    uint4 *d = dst + ((size_t)b * nblk + blk) * HW;
    T v[PPL][CB];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int pix = pix0 + i * 256;
#pragma unroll
        for (int j = 0; j < CB; ++j) v[i][j] = pix < HW ? s[(size_t)j * HW + pix] : T(0);
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int pix = pix0 + i * 256;
        if (pix < HW) { uint4 q; memcpy(&q, v[i], 16); d[pix] = q; }
    }
}

// v1: 4-byte loads (two pixels per lane and channel; needs HW even), two 16-byte stores per lane
template <int PPL>  // pixel PAIRS per lane
__global__ __launch_bounds__(256) void pack_v1(const T *__restrict__ src, uint4 *__restrict__ dst, int C, int HW, int nblk)
{
    const int pair0 = blockIdx.x * (256 * PPL) + threadIdx.x;
    const int blk = blockIdx.y, b = blockIdx.z;
    const uint32_t *s = (const uint32_t *)(src + ((size_t)b * C + (size_t)blk * CB) * HW);
    uint4 *d = dst + ((size_t)b * nblk + blk) * HW;
    const int HP = HW / 2;
    uint32_t v[PPL][CB];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int pr = pair0 + i * 256;
#pragma unroll
        for (int j = 0; j < CB; ++j) v[i][j] = pr < HP ? s[(size_t)j * HP + pr] : 0u;
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int pr = pair0 + i * 256;
        if (pr < HP) {
            uint4 lo, hi;
            lo.x = (v[i][0] & 0xffffu) | (v[i][1] << 16); hi.x = (v[i][0] >> 16) | (v[i][1] & 0xffff0000u);
            lo.y = (v[i][2] & 0xffffu) | (v[i][3] << 16); hi.y = (v[i][2] >> 16) | (v[i][3] & 0xffff0000u);
            lo.z = (v[i][4] & 0xffffu) | (v[i][5] << 16); hi.z = (v[i][4] >> 16) | (v[i][5] & 0xffff0000u);
            lo.w = (v[i][6] & 0xffffu) | (v[i][7] << 16); hi.w = (v[i][6] >> 16) | (v[i][7] & 0xffff0000u);
            d[2 * pr] = lo;
            d[2 * pr + 1] = hi;
        }
    }
}

// v2: through LDS: a workgroup moves 8 channels x 2048 pixels; reads are 4-byte coalesced rows,
// the transposed image is written to LDS as 2-byte elements and read back as 16-byte pixels
__global__ __launch_bounds__(256) void pack_v2(const T *__restrict__ src, uint4 *__restrict__ dst, int C, int HW, int nblk)
{
    __shared__ __attribute__((aligned(16))) T tile[2048 * CB];
    const int p0 = blockIdx.x * 2048;
    const int blk = blockIdx.y, b = blockIdx.z;
    const uint32_t *s = (const uint32_t *)(src + ((size_t)b * C + (size_t)blk * CB) * HW);
    const int HP = HW / 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pr = p0 / 2 + i * 256 + threadIdx.x;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const uint32_t v = pr < HP ? s[(size_t)j * HP + pr] : 0u;
            const int lp = 2 * (i * 256 + threadIdx.x);
            tile[lp * CB + j] = (T)(v & 0xffffu);
            tile[(lp + 1) * CB + j] = (T)(v >> 16);
        }
    }
    __syncthreads();
    uint4 *d = dst + ((size_t)b * nblk + blk) * HW;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lp = i * 256 + threadIdx.x;
        if (p0 + lp < HW) d[p0 + lp] = ((const uint4 *)tile)[lp];
    }
}

__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

int main()
{
    const int B = 16, C = 256, H = 94, W = 311, HW = H * W, nblk = C / CB;
    const size_t n = (size_t)B * C * HW;
    T *src; uint4 *dst, *ref;
    CK(hipMalloc(&src, n * 2)); CK(hipMalloc(&dst, n * 2)); CK(hipMalloc(&ref, n * 2));
    std::vector<T> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (T)(i * 2654435761u >> 13);
    CK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, bool check) {
        std::vector<float> ts;
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ts.push_back(t);
        }
        std::sort(ts.begin(), ts.end());
        bool ok = true;
        if (check) {
            std::vector<T> a(n), b(n);
            CK(hipMemcpy(a.data(), dst, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref, n * 2, hipMemcpyDeviceToHost));
            ok = memcmp(a.data(), b.data(), n * 2) == 0;
            CK(hipMemset(dst, 0, n * 2));
        }
        printf("%-28s median %7.4f ms  min %7.4f ms  %7.1f GB/s (read+write)%s\n", name, ts[ts.size() / 2], ts[0],
               2.0 * n * 2 / (ts[ts.size() / 2] * 1e-3) / 1e9, check ? (ok ? "  OK" : "  MISMATCH") : "");
    };
    // reference result
    hipLaunchKernelGGL(pack_v0<1>, dim3((HW + 255) / 256, nblk, B), dim3(256), 0, 0, src, ref, C, HW, nblk);
    CK(hipDeviceSynchronize());
    timeit("copy16 (same bytes)", [&] { hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, (const uint4 *)src, dst, n / 8); }, false);
    timeit("v0 ppl=1 (round 1)", [&] { hipLaunchKernelGGL(pack_v0<1>, dim3((HW + 255) / 256, nblk, B), dim3(256), 0, 0, src, dst, C, HW, nblk); }, true);
    timeit("v0 ppl=4 (shipped)", [&] { hipLaunchKernelGGL(pack_v0<4>, dim3((HW + 1023) / 1024, nblk, B), dim3(256), 0, 0, src, dst, C, HW, nblk); }, true);
    timeit("v1 4-byte loads, 1 pair", [&] { hipLaunchKernelGGL(pack_v1<1>, dim3((HW / 2 + 255) / 256, nblk, B), dim3(256), 0, 0, src, dst, C, HW, nblk); }, true);
    timeit("v1 4-byte loads, 2 pairs", [&] { hipLaunchKernelGGL(pack_v1<2>, dim3((HW / 2 + 511) / 512, nblk, B), dim3(256), 0, 0, src, dst, C, HW, nblk); }, true);
    timeit("v1 4-byte loads, 4 pairs", [&] { hipLaunchKernelGGL(pack_v1<4>, dim3((HW / 2 + 1023) / 1024, nblk, B), dim3(256), 0, 0, src, dst, C, HW, nblk); }, true);
    timeit("v2 LDS transpose", [&] { hipLaunchKernelGGL(pack_v2, dim3((HW + 2047) / 2048, nblk, B), dim3(256), 0, 0, src, dst, C, HW, nblk); }, true);
    return 0;
}

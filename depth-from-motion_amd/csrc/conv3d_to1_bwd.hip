// conv3d_to1_bwd.hip -- backward of the prediction heads' Conv3d(32 -> 1, 3, 1, 1) (mmdet3d/models/backbones/
// dfm_backbone.py:120-127), round 6.
//
// The forward is out[v] = sum_c sum_t W[c][t] x[c][v + t - 1] (t = (kd, kh, kw), zero padding).  Until round 6 the
// backward ran the 32 -> 32 machinery on a gradient zero-padded to 32 channels: a 118 MB zero fill, a strided copy,
// the 32 -> 32 MFMA convolution (backward-data), the 32 x 32 weight-gradient kernel and its 56 MB reduction -- ~0.4 ms
// and ten launches per head for two operations that are memory-bound at ~30 us each (a 118 MB write; a 118 MB read).
// Both are one matrix product per 32 (or 16) voxels with the 27 TAPS as a matrix dimension:
//
//   backward-data   gx[c][u] = sum_t W[c][t] g[u - t + 1]         D[c][voxel] = A[c][tap] B[tap][voxel]
//                   A = the weight (32 x 27, zero-padded to 32 taps: two k-steps of v_mfma_f32_32x32x16_bf16),
//                   B = the one-channel gradient read at the 27 shifted positions (2-byte loads, L1 / L2 resident:
//                   the gradient is 3.7 MB), output rows stored channels-last with 16-byte stores;
//   weight gradient gw[c][t] = sum_u x[c][u] g[u - t + 1]         D[tap][c] += A[tap][voxel] B[voxel][c]
//                   A = the gradient at the lane's tap offset for 16 consecutive voxels, B = x as it lies in memory
//                   (lane = channel: the 32 lanes of a voxel read its 64 contiguous bytes); a wave keeps the 32 x 32
//                   accumulator tile over all its rows, a workgroup adds its four waves' tiles in LDS and writes
//                   ONE 4 KiB partial; a second kernel adds the partials in a fixed order (deterministic, no atomics).
// No LDS, no packed weights, no padded tensors.
#include <algorithm>

#include "dfm_common.h"
#include "dfm_hip.h"

using namespace dfm;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct T1bGeom {
    int32_t N, D, H, W, rows_w, nrows;  // rows_w = ceil(W / 32); nrows = N * D * H * rows_w
};

__device__ __forceinline__ uint16_t ld_u16(const bf16_t *p) { return *(const uint16_t *)p; }

// tap k of this lane in k-step ks, element j: (kd, kh, kw) -> the offsets of g relative to the voxel, packed
// (dd + 1) | (dh + 1) << 2 | (dw + 1) << 4, or -1 for the padding taps 27 .. 31
__device__ __forceinline__ int tap_delta(int tap)
{
    if (tap >= 27) return -1;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    return (2 - kd) | ((2 - kh) << 2) | ((2 - kw) << 4);  // (1 - k) + 1
}

// ---- backward-data -----------------------------------------------------------------------------------------------
template <typename TW>
__global__ __launch_bounds__(256) void conv3d_to1_bwd_data_kernel(T1bGeom g, const bf16_t *__restrict__ gy,
                                                                  const TW *__restrict__ weight,
                                                                  bf16_t *__restrict__ gx)
{
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    // A fragments: lane (c = l32, taps ks * 16 + half * 8 + j)
    bf16x8_t wa[2];
    int delta[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = ks * 16 + half * 8 + j;
            const float wv = tap < 27 ? elem<TW>::load(weight[l32 * 27 + tap]) : 0.0f;
            const bf16_t wb = f32_to_bf16(wv);
            __builtin_memcpy((char *)&wa[ks] + 2 * j, &wb, 2);
            delta[ks][j] = tap_delta(tap);
        }
    // two rows in flight per wave: the second row's 16 loads are issued before the first row's MFMAs wait for theirs
    constexpr int RU = 2;
    for (int row0 = wave_g; row0 < g.nrows; row0 += RU * nwaves) {
        int n[RU], d[RU], h[RU], w[RU];
        bool live[RU];
        bf16x8_t gb[RU][2];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = row0 + u * nwaves;
            live[u] = row < g.nrows;
            int t = live[u] ? row : 0;
            const int w0 = (t % g.rows_w) * 32; t /= g.rows_w;
            h[u] = t % g.H; t /= g.H;
            d[u] = t % g.D;
            n[u] = t / g.D;
            w[u] = w0 + l32;   // this lane's voxel (B operand column)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int dl = delta[ks][j];
                    const int dd = d[u] + (dl & 3) - 1, hh = h[u] + ((dl >> 2) & 3) - 1, ww = w[u] + ((dl >> 4) & 3) - 1;
                    const bool ok = live[u] && dl >= 0 && (unsigned)dd < (unsigned)g.D && (unsigned)hh < (unsigned)g.H &&
                                    (unsigned)ww < (unsigned)g.W;
                    const uint16_t v = ok ? ld_u16(gy + (((size_t)n[u] * g.D + dd) * g.H + hh) * g.W + ww) : (uint16_t)0;
                    __builtin_memcpy((char *)&gb[u][ks] + 2 * j, &v, 2);
                }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], gb[u][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], gb[u][1], acc, 0, 0, 0);
            // acc[r]: row (channel) 8 (r >> 2) + 4 half + (r & 3), column (voxel) l32 -> the layout of the
            // convolutions' epilogues: two 16-byte stores per lane after the halves' exchange
            dfm_u32x2 pk[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                pk[gq] = dfm_u32x2{pack_bf16x2(acc[4 * gq], acc[4 * gq + 1]), pack_bf16x2(acc[4 * gq + 2], acc[4 * gq + 3])};
            dfm_u32x4 q16[2];
            acc_rows_to_16B(pk, q16);
            if (live[u] && w[u] < g.W) {
                bf16_t *o = gx + ((((size_t)n[u] * g.D + d[u]) * g.H + h[u]) * g.W + w[u]) * 32;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) *(dfm_u32x4 *)(o + 16 * pr + 8 * half) = q16[pr];
            }
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3d_to1_wgrad_kernel(T1bGeom g, const bf16_t *__restrict__ x,
                                                               const bf16_t *__restrict__ gy,
                                                               float *__restrict__ part)
{
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int dl = tap_delta(l32);  // the lane's tap (A operand row)
    const int od = (dl & 3) - 1, oh = ((dl >> 2) & 3) - 1, ow = ((dl >> 4) & 3) - 1;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    constexpr int RU = 2;   // rows in flight per wave
    for (int row0 = wave_g; row0 < g.nrows; row0 += RU * nwaves) {
        bf16x8_t ga[RU][2], xb[RU][2];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = row0 + u * nwaves;
            const bool live = row < g.nrows;
            int t = live ? row : 0;
            const int w0 = (t % g.rows_w) * 32; t /= g.rows_w;
            const int h = t % g.H; t /= g.H;
            const int d = t % g.D;
            const int n = t / g.D;
            const int dd = d + od, hh = h + oh;
            const bool rok = live && dl >= 0 && (unsigned)dd < (unsigned)g.D && (unsigned)hh < (unsigned)g.H;
            const bf16_t *grow = gy + (((size_t)n * g.D + (rok ? dd : 0)) * g.H + (rok ? hh : 0)) * g.W;
            const bf16_t *xrow = x + ((((size_t)n * g.D + d) * g.H + h) * g.W) * 32 + l32;  // lane = channel (B column)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int w = w0 + ks * 16 + half * 8 + j;   // the voxel (k index)
                    const int ww = w + ow;
                    const uint16_t gv = (rok && w < g.W && (unsigned)ww < (unsigned)g.W) ? ld_u16(grow + ww) : (uint16_t)0;
                    const uint16_t xv = (live && w < g.W) ? ld_u16(xrow + (size_t)w * 32) : (uint16_t)0;
                    __builtin_memcpy((char *)&ga[u][ks] + 2 * j, &gv, 2);
                    __builtin_memcpy((char *)&xb[u][ks] + 2 * j, &xv, 2);
                }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[u][ks], xb[u][ks], acc, 0, 0, 0);
    }
    // the workgroup's partial: its four waves' tiles added in a fixed order through LDS -> part[workgroup][tap][channel]
    __shared__ float sh[4][1024];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) sh[wave][(8 * (r >> 2) + 4 * half + (r & 3)) * 32 + l32] = acc[r];
    __syncthreads();
    float *p = part + (size_t)blockIdx.x * 1024;
    for (int i = threadIdx.x; i < 1024; i += 256) p[i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

// out[c][tap] = sum over the waves' partials, in a fixed order: one workgroup per tap, thread = (channel, slice of waves)
template <typename TO>
__global__ __launch_bounds__(1024) void conv3d_to1_wgrad_reduce_kernel(const float *__restrict__ part, int nparts,
                                                                       TO *__restrict__ out)
{
    __shared__ float sh[32][32];
    const int tap = blockIdx.x, c = threadIdx.x & 31, s = threadIdx.x >> 5;
    float a = 0.0f;
    for (int w = s; w < nparts; w += 32) a += part[(size_t)w * 1024 + tap * 32 + c];
    sh[s][c] = a;
    __syncthreads();
    if (s == 0) {
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) r += sh[k][c];
        out[c * 27 + tap] = elem<TO>::store(r);
    }
}

constexpr int T1B_WGS = 1024;  // workgroups of the weight gradient (four per CU): as many partials of 4 KiB

int t1b_geom(int32_t n, int32_t d, int32_t h, int32_t w, T1bGeom &g)
{
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    const long long rows_w = (w + 31) / 32, nrows = (long long)n * d * h * rows_w;
    if (nrows >= (1ll << 31) || (long long)n * d * h * w * 32 >= (1ll << 40))
        return set_error(DFM_ERR_UNSUPPORTED, "volume too large");
    g.N = n; g.D = d; g.H = h; g.W = w; g.rows_w = (int)rows_w; g.nrows = (int)nrows;
    return DFM_OK;
}

}  // namespace

// grad_out : (n, d, h, w) bf16 (the (n, 1, d, h, w) gradient of the convolution's output) [device]
// weight   : (1, 32, 3, 3, 3) in weight_dtype (DFM_F32 | DFM_BF16) [device]
// grad_x   : (n, d, h, w, 32) bf16 channels-last, 16-byte aligned, OVERWRITTEN
extern "C" DFM_API int dfm_conv3d_to1_bwd_data(int32_t n, int32_t d, int32_t h, int32_t w, const void *grad_out,
                                               const void *weight, int32_t weight_dtype, void *grad_x, void *stream)
{
    T1bGeom g;
    const int rc = t1b_geom(n, d, h, w, g);
    if (rc != DFM_OK) return rc;
    if (!grad_out || !weight || !grad_x) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    if ((uintptr_t)grad_x & 15) return set_error(DFM_ERR_INVALID_ARG, "grad_x must be 16-byte aligned");
    const int wgs = (int)std::min<long long>(2048, ((long long)g.nrows + 7) / 8);
    hipStream_t st = (hipStream_t)stream;
    if (weight_dtype == DFM_F32)
        hipLaunchKernelGGL(conv3d_to1_bwd_data_kernel<float>, dim3(wgs), dim3(256), 0, st, g, (const bf16_t *)grad_out,
                           (const float *)weight, (bf16_t *)grad_x);
    else
        hipLaunchKernelGGL(conv3d_to1_bwd_data_kernel<bf16_t>, dim3(wgs), dim3(256), 0, st, g, (const bf16_t *)grad_out,
                           (const bf16_t *)weight, (bf16_t *)grad_x);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API size_t dfm_conv3d_to1_wgrad_workspace_bytes(void) { return (size_t)T1B_WGS * 1024 * sizeof(float); }

// x : (n, d, h, w, 32) bf16 channels-last; grad_out : (n, d, h, w) bf16; grad_weight : (1, 32, 3, 3, 3) in out_dtype,
// OVERWRITTEN; workspace >= dfm_conv3d_to1_wgrad_workspace_bytes()
extern "C" DFM_API int dfm_conv3d_to1_wgrad(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                            const void *grad_out, void *grad_weight, int32_t out_dtype,
                                            void *workspace, size_t workspace_bytes, void *stream)
{
    T1bGeom g;
    const int rc = t1b_geom(n, d, h, w, g);
    if (rc != DFM_OK) return rc;
    if (!x || !grad_out || !grad_weight || !workspace) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (out_dtype != DFM_F32 && out_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "gradient dtype must be DFM_F32 or DFM_BF16");
    if (workspace_bytes < dfm_conv3d_to1_wgrad_workspace_bytes())
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_conv3d_to1_wgrad_workspace_bytes");
    const int wgs = (int)std::min<long long>(T1B_WGS, ((long long)g.nrows + 7) / 8);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv3d_to1_wgrad_kernel, dim3(wgs), dim3(256), 0, st, g, (const bf16_t *)x,
                       (const bf16_t *)grad_out, (float *)workspace);
    if (out_dtype == DFM_F32)
        hipLaunchKernelGGL(conv3d_to1_wgrad_reduce_kernel<float>, dim3(27), dim3(1024), 0, st, (const float *)workspace,
                           wgs, (float *)grad_weight);
    else
        hipLaunchKernelGGL(conv3d_to1_wgrad_reduce_kernel<bf16_t>, dim3(27), dim3(1024), 0, st, (const float *)workspace,
                           wgs, (bf16_t *)grad_weight);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

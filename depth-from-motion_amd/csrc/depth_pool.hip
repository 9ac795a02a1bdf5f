// depth_pool.hip -- AvgPool3d((k, 1, 1)) of FrustumToVoxel on the NDHWC stack, forward and backward
// (mmdet3d/models/necks/feature_transformation.py:167: the voxel volume's height axis pooled 20 -> 5).
//
// On a channels-last (N, C, D, H, W) tensor the pooled axis is the slowest one inside a sample: the tensor is
// (outer = N * D / k, k, inner = H * W * C) and the pool a mean over the middle axis.  torch ran it as
// float() -> mean(dim) -> to(bf16) (three kernels, the fp32 copy of the 112 MB volume in between) and the same
// chain backwards: 0.4 ms of a training step at config K.  Here: one pass each way, 16-byte vectors, fp32 sums,
// one rounding -- the values of the torch chain (a sum of k <= 8 bf16 values is exact in fp32 unless their exponents
// differ by more than 16 bits; 1 / k is exact for k a power of two).
//   forward : y[o][i] = (sum_j x[o][j][i]) / k
//   backward: gx[o][j][i] = gy[o][i] / k
#include <algorithm>

#include "dfm_common.h"

using namespace dfm;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void depth_pool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                             long long outer, int k, long long inner_vec)
{
    constexpr int VEC = vec16<T>::N;
    const long long total = outer * inner_vec;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const long long o = v / inner_vec, i = v - o * inner_vec;
        const T *src = x + ((size_t)o * k * inner_vec + i) * VEC;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
        for (int j = 0; j < k; ++j) {
            float f[VEC];
            load16<T>(src + (size_t)j * inner_vec * VEC, f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = acc[e] / (float)k;
        store16<T>(y + (size_t)v * VEC, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void depth_pool_bwd_kernel(const T *__restrict__ gy, T *__restrict__ gx,
                                                             long long outer, int k, long long inner_vec)
{
    constexpr int VEC = vec16<T>::N;
    const long long total = outer * inner_vec;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const long long o = v / inner_vec, i = v - o * inner_vec;
        float f[VEC];
        load16<T>(gy + (size_t)v * VEC, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = f[e] / (float)k;
        T *dst = gx + ((size_t)o * k * inner_vec + i) * VEC;
        for (int j = 0; j < k; ++j) store16<T>(dst + (size_t)j * inner_vec * VEC, f);
    }
}

template <typename T>
int pool_launch(bool bwd, const void *a, void *b, long long outer, int k, long long inner, hipStream_t st)
{
    constexpr int VEC = vec16<T>::N;
    const long long iv = inner / VEC, total = outer * iv;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 16);
    if (bwd)
        hipLaunchKernelGGL(depth_pool_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T *)a, (T *)b, outer, k, iv);
    else
        hipLaunchKernelGGL(depth_pool_fwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T *)a, (T *)b, outer, k, iv);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

int pool_check(const void *a, const void *b, int64_t outer, int32_t k, int64_t inner, int32_t dtype)
{
    if (!a || !b) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (outer <= 0 || k <= 0 || inner <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    if (dtype != DFM_F32 && dtype != DFM_BF16) return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    const int vec = dtype == DFM_BF16 ? 8 : 4;
    if (inner % vec || ((uintptr_t)a & 15) || ((uintptr_t)b & 15))
        return set_error(DFM_ERR_UNSUPPORTED, "depth pool: inner extent in whole 16-byte vectors, 16-byte aligned buffers");
    return DFM_OK;
}

}  // namespace

extern "C" DFM_API int dfm_depth_pool_fwd(int64_t outer, int32_t k, int64_t inner, int32_t dtype, const void *x,
                                          void *y, void *stream)
{
    const int rc = pool_check(x, y, outer, k, inner, dtype);
    if (rc != DFM_OK) return rc;
    return dtype == DFM_BF16 ? pool_launch<bf16_t>(false, x, y, outer, k, inner, (hipStream_t)stream)
                             : pool_launch<float>(false, x, y, outer, k, inner, (hipStream_t)stream);
}

extern "C" DFM_API int dfm_depth_pool_bwd(int64_t outer, int32_t k, int64_t inner, int32_t dtype, const void *grad_y,
                                          void *grad_x, void *stream)
{
    const int rc = pool_check(grad_y, grad_x, outer, k, inner, dtype);
    if (rc != DFM_OK) return rc;
    return dtype == DFM_BF16 ? pool_launch<bf16_t>(true, grad_y, grad_x, outer, k, inner, (hipStream_t)stream)
                             : pool_launch<float>(true, grad_y, grad_x, outer, k, inner, (hipStream_t)stream);
}

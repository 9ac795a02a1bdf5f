// bilinear_bwd.hip -- backward of F.interpolate(mode='bilinear') on an NHWC map as a GATHER (round 6).
//
// The 2-D necks either side of the path up-sample their maps bilinearly (mmdet3d/models/necks/spp_unet_neck.py:60-70,
// 83-91: nn.Upsample(scale_factor=2) between the up-convolutions, the SPP branches' resize to the map's size).
// Bilinear resampling is linear and separable: gX = A_h^T gY A_w with the 1-D interpolation matrices A.  ATen's
// backward scatters with atomics (0.67 ms per call on these maps, round 4); round 4 replaced it with the two dense
// matrix products -- exact, but an interpolation matrix has two non-zeros per row, and in fp32 the products were the
// largest ATen item left in the training step (0.75 ms, profiles/r06_c32_*).  Here a lane owns one input pixel's
// 16-byte channel vector and walks the few output pixels that interpolate from it: the non-zeros of its row of
// A_h^T and of A_w^T, handed over as padded index / weight tables built from ATen's own forward (the host mirrors
// F.interpolate's weights exactly: modules._interp_matrix).  One read of the gradient, one write of the result, fp32
// sums rounded once.
#include "dfm_common.h"

using namespace dfm;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_nhwc_kernel(int C, int h_in, int w_in, int h_out, int w_out,
                                                                const T *__restrict__ gy,
                                                                const int32_t *__restrict__ row_idx,
                                                                const float *__restrict__ row_w, int kh,
                                                                const int32_t *__restrict__ col_idx,
                                                                const float *__restrict__ col_w, int kw,
                                                                T *__restrict__ gx, long long total)
{
    constexpr int VEC = vec16<T>::N;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int nvb = C / VEC;
    const int vb = (int)(i % nvb);
    long long p = i / nvb;
    const int wi = (int)(p % w_in);
    p /= w_in;
    const int hi = (int)(p % h_in);
    const int b = (int)(p / h_in);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    const T *gb = gy + (size_t)b * h_out * w_out * C + (size_t)vb * VEC;
    for (int a = 0; a < kh; ++a) {
        const float wh = row_w[hi * kh + a];
        if (wh == 0.0f) continue;  // (padding of the table)
        const T *gr = gb + (size_t)row_idx[hi * kh + a] * w_out * C;
        for (int c = 0; c < kw; ++c) {
            const float ww = col_w[wi * kw + c];
            if (ww == 0.0f) continue;
            float f[VEC];
            load16<T>(gr + (size_t)col_idx[wi * kw + c] * C, f);
            const float wt = wh * ww;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(f[k], wt, acc[k]);
        }
    }
    store16<T>(gx + (((size_t)b * h_in + hi) * w_in + wi) * C + (size_t)vb * VEC, acc);
}

}  // namespace

// gy: (n, h_out, w_out, c) [device], gx: (n, h_in, w_in, c), both `dtype`, 16-byte aligned, c a multiple of the
// 16-byte vector (8 bf16 / 4 fp32).  row_idx / row_w: [h_in][kh] -- the output rows that interpolate from input row
// hi and their weights, padded with weight 0; col_idx / col_w: [w_in][kw] likewise.  gx is OVERWRITTEN.
extern "C" DFM_API int dfm_bilinear_resize_bwd_nhwc(int32_t n, int32_t c, int32_t h_in, int32_t w_in, int32_t h_out,
                                                    int32_t w_out, int32_t dtype, const void *gy,
                                                    const int32_t *row_idx, const float *row_w, int32_t kh,
                                                    const int32_t *col_idx, const float *col_w, int32_t kw,
                                                    void *gx, void *stream)
{
    if (n <= 0 || c <= 0 || h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0 || kh <= 0 || kw <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_bilinear_resize_bwd_nhwc");
    if (dtype != DFM_F32 && dtype != DFM_BF16) return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!gy || !gx || !row_idx || !row_w || !col_idx || !col_w) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const int vec = dtype == DFM_BF16 ? 8 : 4;
    if (c % vec || ((uintptr_t)gy & 15) || ((uintptr_t)gx & 15))
        return set_error(DFM_ERR_UNSUPPORTED, "channels must be whole 16-byte vectors of 16-byte aligned tensors");
    const long long total = (long long)n * h_in * w_in * (c / vec);
    const long long blocks = (total + 255) / 256;
    if (blocks >= (1ll << 31)) return set_error(DFM_ERR_UNSUPPORTED, "map too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFM_F32)
        hipLaunchKernelGGL(bilinear_bwd_nhwc_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, c, h_in, w_in, h_out,
                           w_out, (const float *)gy, row_idx, row_w, kh, col_idx, col_w, kw, (float *)gx, total);
    else
        hipLaunchKernelGGL(bilinear_bwd_nhwc_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, c, h_in, w_in, h_out,
                           w_out, (const bf16_t *)gy, row_idx, row_w, kh, col_idx, col_w, kw, (bf16_t *)gx, total);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

// depth_head.hip -- DepthHead.forward (with_convs=False) fused (gfx950)
//
// Reference: mmdet3d/models/dense_heads/depth_head.py:205-210 --
//   depth_volumes = Upsample(x4, trilinear, align_corners=True)(cost)
//   softmax       = softmax(depth_volumes, dim=depth)
//   depth_preds   = sum(softmax * depth_samples, dim=depth)
// The reference materialises and re-reads three (B,1,4D,4H,4W) tensors; here one
// launch writes the two volumes and the map, reading only the 1/64-size cost.
// One lane = one output pixel column (all 4D depths); lanes adjacent along w,
// so every store instruction of a wave is one contiguous run.  The logits are
// recomputed from the (L1/L2-resident) cost in each of the three column passes
// (max, sum of exp, normalise) instead of being re-read from HBM.
// Upsample arithmetic = ATen's: fma(w0, a, w1*b) nested W -> H -> D (bit-exact
// volume); softmax uses expf (equal to torch's Sleef exp to rounding error).
// Bound: HBM write (2 volumes), ~8 cached loads + 1 exp per element.
#include "dfm_common.h"

using namespace dfm;

namespace {

struct UpIdx {
    int i0, i1;
    float w0, w1;
};

__device__ __forceinline__ UpIdx up_index(int i, int in, int out)
{
    UpIdx u;
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    const float real = scale * (float)i;
    int a = (int)floorf(real);
    a = min(a, in - 1);
    float l = real - (float)a;
    l = fminf(fmaxf(l, 0.0f), 1.0f);
    u.i0 = a;
    u.i1 = min(a + 1, in - 1);
    u.w1 = l;
    u.w0 = 1.0f - l;
    return u;
}

__device__ __forceinline__ float lerp_fma(float w0, float a, float w1, float b)
{
    return __builtin_fmaf(w0, a, w1 * b);
}

template <typename T>
__global__ __launch_bounds__(256) void depth_head_kernel(const T *__restrict__ in, int D, int H,
                                                         int W, int s,
                                                         const float *__restrict__ depth_samples,
                                                         T *__restrict__ vol, T *__restrict__ soft,
                                                         T *__restrict__ pred)
{
    const int Do = D * s, Ho = H * s, Wo = W * s;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (pix >= Ho * Wo) return;
    const int h = pix / Wo, w = pix - h * Wo;
    const UpIdx uh = up_index(h, H, Ho), uw = up_index(w, W, Wo);
    const T *x = in + (size_t)b * D * H * W;
    const int o00 = uh.i0 * W + uw.i0, o01 = uh.i0 * W + uw.i1;
    const int o10 = uh.i1 * W + uw.i0, o11 = uh.i1 * W + uw.i1;
    const size_t plane_o = (size_t)Ho * Wo;
    T *vcol = vol + (size_t)b * Do * plane_o + pix;
    T *scol = soft + (size_t)b * Do * plane_o + pix;

    auto logit = [&](int d) {
        const UpIdx ud = up_index(d, D, Do);
        const T *p0 = x + (size_t)ud.i0 * H * W, *p1 = x + (size_t)ud.i1 * H * W;
        const float a0 = lerp_fma(uw.w0, elem<T>::load(p0[o00]), uw.w1, elem<T>::load(p0[o01]));
        const float b0 = lerp_fma(uw.w0, elem<T>::load(p0[o10]), uw.w1, elem<T>::load(p0[o11]));
        const float a1 = lerp_fma(uw.w0, elem<T>::load(p1[o00]), uw.w1, elem<T>::load(p1[o01]));
        const float b1 = lerp_fma(uw.w0, elem<T>::load(p1[o10]), uw.w1, elem<T>::load(p1[o11]));
        const float v = lerp_fma(ud.w0, lerp_fma(uh.w0, a0, uh.w1, b0), ud.w1,
                                 lerp_fma(uh.w0, a1, uh.w1, b1));
        // the reference's softmax reads depth_volumes in its storage type
        return elem<T>::load(elem<T>::store(v));
    };

    float mx = -INFINITY;
    for (int d = 0; d < Do; ++d) {
        const float v = logit(d);
        vcol[(size_t)d * plane_o] = elem<T>::store(v);
        mx = fmaxf(mx, v);
    }
    float sum = 0.0f;
    for (int d = 0; d < Do; ++d) sum = sum + expf(logit(d) - mx);
    float acc = 0.0f;
    for (int d = 0; d < Do; ++d) {
        const float pr = expf(logit(d) - mx) / sum;
        const T st = elem<T>::store(pr);
        scol[(size_t)d * plane_o] = st;
        acc = acc + elem<T>::load(st) * depth_samples[d];
    }
    pred[(size_t)b * plane_o + pix] = elem<T>::store(acc);
}

}  // namespace

extern "C" {

DFM_API int dfm_depth_head_fwd(int32_t batch, int32_t d, int32_t h, int32_t w, int32_t scale,
                               int32_t dtype, const void *cost, const float *depth_samples,
                               void *depth_volumes, void *softmax, void *depth_preds, void *stream)
{
    if (batch <= 0 || d <= 0 || h <= 0 || w <= 0 || scale <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_depth_head_fwd");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!cost || !depth_samples || !depth_volumes || !softmax || !depth_preds)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (batch > 65535 || (long long)h * scale * w * scale >= (1ll << 31))
        return set_error(DFM_ERR_UNSUPPORTED, "shape too large");
    const int npix = h * scale * w * scale;
    dim3 grid((npix + 255) / 256, batch);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFM_F32)
        hipLaunchKernelGGL(depth_head_kernel<float>, grid, dim3(256), 0, st, (const float *)cost, d,
                           h, w, scale, depth_samples, (float *)depth_volumes, (float *)softmax,
                           (float *)depth_preds);
    else
        hipLaunchKernelGGL(depth_head_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)cost,
                           d, h, w, scale, depth_samples, (bf16_t *)depth_volumes,
                           (bf16_t *)softmax, (bf16_t *)depth_preds);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// backward: gradients arriving on depth_volumes, softmax and depth_preds are
// folded into one logit gradient per column
//   gl[d] = g_vol[d] + p[d] * (s[d] - sum_k p[k] s[k]),  s[d] = g_soft[d] + g_pred * depth[d]
// and scattered through the trilinear weights (fp32 atomics, zero-initialised).
// Any of the three incoming gradients may be NULL.
// ---------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void depth_head_bwd_kernel(
    const T *__restrict__ in, int D, int H, int W, int s, const float *__restrict__ depth_samples,
    const T *__restrict__ gvol, const T *__restrict__ gsoft, const T *__restrict__ gpred,
    float *__restrict__ gin)
{
    const int Do = D * s, Ho = H * s, Wo = W * s;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (pix >= Ho * Wo) return;
    const int h = pix / Wo, w = pix - h * Wo;
    const UpIdx uh = up_index(h, H, Ho), uw = up_index(w, W, Wo);
    const T *x = in + (size_t)b * D * H * W;
    float *gx = gin + (size_t)b * D * H * W;
    const int o00 = uh.i0 * W + uw.i0, o01 = uh.i0 * W + uw.i1;
    const int o10 = uh.i1 * W + uw.i0, o11 = uh.i1 * W + uw.i1;
    const size_t plane_o = (size_t)Ho * Wo;
    const size_t col = (size_t)b * Do * plane_o + pix;
    auto logit = [&](int d) {
        const UpIdx ud = up_index(d, D, Do);
        const T *p0 = x + (size_t)ud.i0 * H * W, *p1 = x + (size_t)ud.i1 * H * W;
        const float a0 = lerp_fma(uw.w0, elem<T>::load(p0[o00]), uw.w1, elem<T>::load(p0[o01]));
        const float b0 = lerp_fma(uw.w0, elem<T>::load(p0[o10]), uw.w1, elem<T>::load(p0[o11]));
        const float a1 = lerp_fma(uw.w0, elem<T>::load(p1[o00]), uw.w1, elem<T>::load(p1[o01]));
        const float b1 = lerp_fma(uw.w0, elem<T>::load(p1[o10]), uw.w1, elem<T>::load(p1[o11]));
        return lerp_fma(ud.w0, lerp_fma(uh.w0, a0, uh.w1, b0), ud.w1, lerp_fma(uh.w0, a1, uh.w1, b1));
    };
    const bool soft_path = gsoft || gpred;
    float mx = -INFINITY, sum = 0.0f, dotps = 0.0f;
    const float gp = gpred ? elem<T>::load(gpred[(size_t)b * plane_o + pix]) : 0.0f;
    if (soft_path) {
        for (int d = 0; d < Do; ++d) mx = fmaxf(mx, logit(d));
        for (int d = 0; d < Do; ++d) sum += expf(logit(d) - mx);
        for (int d = 0; d < Do; ++d) {
            const float p = expf(logit(d) - mx) / sum;
            const float sd = (gsoft ? elem<T>::load(gsoft[col + (size_t)d * plane_o]) : 0.0f) +
                             gp * depth_samples[d];
            dotps += p * sd;
        }
    }
    for (int d = 0; d < Do; ++d) {
        float gl = gvol ? elem<T>::load(gvol[col + (size_t)d * plane_o]) : 0.0f;
        if (soft_path) {
            const float p = expf(logit(d) - mx) / sum;
            const float sd = (gsoft ? elem<T>::load(gsoft[col + (size_t)d * plane_o]) : 0.0f) +
                             gp * depth_samples[d];
            gl += p * (sd - dotps);
        }
        if (gl == 0.0f) continue;
        const UpIdx ud = up_index(d, D, Do);
        float *g0 = gx + (size_t)ud.i0 * H * W, *g1 = gx + (size_t)ud.i1 * H * W;
        const float c0 = gl * ud.w0, c1 = gl * ud.w1;
        atomicAdd(g0 + o00, c0 * uh.w0 * uw.w0); atomicAdd(g0 + o01, c0 * uh.w0 * uw.w1);
        atomicAdd(g0 + o10, c0 * uh.w1 * uw.w0); atomicAdd(g0 + o11, c0 * uh.w1 * uw.w1);
        atomicAdd(g1 + o00, c1 * uh.w0 * uw.w0); atomicAdd(g1 + o01, c1 * uh.w0 * uw.w1);
        atomicAdd(g1 + o10, c1 * uh.w1 * uw.w0); atomicAdd(g1 + o11, c1 * uh.w1 * uw.w1);
    }
}

}  // namespace

extern "C" DFM_API int dfm_depth_head_bwd(int32_t batch, int32_t d, int32_t h, int32_t w,
                                          int32_t scale, int32_t dtype, const void *cost,
                                          const float *depth_samples, const void *grad_volumes,
                                          const void *grad_softmax, const void *grad_preds,
                                          float *grad_cost, void *stream)
{
    if (batch <= 0 || d <= 0 || h <= 0 || w <= 0 || scale <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_depth_head_bwd");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!cost || !depth_samples || !grad_cost)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const int npix = h * scale * w * scale;
    dim3 grid((npix + 255) / 256, batch);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFM_F32)
        hipLaunchKernelGGL(depth_head_bwd_kernel<float>, grid, dim3(256), 0, st, (const float *)cost,
                           d, h, w, scale, depth_samples, (const float *)grad_volumes,
                           (const float *)grad_softmax, (const float *)grad_preds, grad_cost);
    else
        hipLaunchKernelGGL(depth_head_bwd_kernel<bf16_t>, grid, dim3(256), 0, st,
                           (const bf16_t *)cost, d, h, w, scale, depth_samples,
                           (const bf16_t *)grad_volumes, (const bf16_t *)grad_softmax,
                           (const bf16_t *)grad_preds, grad_cost);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

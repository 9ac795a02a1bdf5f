// depth_loss.hip -- per-pixel depth-distribution losses of DepthHead.loss
// (reference: mmdet3d/models/dense_heads/depth_head.py:75-188, called from
// mmdet3d/models/detectors/dfm.py:348-356).
//
// The reference gathers the D logits of every pixel with a valid ground-truth depth
// (`depth_volumes.permute(0,2,3,1)[mask]`, a (n_valid, D) copy), takes log_softmax, builds a
// soft target over the bins and reduces.  Here one lane owns one pixel and walks its depth
// column in place -- lanes are consecutive pixels of a row, so every depth step of a wave is
// one coalesced access and invalid pixels (typically > 90 % with LiDAR supervision) cost
// nothing but their zero -- and writes the UNREDUCED per-pixel loss (loss_i of the reference
// before `.mean()` / the fg-bg weighted sum); the (B,H,W)-sized reductions stay on the host side.
// Backward: grad_volumes[b,:,h,w] = g[b,h,w] * dloss_i/dlogits, zeros for invalid pixels (the
// whole gradient volume is written once).
//
// target p_d (depth_head.py:111-183), dist_d = |depth_samples[d] - gt|:
//   DFM_DL_LINEAR   : 1 - min(dist_d / interval, 1)            ce, balanced_ce, focal, balanced_focal
//   DFM_DL_HARD     : the above, then >= 0.5 -> 1 else 0        hard_ce
//   DFM_DL_GAUSSIAN : exp(-0.5 dist_d^2 / sigma^2) / max(sum, 1)   gaussian_<sigma>
//   DFM_DL_LAPLACIAN: exp(-dist_d / sigma) / max(sum, 1)           laplacian_<sigma>
// loss_i = -sum_d p_d * f(lp_d), lp = log_softmax(logits);  f(lp) = lp, or with focal != 0
//   f(lp) = alpha * (1 - exp(lp))^gamma * lp                    (depth_head.py:131-139)
//
// Fused with the depth head (training): dfm_depth_loss_fused_fwd / _bwd take the
// LOW-RESOLUTION cost (B, 1, D/s, H/s, W/s) instead of depth_volumes = Upsample_xs(cost) and evaluate a
// valid pixel's column of logits on the fly with the depth-head kernel's own arithmetic (up_index /
// lerp_fma, rounded through the storage type: bit for bit what dfm_depth_head_fwd would have stored), so
// DepthHead.loss needs no (B, 1, sD, sH, sW) tensor; the backward adds d loss / d logits through the
// transposed upsample straight into grad_cost (fp32), a column's contributions to one coarse depth plane
// summed in registers first -- no gradient volume either.
#include <type_traits>

#include "dfm_common.h"

using namespace dfm;

namespace {

struct DlGeom {
    int32_t B, D, H, W, target, focal;
    float min_depth, max_depth, interval, sigma, alpha, gamma;
    int32_t cd, ch, cw;  // fused: size of the low-resolution cost (D = s * cd ...)
};

// a pixel's column of logits: read from the materialised volume ...
template <typename T>
struct ColVolume {
    const T *col;
    size_t HW;
    __device__ __forceinline__ float at(int d) { return elem<T>::load(col[(size_t)d * HW]); }
};
// ... or evaluated from the low-resolution cost (nested W -> H -> D fma upsample, align_corners=True:
// depth_head_kernel's expressions; frustum_to_voxel.hip's fused_disp uses the same)
template <typename T>
struct ColFused {
    const T *cost;   // sample b's (cd, ch, cw) volume
    int cd, D, plane;
    int o00, o01, o10, o11;  // the pixel's four (h, w) neighbours in a coarse plane
    float ww0, ww1, hw0, hw1;
    __device__ __forceinline__ float plane_at(int z) const
    {
        const T *p = cost + (size_t)z * plane;
        const float a = lerp_fma(ww0, elem<T>::load(p[o00]), ww1, elem<T>::load(p[o01]));
        const float b = lerp_fma(ww0, elem<T>::load(p[o10]), ww1, elem<T>::load(p[o11]));
        return lerp_fma(hw0, a, hw1, b);
    }
    // The two coarse planes a fine depth blends stay in registers while the walk is between them (round 6): the lower
    // index does not decrease with d and moves by one plane at a time, so a pass over the column evaluates each
    // plane ONCE (4 loads, 3 blends) instead of twice per fine depth (8 loads per logit: 4600 loads per pixel and
    // forward call at config K).  Same expressions on the same values: the logits are bit for bit what they were.
    int zc = -2;
    float v0 = 0.0f, v1 = 0.0f;  // plane_at(zc), plane_at(min(zc + 1, cd - 1))
    __device__ __forceinline__ float at(int d)
    {
        const UpIdx ud = up_index(d, cd, D);
        if (ud.i0 != zc) {
            v0 = ud.i0 == zc + 1 ? v1 : plane_at(ud.i0);
            zc = ud.i0;
            v1 = plane_at(min(zc + 1, cd - 1));
        }
        return elem<T>::load(elem<T>::store(lerp_fma(ud.w0, v0, ud.w1, ud.i1 == ud.i0 ? v0 : v1)));
    }
};

__device__ __forceinline__ float dl_target(const DlGeom &g, float ds, float gt)
{
    const float dist = fabsf(ds - gt);
    if (g.target == DFM_DL_GAUSSIAN) return expf(-0.5f * (dist * dist) / (g.sigma * g.sigma));
    if (g.target == DFM_DL_LAPLACIAN) return expf(-dist / g.sigma);
    float p = 1.0f - fminf(dist / g.interval, 1.0f);
    if (g.target == DFM_DL_HARD) p = p >= 0.5f ? 1.0f : 0.0f;
    return p;
}

__device__ __forceinline__ float dl_pow(float x, float gamma)
{
    if (gamma == 2.0f) return x * x;  // torch.pow(x, 2) is x*x
    if (gamma == 1.0f) return x;
    if (gamma == 0.0f) return 1.0f;
    return powf(x, gamma);
}

// f(lp) and f'(lp)
__device__ __forceinline__ void dl_f(const DlGeom &g, float lp, float &f, float &df)
{
    if (!g.focal) { f = lp; df = 1.0f; return; }
    const float pr = expf(lp), om = 1.0f - pr;
    const float w = g.alpha * dl_pow(om, g.gamma);
    f = w * lp;
    // d/dlp [alpha (1-e^lp)^gamma lp] = alpha (1-e^lp)^gamma - alpha gamma (1-e^lp)^(gamma-1) e^lp lp
    const float dw = (g.gamma == 0.0f) ? 0.0f : g.alpha * g.gamma * dl_pow(om, g.gamma - 1.0f) * pr;
    df = w - dw * lp;
}

// BWD == false: loss_out[b,h,w] (0 for invalid), valid_out[b,h,w]
// BWD == true : grad_vol[b,:,h,w] = gpix[b,h,w] * dloss/dlogits; FUSED: added into grad_cost through the
//               transposed upsample instead (vol = the low-resolution cost)
template <typename T, bool BWD, bool FUSED>
__global__ __launch_bounds__(256) void depth_loss_kernel(DlGeom g, const T *__restrict__ vol,
                                                         const float *__restrict__ depth_img,
                                                         const float *__restrict__ ds,
                                                         const float *__restrict__ gpix,
                                                         float *__restrict__ loss_out,
                                                         unsigned char *__restrict__ valid_out,
                                                         T *__restrict__ grad_vol,
                                                         float *__restrict__ grad_cost)
{
    const int HW = g.H * g.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (pix >= HW) return;
    const float gt = depth_img[(size_t)b * HW + pix];
    const bool valid = gt > g.min_depth && gt < g.max_depth;  // depth_head.py:89
    if (!BWD) {
        valid_out[(size_t)b * HW + pix] = valid ? 1 : 0;
        if (!valid) { loss_out[(size_t)b * HW + pix] = 0.0f; return; }
    } else {
        const float gp = valid ? gpix[(size_t)b * HW + pix] : 0.0f;
        if (gp == 0.0f) {
            if constexpr (!FUSED) {
                T *gc = grad_vol + (size_t)b * g.D * HW + pix;
                for (int d = 0; d < g.D; ++d) gc[(size_t)d * HW] = T(0);
            }
            return;
        }
    }
    typedef typename std::conditional<FUSED, ColFused<T>, ColVolume<T>>::type Col;
    Col col;
    UpIdx uw{}, uh{};
    if constexpr (FUSED) {
        const int y = pix / g.W, x = pix - y * g.W;
        uw = up_index(x, g.cw, g.W);
        uh = up_index(y, g.ch, g.H);
        col.cost = vol + (size_t)b * g.cd * g.ch * g.cw;
        col.cd = g.cd; col.D = g.D; col.plane = g.ch * g.cw;
        col.o00 = uh.i0 * g.cw + uw.i0; col.o01 = uh.i0 * g.cw + uw.i1;
        col.o10 = uh.i1 * g.cw + uw.i0; col.o11 = uh.i1 * g.cw + uw.i1;
        col.ww0 = uw.w0; col.ww1 = uw.w1; col.hw0 = uh.w0; col.hw1 = uh.w1;
    } else {
        col.col = vol + (size_t)b * g.D * HW + pix;
        col.HW = (size_t)HW;
    }
    // log-sum-exp of the column (online), and the target's normaliser
    float mx = -3.0e38f, se = 0.0f, psum = 0.0f;
    for (int d = 0; d < g.D; ++d) {
        const float x = col.at(d);
        if (x > mx) { se = se * expf(mx - x) + 1.0f; mx = x; }
        else se += expf(x - mx);
        if (g.target >= DFM_DL_GAUSSIAN) psum += dl_target(g, ds[d], gt);
    }
    const float lse = mx + logf(se);
    const float pnorm = g.target >= DFM_DL_GAUSSIAN ? 1.0f / fmaxf(psum, 1.0f) : 1.0f;
    float loss = 0.0f, A = 0.0f;  // A = sum_d p_d f'(lp_d)
    for (int d = 0; d < g.D; ++d) {
        const float p = dl_target(g, ds[d], gt) * pnorm;
        if (p == 0.0f) continue;
        const float lp = col.at(d) - lse;
        float f, df;
        dl_f(g, lp, f, df);
        loss -= p * f;
        A += p * df;
    }
    if (!BWD) {
        loss_out[(size_t)b * HW + pix] = loss;
        return;
    }
    // dloss/dx_k = -(p_k f'_k - softmax_k * A)
    const float gp = gpix[(size_t)b * HW + pix];
    if constexpr (!FUSED) {
        T *gc = grad_vol + (size_t)b * g.D * HW + pix;
        for (int d = 0; d < g.D; ++d) {
            const float lp = col.at(d) - lse;
            const float p = dl_target(g, ds[d], gt) * pnorm;
            float f, df;
            dl_f(g, lp, f, df);
            gc[(size_t)d * HW] = elem<T>::store(gp * (expf(lp) * A - p * df));
        }
    } else {
        // transposed upsample: logit_d = w0(d) * plane[i0(d)] + w1(d) * plane[i1(d)], plane[z] = the
        // bilinear blend of four coarse cells.  i0(d) does not decrease with d: the column's
        // contributions to coarse planes `zc` and `zc + 1` are summed in two registers and leave as four
        // atomics per plane when the walk moves on.
        float *gcb = grad_cost + (size_t)b * g.cd * col.plane;
        auto flush_plane = [&](int z, float v) {
            if (v == 0.0f) return;
            float *p = gcb + (size_t)z * col.plane;
            atomicAdd(p + col.o00, v * col.hw0 * col.ww0);
            atomicAdd(p + col.o01, v * col.hw0 * col.ww1);
            atomicAdd(p + col.o10, v * col.hw1 * col.ww0);
            atomicAdd(p + col.o11, v * col.hw1 * col.ww1);
        };
        int zc = 0;
        float a0 = 0.0f, a1 = 0.0f;  // planes zc, zc + 1
        for (int d = 0; d < g.D; ++d) {
            const float lp = col.at(d) - lse;
            const float p = dl_target(g, ds[d], gt) * pnorm;
            float f, df;
            dl_f(g, lp, f, df);
            const float gl = gp * (expf(lp) * A - p * df);
            const UpIdx ud = up_index(d, g.cd, g.D);
            while (zc < ud.i0) {  // the walk has left plane zc
                flush_plane(zc, a0);
                a0 = a1;
                a1 = 0.0f;
                ++zc;
            }
            a0 += gl * ud.w0;
            if (ud.i1 == ud.i0) a0 += gl * ud.w1; else a1 += gl * ud.w1;
        }
        flush_plane(zc, a0);
        if (zc + 1 < g.cd) flush_plane(zc + 1, a1);
    }
}

int check(const dfm_depth_loss_desc *d)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->batch <= 0 || d->num_depths <= 0 || d->h <= 0 || d->w <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_depth_loss_desc");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (d->target < DFM_DL_LINEAR || d->target > DFM_DL_LAPLACIAN)
        return set_error(DFM_ERR_INVALID_ARG, "unknown target kind");
    if (d->target >= DFM_DL_GAUSSIAN && !(d->sigma > 0.0f))
        return set_error(DFM_ERR_INVALID_ARG, "sigma must be positive");
    if (d->target <= DFM_DL_HARD && !(d->interval != 0.0f))
        return set_error(DFM_ERR_INVALID_ARG, "depth interval must be non-zero");
    if (d->batch > 65535) return set_error(DFM_ERR_UNSUPPORTED, "batch > 65535");
    return DFM_OK;
}

DlGeom geom(const dfm_depth_loss_desc *d)
{
    DlGeom g;
    g.B = d->batch; g.D = d->num_depths; g.H = d->h; g.W = d->w;
    g.target = d->target; g.focal = d->focal;
    g.min_depth = d->min_depth; g.max_depth = d->max_depth; g.interval = d->interval;
    g.sigma = d->sigma; g.alpha = d->alpha; g.gamma = d->gamma;
    g.cd = g.ch = g.cw = 0;
    return g;
}

}  // namespace

extern "C" DFM_API int dfm_depth_loss_fwd(const dfm_depth_loss_desc *d, const void *depth_volumes,
                                          const float *depth_img, const float *depth_samples,
                                          float *pixel_loss, unsigned char *valid, void *stream)
{
    int rc = check(d);
    if (rc != DFM_OK) return rc;
    if (!depth_volumes || !depth_img || !depth_samples || !pixel_loss || !valid)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const DlGeom g = geom(d);
    dim3 grid((d->h * d->w + 255) / 256, d->batch);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL((depth_loss_kernel<float, false, false>), grid, dim3(256), 0, st, g,
                           (const float *)depth_volumes, depth_img, depth_samples,
                           (const float *)nullptr, pixel_loss, valid, (float *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL((depth_loss_kernel<bf16_t, false, false>), grid, dim3(256), 0, st, g,
                           (const bf16_t *)depth_volumes, depth_img, depth_samples,
                           (const float *)nullptr, pixel_loss, valid, (bf16_t *)nullptr, (float *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_depth_loss_bwd(const dfm_depth_loss_desc *d, const void *depth_volumes,
                                          const float *depth_img, const float *depth_samples,
                                          const float *grad_pixel_loss, void *grad_volumes,
                                          void *stream)
{
    int rc = check(d);
    if (rc != DFM_OK) return rc;
    if (!depth_volumes || !depth_img || !depth_samples || !grad_pixel_loss || !grad_volumes)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const DlGeom g = geom(d);
    dim3 grid((d->h * d->w + 255) / 256, d->batch);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL((depth_loss_kernel<float, true, false>), grid, dim3(256), 0, st, g,
                           (const float *)depth_volumes, depth_img, depth_samples, grad_pixel_loss,
                           (float *)nullptr, (unsigned char *)nullptr, (float *)grad_volumes, (float *)nullptr);
    else
        hipLaunchKernelGGL((depth_loss_kernel<bf16_t, true, false>), grid, dim3(256), 0, st, g,
                           (const bf16_t *)depth_volumes, depth_img, depth_samples, grad_pixel_loss,
                           (float *)nullptr, (unsigned char *)nullptr, (bf16_t *)grad_volumes, (float *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

namespace {
int fused_geom(const dfm_depth_loss_desc *d, int scale, DlGeom &g)
{
    int rc = check(d);
    if (rc != DFM_OK) return rc;
    if (scale < 1 || d->num_depths % scale || d->h % scale || d->w % scale)
        return set_error(DFM_ERR_INVALID_ARG, "fused depth loss: num_depths, h, w must be multiples of head_scale");
    g = geom(d);
    g.cd = d->num_depths / scale; g.ch = d->h / scale; g.cw = d->w / scale;
    return DFM_OK;
}
}  // namespace

extern "C" DFM_API int dfm_depth_loss_fused_fwd(const dfm_depth_loss_desc *d, const void *cost, int32_t head_scale,
                                                const float *depth_img, const float *depth_samples,
                                                float *pixel_loss, unsigned char *valid, void *stream)
{
    DlGeom g;
    int rc = fused_geom(d, head_scale, g);
    if (rc != DFM_OK) return rc;
    if (!cost || !depth_img || !depth_samples || !pixel_loss || !valid)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    dim3 grid((d->h * d->w + 255) / 256, d->batch);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL((depth_loss_kernel<float, false, true>), grid, dim3(256), 0, st, g, (const float *)cost,
                           depth_img, depth_samples, (const float *)nullptr, pixel_loss, valid, (float *)nullptr,
                           (float *)nullptr);
    else
        hipLaunchKernelGGL((depth_loss_kernel<bf16_t, false, true>), grid, dim3(256), 0, st, g, (const bf16_t *)cost,
                           depth_img, depth_samples, (const float *)nullptr, pixel_loss, valid, (bf16_t *)nullptr,
                           (float *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_depth_loss_fused_bwd(const dfm_depth_loss_desc *d, const void *cost, int32_t head_scale,
                                                const float *depth_img, const float *depth_samples,
                                                const float *grad_pixel_loss, float *grad_cost, void *stream)
{
    DlGeom g;
    int rc = fused_geom(d, head_scale, g);
    if (rc != DFM_OK) return rc;
    if (!cost || !depth_img || !depth_samples || !grad_pixel_loss || !grad_cost)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    dim3 grid((d->h * d->w + 255) / 256, d->batch);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL((depth_loss_kernel<float, true, true>), grid, dim3(256), 0, st, g, (const float *)cost,
                           depth_img, depth_samples, grad_pixel_loss, (float *)nullptr, (unsigned char *)nullptr,
                           (float *)nullptr, grad_cost);
    else
        hipLaunchKernelGGL((depth_loss_kernel<bf16_t, true, true>), grid, dim3(256), 0, st, g, (const bf16_t *)cost,
                           depth_img, depth_samples, grad_pixel_loss, (float *)nullptr, (unsigned char *)nullptr,
                           (bf16_t *)nullptr, grad_cost);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

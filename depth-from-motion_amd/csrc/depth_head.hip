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
// volume); softmax uses exp_nonpos (dfm_common.h; equal to torch's Sleef exp to rounding error)
// and one reciprocal per column.
// Bound: HBM write (2 volumes), ~8 cached loads + 1 exp per element.
#include "dfm_common.h"

#include <algorithm>

using namespace dfm;

namespace {

// V output pixels (consecutive along w) per lane: one V*sizeof(T)-byte store per depth.
// The upsample is separable and ATen nests it W -> H -> D, so a lane first reduces
// its 4 (W,H) taps of input plane i to one value col(i) and every output depth is a
// single lerp of col(i0), col(i1); i0 advances once per `s` output depths, so each
// pass over the column evaluates col() D times (not 4D times 2).
// STORE = false: the volumes are not written; instead the column maximum and the sum of
// exp(logit - max) go to col_max / col_sum (fp32, (B, sH, sW)): everything the fused
// FrustumToVoxel needs to evaluate softmax(upsample(cost)) at any lattice point on the fly
// (dfm_frustum_to_voxel_fused_fwd) -- the three 472 MB tensors are never materialised.
template <typename T, int V, bool STORE>
__global__ __launch_bounds__(128) void depth_head_kernel(const T *__restrict__ in, int D, int H,
                                                         int W, int s,
                                                         const float *__restrict__ depth_samples,
                                                         T *__restrict__ vol, T *__restrict__ soft,
                                                         T *__restrict__ pred,
                                                         float *__restrict__ col_max,
                                                         float *__restrict__ col_sum)
{
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int Do = D * s, Ho = H * s, Wo = W * s;
    const int pix = (blockIdx.x * 128 + threadIdx.x) * V;
    const int b = blockIdx.y;
    // the depth interpolation (input planes and weights of output depth d) is the same for every
    // lane: one LDS table per workgroup, read back as a broadcast, instead of ~20 VALU operations
    // (a division among them) per lane, depth and pass
    extern __shared__ float4 dtab[];  // {i0, i1 (as int bits), w0, w1}
    for (int d = threadIdx.x; d < Do; d += 128) {
        const UpIdx u = up_index(d, D, Do);
        dtab[d] = make_float4(__int_as_float(u.i0), __int_as_float(u.i1), u.w0, u.w1);
    }
    __syncthreads();
    if (pix >= Ho * Wo) return;
    const int h = pix / Wo, w = pix - h * Wo;  // Wo % V == 0: the V pixels share the row
    const UpIdx uh = up_index(h, H, Ho);
    const T *x = in + (size_t)b * D * H * W;
    int o0[V], o1[V];
    float ww0[V], ww1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const UpIdx uw = up_index(w + j, W, Wo);
        o0[j] = uw.i0;
        o1[j] = uw.i1;
        ww0[j] = uw.w0;
        ww1[j] = uw.w1;
    }
    const int r0 = uh.i0 * W, r1 = uh.i1 * W;
    const size_t plane_o = (size_t)Ho * Wo;
    T *vcol = vol + (size_t)b * Do * plane_o + pix;
    T *scol = soft + (size_t)b * Do * plane_o + pix;

    auto column = [&](int i, float (&c)[V]) {
        const T *p = x + (size_t)i * H * W;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float a = lerp_fma(ww0[j], elem<T>::load(p[r0 + o0[j]]), ww1[j],
                                     elem<T>::load(p[r0 + o1[j]]));
            const float bb = lerp_fma(ww0[j], elem<T>::load(p[r1 + o0[j]]), ww1[j],
                                      elem<T>::load(p[r1 + o1[j]]));
            c[j] = lerp_fma(uh.w0, a, uh.w1, bb);
        }
    };
    // walks d = 0..Do-1 keeping col(i0), col(i1) of the current input interval
    float c0[V], c1[V];
    int have = -1;
    auto logits = [&](int d, float (&v)[V]) {
        const float4 te = dtab[d];
        UpIdx ud;
        ud.i0 = __float_as_int(te.x); ud.i1 = __float_as_int(te.y); ud.w0 = te.z; ud.w1 = te.w;
        if (ud.i0 != have) {
            if (ud.i0 == have + 1 && have >= 0) {
#pragma unroll
                for (int j = 0; j < V; ++j) c0[j] = c1[j];
            } else {
                column(ud.i0, c0);
            }
            column(ud.i1, c1);
            have = ud.i0;
        }
#pragma unroll
        for (int j = 0; j < V; ++j)
            // the reference's softmax reads depth_volumes in its storage type
            v[j] = elem<T>::load(elem<T>::store(lerp_fma(ud.w0, c0[j], ud.w1, c1[j])));
    };

    float mx[V], sum[V], acc[V], v[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { mx[j] = -INFINITY; sum[j] = 0.0f; acc[j] = 0.0f; }
    // pass 1: the logits (stored as depth_volumes) with an ONLINE maximum / sum of exponentials over
    // chunks of CH depths held in registers: the running sum is rescaled once per chunk, when the
    // maximum moved (exp(0) == 1 exactly otherwise).  The separate sum pass -- a third evaluation of
    // every logit -- is gone.
    constexpr int CH = 8;
    for (int d0 = 0; d0 < Do; d0 += CH) {
        float vv[CH][V];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (d0 + c < Do) {
                logits(d0 + c, vv[c]);
                if constexpr (STORE) {
                    vec_t st;
#pragma unroll
                    for (int j = 0; j < V; ++j) st[j] = elem<T>::store(vv[c][j]);
                    __builtin_nontemporal_store(st, (vec_t *)(vcol + (size_t)(d0 + c) * plane_o));
                }
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) vv[c][j] = -INFINITY;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float cm = vv[0][j];
#pragma unroll
            for (int c = 1; c < CH; ++c) cm = fmaxf(cm, vv[c][j]);
            const float mnew = fmaxf(mx[j], cm);
            float sj = sum[j] * exp_nonpos(mx[j] - mnew);
#pragma unroll
            for (int c = 0; c < CH; ++c) sj = sj + exp_nonpos(vv[c][j] - mnew);
            sum[j] = sj;
            mx[j] = mnew;
        }
    }
    have = -1;
    if (!STORE && !pred) {  // statistics only: the expectation pass is not needed
#pragma unroll
        for (int j = 0; j < V; ++j) {
            col_max[(size_t)b * plane_o + pix + j] = mx[j];
            col_sum[(size_t)b * plane_o + pix + j] = sum[j];
        }
        return;
    }
    // one IEEE division per column, one multiplication per element (x / sum and x * (1 / sum) differ by
    // at most one rounding; the fused FrustumToVoxel evaluates the same expression)
    float inv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) inv[j] = 1.0f / sum[j];
    for (int d = 0; d < Do; ++d) {
        logits(d, v);
        const float ds = depth_samples[d];
        vec_t st;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            st[j] = elem<T>::store(exp_nonpos(v[j] - mx[j]) * inv[j]);
            acc[j] = acc[j] + elem<T>::load(st[j]) * ds;
        }
        if constexpr (STORE) __builtin_nontemporal_store(st, (vec_t *)(scol + (size_t)d * plane_o));
    }
    if constexpr (!STORE) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            col_max[(size_t)b * plane_o + pix + j] = mx[j];
            col_sum[(size_t)b * plane_o + pix + j] = sum[j];
        }
    }
    if (pred) {
        vec_t pr;
#pragma unroll
        for (int j = 0; j < V; ++j) pr[j] = elem<T>::store(acc[j]);
        *(vec_t *)(pred + (size_t)b * plane_o + pix) = pr;
    }
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
    if (batch > 65535 || (long long)h * scale * w * scale >= (1ll << 31) || (long long)d * scale > 4000)
        return set_error(DFM_ERR_UNSUPPORTED, "shape too large");
    const int npix = h * scale * w * scale;
    // 4 pixels per lane (16-byte fp32 / 8-byte bf16 stores) when rows and the three output base
    // pointers allow it, else one (8 bf16 pixels per lane need 231 VGPRs: 2 waves per SIMD, slower)
    const size_t esz = dtype == DFM_F32 ? 4 : 2;
    const uintptr_t bases = (uintptr_t)depth_volumes | (uintptr_t)softmax | (uintptr_t)depth_preds;
    int v = 1;
    if ((w * scale) % 4 == 0 && bases % (4 * esz) == 0) v = 4;
    dim3 grid((npix / v + 127) / 128, batch);
    hipStream_t st = (hipStream_t)stream;
#define DFM_DH_LAUNCH(T, V)                                                                       \
    hipLaunchKernelGGL((depth_head_kernel<T, V, true>), grid, dim3(128), (size_t)d * scale * 16, st, (const T *)cost, d, h, \
                       w, scale, depth_samples, (T *)depth_volumes, (T *)softmax, (T *)depth_preds,   \
                       (float *)nullptr, (float *)nullptr)
    if (dtype == DFM_F32) {
        if (v == 4) DFM_DH_LAUNCH(float, 4); else DFM_DH_LAUNCH(float, 1);
    } else {
        if (v == 4) DFM_DH_LAUNCH(bf16_t, 4); else DFM_DH_LAUNCH(bf16_t, 1);
    }
#undef DFM_DH_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_depth_head_stats_fwd(int32_t batch, int32_t d, int32_t h, int32_t w, int32_t scale,
                                     int32_t dtype, const void *cost, const float *depth_samples,
                                     float *col_max, float *col_sum, void *depth_preds, void *stream)
{
    if (batch <= 0 || d <= 0 || h <= 0 || w <= 0 || scale <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_depth_head_stats_fwd");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!cost || !depth_samples || !col_max || !col_sum)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (batch > 65535 || (long long)h * scale * w * scale >= (1ll << 31) || (long long)d * scale > 4000)
        return set_error(DFM_ERR_UNSUPPORTED, "shape too large");
    const int npix = h * scale * w * scale;
    // one pixel per lane: there are no wide stores to feed, and 4x the waves hide the latency of the
    // cached cost loads and of expf (a B = 1 launch with 4 pixels per lane has 1.5 waves per SIMD)
    const bool vec4 = false;
    dim3 grid((npix + 127) / 128, batch);
    hipStream_t st = (hipStream_t)stream;
#define DFM_DH_LAUNCH(T, V)                                                                            \
    hipLaunchKernelGGL((depth_head_kernel<T, V, false>), grid, dim3(128), (size_t)d * scale * 16, st, (const T *)cost, d, h, \
                       w, scale, depth_samples, (T *)nullptr, (T *)nullptr, (T *)depth_preds, col_max,  \
                       col_sum)
    if (dtype == DFM_F32) {
        if (vec4) DFM_DH_LAUNCH(float, 4); else DFM_DH_LAUNCH(float, 1);
    } else {
        if (vec4) DFM_DH_LAUNCH(bf16_t, 4); else DFM_DH_LAUNCH(bf16_t, 1);
    }
#undef DFM_DH_LAUNCH
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
        for (int d = 0; d < Do; ++d) sum += exp_nonpos(logit(d) - mx);
        for (int d = 0; d < Do; ++d) {
            const float p = exp_nonpos(logit(d) - mx) / sum;
            const float sd = (gsoft ? elem<T>::load(gsoft[col + (size_t)d * plane_o]) : 0.0f) +
                             gp * depth_samples[d];
            dotps += p * sd;
        }
    }
    for (int d = 0; d < Do; ++d) {
        float gl = gvol ? elem<T>::load(gvol[col + (size_t)d * plane_o]) : 0.0f;
        if (soft_path) {
            const float p = exp_nonpos(logit(d) - mx) / sum;
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

// Tiled backward: a workgroup owns 4 output rows x 64 output columns (lane = one output
// pixel column, all 4D depths) and the few input pixels under them.  The logit gradients of
// a column are first reduced along depth in registers (the ~s outputs that share an input
// depth interval), then added into an LDS tile [D][rows][cols] of the input -- 4 LDS atomics
// per input depth and lane instead of 8 global atomics per OUTPUT element -- and the tile goes
// to memory with one atomic per input element.  The logits are recomputed with the forward's
// rolling columns.  (The scatter-per-output kernel above stays as the fallback for tiles
// that do not fit the LDS: scale 1, very deep volumes.)
template <typename T>
__global__ __launch_bounds__(256) void depth_head_bwd_tile_kernel(
    const T *__restrict__ in, int D, int H, int W, int s, const float *__restrict__ depth_samples,
    const T *__restrict__ gvol, const T *__restrict__ gsoft, const T *__restrict__ gpred,
    float *__restrict__ gin, int nr, int ncol)
{
    extern __shared__ float dh_tile[];  // [D][nr][ncol]
    const int Do = D * s, Ho = H * s, Wo = W * s;
    const int b = blockIdx.z;
    const int h0 = blockIdx.y * 4, w0 = blockIdx.x * 64;
    const int h = h0 + (threadIdx.x >> 6), w = w0 + (threadIdx.x & 63);
    const bool live = h < Ho && w < Wo;
    // input window under the tile
    const int ih_lo = up_index(h0, H, Ho).i0, iw_lo = up_index(w0, W, Wo).i0;
    const int tile_n = D * nr * ncol;
    for (int i = threadIdx.x; i < tile_n; i += 256) dh_tile[i] = 0.0f;
    __syncthreads();
    const T *x = in + (size_t)b * D * H * W;
    if (live) {
        const UpIdx uh = up_index(h, H, Ho), uw = up_index(w, W, Wo);
        const int r0 = uh.i0 * W, r1 = uh.i1 * W;
        auto column = [&](int i) {
            const T *p = x + (size_t)i * H * W;
            const float a = lerp_fma(uw.w0, elem<T>::load(p[r0 + uw.i0]), uw.w1, elem<T>::load(p[r0 + uw.i1]));
            const float bb = lerp_fma(uw.w0, elem<T>::load(p[r1 + uw.i0]), uw.w1, elem<T>::load(p[r1 + uw.i1]));
            return lerp_fma(uh.w0, a, uh.w1, bb);
        };
        float c0 = 0.0f, c1 = 0.0f;
        int have = -1;
        auto logit = [&](int d) {
            const UpIdx ud = up_index(d, D, Do);
            if (ud.i0 != have) {
                c0 = (ud.i0 == have + 1 && have >= 0) ? c1 : column(ud.i0);
                c1 = column(ud.i1);
                have = ud.i0;
            }
            return lerp_fma(ud.w0, c0, ud.w1, c1);
        };
        const size_t plane_o = (size_t)Ho * Wo;
        const size_t pix = (size_t)h * Wo + w;
        const size_t col = (size_t)b * Do * plane_o + pix;
        const bool soft_path = gsoft || gpred;
        float mx = -INFINITY, sum = 0.0f, dotps = 0.0f;
        const float gp = gpred ? elem<T>::load(gpred[(size_t)b * plane_o + pix]) : 0.0f;
        if (soft_path) {
            for (int d = 0; d < Do; ++d) mx = fmaxf(mx, logit(d));
            have = -1;
            for (int d = 0; d < Do; ++d) sum += exp_nonpos(logit(d) - mx);
            have = -1;
            for (int d = 0; d < Do; ++d) {
                const float p = exp_nonpos(logit(d) - mx) / sum;
                const float sd = (gsoft ? elem<T>::load(gsoft[col + (size_t)d * plane_o]) : 0.0f) +
                                 gp * depth_samples[d];
                dotps += p * sd;
            }
            have = -1;
        }
        // the four (h, w) taps of this column inside the tile
        const int t00 = (uh.i0 - ih_lo) * ncol + (uw.i0 - iw_lo), t01 = (uh.i0 - ih_lo) * ncol + (uw.i1 - iw_lo);
        const int t10 = (uh.i1 - ih_lo) * ncol + (uw.i0 - iw_lo), t11 = (uh.i1 - ih_lo) * ncol + (uw.i1 - iw_lo);
        const float w00 = uh.w0 * uw.w0, w01 = uh.w0 * uw.w1, w10 = uh.w1 * uw.w0, w11 = uh.w1 * uw.w1;
        const int dstride = nr * ncol;
        auto put = [&](int i, float a) {  // input depth i receives a (already weighted along depth)
            if (a == 0.0f) return;
            float *t = dh_tile + i * dstride;
            atomicAdd(t + t00, a * w00); atomicAdd(t + t01, a * w01);
            atomicAdd(t + t10, a * w10); atomicAdd(t + t11, a * w11);
        };
        float a_lo = 0.0f, a_hi = 0.0f;  // pending sums for input depths cur and cur + 1
        int cur = 0;
        for (int d = 0; d < Do; ++d) {
            float gl = gvol ? elem<T>::load(gvol[col + (size_t)d * plane_o]) : 0.0f;
            if (soft_path) {
                const float p = exp_nonpos(logit(d) - mx) / sum;
                const float sd = (gsoft ? elem<T>::load(gsoft[col + (size_t)d * plane_o]) : 0.0f) +
                                 gp * depth_samples[d];
                gl += p * (sd - dotps);
            }
            const UpIdx ud = up_index(d, D, Do);
            while (cur < ud.i0) {  // the interval moved on: the lower depth is complete
                put(cur, a_lo);
                a_lo = a_hi;
                a_hi = 0.0f;
                ++cur;
            }
            a_lo += gl * ud.w0;
            if (ud.i1 != ud.i0) a_hi += gl * ud.w1;
            else a_lo += gl * ud.w1;  // last plane: i1 == i0
        }
        put(cur, a_lo);
        if (cur + 1 < D) put(cur + 1, a_hi);
    }
    __syncthreads();
    float *gx = gin + (size_t)b * D * H * W;
    for (int i = threadIdx.x; i < tile_n; i += 256) {
        const float v = dh_tile[i];
        if (v == 0.0f) continue;
        const int di = i / (nr * ncol), rem = i - di * (nr * ncol);
        const int r = rem / ncol, c = rem - r * ncol;
        const int ih = ih_lo + r, iw = iw_lo + c;
        if (ih < H && iw < W) atomicAdd(gx + ((size_t)di * H + ih) * W + iw, v);
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
    hipStream_t st = (hipStream_t)stream;
    {
        // tiled kernel when the input window of a 4 x 64 output tile fits 64 KiB of LDS
        const int Ho = h * scale, Wo = w * scale;
        const double sh = Ho > 1 ? (double)(h - 1) / (Ho - 1) : 0.0, sw = Wo > 1 ? (double)(w - 1) / (Wo - 1) : 0.0;
        const int nr = std::min(h, (int)(3 * sh) + 3), ncol = std::min(w, (int)(63 * sw) + 3);
        const size_t lds = (size_t)d * nr * ncol * sizeof(float);
        if (lds <= 64 * 1024 && batch <= 65535 && (Ho + 3) / 4 <= 65535) {
            dim3 tg((Wo + 63) / 64, (Ho + 3) / 4, batch);
            if (dtype == DFM_F32)
                hipLaunchKernelGGL(depth_head_bwd_tile_kernel<float>, tg, dim3(256), lds, st,
                                   (const float *)cost, d, h, w, scale, depth_samples,
                                   (const float *)grad_volumes, (const float *)grad_softmax,
                                   (const float *)grad_preds, grad_cost, nr, ncol);
            else
                hipLaunchKernelGGL(depth_head_bwd_tile_kernel<bf16_t>, tg, dim3(256), lds, st,
                                   (const bf16_t *)cost, d, h, w, scale, depth_samples,
                                   (const bf16_t *)grad_volumes, (const bf16_t *)grad_softmax,
                                   (const bf16_t *)grad_preds, grad_cost, nr, ncol);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
            return DFM_OK;
        }
    }
    const int npix = h * scale * w * scale;
    dim3 grid((npix + 255) / 256, batch);
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

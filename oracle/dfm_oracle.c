/*
 * dfm_oracle.c -- CPU restatement of the Depth-from-Motion plane-sweep path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (depth-from-motion_amd/) never links or calls anything here.
 *
 * Every function follows the reference's fp32 arithmetic operation by
 * operation (one IEEE-754 binary32 rounding per torch op) so that the
 * sampling coordinates agree BITWISE with the reference's PyTorch-CPU run;
 * tests/golden/make_golden.py pins that against the reference executed in the
 * build container (tests/test_oracle_golden.py replays the pins).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile);
 * -ffp-contract=off matters: a*b+c must stay two roundings except where this
 * file calls fmaf() explicitly.
 *
 * Third-party arithmetic that is not under /root/reference: torch (any
 * >=1.3; the container has 2.10) F.grid_sample and torch.mm / `@`:
 *  - (N,4)@(4,4) fp32 on CPU (MKL sgemm): each output is the k-ordered fmaf
 *    chain fma(a3,b3,fma(a2,b2,fma(a1,b1,a0*b0))) -- pinned empirically by
 *    the golden grids (make_golden.py stores the reference's grids).
 *  - grid_sample(bilinear|nearest, zeros, align_corners=True): ATen
 *    native/cpu/GridSamplerKernel.cpp (2-D, vectorised) and
 *    native/GridSampler.cpp (3-D, scalar): unnormalise ((g+1)/2)*(size-1),
 *    floor, corner weights (1-w)(1-n)..., per-corner bounds test, zero fill.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* helpers                                                                   */
/* ------------------------------------------------------------------------ */

/* row `r` of (v[0..3] @ M^T), M row-major 4x4: sum_k v[k]*M[r][k], evaluated
 * as the k-ordered fma chain MKL's sgemm micro-kernel produces. */
static inline float dot4_chain(const float v[4], const float *Mrow)
{
    float acc = v[0] * Mrow[0];
    acc = fmaf(v[1], Mrow[1], acc);
    acc = fmaf(v[2], Mrow[2], acc);
    acc = fmaf(v[3], Mrow[3], acc);
    return acc;
}

/* points_cam2img(points_3d, proj_mat 4x4)[:, :2]
 * reference: mmdet3d/core/bbox/structures/utils.py:206-209 */
static inline void cam2img_4x4(const float X[3], const float *P, float *u, float *v,
                               float *z)
{
    float p4[4] = {X[0], X[1], X[2], 1.0f};
    float a = dot4_chain(p4, P + 0);
    float b = dot4_chain(p4, P + 4);
    float c = dot4_chain(p4, P + 8);
    *u = a / c;
    *v = b / c;
    if (z) *z = c;
}

/* ------------------------------------------------------------------------ */
/* plane-sweep lattice -> normalised sampling grids                          */
/* reference: mmdet3d/models/backbones/dfm_backbone.py:247-294               */
/* ------------------------------------------------------------------------ */

typedef struct {
    int32_t D, h_out, w_out, h_in, w_in;
    float fsf;          /* feat_sample_factor, cast to fp32 like torch does */
    float csf;          /* cost_sample_factor                              */
    float scale;        /* img_scale_factor                                */
    float crop_x, crop_y; /* img_crop_offset as fp32                        */
    int32_t flip;
    float org_w;        /* img_shape[1]                                     */
    float P[16];        /* ori_cam2img padded 4x4 (row major)               */
    float Pinv[16];     /* torch.inverse(pad(cam2img[:3])) (row major)      */
    float T[16];        /* cur2prev 4x4 (row major)                         */
} dfm_oracle_sweep_params;

/* one lattice point -> normalised (x,y) for cur and prev */
static inline void sweep_point(const dfm_oracle_sweep_params *p, float depth, int hi, int wi,
                               float cur[2], float prev[2])
{
    /* :247-250  ws = linspace(0,w_out-1,w_out)*fsf*csf  (linspace step == 1) */
    float x = ((float)wi * p->fsf) * p->csf;
    float y = ((float)hi * p->fsf) * p->csf;
    /* :259-263 crop back -> scale back -> flip back */
    x = x + p->crop_x;
    y = y + p->crop_y;
    x = x / p->scale;
    y = y / p->scale;
    if (p->flip) x = p->org_w - x;
    /* points_img2cam, utils.py:235-246 */
    float homo[4] = {x * depth, y * depth, depth, 1.0f};
    float X[3];
    /* torch.mm(homo, inv.T)[:, :3] -> X[j] = sum_k homo[k]*inv[j][k] */
    X[0] = dot4_chain(homo, p->Pinv + 0);
    X[1] = dot4_chain(homo, p->Pinv + 4);
    X[2] = dot4_chain(homo, p->Pinv + 8);
    /* :269 cur projection with the full 4x4 */
    float cu, cv;
    cam2img_4x4(X, p->P, &cu, &cv, NULL);
    /* :268,270 homo_grid3d @ cur2prev.T, [:, :3] */
    float X4[4] = {X[0], X[1], X[2], 1.0f};
    float Y[3];
    Y[0] = dot4_chain(X4, p->T + 0);
    Y[1] = dot4_chain(X4, p->T + 4);
    Y[2] = dot4_chain(X4, p->T + 8);
    float pu, pv;
    cam2img_4x4(Y, p->P, &pu, &pv, NULL);
    /* :278-285 flip -> scale -> crop */
    if (p->flip) {
        cu = p->org_w - cu;
        pu = p->org_w - pu;
    }
    cu = cu * p->scale; cv = cv * p->scale;
    pu = pu * p->scale; pv = pv * p->scale;
    cu = cu - p->crop_x; cv = cv - p->crop_y;
    pu = pu - p->crop_x; pv = pv - p->crop_y;
    /* :287-288 */
    cu = cu / p->fsf; cv = cv / p->fsf;
    pu = pu / p->fsf; pv = pv / p->fsf;
    /* :291-294 */
    float wm1 = (float)(p->w_in - 1), hm1 = (float)(p->h_in - 1);
    cur[0] = cu / wm1 * 2.0f - 1.0f;
    cur[1] = cv / hm1 * 2.0f - 1.0f;
    prev[0] = pu / wm1 * 2.0f - 1.0f;
    prev[1] = pv / hm1 * 2.0f - 1.0f;
}

/* grids: (D*h_out*w_out, 2) each */
ORACLE_API void dfm_oracle_plane_sweep_grid(const dfm_oracle_sweep_params *p, const float *depths,
                                            float *cur_grid, float *prev_grid)
{
    size_t n = 0;
    for (int d = 0; d < p->D; ++d)
        for (int h = 0; h < p->h_out; ++h)
            for (int w = 0; w < p->w_out; ++w, ++n)
                sweep_point(p, depths[d], h, w, cur_grid + 2 * n, prev_grid + 2 * n);
}

/* ------------------------------------------------------------------------ */
/* F.grid_sample 2-D, zeros padding, align_corners=True                      */
/* ------------------------------------------------------------------------ */

static inline float unnormalize_ac(float g, int size)
{
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}

/* float -> int32 with the x86 cvttps2dq convention ATen's vectorised kernel
 * gets: NaN / out-of-range -> INT32_MIN (always out of bounds). */
static inline int32_t f2i_sat(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

/* bilinear sample of one (H,W) plane; non-finite coordinates give 0 (the
 * torch GPU convention; torch's CPU kernel yields NaN there -- documented
 * deviation, DESIGN.md "non-finite coordinates"). */
static inline float bilinear_plane(const float *plane, int H, int W, float gx, float gy)
{
    float x = unnormalize_ac(gx, W);
    float y = unnormalize_ac(gy, H);
    if (!isfinite(x) || !isfinite(y)) return 0.0f;
    float xw = floorf(x), yn = floorf(y);
    float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    float nw = s * e, ne = s * w, sw = n * e, se = n * w;
    int32_t ix = f2i_sat(xw), iy = f2i_sat(yn);
    int32_t ix1 = (ix == INT32_MIN) ? INT32_MIN : ix + 1;
    int32_t iy1 = (iy == INT32_MIN) ? INT32_MIN : iy + 1;
    int wok = ix > -1 && ix < W, eok = ix1 > -1 && ix1 < W;
    int nok = iy > -1 && iy < H, sok = iy1 > -1 && iy1 < H;
    float vnw = (wok && nok) ? plane[(size_t)iy * W + ix] : 0.0f;
    float vne = (eok && nok) ? plane[(size_t)iy * W + ix1] : 0.0f;
    float vsw = (wok && sok) ? plane[(size_t)iy1 * W + ix] : 0.0f;
    float vse = (eok && sok) ? plane[(size_t)iy1 * W + ix1] : 0.0f;
    return fmaf(vse, se, fmaf(vsw, sw, fmaf(vne, ne, vnw * nw)));
}

/* nearest: nearbyint (round-half-even) of the unnormalised coordinate */
static inline float nearest_plane(const float *plane, int H, int W, float gx, float gy)
{
    float x = unnormalize_ac(gx, W);
    float y = unnormalize_ac(gy, H);
    if (!isfinite(x) || !isfinite(y)) return 0.0f;
    int32_t ix = f2i_sat(nearbyintf(x)), iy = f2i_sat(nearbyintf(y));
    if (ix > -1 && ix < W && iy > -1 && iy < H) return plane[(size_t)iy * W + ix];
    return 0.0f;
}

/* input (C,H,W); grid (N,2) normalised; out (C,N).  mode 0 bilinear, 1 nearest */
ORACLE_API void dfm_oracle_grid_sample2d(const float *input, int C, int H, int W, const float *grid,
                                         int64_t N, int mode, float *out)
{
    for (int c = 0; c < C; ++c) {
        const float *plane = input + (size_t)c * H * W;
        float *o = out + (size_t)c * N;
        if (mode == 0)
            for (int64_t i = 0; i < N; ++i) o[i] = bilinear_plane(plane, H, W, grid[2 * i], grid[2 * i + 1]);
        else
            for (int64_t i = 0; i < N; ++i) o[i] = nearest_plane(plane, H, W, grid[2 * i], grid[2 * i + 1]);
    }
}

/* ------------------------------------------------------------------------ */
/* build_dfm_cost for ONE sample (reference semantics are B=1,               */
/* dfm_backbone.py:257-275); caller loops the batch.                         */
/* cur/prev (C,h_in,w_in) -> out (2C, D, h_out, w_out)                       */
/* ------------------------------------------------------------------------ */
ORACLE_API void dfm_oracle_build_dfm_cost(const dfm_oracle_sweep_params *p, const float *depths,
                                          const float *cur, const float *prev, int C, float *out)
{
    const int64_t N = (int64_t)p->D * p->h_out * p->w_out;
    const size_t plane = (size_t)p->h_in * p->w_in;
    const int W = p->w_out;
    /* (d,h) rows are independent: OpenMP only changes who computes which row.
     * Per row: the grids once, then channel-outer / w-inner so writes stream. */
#pragma omp parallel
    {
        float *g = (float *)malloc((size_t)W * 4 * sizeof(float));
#pragma omp for collapse(2) schedule(static)
        for (int d = 0; d < p->D; ++d)
            for (int h = 0; h < p->h_out; ++h) {
                const int64_t n0 = ((int64_t)d * p->h_out + h) * W;
                for (int w = 0; w < W; ++w) sweep_point(p, depths[d], h, w, g + 4 * w, g + 4 * w + 2);
                for (int c = 0; c < C; ++c) {
                    float *oc = out + (size_t)c * N + n0;
                    float *op = out + (size_t)(C + c) * N + n0;
                    const float *pc = cur + c * plane, *pp = prev + c * plane;
                    for (int w = 0; w < W; ++w) {
                        oc[w] = bilinear_plane(pc, p->h_in, p->w_in, g[4 * w], g[4 * w + 1]);
                        op[w] = bilinear_plane(pp, p->h_in, p->w_in, g[4 * w + 2], g[4 * w + 3]);
                    }
                }
            }
        free(g);
    }
}

/* ------------------------------------------------------------------------ */
/* point_sample: mmdet3d/models/fusion_layers/point_fusion.py:14-106         */
/* (apply_3d_transformation is the identity for the DfM configs: no          */
/*  transformation_3d_flow in the multi-view pipelines)                      */
/* feat (C,Hf,Wf); points (N,3); proj 4x4 row major; out (N,C); valid (N)     */
/* mode 0 = nearest (aligned=False), 1 = bilinear (aligned=True)             */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t C, Hf, Wf;
    float scale_x, scale_y;   /* img_scale_factor (w, h)                       */
    float crop_x, crop_y;     /* img_crop_offset                               */
    int32_t flip;
    float ori_w;              /* img_shape[1]                                  */
    float pad_h, pad_w;       /* img_pad_shape                                 */
    int32_t mode;
    float proj[16];
} dfm_oracle_ps_params;

ORACLE_API void dfm_oracle_point_sample(const dfm_oracle_ps_params *p, const float *feat,
                                        const float *points, int64_t N, float *out,
                                        uint8_t *valid)
{
    const size_t plane = (size_t)p->Hf * p->Wf;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        /* points_cam2img(points, proj_mat, with_depth=True), utils.py:206-212 */
        const float p4[4] = {points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.0f};
        const float a = dot4_chain(p4, p->proj + 0);
        const float b = dot4_chain(p4, p->proj + 4);
        const float c = dot4_chain(p4, p->proj + 8);
        float x = a / c, y = b / c;
        const float depth = c;
        /* :70-80 scale -> crop -> flip */
        x = x * p->scale_x;
        y = y * p->scale_y;
        x = x - p->crop_x;
        y = y - p->crop_y;
        if (p->flip) x = p->ori_w - x;
        /* :82-84 */
        const float ny = y / p->pad_h * 2.0f - 1.0f;
        const float nx = x / p->pad_w * 2.0f - 1.0f;
        /* :99-101 */
        const int ok = (x < p->pad_w) && (x > 0.0f) && (y < p->pad_h) && (y > 0.0f) && (depth > 0.0f);
        if (valid) valid[i] = (uint8_t)ok;
        for (int ch = 0; ch < p->C; ++ch) {
            float v = 0.0f;
            if (ok || !valid)
                v = p->mode ? bilinear_plane(feat + ch * plane, p->Hf, p->Wf, nx, ny)
                            : nearest_plane(feat + ch * plane, p->Hf, p->Wf, nx, ny);
            out[(size_t)i * p->C + ch] = v; /* valid_features[~valid] = 0, :103 */
        }
    }
}

ORACLE_API int dfm_oracle_version(void) { return 1; }

#ifdef _OPENMP
#include <omp.h>
ORACLE_API int dfm_oracle_max_threads(void) { return omp_get_max_threads(); }
ORACLE_API void dfm_oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
ORACLE_API int dfm_oracle_max_threads(void) { return 1; }
ORACLE_API void dfm_oracle_set_threads(int n) { (void)n; }
#endif

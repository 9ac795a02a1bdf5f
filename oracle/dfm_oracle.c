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

/* ------------------------------------------------------------------------ */
/* FrustumToVoxel sampling stage                                             */
/* reference: mmdet3d/models/necks/feature_transformation.py:82-158          */
/* 3-D grid_sample = ATen native/GridSampler.cpp grid_sampler_3d_cpu_impl    */
/* (scalar path): unnormalise ((g+1)/2)*(size-1); corner weights             */
/*   tnw = (ix_bse-ix)*(iy_bse-iy)*(iz_bse-iz) ...; out = 0; out += v*w for   */
/* the in-bounds corners in the order tnw,tne,tsw,tse,bnw,bne,bsw,bse (plain  */
/* multiply then add: that file is not built with FMA).                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t C;           /* cost-volume channels                              */
    int32_t D, H, W;     /* cost volume (stereo_feat) size                    */
    int32_t Ds, Hs, Ws;  /* depth-distribution volume size (softmax)          */
    int32_t Csem;        /* semantic channels (0: cat_img_feature=False)      */
    int32_t Hsem, Wsem;
    int32_t Nz, Ny, Nx;
    float pad_h, pad_w;  /* img_metas[0]['pad_shape']                         */
    float depth_min;     /* coordinate offset  (fp32 of the python number)    */
    float depth_span;    /* fp32(depth_max - depth_min)                       */
    float P[12];         /* cam2img[:3] (3x4), row major                      */
    int32_t stereo_atten;      /* stereo_atten_feat (:141-142): Voxel *= pred_disp           */
    int32_t no_sem_atten;      /* sem_atten_feat=False (:154-155): Voxel_2D not depth-weighted */
} dfm_oracle_f2v_params;

static inline float trilinear(const float *vol, int D, int H, int W, float gx, float gy, float gz)
{
    const float ix = unnormalize_ac(gx, W), iy = unnormalize_ac(gy, H), iz = unnormalize_ac(gz, D);
    if (!isfinite(ix) || !isfinite(iy) || !isfinite(iz)) return 0.0f;
    if (fabsf(ix) > 1e9f || fabsf(iy) > 1e9f || fabsf(iz) > 1e9f) return 0.0f;
    const int64_t x0 = (int64_t)floorf(ix), y0 = (int64_t)floorf(iy), z0 = (int64_t)floorf(iz);
    const int64_t x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    const float fz0 = (float)z0, fz1 = (float)z1;
    const float tnw = (fx1 - ix) * (fy1 - iy) * (fz1 - iz);
    const float tne = (ix - fx0) * (fy1 - iy) * (fz1 - iz);
    const float tsw = (fx1 - ix) * (iy - fy0) * (fz1 - iz);
    const float tse = (ix - fx0) * (iy - fy0) * (fz1 - iz);
    const float bnw = (fx1 - ix) * (fy1 - iy) * (iz - fz0);
    const float bne = (ix - fx0) * (fy1 - iy) * (iz - fz0);
    const float bsw = (fx1 - ix) * (iy - fy0) * (iz - fz0);
    const float bse = (ix - fx0) * (iy - fy0) * (iz - fz0);
#define IN3(z, y, x) ((z) >= 0 && (z) < D && (y) >= 0 && (y) < H && (x) >= 0 && (x) < W)
#define AT3(z, y, x) vol[((size_t)(z) * H + (y)) * W + (x)]
    float out = 0.0f;
    if (IN3(z0, y0, x0)) out += AT3(z0, y0, x0) * tnw;
    if (IN3(z0, y0, x1)) out += AT3(z0, y0, x1) * tne;
    if (IN3(z0, y1, x0)) out += AT3(z0, y1, x0) * tsw;
    if (IN3(z0, y1, x1)) out += AT3(z0, y1, x1) * tse;
    if (IN3(z1, y0, x0)) out += AT3(z1, y0, x0) * bnw;
    if (IN3(z1, y0, x1)) out += AT3(z1, y0, x1) * bne;
    if (IN3(z1, y1, x0)) out += AT3(z1, y1, x0) * bsw;
    if (IN3(z1, y1, x1)) out += AT3(z1, y1, x1) * bse;
#undef IN3
#undef AT3
    return out;
}

/* one sample: stereo (C,D,H,W), soft (1,Ds,Hs,Ws), sem (Csem,Hsem,Wsem) or NULL,
 * coords (Nz,Ny,Nx,3) pseudo-LiDAR voxel centres -> out (C+Csem, Nz,Ny,Nx) */
ORACLE_API void dfm_oracle_frustum_to_voxel(const dfm_oracle_f2v_params *p, const float *stereo,
                                            const float *soft, const float *sem,
                                            const float *coords, float *out)
{
    const int64_t N = (int64_t)p->Nz * p->Ny * p->Nx;
    const size_t vol = (size_t)p->D * p->H * p->W;
    const size_t semplane = (size_t)p->Hsem * p->Wsem;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
        /* :176-178 pseudo-LiDAR -> rect camera; :181-188 project with the 3x4 */
        const float r4[4] = {-ys, -zs, xs, 1.0f};
        const float a = dot4_chain(r4, p->P + 0), b = dot4_chain(r4, p->P + 4);
        const float c = dot4_chain(r4, p->P + 8);
        const float u = a / c, v = b / c, depth = xs; /* :95 cat c3d[..., 2:] */
        /* :102-105 */
        const int valid2d = (u >= 0.0f) && (u <= p->pad_w) && (v >= 0.0f) && (v <= p->pad_h);
        /* :110-119 */
        float gx = (u - 0.0f) / (p->pad_w - 1.0f), gy = (v - 0.0f) / (p->pad_h - 1.0f);
        float gz = (depth - p->depth_min) / p->depth_span;
        gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
        /* :125-127 */
        const float valid = (valid2d && gz >= -1.0f && gz <= 1.0f) ? 1.0f : 0.0f;
        /* :130-139: pred_disp only when one of the attentions wants it */
        float disp = 1.0f;
        if (p->stereo_atten || (p->Csem > 0 && !p->no_sem_atten))
            disp = trilinear(soft, p->Ds, p->Hs, p->Ws, gx, gy, gz) * valid;
        for (int ch = 0; ch < p->C; ++ch) {
            float s3 = trilinear(stereo + ch * vol, p->D, p->H, p->W, gx, gy, gz) * valid;
            if (p->stereo_atten) s3 = s3 * disp; /* :141-142 */
            out[(size_t)ch * N + i] = s3;
        }
        if (p->Csem > 0) {
            const float v2d = valid2d ? 1.0f : 0.0f;
            /* :146-155 semantic feature at z := 0, masked, depth-weighted */
            for (int ch = 0; ch < p->Csem; ++ch) {
                float s2 = trilinear(sem + ch * semplane, 1, p->Hsem, p->Wsem, gx, gy, 0.0f);
                s2 = s2 * v2d;
                if (!p->no_sem_atten) s2 = s2 * disp;
                out[(size_t)(p->C + ch) * N + i] = s2;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* DepthHead.forward (with_convs=False)                                      */
/* reference: mmdet3d/models/dense_heads/depth_head.py:205-210               */
/*  nn.Upsample(scale, 'trilinear', align_corners=True): ATen                 */
/*  native/cpu/UpSampleKernel.cpp -- index0 = min(floor(scale*i), in-1),      */
/*  lambda1 = clamp(scale*i - index0, 0, 1), scale = (in-1)/(out-1) in fp32,  */
/*  value = fma(w0, a, w1*b) nested W -> H -> D (pinned bit-exact by           */
/*  tests/golden/depth_head_*.npz).                                           */
/*  softmax(dim=2) and sum(softmax*depth): torch uses a vectorised Sleef exp; */
/*  this restatement uses libm expf -- equal to rounding error, NOT bitwise   */
/*  (parity tolerance rtol 2e-6 for softmax, 1e-5 for the expectation).       */
/* ------------------------------------------------------------------------ */
static inline void up_index(int i, int in, int out, int *i0, int *i1, float *w0, float *w1)
{
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    const float real = scale * (float)i;
    int a = (int)floorf(real);
    if (a > in - 1) a = in - 1;
    float l = real - (float)a;
    l = l < 0.0f ? 0.0f : (l > 1.0f ? 1.0f : l);
    *i0 = a;
    *i1 = a + 1 < in ? a + 1 : in - 1;
    *w1 = l;
    *w0 = 1.0f - l;
}
static inline float lerp_fma(float w0, float a, float w1, float b) { return fmaf(w0, a, w1 * b); }

/* in (D,H,W) one sample; s = upsample factor; depth_samples (s*D);
 * vol, soft (s*D, s*H, s*W); pred (s*H, s*W) */
ORACLE_API void dfm_oracle_depth_head(const float *in, int D, int H, int W, int s,
                                      const float *depth_samples, float *vol, float *soft,
                                      float *pred)
{
    const int Do = D * s, Ho = H * s, Wo = W * s;
#pragma omp parallel for collapse(2) schedule(static)
    for (int h = 0; h < Ho; ++h)
        for (int w = 0; w < Wo; ++w) {
            int h0, h1, w0, w1;
            float wh0, wh1, ww0, ww1;
            up_index(h, H, Ho, &h0, &h1, &wh0, &wh1);
            up_index(w, W, Wo, &w0, &w1, &ww0, &ww1);
            float mx = -INFINITY;
            for (int d = 0; d < Do; ++d) {
                int d0, d1;
                float wd0, wd1;
                up_index(d, D, Do, &d0, &d1, &wd0, &wd1);
                const float *p0 = in + (size_t)d0 * H * W, *p1 = in + (size_t)d1 * H * W;
                const float a0 = lerp_fma(ww0, p0[h0 * W + w0], ww1, p0[h0 * W + w1]);
                const float b0 = lerp_fma(ww0, p0[h1 * W + w0], ww1, p0[h1 * W + w1]);
                const float a1 = lerp_fma(ww0, p1[h0 * W + w0], ww1, p1[h0 * W + w1]);
                const float b1 = lerp_fma(ww0, p1[h1 * W + w0], ww1, p1[h1 * W + w1]);
                const float v = lerp_fma(wd0, lerp_fma(wh0, a0, wh1, b0), wd1, lerp_fma(wh0, a1, wh1, b1));
                vol[((size_t)d * Ho + h) * Wo + w] = v;
                if (v > mx) mx = v;
            }
            float sum = 0.0f;
            for (int d = 0; d < Do; ++d) {
                const float e = expf(vol[((size_t)d * Ho + h) * Wo + w] - mx);
                soft[((size_t)d * Ho + h) * Wo + w] = e;
                sum += e;
            }
            float acc = 0.0f;
            for (int d = 0; d < Do; ++d) {
                const float pr = soft[((size_t)d * Ho + h) * Wo + w] / sum;
                soft[((size_t)d * Ho + h) * Wo + w] = pr;
                acc += pr * depth_samples[d];
            }
            pred[(size_t)h * Wo + w] = acc;
        }
}

/* ------------------------------------------------------------------------ */
/* voxel_sample: mmdet3d/models/fusion_layers/point_fusion.py:324-410        */
/* frustum lattice -> LiDAR (points_img2cam with the inverse of lidar2img)    */
/* -> voxel index -> normalised (z,y,x) grid -> 3-D grid_sample of the        */
/* (C, Nx, Ny, Nz) voxel volume.  mode 1 trilinear (aligned), 0 nearest.      */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t C, Nx, Ny, Nz;     /* voxel_features (1, C, Nx, Ny, Nz)            */
    int32_t D, h_out, w_out;   /* output lattice                               */
    float ds;                  /* downsample_factor                            */
    float scale_x, scale_y, crop_x, crop_y;
    int32_t flip;
    float ori_w;
    float range[6];            /* voxel_range                                  */
    float vsize[3];            /* voxel_size                                   */
    int32_t mode;
    float Minv[16];            /* torch.inverse(proj_mat), row major           */
} dfm_oracle_vs_params;

static inline float nearest3(const float *vol, int D, int H, int W, float gx, float gy, float gz)
{
    const float ix = unnormalize_ac(gx, W), iy = unnormalize_ac(gy, H), iz = unnormalize_ac(gz, D);
    if (!isfinite(ix) || !isfinite(iy) || !isfinite(iz)) return 0.0f;
    if (fabsf(ix) > 1e9f || fabsf(iy) > 1e9f || fabsf(iz) > 1e9f) return 0.0f;
    const int64_t x = (int64_t)nearbyintf(ix), y = (int64_t)nearbyintf(iy), z = (int64_t)nearbyintf(iz);
    if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) return vol[((size_t)z * H + y) * W + x];
    return 0.0f;
}

/* depths: the D plane depths actually used (depth_samples[::downsample_factor]) */
ORACLE_API void dfm_oracle_voxel_sample(const dfm_oracle_vs_params *p, const float *vox,
                                        const float *depths, float *out)
{
    const int64_t N = (int64_t)p->D * p->h_out * p->w_out;
    const size_t vol = (size_t)p->Nx * p->Ny * p->Nz;
    /* grid_size = (range[3:] - range[:3]) / voxel_size, fp32 tensor ops (:393) */
    float gsz[3];
    for (int k = 0; k < 3; ++k) gsz[k] = (p->range[3 + k] - p->range[k]) / p->vsize[k];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const int w = (int)(i % p->w_out), h = (int)((i / p->w_out) % p->h_out);
        const int d = (int)(i / ((int64_t)p->w_out * p->h_out));
        float x = (float)w * p->ds, y = (float)h * p->ds; /* :371-372 */
        const float depth = depths[d];
        if (p->flip) x = p->ori_w - x;                    /* :380-384 */
        x = x + p->crop_x; y = y + p->crop_y;             /* :385 */
        x = x / p->scale_x; y = y / p->scale_y;           /* :386 */
        const float homo[4] = {x * depth, y * depth, depth, 1.0f};
        float X[3];
        for (int k = 0; k < 3; ++k) X[k] = dot4_chain(homo, p->Minv + 4 * k); /* points_img2cam */
        float g[3];
        for (int k = 0; k < 3; ++k) {
            float v = (X[k] - p->range[k]) / p->vsize[k] - 0.5f; /* :392 */
            g[k] = v / gsz[k] * 2.0f - 1.0f;                      /* :395 */
        }
        /* :397 (x,y,z) -> (z,y,x): grid x indexes Nz, grid z indexes Nx */
        for (int c = 0; c < p->C; ++c)
            out[(size_t)c * N + i] =
                p->mode ? trilinear(vox + c * vol, p->Nx, p->Ny, p->Nz, g[2], g[1], g[0])
                        : nearest3(vox + c * vol, p->Nx, p->Ny, p->Nz, g[2], g[1], g[0]);
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

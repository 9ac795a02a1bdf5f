// point_sample.hip -- multi-view voxel lifting (gfx950)
//
// Reference: point_sample (mmdet3d/models/fusion_layers/point_fusion.py:14-106)
// called once per (frame, view) by MultiViewDfM.feature_transformation
// (mmdet3d/models/detectors/multiview_dfm.py:119-208), followed by the
// valid-count reduction over views and frames and the (Nz,Ny,Nx,C) ->
// (C,Nx,Ny,Nz) permute.  Here ONE launch does all views and frames of a sample
// and writes the final volume: no (N,C) per-view temporaries, no stack/sum/
// permute passes.
//
// Layout in HBM
//   feats     : caller tensor (F*Nv, C, Hf, Wf), f32 or bf16
//   workspace : the same, pixel-major [F*Nv][Hf][Wf][Cp] (Cp = C rounded up to a whole
//               16-byte channel block): the sampling is sparse (neighbouring voxels hit
//               pixels several columns apart), so a tap should be ONE contiguous run
//               of Cp*sizeof(T) bytes, not C/CB pieces a plane apart
//   out       : volume (C*F', Nx, Ny, Nz) or flat (N, C)
// Bound: HBM write of the volume + L2-resident gathers; no reuse to stage.
#include "dfm_common.h"

// The lifted volume (hundreds of MB, written once, read by a later kernel) leaves the CU with
// non-temporal 16-byte stores: the batched channels-last kernel writes whole contiguous KiBs per
// instruction and gains 6 % (waymo_cl 0.2366 -> 0.2235 ms, profiles/archive/r04_c7_lift_nt_vs_plain.txt);
// -DDFM_LIFT_PLAIN builds the plain-store variant.  (FrustumToVoxel's lane-per-voxel kernel must NOT
// use nt: its stores are partial lines per instruction, see frustum_to_voxel.hip.)
template <typename T>
__device__ __forceinline__ void lift_store16(T *p, const float (&f)[dfm::vec16<T>::N])
{
#ifdef DFM_LIFT_PLAIN
    dfm::store16<T>(p, f);
#else
    typedef uint32_t lift_u32x4 __attribute__((ext_vector_type(4)));
    lift_u32x4 v;
    if constexpr (sizeof(T) == 4) {
        v = lift_u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    } else {
        v = lift_u32x4{dfm::pack_bf16x2(f[0], f[1]), dfm::pack_bf16x2(f[2], f[3]), dfm::pack_bf16x2(f[4], f[5]),
                       dfm::pack_bf16x2(f[6], f[7])};
    }
    __builtin_nontemporal_store(v, (lift_u32x4 *)p);
#endif
}


#include <stdio.h>

#include <algorithm>

using namespace dfm;

namespace {

int fail_ps(int code, const char *msg) { return dfm::set_error(code, msg); }

struct MvGeom {
    int32_t num_views, num_frames, C, Hf, Wf, nblk;
    int32_t nx, ny, nz;  // nz == 0 : flat (N, C) output
    long long N;
    float scale_x, scale_y, crop_x, crop_y, pad_h, pad_w;
    int32_t flip, mode, aggregate, valid_sample;
    int32_t out_cl;  // volume stored channels-last: (nx, ny, nz, C*F') -- torch channels_last_3d
};

// projection + image transform of one point into one view; returns validity
// (point_fusion.py:61-84,99-101) and the normalised grid coordinates
__device__ __forceinline__ bool project_view(const MvGeom &g, const float *__restrict__ M,
                                             float ori_w, float px, float py, float pz, float &nx,
                                             float &ny)
{
    const float a = dot4_chain(px, py, pz, 1.0f, M + 0);
    const float b = dot4_chain(px, py, pz, 1.0f, M + 4);
    const float c = dot4_chain(px, py, pz, 1.0f, M + 8);
    float x = a / c, y = b / c;
    x = x * g.scale_x;
    y = y * g.scale_y;
    x = x - g.crop_x;
    y = y - g.crop_y;
    if (g.flip) x = ori_w - x;
    ny = y / g.pad_h * 2.0f - 1.0f;
    nx = x / g.pad_w * 2.0f - 1.0f;
    return (x < g.pad_w) && (x > 0.0f) && (y < g.pad_h) && (y > 0.0f) && (c > 0.0f);
}

// one thread = one voxel (output order).  32 channels (NB 16-byte blocks) are
// accumulated per pass so every (frame, view) is projected once per pass, and the NB
// loads of a tap are adjacent.  Per channel the additions run in the reference's
// order (views, then frames); a skipped out-of-image tap adds nothing, which equals
// adding its zeros (the accumulators start at +0 and can never become -0).
template <typename T>
__global__ __launch_bounds__(256) void mv_sample_kernel(
    MvGeom g, const uint4 *__restrict__ feats, const float *__restrict__ points,
    const float *__restrict__ proj, const float *__restrict__ ori_w, T *__restrict__ out,
    unsigned char *__restrict__ valid_out)
{
    constexpr int CB = elem<T>::CB;
    constexpr int NB = 32 / CB;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= g.N) return;
    long long pidx = o;
    if (g.nz > 0) {
        const int z = (int)(o % g.nz);
        const long long t = o / g.nz;
        const int y = (int)(t % g.ny);
        const int x = (int)(t / g.ny);
        pidx = ((long long)z * g.ny + y) * g.nx + x;  // anchor order: z-major, then y, then x
    }
    const float px = points[3 * pidx], py = points[3 * pidx + 1], pz = points[3 * pidx + 2];
    const int HW = g.Hf * g.Wf;
    // volume: (C, N); flat, and the channels-last volume (what the NDHWC neck convolutions read): (N, C)
    const bool pm_out = g.nz == 0 || g.out_cl;
    const size_t chan_stride = pm_out ? 1 : (size_t)g.N;
    T *obase = pm_out ? out + (size_t)o * g.C * (g.aggregate ? g.num_frames : 1) : out + o;

    for (int blk0 = 0; blk0 < g.nblk; blk0 += NB) {
        float tot[NB][CB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int k = 0; k < CB; ++k) tot[j][k] = 0.0f;
        int tot_cnt = 0, nvalid = 0;
        for (int f = 0; f < g.num_frames; ++f) {
            float acc[NB][CB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int k = 0; k < CB; ++k) acc[j][k] = 0.0f;
            int cnt = 0;
            for (int v = 0; v < g.num_views; ++v) {
                const int i = f * g.num_views + v;
                float nx, ny;
                const bool ok = project_view(g, proj + 16 * i, ori_w[i], px, py, pz, nx, ny);
                nvalid += ok ? 1 : 0;
                if (g.valid_sample && !ok) continue;  // valid_features[~valid] = 0
                ++cnt;
                const float x = ((nx + 1.0f) * 0.5f) * (float)(g.Wf - 1);
                const float y = ((ny + 1.0f) * 0.5f) * (float)(g.Hf - 1);
                const uint4 *fb = feats + (size_t)i * HW * g.nblk + blk0;
                if (g.mode == 0) {
                    // nearest: nearbyint (round half to even), zeros outside
                    const float xr = rintf(x), yr = rintf(y);
                    const bool in = (fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f) && xr >= 0.0f &&
                                    xr <= (float)(g.Wf - 1) && yr >= 0.0f && yr <= (float)(g.Hf - 1);
                    if (!in) continue;
                    const uint4 *p = fb + (size_t)((int)yr * g.Wf + (int)xr) * g.nblk;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (blk0 + j < g.nblk) {
                            float r[CB];
                            unpack16(p[j], r);
#pragma unroll
                            for (int k = 0; k < CB; ++k) acc[j][k] = acc[j][k] + r[k];  // stack(views).sum(0)
                        }
                    }
                } else {
                    const Tap t = make_tap(x, y, g.Hf, g.Wf);
                    const int i00 = t.iy * g.Wf + t.ix, i01 = i00 + t.dx;
                    const int i10 = i00 + t.dy * g.Wf, i11 = i10 + t.dx;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (blk0 + j >= g.nblk) continue;
                        const uint4 q0 = fb[(size_t)i00 * g.nblk + j], q1 = fb[(size_t)i01 * g.nblk + j];
                        const uint4 q2 = fb[(size_t)i10 * g.nblk + j], q3 = fb[(size_t)i11 * g.nblk + j];
                        float a[CB], b2[CB], c2[CB], d2[CB];
                        unpack16(q0, a); unpack16(q1, b2); unpack16(q2, c2); unpack16(q3, d2);
#pragma unroll
                        for (int k = 0; k < CB; ++k) {
                            const float vnw = (t.ok & 1u) ? a[k] : 0.0f, vne = (t.ok & 2u) ? b2[k] : 0.0f;
                            const float vsw = (t.ok & 4u) ? c2[k] : 0.0f, vse = (t.ok & 8u) ? d2[k] : 0.0f;
                            float s = vnw * t.nw;
                            s = __builtin_fmaf(vne, t.ne, s);
                            s = __builtin_fmaf(vsw, t.sw, s);
                            s = __builtin_fmaf(vse, t.se, s);
                            acc[j][k] = acc[j][k] + s;
                        }
                    }
                }
            }
            if (g.aggregate) {
                // 'concat': per-frame mean over its valid views, multiview_dfm.py:196-203
                const float den = (float)max(cnt, 1);
                if (g.out_cl && g.C % CB == 0) {  // 16-byte stores of CB consecutive channels
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (blk0 + j < g.nblk) {
                            float r[CB];
#pragma unroll
                            for (int k = 0; k < CB; ++k) r[k] = acc[j][k] / den;
                            lift_store16<T>(obase + (size_t)f * g.C + (size_t)(blk0 + j) * CB, r);
                        }
                } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int k = 0; k < CB; ++k) {
                        const int c = (blk0 + j) * CB + k;
                        if (c < g.C)
                            obase[(size_t)(f * g.C + c) * chan_stride] = elem<T>::store(acc[j][k] / den);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int k = 0; k < CB; ++k) tot[j][k] = tot[j][k] + acc[j][k];  // stack(frames).sum(0)
                tot_cnt += cnt;
            }
        }
        if (!g.aggregate) {
            // 'mean': sum over frames / clamp(total valid, 1), multiview_dfm.py:188-195
            const float den = (float)max(tot_cnt, 1);
            if (g.out_cl && g.C % CB == 0) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (blk0 + j < g.nblk) {
                        float r[CB];
#pragma unroll
                        for (int k = 0; k < CB; ++k) r[k] = tot[j][k] / den;
                        lift_store16<T>(obase + (size_t)(blk0 + j) * CB, r);
                    }
            } else {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const int c = (blk0 + j) * CB + k;
                    if (c < g.C) obase[(size_t)c * chan_stride] = elem<T>::store(tot[j][k] / den);
                }
            }
        }
        if (valid_out && blk0 == 0) valid_out[o] = nvalid > 0;
    }
}


// ---------------------------------------------------------------------------
// Channels-last lifting, a whole batch per launch (the multi-view configs' bf16 path): LPV lanes
// per voxel, one per 16-byte channel block.
//   * lane = voxel made a wave's stores 16-byte pieces 128-256 B apart (one per voxel row of the
//     channels-last volume) and its tap loads 64 different cache lines per instruction; here the
//     LPV lanes of a voxel read one contiguous (C * sizeof(T))-byte tap and write one contiguous
//     row piece, and a wave's 64 / LPV voxels are consecutive rows of the volume: whole KiBs;
//   * the projections of a voxel's F * Nv views are computed ONCE, spread over its LPV lanes
//     (lane j takes views j, j + LPV, ...), and handed round with wave shuffles;
//   * blockIdx.y = sample: the per-sample image transform comes from a small by-value table
//     (one launch per batch instead of one per sample from Python).
// Nearest sampling only (aligned=False: what MultiViewDfM / ImVoxelNet pass,
// multiview_dfm.py:169, imvoxelnet.py:71); bilinear keeps the kernel above.  Same per-channel
// addition order as the reference (views, then frames) -- bit-identical results.
// ---------------------------------------------------------------------------
constexpr int MV_MAX_BATCH = 16;
struct MvBatch {
    float scale_x[MV_MAX_BATCH], scale_y[MV_MAX_BATCH], crop_x[MV_MAX_BATCH], crop_y[MV_MAX_BATCH];
    float pad_h[MV_MAX_BATCH], pad_w[MV_MAX_BATCH];
    int32_t flip[MV_MAX_BATCH];
    long long points_stride;  // 0: one point set shared by the batch
};

template <typename T, int LPV>
__global__ __launch_bounds__(256) void mv_sample_cl_kernel(
    MvGeom g0, MvBatch mb, const uint4 *__restrict__ feats, const float *__restrict__ points,
    const float *__restrict__ proj, const float *__restrict__ ori_w, T *__restrict__ out,
    unsigned char *__restrict__ valid_out)
{
    constexpr int CB = elem<T>::CB;
    constexpr int VPW = 64 / LPV;   // voxels per wave
    constexpr int KMAX = 4;         // up to 4 * LPV (frame, view) pairs
    const int b = blockIdx.y;
    MvGeom g = g0;
    g.scale_x = mb.scale_x[b]; g.scale_y = mb.scale_y[b]; g.crop_x = mb.crop_x[b]; g.crop_y = mb.crop_y[b];
    g.pad_h = mb.pad_h[b]; g.pad_w = mb.pad_w[b]; g.flip = mb.flip[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = lane % LPV, vgrp = lane / LPV;
    const long long o = ((long long)blockIdx.x * 4 + wave) * VPW + vgrp;
    const bool live = o < g.N;
    const long long oc = live ? o : g.N - 1;
    long long pidx = oc;
    if (g.nz > 0) {
        const int z = (int)(oc % g.nz);
        const long long t = oc / g.nz;
        const int y = (int)(t % g.ny);
        const int x = (int)(t / g.ny);
        pidx = ((long long)z * g.ny + y) * g.nx + x;  // anchor order: z-major, then y, then x
    }
    const float *pts = points + (size_t)b * mb.points_stride;
    const float px = pts[3 * pidx], py = pts[3 * pidx + 1], pz = pts[3 * pidx + 2];
    const int HW = g.Hf * g.Wf, nvf = g.num_views * g.num_frames;
    const float *pj = proj + (size_t)b * nvf * 16;
    const float *ow = ori_w + (size_t)b * nvf;
    const uint4 *fb = feats + (size_t)b * nvf * HW * LPV + blk;
    const int c_out = g.C * (g.aggregate ? g.num_frames : 1);
    T *orow = out + ((size_t)b * g.N + (size_t)oc) * c_out + (size_t)blk * CB;

    // this lane's share of the voxel's projections: code = -2 not valid, -1 valid but outside the
    // map (counts, adds zeros), >= 0 the nearest pixel
    int code[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int i = k * LPV + blk;
        code[k] = -2;
        if (i < nvf) {
            float nx, ny;
            const bool ok = project_view(g, pj + 16 * i, ow[i], px, py, pz, nx, ny);
            if (ok) {  // valid_features[~valid] = 0: a view that does not see the point contributes nothing
                const float x = ((nx + 1.0f) * 0.5f) * (float)(g.Wf - 1);
                const float y = ((ny + 1.0f) * 0.5f) * (float)(g.Hf - 1);
                const float xr = rintf(x), yr = rintf(y);  // nearbyint: round half to even
                const bool in = (fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f) && xr >= 0.0f &&
                                xr <= (float)(g.Wf - 1) && yr >= 0.0f && yr <= (float)(g.Hf - 1);
                code[k] = in ? (int)yr * g.Wf + (int)xr : -1;
            }
        }
    }
    float tot[CB], acc[CB];
#pragma unroll
    for (int e = 0; e < CB; ++e) { tot[e] = 0.0f; acc[e] = 0.0f; }
    int tot_cnt = 0, cnt = 0, nvalid = 0, f = 0, v = 0;
    const int base = lane - blk;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        for (int j = 0; j < LPV; ++j) {
            const int i = k * LPV + j;
            if (i >= nvf) break;
            const int cd = __shfl(code[k], base + j);
            if (cd != -2) {
                ++cnt;
                ++nvalid;
                if (cd >= 0) {
                    float r[CB];
                    unpack16(fb[((size_t)i * HW + cd) * LPV], r);
#pragma unroll
                    for (int e = 0; e < CB; ++e) acc[e] = acc[e] + r[e];  // stack(views).sum(0)
                }
            }
            if (++v == g.num_views) {  // end of frame f
                if (g.aggregate) {
                    const float den = (float)max(cnt, 1);
                    float r[CB];
#pragma unroll
                    for (int e = 0; e < CB; ++e) r[e] = acc[e] / den;
                    if (live) lift_store16<T>(orow + (size_t)f * g.C, r);
                } else {
#pragma unroll
                    for (int e = 0; e < CB; ++e) tot[e] = tot[e] + acc[e];  // stack(frames).sum(0)
                    tot_cnt += cnt;
                }
#pragma unroll
                for (int e = 0; e < CB; ++e) acc[e] = 0.0f;
                cnt = 0; v = 0; ++f;
            }
        }
    }
    if (!g.aggregate) {
        const float den = (float)max(tot_cnt, 1);
        float r[CB];
#pragma unroll
        for (int e = 0; e < CB; ++e) r[e] = tot[e] / den;
        if (live) lift_store16<T>(orow, r);
    }
    if (valid_out && live && blk == 0) valid_out[(size_t)b * g.N + o] = nvalid > 0;
}

}  // namespace

extern "C" {

DFM_API size_t dfm_point_sample_mv_workspace_bytes(const dfm_mv_desc *d)
{
    if (!d || d->num_views <= 0 || d->num_frames <= 0 || d->channels <= 0 || d->feat_h <= 0 ||
        d->feat_w <= 0)
        return 0;
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    const size_t nblk = (d->channels + CB - 1) / CB;
    const size_t bytes = (size_t)d->num_views * d->num_frames * nblk * d->feat_h * d->feat_w * 16;
    return (bytes + 255) & ~(size_t)255;
}

DFM_API int dfm_point_sample_mv_fwd(const dfm_mv_desc *d, const void *feats, const float *points,
                                    const float *proj, const float *ori_w, void *out,
                                    unsigned char *valid_out, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (!d) return fail_ps(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->num_views <= 0 || d->num_frames <= 0 || d->channels <= 0 || d->feat_h <= 0 ||
        d->feat_w <= 0 || d->num_points <= 0)
        return fail_ps(DFM_ERR_INVALID_ARG, "non-positive size in dfm_mv_desc");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return fail_ps(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (d->mode != 0 && d->mode != 1) return fail_ps(DFM_ERR_UNSUPPORTED, "mode must be 0 or 1");
    if (d->nz > 0 && (long long)d->nx * d->ny * d->nz != d->num_points)
        return fail_ps(DFM_ERR_INVALID_ARG, "nx*ny*nz != num_points");
    if (!feats || !points || !proj || !ori_w || !out)
        return fail_ps(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (!d->feats_channels_last && (!workspace || workspace_bytes < dfm_point_sample_mv_workspace_bytes(d)))
        return fail_ps(DFM_ERR_WORKSPACE, "workspace smaller than dfm_point_sample_mv_workspace_bytes");
    MvGeom g;
    g.num_views = d->num_views; g.num_frames = d->num_frames; g.C = d->channels;
    g.Hf = d->feat_h; g.Wf = d->feat_w;
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    g.nblk = (d->channels + CB - 1) / CB;
    g.nx = d->nx; g.ny = d->ny; g.nz = d->nz; g.N = d->num_points;
    g.scale_x = d->scale_x; g.scale_y = d->scale_y; g.crop_x = d->crop_x; g.crop_y = d->crop_y;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.flip = d->flip; g.mode = d->mode;
    g.aggregate = d->aggregate;
    g.valid_sample = d->valid_sample;
    g.out_cl = d->out_channels_last && d->nz > 0 ? 1 : 0;
    if (!d->valid_sample && (d->num_views != 1 || d->num_frames != 1))
        return fail_ps(DFM_ERR_UNSUPPORTED, "valid_sample=0 is only supported for a single view");
    hipStream_t st = (hipStream_t)stream;
    const int HW = d->feat_h * d->feat_w;
    const int nvf = d->num_views * d->num_frames;
    const int Cp = g.nblk * CB;
    dim3 pg((HW + 63) / 64, (Cp + 31) / 32, nvf);
    const long long nb = (g.N + 255) / 256;
    if (nb > 2147483647ll) return fail_ps(DFM_ERR_UNSUPPORTED, "too many points");
    // channels-last view features ARE the pixel-major layout: sampled where they lie
    const bool in_place = d->feats_channels_last != 0;
    if (in_place && (d->channels % CB != 0 || ((uintptr_t)feats & 15)))
        return fail_ps(DFM_ERR_UNSUPPORTED, "channels-last view features need whole 16-byte channel blocks");
    const void *maps = in_place ? feats : workspace;
    if (d->dtype == DFM_F32) {
        if (!in_place)
            hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg, dim3(256), 0, st,
                               (const float *)feats, (float *)workspace, g.C, Cp, (long long)HW);
        hipLaunchKernelGGL(mv_sample_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, g,
                           (const uint4 *)maps, points, proj, ori_w, (float *)out, valid_out);
    } else {
        if (!in_place)
            hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg, dim3(256), 0, st,
                               (const bf16_t *)feats, (bf16_t *)workspace, g.C, Cp, (long long)HW);
        hipLaunchKernelGGL(mv_sample_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, g,
                           (const uint4 *)maps, points, proj, ori_w, (bf16_t *)out, valid_out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_ps(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}


DFM_API int dfm_point_sample_mv_fwd_batched(const dfm_mv_desc *descs, int32_t batch, const void *feats,
                                            const float *points, int32_t points_per_sample, const float *proj,
                                            const float *ori_w, void *out, unsigned char *valid_out, void *stream)
{
    if (!descs || batch <= 0) return fail_ps(DFM_ERR_INVALID_ARG, "no descriptors");
    const dfm_mv_desc *d = &descs[0];
    if (d->num_views <= 0 || d->num_frames <= 0 || d->channels <= 0 || d->feat_h <= 0 || d->feat_w <= 0 ||
        d->num_points <= 0)
        return fail_ps(DFM_ERR_INVALID_ARG, "non-positive size in dfm_mv_desc");
    if (!feats || !points || !proj || !ori_w || !out) return fail_ps(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (d->nz > 0 && (long long)d->nx * d->ny * d->nz != d->num_points)
        return fail_ps(DFM_ERR_INVALID_ARG, "nx*ny*nz != num_points");
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    const int nblk = d->channels / CB;
    // what the lanes-per-voxel kernel covers; anything else: DFM_ERR_UNSUPPORTED, use dfm_point_sample_mv_fwd
    if ((d->dtype != DFM_F32 && d->dtype != DFM_BF16) || d->mode != 0 || !d->valid_sample || !d->feats_channels_last ||
        (d->nz > 0 && !d->out_channels_last) || d->channels % CB != 0 || (nblk != 4 && nblk != 8 && nblk != 16) ||
        d->num_views * d->num_frames > 4 * nblk || ((uintptr_t)feats & 15) || ((uintptr_t)out & 15))
        return fail_ps(DFM_ERR_UNSUPPORTED, "batched lifting: nearest mode, valid_sample, channels-last views and volume, "
                                            "4 / 8 / 16 channel blocks, <= 4 x blocks (frame, view) pairs");
    for (int b = 1; b < batch; ++b) {
        const dfm_mv_desc &e = descs[b];
        if (e.num_views != d->num_views || e.num_frames != d->num_frames || e.channels != d->channels ||
            e.feat_h != d->feat_h || e.feat_w != d->feat_w || e.nx != d->nx || e.ny != d->ny || e.nz != d->nz ||
            e.num_points != d->num_points || e.mode != d->mode || e.aggregate != d->aggregate ||
            e.valid_sample != d->valid_sample || e.dtype != d->dtype || e.out_channels_last != d->out_channels_last ||
            e.feats_channels_last != d->feats_channels_last)
            return fail_ps(DFM_ERR_INVALID_ARG, "the samples of a batch differ in more than their image transform");
    }
    MvGeom g;
    g.num_views = d->num_views; g.num_frames = d->num_frames; g.C = d->channels;
    g.Hf = d->feat_h; g.Wf = d->feat_w; g.nblk = nblk;
    g.nx = d->nx; g.ny = d->ny; g.nz = d->nz; g.N = d->num_points;
    g.mode = 0; g.aggregate = d->aggregate; g.valid_sample = 1; g.out_cl = d->nz > 0 ? 1 : 0;
    g.scale_x = g.scale_y = 1.0f; g.crop_x = g.crop_y = 0.0f; g.pad_h = g.pad_w = 1.0f; g.flip = 0;
    hipStream_t st = (hipStream_t)stream;
    const int vpw = 64 / nblk;
    const long long nb = (g.N + 4 * vpw - 1) / (4 * vpw);
    if (nb > 2147483647ll) return fail_ps(DFM_ERR_UNSUPPORTED, "too many points");
    const size_t esz = d->dtype == DFM_BF16 ? 2 : 4;
    const size_t feat_stride = (size_t)g.num_views * g.num_frames * g.Hf * g.Wf * g.C * esz;
    const size_t out_stride = (size_t)g.N * g.C * (g.aggregate ? g.num_frames : 1) * esz;
    for (int b0 = 0; b0 < batch; b0 += MV_MAX_BATCH) {
        const int nbatch = std::min(MV_MAX_BATCH, batch - b0);
        MvBatch mb;
        for (int i = 0; i < nbatch; ++i) {
            const dfm_mv_desc &e = descs[b0 + i];
            mb.scale_x[i] = e.scale_x; mb.scale_y[i] = e.scale_y; mb.crop_x[i] = e.crop_x; mb.crop_y[i] = e.crop_y;
            mb.pad_h[i] = e.pad_h; mb.pad_w[i] = e.pad_w; mb.flip[i] = e.flip;
        }
        mb.points_stride = points_per_sample ? (long long)g.N * 3 : 0;
        const dim3 grid((unsigned)nb, nbatch);
        const uint4 *fp = (const uint4 *)((const char *)feats + b0 * feat_stride);
        const float *pp = points + (size_t)b0 * mb.points_stride;
        const float *pj = proj + (size_t)b0 * g.num_views * g.num_frames * 16;
        const float *owp = ori_w + (size_t)b0 * g.num_views * g.num_frames;
        void *op = (char *)out + b0 * out_stride;
        unsigned char *vp = valid_out ? valid_out + (size_t)b0 * g.N : nullptr;
#define MV_CL_LAUNCH(T, LPV) \
        hipLaunchKernelGGL((mv_sample_cl_kernel<T, LPV>), grid, dim3(256), 0, st, g, mb, fp, pp, pj, owp, (T *)op, vp)
        if (d->dtype == DFM_BF16) {
            if (nblk == 4) MV_CL_LAUNCH(bf16_t, 4); else if (nblk == 8) MV_CL_LAUNCH(bf16_t, 8); else MV_CL_LAUNCH(bf16_t, 16);
        } else {
            if (nblk == 4) MV_CL_LAUNCH(float, 4); else if (nblk == 8) MV_CL_LAUNCH(float, 8); else MV_CL_LAUNCH(float, 16);
        }
#undef MV_CL_LAUNCH
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_ps(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// backward of the multi-view lifting w.r.t. the view features: every valid
// (frame, view) tap receives grad_out / (number of valid views), fp32 atomics
// into a zero-initialised (F*Nv, C, Hf, Wf) tensor.
// ---------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void mv_sample_bwd_kernel(
    MvGeom g, const T *__restrict__ gout, const float *__restrict__ points,
    const float *__restrict__ proj, const float *__restrict__ ori_w, float *__restrict__ gfeats)
{
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= g.N) return;
    long long pidx = o;
    if (g.nz > 0) {
        const int z = (int)(o % g.nz);
        const long long t = o / g.nz;
        const int y = (int)(t % g.ny);
        const int x = (int)(t / g.ny);
        pidx = ((long long)z * g.ny + y) * g.nx + x;
    }
    const float px = points[3 * pidx], py = points[3 * pidx + 1], pz = points[3 * pidx + 2];
    const int HW = g.Hf * g.Wf;
    const size_t chan_stride = g.nz > 0 ? (size_t)g.N : 1;
    const T *gbase = g.nz > 0 ? gout + o : gout + (size_t)o * g.C * (g.aggregate ? g.num_frames : 1);
    int tot_cnt = 0;
    for (int i = 0; i < g.num_views * g.num_frames; ++i) {
        float nx, ny;
        tot_cnt += project_view(g, proj + 16 * i, ori_w[i], px, py, pz, nx, ny) ? 1 : 0;
    }
    for (int f = 0; f < g.num_frames; ++f) {
        int cnt = 0;
        for (int v = 0; v < g.num_views; ++v) {
            float nx, ny;
            cnt += project_view(g, proj + 16 * (f * g.num_views + v), ori_w[f * g.num_views + v], px,
                                py, pz, nx, ny) ? 1 : 0;
        }
        const float den = g.valid_sample ? (float)max(g.aggregate ? cnt : tot_cnt, 1) : 1.0f;
        for (int v = 0; v < g.num_views; ++v) {
            const int i = f * g.num_views + v;
            float nx, ny;
            const bool ok = project_view(g, proj + 16 * i, ori_w[i], px, py, pz, nx, ny);
            if (g.valid_sample && !ok) continue;
            const float x = ((nx + 1.0f) * 0.5f) * (float)(g.Wf - 1);
            const float y = ((ny + 1.0f) * 0.5f) * (float)(g.Hf - 1);
            float *gf = gfeats + (size_t)i * g.C * HW;
            int idx[4];
            float wt[4];
            int ntap = 0;
            if (g.mode == 0) {
                const float xr = rintf(x), yr = rintf(y);
                if ((fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f) && xr >= 0.0f &&
                    xr <= (float)(g.Wf - 1) && yr >= 0.0f && yr <= (float)(g.Hf - 1)) {
                    idx[0] = (int)yr * g.Wf + (int)xr; wt[0] = 1.0f; ntap = 1;
                }
            } else {
                const Tap t = make_tap(x, y, g.Hf, g.Wf);
                const int i00 = t.iy * g.Wf + t.ix, i01 = i00 + t.dx;
                const int i10 = i00 + t.dy * g.Wf, i11 = i10 + t.dx;
                if (t.ok & 1u) { idx[ntap] = i00; wt[ntap++] = t.nw; }
                if (t.ok & 2u) { idx[ntap] = i01; wt[ntap++] = t.ne; }
                if (t.ok & 4u) { idx[ntap] = i10; wt[ntap++] = t.sw; }
                if (t.ok & 8u) { idx[ntap] = i11; wt[ntap++] = t.se; }
            }
            if (!ntap) continue;
            for (int c = 0; c < g.C; ++c) {
                const int co = g.aggregate ? f * g.C + c : c;
                const float gv = elem<T>::load(gbase[(size_t)co * chan_stride]) / den;
                for (int k = 0; k < ntap; ++k) atomicAdd(gf + (size_t)c * HW + idx[k], gv * wt[k]);
            }
        }
    }
}

// Pixel-major backward (same idea as f2v_bwd_pm_kernel): the lanes of a wave are the channels
// of one voxel, so the C atomics of a tap are one contiguous run of the pixel-major scratch
// [frame*view][Hf*Wf][C]; a transpose pass adds the scratch into grad_feats.
constexpr int MV_VT = 64;  // voxels per workgroup

struct MvFoot {
    int idx[4];   // pixel per tap, -1 = none
    float w[4];   // tap weight / number of valid views
};

template <typename T>
__global__ __launch_bounds__(256) void mv_sample_bwd_pm_kernel(
    MvGeom g, const T *__restrict__ gout, const float *__restrict__ points,
    const float *__restrict__ proj, const float *__restrict__ ori_w, float *__restrict__ gf_pm)
{
    extern __shared__ float mv_lds[];
    const int nvf = g.num_views * g.num_frames;
    const int c_out = g.C * (g.aggregate ? g.num_frames : 1);
    float *gt = mv_lds;                                  // [c_out][MV_VT + 1]
    MvFoot *foot = (MvFoot *)(gt + c_out * (MV_VT + 1));  // [MV_VT][nvf]
    const long long v0 = (long long)blockIdx.x * MV_VT;
    const int nv = (int)min((long long)MV_VT, g.N - v0);
    const int tid = threadIdx.x;
    const size_t cstride = g.nz > 0 ? (size_t)g.N : 1, vstride = g.nz > 0 ? 1 : (size_t)c_out;
    for (int i = tid; i < c_out * MV_VT; i += 256) {
        const int c = i / MV_VT, v = i - c * MV_VT;
        gt[c * (MV_VT + 1) + v] = v < nv ? elem<T>::load(gout[(size_t)(v0 + v) * vstride + c * cstride]) : 0.0f;
    }
    if (tid < MV_VT) {
        for (int i = 0; i < nvf; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { foot[tid * nvf + i].idx[k] = -1; foot[tid * nvf + i].w[k] = 0.0f; }
        if (tid < nv) {
            const long long o = v0 + tid;
            long long pidx = o;
            if (g.nz > 0) {
                const int z = (int)(o % g.nz);
                const long long t = o / g.nz;
                const int y = (int)(t % g.ny);
                const int x = (int)(t / g.ny);
                pidx = ((long long)z * g.ny + y) * g.nx + x;
            }
            const float px = points[3 * pidx], py = points[3 * pidx + 1], pz = points[3 * pidx + 2];
            int tot_cnt = 0;
            for (int i = 0; i < nvf; ++i) {
                float nx, ny;
                tot_cnt += project_view(g, proj + 16 * i, ori_w[i], px, py, pz, nx, ny) ? 1 : 0;
            }
            for (int f = 0; f < g.num_frames; ++f) {
                int cnt = 0;
                for (int v = 0; v < g.num_views; ++v) {
                    float nx, ny;
                    cnt += project_view(g, proj + 16 * (f * g.num_views + v), ori_w[f * g.num_views + v],
                                        px, py, pz, nx, ny) ? 1 : 0;
                }
                const float den = g.valid_sample ? (float)max(g.aggregate ? cnt : tot_cnt, 1) : 1.0f;
                for (int v = 0; v < g.num_views; ++v) {
                    const int i = f * g.num_views + v;
                    float nx, ny;
                    const bool ok = project_view(g, proj + 16 * i, ori_w[i], px, py, pz, nx, ny);
                    if (g.valid_sample && !ok) continue;
                    const float x = ((nx + 1.0f) * 0.5f) * (float)(g.Wf - 1);
                    const float y = ((ny + 1.0f) * 0.5f) * (float)(g.Hf - 1);
                    MvFoot &ft = foot[tid * nvf + i];
                    if (g.mode == 0) {
                        const float xr = rintf(x), yr = rintf(y);
                        if ((fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f) && xr >= 0.0f &&
                            xr <= (float)(g.Wf - 1) && yr >= 0.0f && yr <= (float)(g.Hf - 1)) {
                            ft.idx[0] = (int)yr * g.Wf + (int)xr;
                            ft.w[0] = 1.0f / den;
                        }
                    } else {
                        const Tap t = make_tap(x, y, g.Hf, g.Wf);
                        const int i00 = t.iy * g.Wf + t.ix, i01 = i00 + t.dx;
                        const int i10 = i00 + t.dy * g.Wf, i11 = i10 + t.dx;
                        if (t.ok & 1u) { ft.idx[0] = i00; ft.w[0] = t.nw / den; }
                        if (t.ok & 2u) { ft.idx[1] = i01; ft.w[1] = t.ne / den; }
                        if (t.ok & 4u) { ft.idx[2] = i10; ft.w[2] = t.sw / den; }
                        if (t.ok & 8u) { ft.idx[3] = i11; ft.w[3] = t.se / den; }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    int lpv = 1;
    while (lpv < min(g.C, 64)) lpv <<= 1;  // lanes per voxel
    // a wave owns MV_VT/4 voxels: never spread an iteration over more of them (1- and 2-channel
    // maps would otherwise scatter their neighbours' voxels a second time)
    lpv = max(lpv, 64 / (MV_VT / 4));
    const int ch = lane & (lpv - 1), vin = lane / lpv, vpi = 64 / lpv;
    const size_t HW = (size_t)g.Hf * g.Wf;
    for (int it = 0; it < MV_VT / 4; it += vpi) {
        const int v = wave * (MV_VT / 4) + it + vin;
        if (v >= nv) continue;
        for (int i = 0; i < nvf; ++i) {
            const MvFoot &ft = foot[v * nvf + i];
            if (ft.idx[0] < 0 && ft.idx[1] < 0 && ft.idx[2] < 0 && ft.idx[3] < 0) continue;
            const int f = i / g.num_views;
            float *dst = gf_pm + (size_t)i * HW * g.C;
            for (int c = ch; c < g.C; c += lpv) {
                const float gv = gt[(g.aggregate ? f * g.C + c : c) * (MV_VT + 1) + v];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ft.idx[k] >= 0) atomicAdd(dst + (size_t)ft.idx[k] * g.C + c, gv * ft.w[k]);
            }
        }
    }
}

}  // namespace

extern "C" DFM_API size_t dfm_point_sample_mv_bwd_workspace_bytes(const dfm_mv_desc *d)
{
    if (!d || d->num_views <= 0 || d->num_frames <= 0 || d->channels <= 0 || d->feat_h <= 0 ||
        d->feat_w <= 0)
        return 0;
    return (((size_t)d->num_views * d->num_frames * d->channels * d->feat_h * d->feat_w * sizeof(float)) +
            255) & ~(size_t)255;
}

extern "C" DFM_API int dfm_point_sample_mv_bwd(const dfm_mv_desc *d, const void *grad_out,
                                               const float *points, const float *proj,
                                               const float *ori_w, float *grad_feats,
                                               void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d) return fail_ps(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return fail_ps(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!grad_out || !points || !proj || !ori_w || !grad_feats)
        return fail_ps(DFM_ERR_INVALID_ARG, "NULL device pointer");
    MvGeom g;
    g.num_views = d->num_views; g.num_frames = d->num_frames; g.C = d->channels;
    g.Hf = d->feat_h; g.Wf = d->feat_w; g.nblk = 0;
    g.nx = d->nx; g.ny = d->ny; g.nz = d->nz; g.N = d->num_points;
    g.scale_x = d->scale_x; g.scale_y = d->scale_y; g.crop_x = d->crop_x; g.crop_y = d->crop_y;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.flip = d->flip; g.mode = d->mode;
    g.aggregate = d->aggregate; g.valid_sample = d->valid_sample;
    g.out_cl = 0;  // the backward reads an (C, N) gradient volume
    hipStream_t st = (hipStream_t)stream;
    const int nvf = d->num_views * d->num_frames;
    const int c_out = d->channels * (d->aggregate ? d->num_frames : 1);
    const size_t lds = (size_t)c_out * (MV_VT + 1) * sizeof(float) + (size_t)MV_VT * nvf * sizeof(MvFoot);
    if (workspace && lds <= 64 * 1024) {
        // pixel-major accumulation (see mv_sample_bwd_pm_kernel)
        const size_t need = dfm_point_sample_mv_bwd_workspace_bytes(d);
        if (workspace_bytes < need)
            return fail_ps(DFM_ERR_WORKSPACE, "workspace smaller than dfm_point_sample_mv_bwd_workspace_bytes");
        hipError_t e = hipMemsetAsync(workspace, 0, need, st);
        if (e != hipSuccess) return fail_ps(DFM_ERR_HIP, hipGetErrorString(e));
        const long long nb = (g.N + MV_VT - 1) / MV_VT;
        if (d->dtype == DFM_F32)
            hipLaunchKernelGGL(mv_sample_bwd_pm_kernel<float>, dim3((unsigned)nb), dim3(256), lds, st, g,
                               (const float *)grad_out, points, proj, ori_w, (float *)workspace);
        else
            hipLaunchKernelGGL(mv_sample_bwd_pm_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), lds, st,
                               g, (const bf16_t *)grad_out, points, proj, ori_w, (float *)workspace);
        const long long HW = (long long)d->feat_h * d->feat_w;
        dim3 tg((unsigned)((HW + 63) / 64), (d->channels + 31) / 32, nvf);
        hipLaunchKernelGGL(add_from_pixel_major_kernel<float>, tg, dim3(256), 0, st,
                           (const float *)workspace, grad_feats, d->channels, HW);
        e = hipGetLastError();
        if (e != hipSuccess) return fail_ps(DFM_ERR_HIP, hipGetErrorString(e));
        return DFM_OK;
    }
    const long long nb = (g.N + 255) / 256;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL(mv_sample_bwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, g,
                           (const float *)grad_out, points, proj, ori_w, grad_feats);
    else
        hipLaunchKernelGGL(mv_sample_bwd_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, g,
                           (const bf16_t *)grad_out, points, proj, ori_w, grad_feats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_ps(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

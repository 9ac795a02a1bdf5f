// frustum_to_voxel.hip -- sampling stage of FrustumToVoxel.forward (gfx950)
//
// Reference: mmdet3d/models/necks/feature_transformation.py:82-158 -- per voxel
// project with cam2img[:3], normalise (u, v, depth), three 3-D grid_samples
// (cost volume, depth distribution, 2-D semantic feature at z := 0), validity
// masks, depth-probability weighting, channel concat.  One launch writes the
// concatenated (B, C+Cs, Nz, Ny, Nx) volume: no coordinate tensors, no three
// separate sampled volumes, no mask multiplies, no cat.
//
// Arithmetic = ATen grid_sampler_3d_cpu_impl (scalar path): corner weights
// (x1-ix)*(y1-iy)*(z1-iz) ..., out = 0; out += v*w over the in-bounds corners in
// the order tnw,tne,tsw,tse,bnw,bne,bsw,bse, multiply and add unfused.
//
// Layout: caller tensors are NCDHW / NCHW.  stereo_feat and cur_sem_feats are first
// re-laid pixel-major into the workspace ([d][h][w][C], [h][w][Cs]: a corner is one
// contiguous run of C*sizeof(T) bytes read with 16-byte loads -- in NCDHW the C scalar
// loads of a corner sit a whole volume apart and neighbouring voxels along x walk
// through depth planes, so every load touched its own cache line); the 1-channel
// depth distribution is sampled where it lies.  Shapes whose channel counts are not a
// whole number of 16-byte blocks take the scalar kernel on the caller's layout.
// Bound: HBM write of the volume + one read of the sources.
#include <type_traits>

#include "dfm_common.h"

// A lane (= one voxel) writes its channels-last row as 16-byte pieces; the lanes of a store instruction
// are 128-256 B apart, so one instruction touches 64 partial lines that the row's other stores complete.
// PLAIN stores let the L2 merge them: the non-temporal form pushed partial lines out and measured 1.9x
// slower here (f2v_cl 1.93 -> 3.59 ms, profiles/archive/r04_c7_lift_nt_vs_plain.txt) -- while the batched
// multi-view kernel (point_sample.hip), whose lanes write whole contiguous KiBs, gains 6 % from nt.
#define DFM_LIFT_PLAIN 1
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


using namespace dfm;

namespace {

struct F2vGeom {
    int32_t C, D, H, W, Ds, Hs, Ws, Cs, Hsem, Wsem, Nz, Ny, Nx;
    float pad_h, pad_w, depth_min, depth_span;
    int32_t out_cl;      // out stored (B, Nz, Ny, Nx, C + Cs): torch channels_last_3d
    int32_t cd, ch, cw;  // fused depth head: size of the low-resolution cost volume (Ds = scale * cd ...)
    int32_t st_att;      // stereo_atten_feat: Voxel *= pred_disp      (feature_transformation.py:141-142)
    int32_t sem_att;     // sem_atten_feat:    Voxel_2D *= pred_disp   (feature_transformation.py:154-155)
};

// Fused DepthHead: the depth distribution the reference samples,
//   softmax_d(Upsample_x4(cost))          dense_heads/depth_head.py:205-207
// is evaluated at the (up to) 8 lattice corners of the voxel directly from the low-resolution
// cost volume and the per-column softmax statistics (col_max, col_sum from
// dfm_depth_head_stats_fwd), with the arithmetic of depth_head_kernel -- the value at a corner is
// bit for bit what that kernel would have stored -- instead of reading a materialised
// (B, 1, 4D, 4H, 4W) tensor (472 MB per sample at config K, written once and read once).
struct FusedHead {
    const void *cost;      // (B, 1, cd, ch, cw), T
    const float *col_max;  // (B, Hs, Ws)
    const float *col_sum;
};

template <typename T>
__device__ __forceinline__ float fused_disp(const F2vGeom &g, const T *__restrict__ cost,
                                            const float *__restrict__ cmax,
                                            const float *__restrict__ csum, float gx, float gy, float gz)
{
    const int D = g.Ds, H = g.Hs, W = g.Ws;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
    const bool fin = fabsf(ix) <= 1.0e9f && fabsf(iy) <= 1.0e9f && fabsf(iz) <= 1.0e9f;
    float wgt[8];
    wgt[0] = (x1 - ix) * (y1 - iy) * (z1 - iz);
    wgt[1] = (ix - x0) * (y1 - iy) * (z1 - iz);
    wgt[2] = (x1 - ix) * (iy - y0) * (z1 - iz);
    wgt[3] = (ix - x0) * (iy - y0) * (z1 - iz);
    wgt[4] = (x1 - ix) * (y1 - iy) * (iz - z0);
    wgt[5] = (ix - x0) * (y1 - iy) * (iz - z0);
    wgt[6] = (x1 - ix) * (iy - y0) * (iz - z0);
    wgt[7] = (ix - x0) * (iy - y0) * (iz - z0);
    float out = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xf = (k & 1) ? x1 : x0, yf = (k & 2) ? y1 : y0, zf = (k & 4) ? z1 : z0;
        const bool ok = fin && xf >= 0.0f && xf <= (float)(W - 1) && yf >= 0.0f && yf <= (float)(H - 1) &&
                        zf >= 0.0f && zf <= (float)(D - 1);
        if (!ok) continue;
        const int xc = (int)xf, yc = (int)yf, zc = (int)zf;
        const UpIdx uw = up_index(xc, g.cw, W), uh = up_index(yc, g.ch, H), ud = up_index(zc, g.cd, D);
        const int r0 = uh.i0 * g.cw, r1 = uh.i1 * g.cw;
        float col[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const T *p = cost + (size_t)(e ? ud.i1 : ud.i0) * g.ch * g.cw;
            const float a = lerp_fma(uw.w0, elem<T>::load(p[r0 + uw.i0]), uw.w1, elem<T>::load(p[r0 + uw.i1]));
            const float b = lerp_fma(uw.w0, elem<T>::load(p[r1 + uw.i0]), uw.w1, elem<T>::load(p[r1 + uw.i1]));
            col[e] = lerp_fma(uh.w0, a, uh.w1, b);
        }
        // depth_volumes and its softmax are stored (and read back) in T by the unfused pipeline
        const float logit = elem<T>::load(elem<T>::store(lerp_fma(ud.w0, col[0], ud.w1, col[1])));
        const size_t pix = (size_t)yc * W + xc;
        const float prob = elem<T>::load(elem<T>::store(exp_nonpos(logit - cmax[pix]) * (1.0f / csum[pix])));
        out = out + prob * wgt[k];
    }
    return out;
}

struct Tri {
    int o[8];     // element offsets of the 8 corners (valid only where ok bit set)
    float w[8];   // corner weights, ATen order
    uint32_t ok;
};

__device__ __forceinline__ Tri make_tri(float gx, float gy, float gz, int D, int H, int W)
{
    Tri t;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
    t.w[0] = (x1 - ix) * (y1 - iy) * (z1 - iz);
    t.w[1] = (ix - x0) * (y1 - iy) * (z1 - iz);
    t.w[2] = (x1 - ix) * (iy - y0) * (z1 - iz);
    t.w[3] = (ix - x0) * (iy - y0) * (z1 - iz);
    t.w[4] = (x1 - ix) * (y1 - iy) * (iz - z0);
    t.w[5] = (ix - x0) * (y1 - iy) * (iz - z0);
    t.w[6] = (x1 - ix) * (iy - y0) * (iz - z0);
    t.w[7] = (ix - x0) * (iy - y0) * (iz - z0);
    const bool fin = fabsf(ix) <= 1.0e9f && fabsf(iy) <= 1.0e9f && fabsf(iz) <= 1.0e9f;  // no NaN/Inf
    const bool bx0 = fin && x0 >= 0.0f && x0 <= (float)(W - 1), bx1 = fin && x1 >= 0.0f && x1 <= (float)(W - 1);
    const bool by0 = fin && y0 >= 0.0f && y0 <= (float)(H - 1), by1 = fin && y1 >= 0.0f && y1 <= (float)(H - 1);
    const bool bz0 = fin && z0 >= 0.0f && z0 <= (float)(D - 1), bz1 = fin && z1 >= 0.0f && z1 <= (float)(D - 1);
    const int xi = bx0 ? (int)x0 : 0, yi = by0 ? (int)y0 : 0, zi = bz0 ? (int)z0 : 0;
    const int xj = bx1 ? (int)x1 : 0, yj = by1 ? (int)y1 : 0, zj = bz1 ? (int)z1 : 0;
    t.o[0] = (zi * H + yi) * W + xi; t.o[1] = (zi * H + yi) * W + xj;
    t.o[2] = (zi * H + yj) * W + xi; t.o[3] = (zi * H + yj) * W + xj;
    t.o[4] = (zj * H + yi) * W + xi; t.o[5] = (zj * H + yi) * W + xj;
    t.o[6] = (zj * H + yj) * W + xi; t.o[7] = (zj * H + yj) * W + xj;
    t.ok = (uint32_t)(bz0 && by0 && bx0) | ((uint32_t)(bz0 && by0 && bx1) << 1) |
           ((uint32_t)(bz0 && by1 && bx0) << 2) | ((uint32_t)(bz0 && by1 && bx1) << 3) |
           ((uint32_t)(bz1 && by0 && bx0) << 4) | ((uint32_t)(bz1 && by0 && bx1) << 5) |
           ((uint32_t)(bz1 && by1 && bx0) << 6) | ((uint32_t)(bz1 && by1 && bx1) << 7);
    return t;
}

template <typename T>
__device__ __forceinline__ float tri_sample(const Tri &t, const T *__restrict__ vol)
{
    float out = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (t.ok & (1u << k)) out = out + elem<T>::load(vol[t.o[k]]) * t.w[k];
    return out;
}

template <typename T>
__global__ __launch_bounds__(256) void f2v_kernel(F2vGeom g, const T *__restrict__ stereo,
                                                  const T *__restrict__ soft,
                                                  const T *__restrict__ sem,
                                                  const float *__restrict__ coords,
                                                  const float *__restrict__ cam2img,
                                                  T *__restrict__ out, FusedHead fh)
{
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
    const float *P = cam2img + 16 * b;  // rows 0..2 of the 4x4 == cam2img[:3]
    const float a = dot4_chain(-ys, -zs, xs, 1.0f, P + 0);
    const float bb = dot4_chain(-ys, -zs, xs, 1.0f, P + 4);
    const float c = dot4_chain(-ys, -zs, xs, 1.0f, P + 8);
    const float u = a / c, v = bb / c;
    const bool valid2d = (u >= 0.0f) && (u <= g.pad_w) && (v >= 0.0f) && (v <= g.pad_h);
    float gx = (u - 0.0f) / (g.pad_w - 1.0f), gy = (v - 0.0f) / (g.pad_h - 1.0f);
    float gz = (xs - g.depth_min) / g.depth_span;
    gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
    const float valid = (valid2d && gz >= -1.0f && gz <= 1.0f) ? 1.0f : 0.0f;

    // pred_disp = grid_sample(stereo_feat_softmax) * valids, wanted when either attention is on
    // (feature_transformation.py:133-139)
    float disp = 1.0f;
    if (g.st_att || (g.Cs > 0 && g.sem_att)) {
        if (fh.cost) {
            // nothing to evaluate outside the frustum (the unfused product is +0 there as well)
            disp = valid != 0.0f
                       ? fused_disp<T>(g, (const T *)fh.cost + (size_t)b * g.cd * g.ch * g.cw,
                                       fh.col_max + (size_t)b * g.Hs * g.Ws,
                                       fh.col_sum + (size_t)b * g.Hs * g.Ws, gx, gy, gz) * valid
                       : 0.0f;
        } else {
            const Tri ts = make_tri(gx, gy, gz, g.Ds, g.Hs, g.Ws);
            disp = tri_sample<T>(ts, soft + (size_t)b * g.Ds * g.Hs * g.Ws) * valid;
        }
    }
    const float sdisp = g.st_att ? disp : 1.0f;   // x * 1.0f is exact: one code path
    const float mdisp = g.sem_att ? disp : 1.0f;
    const size_t vol = (size_t)g.D * g.H * g.W;
    T *o = out + (size_t)b * (g.C + g.Cs) * N + i;
    {
        const Tri t = make_tri(gx, gy, gz, g.D, g.H, g.W);
        const T *sv = stereo + (size_t)b * g.C * vol;
        for (int ch = 0; ch < g.C; ++ch)
            o[(size_t)ch * N] = elem<T>::store(tri_sample<T>(t, sv + ch * vol) * valid * sdisp);
    }
    if (g.Cs > 0) {
        const Tri t2 = make_tri(gx, gy, 0.0f, 1, g.Hsem, g.Wsem);
        const float v2d = valid2d ? 1.0f : 0.0f;
        const size_t plane = (size_t)g.Hsem * g.Wsem;
        const T *sp = sem + (size_t)b * g.Cs * plane;
        for (int ch = 0; ch < g.Cs; ++ch) {
            float s = tri_sample<T>(t2, sp + ch * plane);
            s = s * v2d;
            o[(size_t)(g.C + ch) * N] = elem<T>::store(s * mdisp);
        }
    }
}

// pixel-major sources: stereo_pm [b][d*h*w][C], sem_pm [b][hsem*wsem][Cs] (16-byte blocks)
// 32 channels per pass; per channel the corners are added in ATen's order.
template <typename T>
__global__ __launch_bounds__(256) void f2v_pm_kernel(F2vGeom g, const uint4 *__restrict__ stereo_pm,
                                                     const T *__restrict__ soft,
                                                     const uint4 *__restrict__ sem_pm,
                                                     const float *__restrict__ coords,
                                                     const float *__restrict__ cam2img,
                                                     T *__restrict__ out, FusedHead fh)
{
    constexpr int CB = elem<T>::CB;
    constexpr int NB = 32 / CB;
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
    const float *P = cam2img + 16 * b;  // rows 0..2 of the 4x4 == cam2img[:3]
    const float a = dot4_chain(-ys, -zs, xs, 1.0f, P + 0);
    const float bb = dot4_chain(-ys, -zs, xs, 1.0f, P + 4);
    const float c = dot4_chain(-ys, -zs, xs, 1.0f, P + 8);
    const float u = a / c, v = bb / c;
    const bool valid2d = (u >= 0.0f) && (u <= g.pad_w) && (v >= 0.0f) && (v <= g.pad_h);
    float gx = (u - 0.0f) / (g.pad_w - 1.0f), gy = (v - 0.0f) / (g.pad_h - 1.0f);
    float gz = (xs - g.depth_min) / g.depth_span;
    gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
    const float valid = (valid2d && gz >= -1.0f && gz <= 1.0f) ? 1.0f : 0.0f;

    // pred_disp = grid_sample(stereo_feat_softmax) * valids, wanted when either attention is on
    // (feature_transformation.py:133-139)
    float disp = 1.0f;
    if (g.st_att || (g.Cs > 0 && g.sem_att)) {
        if (fh.cost) {
            // nothing to evaluate outside the frustum (the unfused product is +0 there as well)
            disp = valid != 0.0f
                       ? fused_disp<T>(g, (const T *)fh.cost + (size_t)b * g.cd * g.ch * g.cw,
                                       fh.col_max + (size_t)b * g.Hs * g.Ws,
                                       fh.col_sum + (size_t)b * g.Hs * g.Ws, gx, gy, gz) * valid
                       : 0.0f;
        } else {
            const Tri ts = make_tri(gx, gy, gz, g.Ds, g.Hs, g.Ws);
            disp = tri_sample<T>(ts, soft + (size_t)b * g.Ds * g.Hs * g.Ws) * valid;
        }
    }
    const float sdisp = g.st_att ? disp : 1.0f;   // x * 1.0f is exact: one code path
    const float mdisp = g.sem_att ? disp : 1.0f;
    T *o = out + (size_t)b * (g.C + g.Cs) * N + i;
    T *ocl = out + ((size_t)b * N + i) * (g.C + g.Cs);  // channels-last: this voxel's C + Cs values
    {
        const Tri t = make_tri(gx, gy, gz, g.D, g.H, g.W);
        const int nblk = g.C / CB;
        const uint4 *sv = stereo_pm + (size_t)b * g.D * g.H * g.W * nblk;
        for (int blk0 = 0; blk0 < nblk; blk0 += NB) {
            float acc[NB][CB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int k = 0; k < CB; ++k) acc[j][k] = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!(t.ok & (1u << k))) continue;
                const uint4 *p = sv + (size_t)t.o[k] * nblk + blk0;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (blk0 + j < nblk) {
                        float r[CB];
                        unpack16(p[j], r);
#pragma unroll
                        for (int e = 0; e < CB; ++e) acc[j][e] = acc[j][e] + r[e] * t.w[k];
                    }
                }
            }
            if (g.out_cl) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (blk0 + j < nblk) {
                        float r[CB];
#pragma unroll
                        for (int e = 0; e < CB; ++e) r[e] = acc[j][e] * valid * sdisp;
                        lift_store16<T>(ocl + (size_t)(blk0 + j) * CB, r);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < CB; ++e)
                        if (blk0 + j < nblk)
                            o[(size_t)((blk0 + j) * CB + e) * N] = elem<T>::store(acc[j][e] * valid * sdisp);
            }
        }
    }
    if (g.Cs > 0) {
        const Tri t2 = make_tri(gx, gy, 0.0f, 1, g.Hsem, g.Wsem);
        const float v2d = valid2d ? 1.0f : 0.0f;
        const int nblk = g.Cs / CB;
        const uint4 *sp = sem_pm + (size_t)b * g.Hsem * g.Wsem * nblk;
        for (int blk0 = 0; blk0 < nblk; blk0 += NB) {
            float acc[NB][CB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int k = 0; k < CB; ++k) acc[j][k] = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // D == 1: the z1 corners are never in bounds
                if (!(t2.ok & (1u << k))) continue;
                const uint4 *p = sp + (size_t)t2.o[k] * nblk + blk0;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (blk0 + j < nblk) {
                        float r[CB];
                        unpack16(p[j], r);
#pragma unroll
                        for (int e = 0; e < CB; ++e) acc[j][e] = acc[j][e] + r[e] * t2.w[k];
                    }
                }
            }
            if (g.out_cl) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (blk0 + j < nblk) {
                        float r[CB];
#pragma unroll
                        for (int e = 0; e < CB; ++e) {
                            const float sval = acc[j][e] * v2d;
                            r[e] = sval * mdisp;
                        }
                        lift_store16<T>(ocl + g.C + (size_t)(blk0 + j) * CB, r);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < CB; ++e)
                        if (blk0 + j < nblk) {
                            float sval = acc[j][e] * v2d;
                            o[(size_t)(g.C + (blk0 + j) * CB + e) * N] = elem<T>::store(sval * mdisp);
                        }
            }
        }
    }
}

// Channels-last output, several lanes per voxel.  f2v_pm_kernel is one lane = one voxel: a tap is 64-128
// contiguous bytes per LANE, so the 64 lanes of every load touch 64 different cache lines, and a voxel's
// channels-last row leaves as 16-byte pieces 128-256 bytes apart.  Here the geometry (corner offsets, weights,
// masks, the depth probability) is computed lane = voxel as before and parked in LDS (128 bytes per voxel);
// then each source in turn (stereo feature: 8 corners, semantic feature: 4) is gathered with lane = (voxel,
// 16-byte channel block of the source): a tap is ONE coalesced 16-byte load per lane -- the 4 or 8 lanes of a
// voxel cover the corner's contiguous channels --, two voxels per lane are in flight (16 / 8 independent taps),
// and a voxel's half of the row leaves as whole 64-byte runs of neighbouring voxels (non-temporal: nothing
// re-reads it here).  Same arithmetic, corner by corner in ATen's order.  f2v_cl (config K, bf16): 1.93 ->
// 1.35 ms (profiles/archive/r04_c51_*); a first version with both sources in one pass (lanes 0-3 stereo, 4-7
// semantic: divergent halves, one voxel per lane in flight) measured 2.57 ms (r04_c50).  PLANAR: the
// reference layout (B, C + Cs, Nz, Ny, Nx) from the same gather (gather_planar below): f2v fp32 3.86 ->
// 2.93 ms (r04_c59).
template <typename T, bool PLANAR>
__global__ __launch_bounds__(256) void f2v_pm8_kernel(F2vGeom g, const uint4 *__restrict__ stereo_pm,
                                                      const T *__restrict__ soft, const uint4 *__restrict__ sem_pm,
                                                      const float *__restrict__ coords,
                                                      const float *__restrict__ cam2img, T *__restrict__ out,
                                                      FusedHead fh)
{
    constexpr int CB = elem<T>::CB;
    struct Rec {
        int o[8];
        float w[8];
        int o2[4];
        float w2[4];
        float valid, sdisp, v2d, mdisp;
        uint32_t ok, ok2, pad0, pad1;
    };
    static_assert(sizeof(Rec) == 128, "one record = 128 bytes");
    __shared__ __attribute__((aligned(16))) Rec rec[256];
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const long long i0 = (long long)blockIdx.x * 256;
    const int b = blockIdx.y;
    {
        const long long i = min(i0 + threadIdx.x, N - 1);
        const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
        const float *P = cam2img + 16 * b;
        const float a = dot4_chain(-ys, -zs, xs, 1.0f, P + 0);
        const float bb = dot4_chain(-ys, -zs, xs, 1.0f, P + 4);
        const float c = dot4_chain(-ys, -zs, xs, 1.0f, P + 8);
        const float u = a / c, v = bb / c;
        const bool valid2d = (u >= 0.0f) && (u <= g.pad_w) && (v >= 0.0f) && (v <= g.pad_h);
        float gx = (u - 0.0f) / (g.pad_w - 1.0f), gy = (v - 0.0f) / (g.pad_h - 1.0f);
        float gz = (xs - g.depth_min) / g.depth_span;
        gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
        const float valid = (valid2d && gz >= -1.0f && gz <= 1.0f) ? 1.0f : 0.0f;
        float disp = 1.0f;
        if (g.st_att || (g.Cs > 0 && g.sem_att)) {
            if (fh.cost) {
                disp = valid != 0.0f
                           ? fused_disp<T>(g, (const T *)fh.cost + (size_t)b * g.cd * g.ch * g.cw,
                                           fh.col_max + (size_t)b * g.Hs * g.Ws,
                                           fh.col_sum + (size_t)b * g.Hs * g.Ws, gx, gy, gz) * valid
                           : 0.0f;
            } else {
                const Tri ts = make_tri(gx, gy, gz, g.Ds, g.Hs, g.Ws);
                disp = tri_sample<T>(ts, soft + (size_t)b * g.Ds * g.Hs * g.Ws) * valid;
            }
        }
        Rec r;
        const Tri t = make_tri(gx, gy, gz, g.D, g.H, g.W);
#pragma unroll
        for (int k = 0; k < 8; ++k) { r.o[k] = t.o[k]; r.w[k] = t.w[k]; }
        r.ok = t.ok;
        r.ok2 = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r.o2[k] = 0; r.w2[k] = 0.0f; }
        if (g.Cs > 0) {
            const Tri t2 = make_tri(gx, gy, 0.0f, 1, g.Hsem, g.Wsem);
#pragma unroll
            for (int k = 0; k < 4; ++k) { r.o2[k] = t2.o[k]; r.w2[k] = t2.w[k]; }
            r.ok2 = t2.ok & 15u;  // D == 1: the z1 corners are never in bounds
        }
        r.valid = valid;
        r.sdisp = g.st_att ? disp : 1.0f;  // x * 1.0f is exact: one code path
        r.v2d = valid2d ? 1.0f : 0.0f;
        r.mdisp = g.sem_att ? disp : 1.0f;
        r.pad0 = r.pad1 = 0u;
        rec[threadIdx.x] = r;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nbs = g.C / CB, nbm = g.Cs / CB, nbt = nbs + nbm;
    const uint4 *sv = stereo_pm + (size_t)b * g.D * g.H * g.W * nbs;
    const uint4 *sp = sem_pm + (size_t)b * g.Hsem * g.Wsem * nbm;
    typedef uint32_t f2v_u32x4 __attribute__((ext_vector_type(4)));
    // One source at a time (all lanes run the same corners), L lanes per voxel (L = 4 or 8 blocks of the
    // source per trip), TWO voxels per lane in flight: 16 (stereo) / 8 (semantic) independent taps per lane.
    auto gather = [&](auto stereo_c, const uint4 *src, int nb, int bi0, int L) {
        constexpr bool ST = decltype(stereo_c)::value;
        constexpr int NK = ST ? 8 : 4;
        const int sh = L == 4 ? 2 : 3, vpi = 64 >> sh;  // voxels per trip
        const int j = lane & (L - 1), vp = lane >> sh;
        for (int bb = j; bb < nb; bb += L) {
            for (int v0 = 0; v0 < 64; v0 += 2 * vpi) {
                float acc[2][CB];
                int qq[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    qq[u] = wave * 64 + v0 + u * vpi + vp;
#pragma unroll
                    for (int e = 0; e < CB; ++e) acc[u][e] = 0.0f;
                }
                uint4 tapv[2][NK];
                uint32_t okv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const Rec &r = rec[qq[u]];
                    okv[u] = ST ? r.ok : r.ok2;
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        // (an out-of-bounds corner's offset is 0: addressable, never used)
                        const int o = ST ? r.o[k] : r.o2[k];
                        tapv[u][k] = src[(size_t)o * nb + bb];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const Rec &r = rec[qq[u]];
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        if (!(okv[u] & (1u << k))) continue;
                        float v[CB];
                        unpack16(tapv[u][k], v);
                        const float w = ST ? r.w[k] : r.w2[k];
#pragma unroll
                        for (int e = 0; e < CB; ++e) acc[u][e] = acc[u][e] + v[e] * w;
                    }
                    float res[CB];
                    if constexpr (ST) {
                        const float va = r.valid, sd = r.sdisp;
#pragma unroll
                        for (int e = 0; e < CB; ++e) res[e] = acc[u][e] * va * sd;
                    } else {
                        const float v2 = r.v2d, md = r.mdisp;
#pragma unroll
                        for (int e = 0; e < CB; ++e) {
                            const float sval = acc[u][e] * v2;
                            res[e] = sval * md;
                        }
                    }
                    const long long i = i0 + qq[u];
                    if (i < N) {
                        f2v_u32x4 pk;
                        if constexpr (sizeof(T) == 4) {
                            pk = f2v_u32x4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]),
                                           __float_as_uint(res[3])};
                        } else {
                            pk = f2v_u32x4{dfm::pack_bf16x2(res[0], res[1]), dfm::pack_bf16x2(res[2], res[3]),
                                           dfm::pack_bf16x2(res[4], res[5]), dfm::pack_bf16x2(res[6], res[7])};
                        }
                        T *dst = out + ((size_t)b * N + i) * (g.C + g.Cs) + (size_t)(bi0 + bb) * CB;
                        __builtin_nontemporal_store(pk, (f2v_u32x4 *)dst);
                    }
                }
            }
        }
    };
    // Planar output (B, C + Cs, Nz, Ny, Nx), the reference layout: the same gather, but a lane keeps ONE channel
    // block for FOUR CONSECUTIVE voxels (four trips, two voxels in flight) and stores, per channel, one vector
    // of those four voxels -- the lanes of a store instruction that share a channel are 8 (fp32) / 16 (bf16)
    // neighbouring voxel groups: whole 128-byte lines per channel plane, no transpose through LDS.
    auto gather_planar = [&](auto stereo_c, const uint4 *src, int nb, int ch0, int L) {
        constexpr bool ST = decltype(stereo_c)::value;
        constexpr int NK = ST ? 8 : 4;
        const int sh = L == 4 ? 2 : 3, ngr = 64 >> sh;  // voxel groups per round
        const int j = lane & (L - 1), vg = lane >> sh;
        for (int bb = j; bb < nb; bb += L) {
            for (int v0 = 0; v0 < 64; v0 += 4 * ngr) {
                const int qb = wave * 64 + v0 + 4 * vg;  // this lane's four voxels: qb .. qb + 3
                float res[4][CB];
#pragma unroll
                for (int t0 = 0; t0 < 4; t0 += 2) {
                    uint4 tapv[2][NK];
                    uint32_t okv[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const Rec &r = rec[qb + t0 + u];
                        okv[u] = ST ? r.ok : r.ok2;
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
                            const int o = ST ? r.o[k] : r.o2[k];
                            tapv[u][k] = src[(size_t)o * nb + bb];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const Rec &r = rec[qb + t0 + u];
                        float acc[CB];
#pragma unroll
                        for (int e = 0; e < CB; ++e) acc[e] = 0.0f;
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
                            if (!(okv[u] & (1u << k))) continue;
                            float v[CB];
                            unpack16(tapv[u][k], v);
                            const float w = ST ? r.w[k] : r.w2[k];
#pragma unroll
                            for (int e = 0; e < CB; ++e) acc[e] = acc[e] + v[e] * w;
                        }
                        if constexpr (ST) {
                            const float va = r.valid, sd = r.sdisp;
#pragma unroll
                            for (int e = 0; e < CB; ++e) res[t0 + u][e] = acc[e] * va * sd;
                        } else {
                            const float v2 = r.v2d, md = r.mdisp;
#pragma unroll
                            for (int e = 0; e < CB; ++e) {
                                const float sval = acc[e] * v2;
                                res[t0 + u][e] = sval * md;
                            }
                        }
                    }
                }
                const long long i = i0 + qb;
                if (i < N) {  // (N % 4 == 0: a group of four is inside or outside as a whole)
                    T *dst = out + ((size_t)b * (g.C + g.Cs) + ch0 + (size_t)bb * CB) * N + i;
#pragma unroll
                    for (int e = 0; e < CB; ++e) {
                        if constexpr (sizeof(T) == 4) {
                            const f2v_u32x4 pk = {__float_as_uint(res[0][e]), __float_as_uint(res[1][e]),
                                                  __float_as_uint(res[2][e]), __float_as_uint(res[3][e])};
                            __builtin_nontemporal_store(pk, (f2v_u32x4 *)(dst + (size_t)e * N));
                        } else {
                            typedef uint32_t f2v_u32x2 __attribute__((ext_vector_type(2)));
                            const f2v_u32x2 pk = {dfm::pack_bf16x2(res[0][e], res[1][e]), dfm::pack_bf16x2(res[2][e], res[3][e])};
                            __builtin_nontemporal_store(pk, (f2v_u32x2 *)(dst + (size_t)e * N));
                        }
                    }
                }
            }
        }
    };
    if constexpr (!PLANAR) {
        gather(std::true_type{}, sv, nbs, 0, nbs % 8 == 0 ? 8 : 4);
        if (nbm > 0) gather(std::false_type{}, sp, nbm, nbs, nbm % 8 == 0 ? 8 : 4);
    } else {
        gather_planar(std::true_type{}, sv, nbs, 0, nbs % 8 == 0 ? 8 : 4);
        if (nbm > 0) gather_planar(std::false_type{}, sp, nbm, g.C, nbm % 8 == 0 ? 8 : 4);
    }
    (void)nbt;
}

}  // namespace

extern "C" {

// pixel-major staging is used when both channel counts are whole 16-byte blocks
static bool f2v_pixel_major(const dfm_f2v_desc *d)
{
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    return d->channels % CB == 0 && d->sem_channels % CB == 0;
}

// the several-lanes-per-voxel kernel: 4 or 8 sixteen-byte blocks of a source per trip
static bool f2v_lanes_per_voxel(const dfm_f2v_desc *d)
{
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    const int nbs = d->channels / CB, nbm = d->sem_channels / CB;
    return nbs % 4 == 0 && nbm % 4 == 0;
}

DFM_API size_t dfm_frustum_to_voxel_workspace_bytes(const dfm_f2v_desc *d)
{
    if (!d || d->batch <= 0 || d->channels <= 0 || d->d <= 0 || d->h <= 0 || d->w <= 0 ||
        d->sem_channels < 0 || (d->dtype != DFM_F32 && d->dtype != DFM_BF16))
        return 0;
    if (!f2v_pixel_major(d)) return 256;
    const size_t esz = d->dtype == DFM_BF16 ? 2 : 4;
    const size_t a = (size_t)d->batch * d->channels * d->d * d->h * d->w * esz;
    const size_t b = (size_t)d->batch * d->sem_channels * d->hsem * d->wsem * esz;
    return ((a + 255) & ~(size_t)255) + ((b + 255) & ~(size_t)255) + 256;
}

static int f2v_fwd_impl(const dfm_f2v_desc *d, const void *stereo, const void *softmax, FusedHead fh,
                        int32_t head_scale, const void *sem, const float *coords, const float *cam2img,
                        void *out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->batch <= 0 || d->channels <= 0 || d->d <= 0 || d->h <= 0 || d->w <= 0 || d->nz <= 0 ||
        d->ny <= 0 || d->nx <= 0 || d->sem_channels < 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_f2v_desc");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    const bool want_disp = d->stereo_atten || (d->sem_channels > 0 && !d->no_sem_atten);
    if (!stereo || !coords || !cam2img || !out || (d->sem_channels > 0 && !sem) || (want_disp && !softmax && !fh.cost))
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if ((long long)d->ds * d->hs * d->ws >= (1ll << 31) || (long long)d->d * d->h * d->w >= (1ll << 31))
        return set_error(DFM_ERR_UNSUPPORTED, "volume too large for 32-bit corner offsets");
    if (d->batch > 65535) return set_error(DFM_ERR_UNSUPPORTED, "batch > 65535");
    F2vGeom g;
    g.C = d->channels; g.D = d->d; g.H = d->h; g.W = d->w;
    g.Ds = d->ds; g.Hs = d->hs; g.Ws = d->ws;
    g.Cs = d->sem_channels; g.Hsem = d->hsem; g.Wsem = d->wsem;
    g.Nz = d->nz; g.Ny = d->ny; g.Nx = d->nx;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.depth_min = d->depth_min; g.depth_span = d->depth_span;
    g.cd = g.ch = g.cw = 0;
    g.st_att = d->stereo_atten ? 1 : 0;
    g.sem_att = d->no_sem_atten ? 0 : 1;
    g.out_cl = d->out_channels_last ? 1 : 0;
    if (g.out_cl && !f2v_pixel_major(d))
        return set_error(DFM_ERR_UNSUPPORTED, "channels-last output needs channel counts of whole 16-byte blocks");
    if (fh.cost) {
        if (head_scale <= 0 || d->ds % head_scale || d->hs % head_scale || d->ws % head_scale)
            return set_error(DFM_ERR_INVALID_ARG, "ds, hs, ws must be multiples of the depth head's scale");
        g.cd = d->ds / head_scale; g.ch = d->hs / head_scale; g.cw = d->ws / head_scale;
    }
    const long long N = (long long)d->nz * d->ny * d->nx;
    dim3 grid((unsigned)((N + 255) / 256), d->batch);
    hipStream_t st = (hipStream_t)stream;
    if (d->sem_channels_last && !f2v_pixel_major(d))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "channels-last cur_sem_feats needs channel counts of whole 16-byte blocks");
    if (d->stereo_channels_last && (!f2v_pixel_major(d) || ((uintptr_t)stereo & 15)))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "channels-last stereo_feat needs channel counts of whole 16-byte blocks");
    if (f2v_pixel_major(d)) {
        if (!workspace || workspace_bytes < dfm_frustum_to_voxel_workspace_bytes(d))
            return set_error(DFM_ERR_WORKSPACE,
                             "workspace smaller than dfm_frustum_to_voxel_workspace_bytes");
        const size_t esz = d->dtype == DFM_BF16 ? 2 : 4;
        const long long vox = (long long)d->d * d->h * d->w, pix = (long long)d->hsem * d->wsem;
        const size_t a = ((size_t)d->batch * d->channels * vox * esz + 255) & ~(size_t)255;
        const bool in_place = d->stereo_channels_last != 0;
        void *stereo_pm = in_place ? const_cast<void *>(stereo) : workspace;
        // a channels-last (NHWC) semantic map is the pixel-major layout already
        const bool sem_in_place = d->sem_channels_last != 0 && d->sem_channels > 0;
        if (sem_in_place && ((uintptr_t)sem & 15))
            return set_error(DFM_ERR_INVALID_ARG, "channels-last cur_sem_feats must be 16-byte aligned");
        void *sem_pm = sem_in_place ? const_cast<void *>(sem) : (void *)((char *)workspace + a);
        dim3 pg1((unsigned)((vox + 63) / 64), (d->channels + 31) / 32, d->batch);
        dim3 pg2((unsigned)((pix + 63) / 64), (d->sem_channels + 31) / 32, d->batch);
        if (d->dtype == DFM_F32) {
            if (!in_place)
                hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg1, dim3(256), 0, st,
                                   (const float *)stereo, (float *)stereo_pm, d->channels,
                                   d->channels, vox);
            if (d->sem_channels > 0 && !sem_in_place)
                hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg2, dim3(256), 0, st,
                                   (const float *)sem, (float *)sem_pm, d->sem_channels,
                                   d->sem_channels, pix);
            if (f2v_lanes_per_voxel(d) && g.out_cl)
                hipLaunchKernelGGL((f2v_pm8_kernel<float, false>), grid, dim3(256), 0, st, g,
                                   (const uint4 *)stereo_pm, (const float *)softmax,
                                   (const uint4 *)sem_pm, coords, cam2img, (float *)out, fh);
            else if (f2v_lanes_per_voxel(d) && N % 4 == 0 && !((uintptr_t)out & 15))
                hipLaunchKernelGGL((f2v_pm8_kernel<float, true>), grid, dim3(256), 0, st, g,
                                   (const uint4 *)stereo_pm, (const float *)softmax,
                                   (const uint4 *)sem_pm, coords, cam2img, (float *)out, fh);
            else
            hipLaunchKernelGGL(f2v_pm_kernel<float>, grid, dim3(256), 0, st, g,
                               (const uint4 *)stereo_pm, (const float *)softmax,
                               (const uint4 *)sem_pm, coords, cam2img, (float *)out, fh);
        } else {
            if (!in_place)
                hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg1, dim3(256), 0, st,
                                   (const bf16_t *)stereo, (bf16_t *)stereo_pm, d->channels,
                                   d->channels, vox);
            if (d->sem_channels > 0 && !sem_in_place)
                hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg2, dim3(256), 0, st,
                                   (const bf16_t *)sem, (bf16_t *)sem_pm, d->sem_channels,
                                   d->sem_channels, pix);
            if (f2v_lanes_per_voxel(d) && g.out_cl)
                hipLaunchKernelGGL((f2v_pm8_kernel<bf16_t, false>), grid, dim3(256), 0, st, g,
                                   (const uint4 *)stereo_pm, (const bf16_t *)softmax,
                                   (const uint4 *)sem_pm, coords, cam2img, (bf16_t *)out, fh);
            else if (f2v_lanes_per_voxel(d) && N % 4 == 0 && !((uintptr_t)out & 15))
                hipLaunchKernelGGL((f2v_pm8_kernel<bf16_t, true>), grid, dim3(256), 0, st, g,
                                   (const uint4 *)stereo_pm, (const bf16_t *)softmax,
                                   (const uint4 *)sem_pm, coords, cam2img, (bf16_t *)out, fh);
            else
            hipLaunchKernelGGL(f2v_pm_kernel<bf16_t>, grid, dim3(256), 0, st, g,
                               (const uint4 *)stereo_pm, (const bf16_t *)softmax,
                               (const uint4 *)sem_pm, coords, cam2img, (bf16_t *)out, fh);
        }
    } else if (d->dtype == DFM_F32)
        hipLaunchKernelGGL(f2v_kernel<float>, grid, dim3(256), 0, st, g, (const float *)stereo,
                           (const float *)softmax, (const float *)sem, coords, cam2img, (float *)out, fh);
    else
        hipLaunchKernelGGL(f2v_kernel<bf16_t>, grid, dim3(256), 0, st, g, (const bf16_t *)stereo,
                           (const bf16_t *)softmax, (const bf16_t *)sem, coords, cam2img,
                           (bf16_t *)out, fh);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_frustum_to_voxel_fwd(const dfm_f2v_desc *d, const void *stereo, const void *softmax,
                                     const void *sem, const float *coords, const float *cam2img,
                                     void *out, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    return f2v_fwd_impl(d, stereo, softmax, FusedHead{nullptr, nullptr, nullptr}, 0, sem, coords, cam2img, out,
                        workspace, workspace_bytes, stream);
}

DFM_API int dfm_frustum_to_voxel_fused_fwd(const dfm_f2v_desc *d, const void *stereo, const void *cost,
                                           const float *col_max, const float *col_sum,
                                           int32_t head_scale, const void *sem, const float *coords,
                                           const float *cam2img, void *out, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    if (!cost || !col_max || !col_sum) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    return f2v_fwd_impl(d, stereo, nullptr, FusedHead{cost, col_max, col_sum}, head_scale, sem, coords, cam2img,
                        out, workspace, workspace_bytes, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// backward of the sampling stage w.r.t. stereo_feat and cur_sem_feats (the depth
// distribution is detached in the reference, feature_transformation.py:136).
// One lane = one voxel; fp32 atomics into zero-initialised gradient tensors.
// ---------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void f2v_bwd_kernel(F2vGeom g, const T *__restrict__ gout,
                                                      const T *__restrict__ soft, FusedHead fh,
                                                      const float *__restrict__ coords,
                                                      const float *__restrict__ cam2img,
                                                      float *__restrict__ gstereo,
                                                      float *__restrict__ gsem)
{
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
    const float *P = cam2img + 16 * b;
    const float a = dot4_chain(-ys, -zs, xs, 1.0f, P + 0);
    const float bb = dot4_chain(-ys, -zs, xs, 1.0f, P + 4);
    const float c = dot4_chain(-ys, -zs, xs, 1.0f, P + 8);
    const float u = a / c, v = bb / c;
    const bool valid2d = (u >= 0.0f) && (u <= g.pad_w) && (v >= 0.0f) && (v <= g.pad_h);
    float gx = (u - 0.0f) / (g.pad_w - 1.0f), gy = (v - 0.0f) / (g.pad_h - 1.0f);
    float gz = (xs - g.depth_min) / g.depth_span;
    gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
    const bool valid = valid2d && gz >= -1.0f && gz <= 1.0f;
    // grad_out in the OUTPUT's layout: planar (B, C + Cs, N) or channels-last (B, N, C + Cs) (g.out_cl)
    const size_t gcs = g.out_cl ? 1 : (size_t)N;  // elements between channels of one voxel
    const T *go = g.out_cl ? gout + ((size_t)b * N + i) * (g.C + g.Cs) : gout + (size_t)b * (g.C + g.Cs) * N + i;
    const size_t vol = (size_t)g.D * g.H * g.W;
    float disp = 1.0f;  // pred_disp (detached): scales the gradients of the attended branches
    if (valid && (g.st_att || (g.Cs > 0 && g.sem_att))) {
        if (fh.cost) {  // the depth head fused (training): the distribution evaluated where it is sampled
            disp = fused_disp<T>(g, (const T *)fh.cost + (size_t)b * g.cd * g.ch * g.cw,
                                 fh.col_max + (size_t)b * g.Hs * g.Ws, fh.col_sum + (size_t)b * g.Hs * g.Ws, gx, gy, gz);
        } else {
            const Tri ts = make_tri(gx, gy, gz, g.Ds, g.Hs, g.Ws);
            disp = tri_sample<T>(ts, soft + (size_t)b * g.Ds * g.Hs * g.Ws);
        }
    }
    if (valid) {
        const Tri t = make_tri(gx, gy, gz, g.D, g.H, g.W);
        float *gs = gstereo + (size_t)b * g.C * vol;
        const float sdisp = g.st_att ? disp : 1.0f;
        for (int ch = 0; ch < g.C; ++ch) {
            const float gv = elem<T>::load(go[(size_t)ch * gcs]) * sdisp;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (t.ok & (1u << k)) atomicAdd(gs + (size_t)ch * vol + t.o[k], gv * t.w[k]);
        }
    }
    // Voxel_2D = sample(sem) * valid2d (* disp * valid): without the attention the 2-D mask alone gates it
    if (g.Cs > 0 && (g.sem_att ? valid : valid2d)) {
        const float mdisp = g.sem_att ? disp : 1.0f;
        const Tri t2 = make_tri(gx, gy, 0.0f, 1, g.Hsem, g.Wsem);
        const size_t plane = (size_t)g.Hsem * g.Wsem;
        float *gm = gsem + (size_t)b * g.Cs * plane;
        for (int ch = 0; ch < g.Cs; ++ch) {
            const float gv = elem<T>::load(go[(size_t)(g.C + ch) * gcs]) * mdisp;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t2.ok & (1u << k)) atomicAdd(gm + (size_t)ch * plane + t2.o[k], gv * t2.w[k]);
        }
    }
}

// Pixel-major backward.  Scattered global fp32 atomics run at ~20 G/s, a wave whose 64
// addresses are consecutive at ~320 G/s (profiles/archive/r01_atomic_microbench.txt).  So the gradients
// are accumulated in pixel-major scratch ([d*h*w][C] and [h*w][Cs], fp32): the lanes of a wave
// are the CHANNELS of a voxel, a tap is one contiguous run of C atomics.  A workgroup takes 64
// voxels: their grad_out rows come in through an LDS tile (read along voxels, used along
// channels), their footprints are computed once per voxel; a transpose pass adds the scratch into
// the caller's NC(D)HW gradients.
constexpr int F2V_VT = 64;  // voxels per workgroup

struct BwdFoot {
    int st[8];     // stereo pixel index per corner, -1 = no contribution
    float sw[8];
    int sm[4];     // semantic-map pixel index per corner, -1 = none
    float mw[4];   // corner weight * depth probability
};

template <typename T>
__global__ __launch_bounds__(256) void f2v_bwd_pm_kernel(F2vGeom g, const T *__restrict__ gout,
                                                         const T *__restrict__ soft, FusedHead fh,
                                                         const float *__restrict__ coords,
                                                         const float *__restrict__ cam2img,
                                                         float *__restrict__ gst_pm,
                                                         float *__restrict__ gsem_pm)
{
    extern __shared__ float f2v_lds[];
    const int CT = g.C + g.Cs;
    float *gt = f2v_lds;                                   // [CT][F2V_VT + 1]
    BwdFoot *foot = (BwdFoot *)(gt + CT * (F2V_VT + 1));   // [F2V_VT]
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const long long v0 = (long long)blockIdx.x * F2V_VT;
    const int b = blockIdx.y;
    const int nv = (int)min((long long)F2V_VT, N - v0);
    const int tid = threadIdx.x;
    // grad_out rows of the tile: one channel per wave pass, 64 consecutive voxels per load
    if (g.out_cl) {
        // channels-last gradient (what an NDHWC voxel_convs backward hands over): the tile's 64 voxels x CT
        // channels are ONE contiguous run, read in place -- torch's strided re-layout of this tensor to the
        // planar form cost 2.1 ms of a 31 ms training step (profiles/archive/r05_c11_*)
        const T *go = gout + ((size_t)b * N + v0) * CT;
        for (int i = tid; i < CT * F2V_VT; i += 256) {
            const int v = i / CT, c = i - v * CT;
            gt[c * (F2V_VT + 1) + v] = v < nv ? elem<T>::load(go[i]) : 0.0f;
        }
    } else {
        const T *go = gout + (size_t)b * CT * N + v0;
        for (int i = tid; i < CT * F2V_VT; i += 256) {
            const int c = i / F2V_VT, v = i - c * F2V_VT;
            gt[c * (F2V_VT + 1) + v] = v < nv ? elem<T>::load(go[(size_t)c * N + v]) : 0.0f;
        }
    }
    if (tid < F2V_VT) {
        BwdFoot f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { f.st[k] = -1; f.sw[k] = 0.0f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { f.sm[k] = -1; f.mw[k] = 0.0f; }
        if (tid < nv) {
            const long long i = v0 + tid;
            const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
            const float *P = cam2img + 16 * b;
            const float a = dot4_chain(-ys, -zs, xs, 1.0f, P + 0);
            const float bb = dot4_chain(-ys, -zs, xs, 1.0f, P + 4);
            const float c = dot4_chain(-ys, -zs, xs, 1.0f, P + 8);
            const float u = a / c, v = bb / c;
            const bool valid2d = (u >= 0.0f) && (u <= g.pad_w) && (v >= 0.0f) && (v <= g.pad_h);
            float gx = (u - 0.0f) / (g.pad_w - 1.0f), gy = (v - 0.0f) / (g.pad_h - 1.0f);
            float gz = (xs - g.depth_min) / g.depth_span;
            gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
            const bool valid = valid2d && gz >= -1.0f && gz <= 1.0f;
            float disp = 1.0f;
            if (valid && (g.st_att || (g.Cs > 0 && g.sem_att))) {
                if (fh.cost) {
                    disp = fused_disp<T>(g, (const T *)fh.cost + (size_t)b * g.cd * g.ch * g.cw,
                                         fh.col_max + (size_t)b * g.Hs * g.Ws, fh.col_sum + (size_t)b * g.Hs * g.Ws,
                                         gx, gy, gz);
                } else {
                    const Tri ts = make_tri(gx, gy, gz, g.Ds, g.Hs, g.Ws);
                    disp = tri_sample<T>(ts, soft + (size_t)b * g.Ds * g.Hs * g.Ws);
                }
            }
            if (valid) {
                const Tri t = make_tri(gx, gy, gz, g.D, g.H, g.W);
                const float sdisp = g.st_att ? disp : 1.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (t.ok & (1u << k)) { f.st[k] = t.o[k]; f.sw[k] = t.w[k] * sdisp; }
            }
            // Voxel_2D = sample(sem) * valid2d (* disp * valid)
            if (g.Cs > 0 && (g.sem_att ? valid : valid2d)) {
                const float mdisp = g.sem_att ? disp : 1.0f;
                const Tri t2 = make_tri(gx, gy, 0.0f, 1, g.Hsem, g.Wsem);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (t2.ok & (1u << k)) { f.sm[k] = t2.o[k]; f.mw[k] = t2.w[k] * mdisp; }
            }
        }
        foot[tid] = f;
    }
    __syncthreads();
    // lanes = channels of a voxel: [voxel in iteration][channel]
    const int wave = tid >> 6, lane = tid & 63;
    int lpv = 1;
    while (lpv < min(max(g.C, g.Cs), 64)) lpv <<= 1;  // lanes per voxel
    // a wave owns F2V_VT/4 voxels: never spread an iteration over more of them (1- and 2-channel
    // maps would otherwise scatter their neighbours' voxels a second time)
    lpv = max(lpv, 64 / (F2V_VT / 4));
    const int ch = lane & (lpv - 1), vin = lane / lpv, vpi = 64 / lpv;
    float *gs = gst_pm + (size_t)b * g.D * g.H * g.W * g.C;
    float *gm = gsem_pm + (size_t)b * g.Hsem * g.Wsem * g.Cs;
    for (int it = 0; it < F2V_VT / 4; it += vpi) {
        const int v = wave * (F2V_VT / 4) + it + vin;
        if (v >= nv) continue;
        const BwdFoot &f = foot[v];
        for (int c = ch; c < g.C; c += lpv) {
            const float gv = gt[c * (F2V_VT + 1) + v];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (f.st[k] >= 0) atomicAdd(gs + (size_t)f.st[k] * g.C + c, gv * f.sw[k]);
        }
        for (int c = ch; c < g.Cs; c += lpv) {
            const float gv = gt[(g.C + c) * (F2V_VT + 1) + v];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (f.sm[k] >= 0) atomicAdd(gm + (size_t)f.sm[k] * g.Cs + c, gv * f.mw[k]);
        }
    }
}


// 8 consecutive channels of one voxel's gradient row, as fp32 (one 16-byte load in bf16, two in fp32)
template <typename T>
__device__ __forceinline__ void gather_row8(const T *__restrict__ gp, float (&v)[8])
{
    if constexpr (sizeof(T) == 4) {
        const uint4 u0 = *(const uint4 *)gp, u1 = *(const uint4 *)(gp + 4);
        v[0] = __uint_as_float(u0.x); v[1] = __uint_as_float(u0.y); v[2] = __uint_as_float(u0.z); v[3] = __uint_as_float(u0.w);
        v[4] = __uint_as_float(u1.x); v[5] = __uint_as_float(u1.y); v[6] = __uint_as_float(u1.z); v[7] = __uint_as_float(u1.w);
    } else {
        const uint4 u = *(const uint4 *)gp;
        const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w4[k] << 16);
            v[2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward as a GATHER (round 5).  The pixel-major scatter above runs at the part's fp32 atomic rate: 1.75 M
// voxels x 8 corners x 64 channels = 0.9 G atomics = 2.9 ms per sample at config K, the third-largest kernel of
// the training step.  Turned around: a lane owns a cell column of the cost volume -- pixel (h, w), a run of depth
// planes -- and finds the voxels whose trilinear footprint holds each cell.  The voxel grid is REGULAR
// (prepare_coordinates_3d: linspace centres, x fastest; the host checks it and passes origin and steps) and the
// voxel -> frustum map is explicit: depth is affine in the voxel's x index, and for a given x the image position
// is a projective function of (y, z) -- so for every x slab that can touch the plane a 2 x 2 linear system gives
// the (y, z) voxel position that projects onto the pixel, and solving it again one pixel further along u and v
// gives the box of voxel indices that can touch the pixel.  Every voxel in the box is then run through the
// FORWARD's own arithmetic (projection, normalisation, floor): it counts only if one of its eight corners IS this
// cell, with the forward's weight -- the (voxel, corner, weight) set is exactly the scatter's.  A hit gathers the
// voxel's gradient row (channels-last: one contiguous run).  The stereo gradient of a cell is STORED when its
// plane is done; the semantic map's gradient (same pixel, every plane: taken from the hits whose LOWER depth
// corner is the cell, so a voxel counts once) leaves as 32 atomics per lane and depth chunk instead of 128 per
// voxel.  The per-voxel factors (validity, pred_disp) come from a lane-per-voxel pre-pass into 8 bytes per voxel.
// ---------------------------------------------------------------------------------------------------------
struct F2vGrid {
    float x0, dx, y0, dy, z0, dz;  // coords[(iz * Ny + iy) * Nx + ix] == (x0 + ix dx, y0 + iy dy, z0 + iz dz)
};

constexpr uint32_t F2G_NONE = 0xffffffffu;  // cell code of a voxel the gather skips

// pre-pass, lane = voxel: the factors the scatter's first phase computes.  sfac = the stereo branch's factor
// (pred_disp when stereo_atten, else 1) or -1 for a voxel that contributes nothing; mfac likewise for the
// semantic branch (pred_disp * valid when sem_atten, else valid2d).  Same arithmetic as f2v_bwd_pm_kernel.
template <typename T>
__global__ __launch_bounds__(256) void f2v_bwd_prep_kernel(F2vGeom g, const T *__restrict__ soft, FusedHead fh,
                                                           const float *__restrict__ coords,
                                                           const float *__restrict__ cam2img,
                                                           float *__restrict__ sfac, float *__restrict__ mfac,
                                                           uint32_t *__restrict__ cell, float4 *__restrict__ pos)
{
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const float xs = coords[3 * i], ys = coords[3 * i + 1], zs = coords[3 * i + 2];
    const float *P = cam2img + 16 * b;
    const float a = dot4_chain(-ys, -zs, xs, 1.0f, P + 0);
    const float bb = dot4_chain(-ys, -zs, xs, 1.0f, P + 4);
    const float c = dot4_chain(-ys, -zs, xs, 1.0f, P + 8);
    const float u = a / c, v = bb / c;
    const bool valid2d = (u >= 0.0f) && (u <= g.pad_w) && (v >= 0.0f) && (v <= g.pad_h);
    float gx = (u - 0.0f) / (g.pad_w - 1.0f), gy = (v - 0.0f) / (g.pad_h - 1.0f);
    float gz = (xs - g.depth_min) / g.depth_span;
    gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f; gz = gz * 2.0f - 1.0f;
    const bool valid = valid2d && gz >= -1.0f && gz <= 1.0f;
    float disp = 1.0f;
    if (valid && (g.st_att || (g.Cs > 0 && g.sem_att))) {
        if (fh.cost) {
            disp = fused_disp<T>(g, (const T *)fh.cost + (size_t)b * g.cd * g.ch * g.cw,
                                 fh.col_max + (size_t)b * g.Hs * g.Ws, fh.col_sum + (size_t)b * g.Hs * g.Ws, gx, gy, gz);
        } else {
            const Tri ts = make_tri(gx, gy, gz, g.Ds, g.Hs, g.Ws);
            disp = tri_sample<T>(ts, soft + (size_t)b * g.Ds * g.Hs * g.Ws);
        }
    }
    const float sf = valid ? (g.st_att ? disp : 1.0f) : -1.0f;
    const float mf = (g.Cs > 0 && (g.sem_att ? valid : valid2d)) ? (g.sem_att ? disp : 1.0f) : -1.0f;
    if (sfac) sfac[(size_t)b * N + i] = sf;
    mfac[(size_t)b * N + i] = mf;
    if (cell) {
        // round 6: the voxel's lower corner cell, packed, and its position in the cost volume -- make_tri's arithmetic,
        // done ONCE per voxel here instead of once per (candidate pixel, plane) in the gather.  A voxel that
        // contributes nothing (or whose position is not finite, or lies outside the packable range: it then has no
        // cell inside the volume either) is F2G_NONE.
        const float px = ((gx + 1.0f) / 2.0f) * (float)(g.W - 1);
        const float py = ((gy + 1.0f) / 2.0f) * (float)(g.H - 1);
        const float pz = ((gz + 1.0f) / 2.0f) * (float)(g.D - 1);
        uint32_t c = F2G_NONE;
        if ((sf >= 0.0f || mf >= 0.0f) && fabsf(px) <= 1.0e9f && fabsf(py) <= 1.0e9f && fabsf(pz) <= 1.0e9f) {
            const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
            if (x0 >= -1.0f && x0 <= 2045.0f && y0 >= -1.0f && y0 <= 1021.0f && z0 >= -1.0f && z0 <= 1021.0f)
                c = (uint32_t)((int)x0 + 1) | ((uint32_t)((int)y0 + 1) << 11) | ((uint32_t)((int)z0 + 1) << 21);
        }
        cell[(size_t)b * N + i] = c;
        pos[(size_t)b * N + i] = make_float4(px, py, pz, sf);
    }
}

// Depth planes per lane.  Round 5 ran 9 (72 planes -> 8 chunks): 800 workgroups of a kernel that held 256 + 16
// registers a lane -- ONE workgroup per CU, 3.1 rounds over the chip, each wave waiting out its own chain of
// dependent loads: 1.80 ms.  Three planes a chunk: 0.97 ms (1, 2 planes measure the same, 4 is 9 % slower; capping
// the registers for 3 or 4 workgroups per CU spilled: 3.3 / 5.7 ms).  Then the candidate test on a packed cell word
// from the pre-pass, four lanes per pixel (150 registers, three waves per SIMD) and the slab positions as affine
// functions of the slab index: 0.75 ms -- profiles/r06_c28_f2v_bwd_depth_chunk.txt
#ifndef DFM_F2G_DCH
#define DFM_F2G_DCH 3
#endif
#ifndef DFM_F2G_WGS
#define DFM_F2G_WGS 1
#endif
constexpr int F2G_DCH = DFM_F2G_DCH;  // depth planes per lane (a chunk): 72 planes -> 24 chunks

// lane = pixel (h, w) of the cost volume x a chunk of depth planes; C == 32, Cs in {0, 32} with the semantic
// map at the cost volume's resolution.  gvs / gcs: element strides of grad_out between voxels / channels.
// GCL: the stereo gradient is written (B, D, H, W, 32) in T -- the layout and type of a channels-last cost volume,
// a lane's 32 sums as one contiguous row (a wave: 64 consecutive rows), rounded once; else planar fp32.
template <typename T, bool SEM, bool GCL>
__global__ __launch_bounds__(256, DFM_F2G_WGS) void f2v_bwd_gather_kernel(F2vGeom g, F2vGrid gr, int dchunks,
                                                             const T *__restrict__ gout, size_t gvs, size_t gcs,
                                                             const float *__restrict__ coords,
                                                             const float *__restrict__ cam2img,
                                                             const uint32_t *__restrict__ cell,
                                                             const float4 *__restrict__ pos,
                                                             const float *__restrict__ mfac,
                                                             void *__restrict__ gstv, float *__restrict__ gsem)
{
    // block id = ((ytile * xtiles + xtile) * dchunks + chunk); blockIdx.y = sample
    // (round 6) FOUR lanes per pixel, 8 channels each: a wave covers 16 pixels of a row.  A hit's gradient row is then
    // one 16-byte load per lane whose four lanes read one contiguous 64-byte row (a wave-level load touches 16 rows,
    // not 4 x 64), the accumulators are 8 + 8 registers instead of 32 + 32, and the candidate test the four lanes
    // repeat is a 4-byte load of the same word.
    const int xtiles = (g.W + 15) / 16;
    int t = blockIdx.x;
    const int chunk = t % dchunks;
    t /= dchunks;
    const int xt = t % xtiles, yt = t / xtiles;
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = yt * 4 + wave, w = xt * 16 + (lane >> 2);
    const int cq = (lane & 3) * 8;  // this lane's channels: cq .. cq + 7
    const bool inside = h < g.H && w < g.W;
    const long long N = (long long)g.Nz * g.Ny * g.Nx;
    const float *P = cam2img + 16 * b;
    const int CT = g.C + g.Cs;
    // image position of this pixel's centre and of its neighbours one pixel further (the box's extent)
    const float su = (g.pad_w - 1.0f) / (float)(g.W - 1), sv = (g.pad_h - 1.0f) / (float)(g.H - 1);
    const float ut = (float)w * su, vt = (float)h * sv;
    // (y, z) of the voxel position that projects onto image point (uu, vv) in the slab x = xs: a 2 x 2 system whose
    // right-hand side is affine in xs -- and xs is affine in the slab index, so the voxel-index position of the
    // solution is iy = Ay + By * ix, iz = Az + Bz * ix: four coefficients per image point, worked out ONCE per lane
    // (round 5 solved the three systems per slab, with a division each: 200 of the ~250 instructions of a slab with
    // no candidate in it).  The box only bounds the candidates (15 % + 0.05 voxels of slack); the voxels inside it are
    // still tested with the forward's own arithmetic.
    const float idx = 1.0f / gr.dx, idy = 1.0f / gr.dy, idz = 1.0f / gr.dz;
    auto affine = [&](float uu, float vv, float &ay, float &by, float &az, float &bz) {
        const float a11 = uu * P[8] - P[0], a12 = uu * P[9] - P[1], c1 = P[2] - uu * P[10], d1 = P[3] - uu * P[11];
        const float a21 = vv * P[8] - P[4], a22 = vv * P[9] - P[5], c2 = P[6] - vv * P[10], d2 = P[7] - vv * P[11];
        const float det = a11 * a22 - a12 * a21, inv = 1.0f / det;
        // ys = ((-c1 xs - d1) a22 + a12 (c2 xs + d2)) inv,  zs = (a11 (-c2 xs - d2) + (c1 xs + d1) a21) inv
        const float ys1 = (a12 * c2 - c1 * a22) * inv, ys0 = (a12 * d2 - d1 * a22) * inv;
        const float zs1 = (c1 * a21 - a11 * c2) * inv, zs0 = (d1 * a21 - a11 * d2) * inv;
        // xs = x0 + ix dx;  index = (pos - origin) / step
        by = ys1 * gr.dx * idy; ay = (ys0 + ys1 * gr.x0 - gr.y0) * idy;
        bz = zs1 * gr.dx * idz; az = (zs0 + zs1 * gr.x0 - gr.z0) * idz;
    };
    float cAy, cBy, cAz, cBz, uAy, uBy, uAz, uBz, vAy, vBy, vAz, vBz;
    affine(ut, vt, cAy, cBy, cAz, cBz);
    affine(ut + su, vt, uAy, uBy, uAz, uBz);
    affine(ut, vt + sv, vAy, vBy, vAz, vBz);
    const float pd_per_x = (float)(g.D - 1) / g.depth_span;  // plane index per metre of depth
    const T *gb = gout + (size_t)b * N * CT;  // (both layouts: a sample is N * (C + Cs) elements)
    float asem[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) asem[c] = 0.0f;
    const int d0 = chunk * F2G_DCH, d1 = min(g.D, d0 + F2G_DCH);
    for (int d = d0; d < d1; ++d) {
        float ast[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) ast[c] = 0.0f;
        // x slabs whose plane position lies in (d - 1, d + 1): uniform for the workgroup
        const float xlo = g.depth_min + ((float)d - 1.0f) / pd_per_x, xhi = g.depth_min + ((float)d + 1.0f) / pd_per_x;
        const float fa = (xlo - gr.x0) * idx, fb = (xhi - gr.x0) * idx;
        const int ixa = max(0, (int)ceilf(fminf(fa, fb) - 0.02f)), ixb = min(g.Nx - 1, (int)floorf(fmaxf(fa, fb) + 0.02f));
        for (int ix = ixa; ix <= ixb; ++ix) {
            const float fx = (float)ix;
            const float iyf = __builtin_fmaf(cBy, fx, cAy), izf = __builtin_fmaf(cBz, fx, cAz);
            const float ty = (fabsf(__builtin_fmaf(uBy, fx, uAy) - iyf) + fabsf(__builtin_fmaf(vBy, fx, vAy) - iyf)) * 1.15f + 0.05f;
            const float tz = (fabsf(__builtin_fmaf(uBz, fx, uAz) - izf) + fabsf(__builtin_fmaf(vBz, fx, vAz) - izf)) * 1.15f + 0.05f;
            const float y0f = ceilf(iyf - ty), y1f = floorf(iyf + ty), z0f = ceilf(izf - tz), z1f = floorf(izf + tz);
            // (non-finite solutions compare false.)  The box is walked in FULL whatever its size, clamped to the
            // grid in fp32 before the int conversion: a coarser cost volume, finer voxels or a long depth range
            // make a cost-volume pixel span many voxels, and the former cut at 8 x 8 / t < 4 dropped those
            // contributions silently.  Larger boxes only cost more iterations.
            const bool some = inside && y0f <= y1f && z0f <= z1f && y1f >= 0.0f && z1f >= 0.0f &&
                              y0f <= (float)(g.Ny - 1) && z0f <= (float)(g.Nz - 1);
            if (!__any(some)) continue;
            const int iy0 = some ? (int)fmaxf(y0f, 0.0f) : 0, iy1 = some ? (int)fminf(y1f, (float)(g.Ny - 1)) : -1;
            const int iz0 = some ? (int)fmaxf(z0f, 0.0f) : 0, iz1 = some ? (int)fminf(z1f, (float)(g.Nz - 1)) : -1;
            // one candidate voxel (ix, iy, iz): its packed lower-corner cell against this lane's (w, h, d) -- a 4-byte
            // load and three subtractions reject it (round 5 re-ran the forward's projection for every candidate: three
            // dependent loads, three dot products and two divisions before the first test)
            auto visit = [&](int iy, int iz) {
                const long long i = ((long long)iz * g.Ny + iy) * g.Nx + ix;
                const uint32_t cc = cell[(size_t)b * N + i];
                if (cc == F2G_NONE) return;
                const int kx = w - ((int)(cc & 0x7ffu) - 1), ky = h - ((int)((cc >> 11) & 0x3ffu) - 1);
                const int kz = d - ((int)(cc >> 21) - 1);
                if ((unsigned)kx > 1u || (unsigned)ky > 1u) return;
                const bool zhit = (unsigned)kz <= 1u;
                if (!zhit && !(SEM && kz == 0)) return;
                const float4 pp = pos[(size_t)b * N + i];
                const float px = pp.x, py = pp.y, pz = pp.z, sf = pp.w;
                const float mf = SEM ? mfac[(size_t)b * N + i] : -1.0f;
                // make_tri's weights, corner by corner (the forward's expressions on the forward's values)
                const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
                const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
                const float wx = kx ? px - x0 : x1 - px, wy = ky ? py - y0 : y1 - py;
                const T *gp = gb + (size_t)i * gvs;
                if (zhit && sf >= 0.0f) {
                    const float wgt = ((wx * wy) * (kz ? pz - z0 : z1 - pz)) * sf;
                    float vv[8];
                    if (gcs == 1) {
                        gather_row8<T>(gp + cq, vv);
                    } else {
#pragma unroll
                        for (int cc2 = 0; cc2 < 8; ++cc2) vv[cc2] = elem<T>::load(gp[(size_t)(cq + cc2) * gcs]);
                    }
#pragma unroll
                    for (int cc2 = 0; cc2 < 8; ++cc2) ast[cc2] += vv[cc2] * wgt;
                }
                if constexpr (SEM) {
                    // the semantic map's pixel (h, w): once per voxel, from the hit on its lower depth corner
                    // (a valid voxel's lower corner is inside the volume); make_tri(gx, gy, 0, 1, H, W): z1 - iz = 1
                    if (kz == 0 && mf >= 0.0f) {
                        const float mw = ((wx * wy) * 1.0f) * mf;
                        float vv[8];
                        if (gcs == 1) {
                            gather_row8<T>(gp + 32 + cq, vv);
                        } else {
#pragma unroll
                            for (int cc2 = 0; cc2 < 8; ++cc2) vv[cc2] = elem<T>::load(gp[(size_t)(32 + cq + cc2) * gcs]);
                        }
#pragma unroll
                        for (int cc2 = 0; cc2 < 8; ++cc2) asem[cc2] += vv[cc2] * mw;
                    }
                }
            };
            // boxes of up to 8 x 8 voxels -- what the path's grids produce -- take the counted loops; a wave that
            // holds a larger box walks it in full (uniform choice)
#ifdef DFM_GATHER_R5_LOOPS   // (A/B builds only: the round-5 form -- boxes cut at 8 x 8)
            const bool small_box = true;
#else
            const bool small_box = !__any(some && (iy1 - iy0 > 7 || iz1 - iz0 > 7));
#endif
            if (small_box) {
                for (int jz = 0; jz < 8; ++jz) {
                    const int iz = iz0 + jz;
                    if (!__any(iz <= iz1)) break;
                    for (int jy = 0; jy < 8; ++jy) {
                        const int iy = iy0 + jy;
                        const bool cand = iz <= iz1 && iy <= iy1;
                        if (!__any(cand)) break;
                        if (cand) visit(iy, iz);
                    }
                }
            } else {
                for (int jz = 0;; ++jz) {
                    const int iz = iz0 + jz;
                    if (!__any(iz <= iz1)) break;
                    for (int jy = 0;; ++jy) {
                        const int iy = iy0 + jy;
                        const bool cand = iz <= iz1 && iy <= iy1;
                        if (!__any(cand)) break;
                        if (cand) visit(iy, iz);
                    }
                }
            }
        }
        if (inside) {
            if constexpr (GCL) {
                constexpr int VEC = dfm::vec16<T>::N;
                T *o = (T *)gstv + ((((size_t)b * g.D + d) * g.H + h) * g.W + w) * 32 + cq;
#pragma unroll
                for (int q = 0; q < 8 / VEC; ++q) {
                    float r[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) r[e] = ast[q * VEC + e];
                    dfm::store16<T>(o + q * VEC, r);
                }
            } else {
                float *o = (float *)gstv + ((size_t)b * g.C * g.D + d) * g.H * g.W + (size_t)h * g.W + w;
                const size_t cs = (size_t)g.D * g.H * g.W;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[(size_t)(cq + c) * cs] = ast[c];
            }
        }
    }
    if constexpr (SEM) {
        if (inside) {
            float *o = gsem + (size_t)b * g.Cs * g.Hsem * g.Wsem + (size_t)h * g.Wsem + w;
            const size_t cs = (size_t)g.Hsem * g.Wsem;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (asem[c] != 0.0f) atomicAdd(o + (size_t)(cq + c) * cs, asem[c]);
        }
    }
}

}  // namespace

extern "C" DFM_API size_t dfm_frustum_to_voxel_bwd_workspace_bytes(const dfm_f2v_desc *d)
{
    if (!d || d->batch <= 0 || d->channels <= 0 || d->d <= 0 || d->h <= 0 || d->w <= 0 ||
        d->sem_channels < 0)
        return 0;
    const size_t a = (size_t)d->batch * d->channels * d->d * d->h * d->w * sizeof(float);
    const size_t b = (size_t)d->batch * d->sem_channels * d->hsem * d->wsem * sizeof(float);
    return ((a + 255) & ~(size_t)255) + ((b + 255) & ~(size_t)255) + 256;
}

static int f2v_bwd_impl(const dfm_f2v_desc *d, const void *grad_out, const void *softmax, FusedHead fh,
                        int head_scale, const float *coords, const float *cam2img, float *grad_stereo,
                        float *grad_sem, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!grad_out || !coords || !cam2img || !grad_stereo || (d->sem_channels > 0 && !grad_sem) ||
        ((d->stereo_atten || (d->sem_channels > 0 && !d->no_sem_atten)) && !softmax && !fh.cost))
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    F2vGeom g;
    g.C = d->channels; g.D = d->d; g.H = d->h; g.W = d->w;
    g.Ds = d->ds; g.Hs = d->hs; g.Ws = d->ws;
    g.Cs = d->sem_channels; g.Hsem = d->hsem; g.Wsem = d->wsem;
    g.Nz = d->nz; g.Ny = d->ny; g.Nx = d->nx;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.depth_min = d->depth_min; g.depth_span = d->depth_span;
    g.cd = g.ch = g.cw = 0;
    if (fh.cost) {
        if (head_scale <= 0 || d->ds % head_scale || d->hs % head_scale || d->ws % head_scale)
            return set_error(DFM_ERR_INVALID_ARG, "ds, hs, ws must be multiples of the depth head's scale");
        g.cd = d->ds / head_scale; g.ch = d->hs / head_scale; g.cw = d->ws / head_scale;
    }
    g.out_cl = d->out_channels_last ? 1 : 0;  // grad_out comes in the layout the forward wrote its output in
    g.st_att = d->stereo_atten ? 1 : 0;
    g.sem_att = d->no_sem_atten ? 0 : 1;
    const long long N = (long long)d->nz * d->ny * d->nx;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(d->channels + d->sem_channels) * (F2V_VT + 1) * sizeof(float) +
                       F2V_VT * sizeof(BwdFoot);
    if (workspace && lds <= 64 * 1024 && d->batch <= 65535) {
        // pixel-major accumulation (see f2v_bwd_pm_kernel)
        if (workspace_bytes < dfm_frustum_to_voxel_bwd_workspace_bytes(d))
            return set_error(DFM_ERR_WORKSPACE,
                             "workspace smaller than dfm_frustum_to_voxel_bwd_workspace_bytes");
        const long long vox = (long long)d->d * d->h * d->w, pix = (long long)d->hsem * d->wsem;
        const size_t a = ((size_t)d->batch * d->channels * vox * sizeof(float) + 255) & ~(size_t)255;
        const size_t bsz = (size_t)d->batch * d->sem_channels * pix * sizeof(float);
        float *gst_pm = (float *)workspace, *gsem_pm = (float *)((char *)workspace + a);
        hipError_t e = hipMemsetAsync(workspace, 0, a + bsz, st);
        if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
        dim3 grid((unsigned)((N + F2V_VT - 1) / F2V_VT), d->batch);
        if (d->dtype == DFM_F32)
            hipLaunchKernelGGL(f2v_bwd_pm_kernel<float>, grid, dim3(256), lds, st, g,
                               (const float *)grad_out, (const float *)softmax, fh, coords, cam2img,
                               gst_pm, gsem_pm);
        else
            hipLaunchKernelGGL(f2v_bwd_pm_kernel<bf16_t>, grid, dim3(256), lds, st, g,
                               (const bf16_t *)grad_out, (const bf16_t *)softmax, fh, coords, cam2img,
                               gst_pm, gsem_pm);
        dim3 t1((unsigned)((vox + 63) / 64), (d->channels + 31) / 32, d->batch);
        hipLaunchKernelGGL(add_from_pixel_major_kernel<float>, t1, dim3(256), 0, st, gst_pm, grad_stereo,
                           d->channels, vox);
        if (d->sem_channels > 0) {
            dim3 t2((unsigned)((pix + 63) / 64), (d->sem_channels + 31) / 32, d->batch);
            hipLaunchKernelGGL(add_from_pixel_major_kernel<float>, t2, dim3(256), 0, st, gsem_pm, grad_sem,
                               d->sem_channels, pix);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
        return DFM_OK;
    }
    // no workspace (or more channels than the LDS tile holds): lane-per-voxel scatter
    dim3 grid((unsigned)((N + 255) / 256), d->batch);
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL(f2v_bwd_kernel<float>, grid, dim3(256), 0, st, g, (const float *)grad_out,
                           (const float *)softmax, fh, coords, cam2img, grad_stereo, grad_sem);
    else
        hipLaunchKernelGGL(f2v_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, g,
                           (const bf16_t *)grad_out, (const bf16_t *)softmax, fh, coords, cam2img,
                           grad_stereo, grad_sem);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_frustum_to_voxel_bwd(const dfm_f2v_desc *d, const void *grad_out,
                                                const void *softmax, const float *coords,
                                                const float *cam2img, float *grad_stereo,
                                                float *grad_sem, void *workspace,
                                                size_t workspace_bytes, void *stream)
{
    return f2v_bwd_impl(d, grad_out, softmax, FusedHead{nullptr, nullptr, nullptr}, 0, coords, cam2img, grad_stereo,
                        grad_sem, workspace, workspace_bytes, stream);
}

extern "C" DFM_API int dfm_frustum_to_voxel_fused_bwd(const dfm_f2v_desc *d, const void *grad_out, const void *cost,
                                                      const float *col_max, const float *col_sum,
                                                      int32_t head_scale, const float *coords,
                                                      const float *cam2img, float *grad_stereo, float *grad_sem,
                                                      void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cost || !col_max || !col_sum) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    return f2v_bwd_impl(d, grad_out, nullptr, FusedHead{cost, col_max, col_sum}, head_scale, coords, cam2img,
                        grad_stereo, grad_sem, workspace, workspace_bytes, stream);
}

// The gather form of the backward (f2v_bwd_gather_kernel).  grid6 (HOST memory): {x0, dx, y0, dy, z0, dz} of the
// regular voxel grid `coords` is (the caller has checked it: coords[(iz * Ny + iy) * Nx + ix] == origin + index *
// step).  grad_stereo is OVERWRITTEN (reference layout, fp32); grad_sem zero-filled by the caller, accumulated.
// workspace: >= dfm_frustum_to_voxel_bwd_gather_workspace_bytes (24 bytes per voxel), 16-byte aligned.  DFM_ERR_UNSUPPORTED unless
// C == 32, Cs in {0, 32} with the semantic map at the cost volume's resolution and sem_atten (a voxel outside
// the depth range then contributes nothing and every contributing voxel has a cell), and non-zero grid steps.
extern "C" DFM_API size_t dfm_frustum_to_voxel_bwd_gather_workspace_bytes(const dfm_f2v_desc *d)
{
    if (!d || d->batch <= 0 || d->nz <= 0 || d->ny <= 0 || d->nx <= 0) return 0;
    // per voxel: its position and stereo factor (16 bytes), its packed cell (4), its semantic factor (4)
    return (((size_t)d->batch * d->nz * d->ny * d->nx * 24) + 255) & ~(size_t)255;
}

static int f2v_bwd_gather_impl(const dfm_f2v_desc *d, const void *grad_out, const void *softmax, const void *cost,
                               const float *col_max, const float *col_sum, int32_t head_scale, const float *coords,
                               const float *grid6, const float *cam2img, void *grad_stereo, bool grad_cl,
                               float *grad_sem, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!grad_out || !coords || !grid6 || !cam2img || !grad_stereo || !workspace || (d->sem_channels > 0 && !grad_sem))
        return set_error(DFM_ERR_INVALID_ARG, "NULL pointer");
    const bool need_disp = d->stereo_atten || (d->sem_channels > 0 && !d->no_sem_atten);
    if (need_disp && !softmax && !cost) return set_error(DFM_ERR_INVALID_ARG, "the attended branches need the depth distribution");
    if (cost && (!col_max || !col_sum)) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (d->channels != 32 || !(d->sem_channels == 0 || (d->sem_channels == 32 && d->hsem == d->h && d->wsem == d->w &&
                                                          !d->no_sem_atten)) ||
        d->d < 2 || d->h < 2 || d->w < 2 || grid6[1] == 0.0f || grid6[3] == 0.0f || grid6[5] == 0.0f ||
        d->batch > 65535 || (d->out_channels_last && ((uintptr_t)grad_out & 15)) ||
        (grad_cl && ((uintptr_t)grad_stereo & 15)) || d->w > 2044 || d->h > 1020 || d->d > 1020 ||
        ((uintptr_t)workspace & 15))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "gather backward: C == 32, Cs in {0, 32} at the cost volume's resolution with sem_atten, regular grid");
    if (workspace_bytes < dfm_frustum_to_voxel_bwd_gather_workspace_bytes(d))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_frustum_to_voxel_bwd_gather_workspace_bytes");
    F2vGeom g;
    g.C = d->channels; g.D = d->d; g.H = d->h; g.W = d->w;
    g.Ds = d->ds; g.Hs = d->hs; g.Ws = d->ws;
    g.Cs = d->sem_channels; g.Hsem = d->hsem; g.Wsem = d->wsem;
    g.Nz = d->nz; g.Ny = d->ny; g.Nx = d->nx;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.depth_min = d->depth_min; g.depth_span = d->depth_span;
    g.cd = g.ch = g.cw = 0;
    FusedHead fh{nullptr, nullptr, nullptr};
    if (cost) {
        if (head_scale <= 0 || d->ds % head_scale || d->hs % head_scale || d->ws % head_scale)
            return set_error(DFM_ERR_INVALID_ARG, "ds, hs, ws must be multiples of the depth head's scale");
        g.cd = d->ds / head_scale; g.ch = d->hs / head_scale; g.cw = d->ws / head_scale;
        fh = FusedHead{cost, col_max, col_sum};
    }
    g.out_cl = d->out_channels_last ? 1 : 0;
    g.st_att = d->stereo_atten ? 1 : 0;
    g.sem_att = d->no_sem_atten ? 0 : 1;
    const F2vGrid gr{grid6[0], grid6[1], grid6[2], grid6[3], grid6[4], grid6[5]};
    const long long N = (long long)d->nz * d->ny * d->nx;
    hipStream_t st = (hipStream_t)stream;
    float4 *pos = (float4 *)workspace;
    uint32_t *cell = (uint32_t *)(pos + (size_t)d->batch * N);
    float *mfac = (float *)(cell + (size_t)d->batch * N);
    const dim3 pgrid((unsigned)((N + 255) / 256), d->batch);
    const int CT = d->channels + d->sem_channels;
    const size_t gvs = g.out_cl ? (size_t)CT : 1, gcs = g.out_cl ? 1 : (size_t)N;
    const int dchunks = (d->d + F2G_DCH - 1) / F2G_DCH;
    const dim3 ggrid((unsigned)(((d->w + 15) / 16) * ((d->h + 3) / 4) * dchunks), d->batch);
#define DFM_F2G_K(T_, SEM_, GCL_)                                                                                 \
    hipLaunchKernelGGL((f2v_bwd_gather_kernel<T_, SEM_, GCL_>), ggrid, dim3(256), 0, st, g, gr, dchunks,         \
                       (const T_ *)grad_out, gvs, gcs, coords, cam2img, (const uint32_t *)cell, (const float4 *)pos,  \
                       (const float *)mfac,                                                                      \
                       grad_stereo, grad_sem)
#define DFM_F2G(T_)                                                                                              \
    do {                                                                                                         \
        hipLaunchKernelGGL(f2v_bwd_prep_kernel<T_>, pgrid, dim3(256), 0, st, g, (const T_ *)softmax, fh, coords, \
                           cam2img, (float *)nullptr, mfac, cell, pos);                                          \
        if (d->sem_channels > 0) {                                                                               \
            if (grad_cl) DFM_F2G_K(T_, true, true); else DFM_F2G_K(T_, true, false);                             \
        } else {                                                                                                 \
            if (grad_cl) DFM_F2G_K(T_, false, true); else DFM_F2G_K(T_, false, false);                           \
        }                                                                                                        \
    } while (0)
    if (d->dtype == DFM_F32) DFM_F2G(float);
    else DFM_F2G(bf16_t);
#undef DFM_F2G
#undef DFM_F2G_K
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_frustum_to_voxel_bwd_gather(const dfm_f2v_desc *d, const void *grad_out, const void *softmax,
                                                       const void *cost, const float *col_max, const float *col_sum,
                                                       int32_t head_scale, const float *coords, const float *grid6,
                                                       const float *cam2img, float *grad_stereo, float *grad_sem,
                                                       void *workspace, size_t workspace_bytes, void *stream)
{
    return f2v_bwd_gather_impl(d, grad_out, softmax, cost, col_max, col_sum, head_scale, coords, grid6, cam2img,
                               grad_stereo, false, grad_sem, workspace, workspace_bytes, stream);
}

// The same, with the stereo gradient in the layout and type of a channels-last cost volume: grad_stereo is
// (B, d, h, w, C) in memory (torch channels_last_3d of (B, C, d, h, w)), desc->dtype, 16-byte aligned, OVERWRITTEN
// -- the fp32 sums rounded once at the store, the bits `dfm_frustum_to_voxel_bwd_gather` + a conversion give.  What
// the NDHWC stack's autograd wants: the engine adds this gradient to the one the prediction convolution's backward
// hands over in that layout (a planar fp32 result cost a zero fill, a conversion and a strided addition: 0.6 ms of
// a 22.5 ms training step).  grad_sem as above (fp32, planar, accumulated).
extern "C" DFM_API int dfm_frustum_to_voxel_bwd_gather_cl(const dfm_f2v_desc *d, const void *grad_out,
                                                          const void *softmax, const void *cost, const float *col_max,
                                                          const float *col_sum, int32_t head_scale, const float *coords,
                                                          const float *grid6, const float *cam2img, void *grad_stereo,
                                                          float *grad_sem, void *workspace, size_t workspace_bytes,
                                                          void *stream)
{
    return f2v_bwd_gather_impl(d, grad_out, softmax, cost, col_max, col_sum, head_scale, coords, grid6, cam2img,
                               grad_stereo, true, grad_sem, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------
// voxel_sample: the inverse op (voxel volume -> frustum), reference
// mmdet3d/models/fusion_layers/point_fusion.py:324-410.  One lane = one lattice
// point (w fastest); same trilinear arithmetic as above, or nearest.
// ---------------------------------------------------------------------------
namespace {

struct VsGeom {
    int32_t C, Nx, Ny, Nz, D, h_out, w_out, flip, mode;
    float ds, scale_x, scale_y, crop_x, crop_y, ori_w;
    float range[6], vsize[3], Minv[16];
};

template <typename T>
__global__ __launch_bounds__(256) void voxel_sample_kernel(VsGeom g, const T *__restrict__ vox,
                                                           const float *__restrict__ depths,
                                                           T *__restrict__ out)
{
    const long long N = (long long)g.D * g.h_out * g.w_out;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int w = (int)(i % g.w_out), h = (int)((i / g.w_out) % g.h_out);
    const int d = (int)(i / ((long long)g.w_out * g.h_out));
    float x = (float)w * g.ds, y = (float)h * g.ds;
    const float depth = depths[d];
    if (g.flip) x = g.ori_w - x;
    x = x + g.crop_x; y = y + g.crop_y;
    x = x / g.scale_x; y = y / g.scale_y;
    const float h0 = x * depth, h1 = y * depth;
    float gr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float X = dot4_chain(h0, h1, depth, 1.0f, g.Minv + 4 * k);
        const float gsz = (g.range[3 + k] - g.range[k]) / g.vsize[k];
        const float v = (X - g.range[k]) / g.vsize[k] - 0.5f;
        gr[k] = v / gsz * 2.0f - 1.0f;
    }
    const size_t vol = (size_t)g.Nx * g.Ny * g.Nz;
    if (g.mode) {
        const Tri t = make_tri(gr[2], gr[1], gr[0], g.Nx, g.Ny, g.Nz);
        for (int c = 0; c < g.C; ++c)
            out[(size_t)c * N + i] = elem<T>::store(tri_sample<T>(t, vox + c * vol));
    } else {
        const float ix = ((gr[2] + 1.0f) / 2.0f) * (float)(g.Nz - 1);
        const float iy = ((gr[1] + 1.0f) / 2.0f) * (float)(g.Ny - 1);
        const float iz = ((gr[0] + 1.0f) / 2.0f) * (float)(g.Nx - 1);
        const float xr = rintf(ix), yr = rintf(iy), zr = rintf(iz);
        const bool in = fabsf(ix) <= 1.0e9f && fabsf(iy) <= 1.0e9f && fabsf(iz) <= 1.0e9f &&
                        xr >= 0.0f && xr <= (float)(g.Nz - 1) && yr >= 0.0f &&
                        yr <= (float)(g.Ny - 1) && zr >= 0.0f && zr <= (float)(g.Nx - 1);
        const int o = in ? ((int)zr * g.Ny + (int)yr) * g.Nz + (int)xr : 0;
        for (int c = 0; c < g.C; ++c)
            out[(size_t)c * N + i] = in ? vox[c * vol + o] : T(0);
    }
}

// backward of voxel_sample w.r.t. the voxel features: the same coordinates, the gradient of every
// lattice point scattered to its <= 8 corners (or its nearest voxel) with fp32 atomics
// (point_fusion.py:396-410 is differentiable through F.grid_sample).
template <typename T>
__global__ __launch_bounds__(256) void voxel_sample_bwd_kernel(VsGeom g, const T *__restrict__ gout,
                                                               const float *__restrict__ depths,
                                                               float *__restrict__ gvox)
{
    const long long N = (long long)g.D * g.h_out * g.w_out;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int w = (int)(i % g.w_out), h = (int)((i / g.w_out) % g.h_out);
    const int d = (int)(i / ((long long)g.w_out * g.h_out));
    float x = (float)w * g.ds, y = (float)h * g.ds;
    const float depth = depths[d];
    if (g.flip) x = g.ori_w - x;
    x = x + g.crop_x; y = y + g.crop_y;
    x = x / g.scale_x; y = y / g.scale_y;
    const float h0 = x * depth, h1 = y * depth;
    float gr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float X = dot4_chain(h0, h1, depth, 1.0f, g.Minv + 4 * k);
        const float gsz = (g.range[3 + k] - g.range[k]) / g.vsize[k];
        const float v = (X - g.range[k]) / g.vsize[k] - 0.5f;
        gr[k] = v / gsz * 2.0f - 1.0f;
    }
    const size_t vol = (size_t)g.Nx * g.Ny * g.Nz;
    if (g.mode) {
        const Tri t = make_tri(gr[2], gr[1], gr[0], g.Nx, g.Ny, g.Nz);
        if (!t.ok) return;
        for (int c = 0; c < g.C; ++c) {
            const float gv = elem<T>::load(gout[(size_t)c * N + i]);
            if (gv == 0.0f) continue;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (t.ok & (1u << k)) atomicAdd(gvox + c * vol + t.o[k], gv * t.w[k]);
        }
    } else {
        const float ix = ((gr[2] + 1.0f) / 2.0f) * (float)(g.Nz - 1);
        const float iy = ((gr[1] + 1.0f) / 2.0f) * (float)(g.Ny - 1);
        const float iz = ((gr[0] + 1.0f) / 2.0f) * (float)(g.Nx - 1);
        const float xr = rintf(ix), yr = rintf(iy), zr = rintf(iz);
        const bool in = fabsf(ix) <= 1.0e9f && fabsf(iy) <= 1.0e9f && fabsf(iz) <= 1.0e9f &&
                        xr >= 0.0f && xr <= (float)(g.Nz - 1) && yr >= 0.0f &&
                        yr <= (float)(g.Ny - 1) && zr >= 0.0f && zr <= (float)(g.Nx - 1);
        if (!in) return;
        const int o = ((int)zr * g.Ny + (int)yr) * g.Nz + (int)xr;
        for (int c = 0; c < g.C; ++c) atomicAdd(gvox + c * vol + o, elem<T>::load(gout[(size_t)c * N + i]));
    }
}

}  // namespace

namespace {
int vs_geom(const dfm_vs_desc *d, VsGeom &g)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "desc is NULL");
    if (d->channels <= 0 || d->nx <= 0 || d->ny <= 0 || d->nz <= 0 || d->num_depths <= 0 ||
        d->h_out <= 0 || d->w_out <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "non-positive size in dfm_vs_desc");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if ((long long)d->nx * d->ny * d->nz >= (1ll << 31))
        return set_error(DFM_ERR_UNSUPPORTED, "volume too large for 32-bit corner offsets");
    g.C = d->channels; g.Nx = d->nx; g.Ny = d->ny; g.Nz = d->nz;
    g.D = d->num_depths; g.h_out = d->h_out; g.w_out = d->w_out;
    g.flip = d->flip; g.mode = d->mode; g.ds = d->downsample_factor;
    g.scale_x = d->scale_x; g.scale_y = d->scale_y; g.crop_x = d->crop_x; g.crop_y = d->crop_y;
    g.ori_w = d->ori_w;
    for (int k = 0; k < 6; ++k) g.range[k] = d->voxel_range[k];
    for (int k = 0; k < 3; ++k) g.vsize[k] = d->voxel_size[k];
    for (int k = 0; k < 16; ++k) g.Minv[k] = d->proj_inv[k];
    return DFM_OK;
}
}  // namespace

extern "C" DFM_API int dfm_voxel_sample_fwd(const dfm_vs_desc *d, const void *voxel_features,
                                            const float *depths, void *out, void *stream)
{
    VsGeom g;
    int rc = vs_geom(d, g);
    if (rc != DFM_OK) return rc;
    if (!voxel_features || !depths || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const long long N = (long long)d->num_depths * d->h_out * d->w_out;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL(voxel_sample_kernel<float>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
                           st, g, (const float *)voxel_features, depths, (float *)out);
    else
        hipLaunchKernelGGL(voxel_sample_kernel<bf16_t>, dim3((unsigned)((N + 255) / 256)), dim3(256),
                           0, st, g, (const bf16_t *)voxel_features, depths, (bf16_t *)out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_voxel_sample_bwd(const dfm_vs_desc *d, const void *grad_out,
                                            const float *depths, float *grad_voxel_features,
                                            void *stream)
{
    VsGeom g;
    int rc = vs_geom(d, g);
    if (rc != DFM_OK) return rc;
    if (!grad_out || !depths || !grad_voxel_features)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const long long N = (long long)d->num_depths * d->h_out * d->w_out;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL(voxel_sample_bwd_kernel<float>, dim3((unsigned)((N + 255) / 256)), dim3(256),
                           0, st, g, (const float *)grad_out, depths, grad_voxel_features);
    else
        hipLaunchKernelGGL(voxel_sample_bwd_kernel<bf16_t>, dim3((unsigned)((N + 255) / 256)),
                           dim3(256), 0, st, g, (const bf16_t *)grad_out, depths, grad_voxel_features);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

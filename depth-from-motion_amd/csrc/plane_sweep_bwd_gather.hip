// plane_sweep_bwd_gather.hip -- backward of STRIDED fp32 sweeps (cost_sample_factor >= 2: config K), PREV map,
// as a GATHER (round 5).  Autograd of the second F.grid_sample of build_dfm_cost
// (mmdet3d/models/backbones/dfm_backbone.py:304-311).
//
// The scatter forms of this gradient all pay for the same thing: a lattice point's four taps move along
// its epipolar line from plane to plane, so every accumulation scheme that follows the lattice (LDS tiles
// over the map rows: 15 of 16 staged pixels receive nothing at csf = 4; per-wave windows flushed when the
// footprint moves; a dense pixel-major scatter -- DESIGN.md 4, profiles/r04_c40..c56) re-visits the lines
// of the 0.42 GB gradient map tens of times with the 2.4 GB of gradient volume streaming through the L2 in
// between: 7.7 ms of config K's 8.7 ms backward.
//
// Turned around: for a fixed depth plane the lattice -> map correspondence of the prev half is a HOMOGRAPHY
// (lattice pixel -> undo augmentation -> un-project at depth d -> rigid motion -> project -> augmentation:
// affine o projective o affine), and with a lattice step of ~4 map pixels a map pixel lies in the 2x2
// footprint of AT MOST one lattice point per plane (a few, when the step shrinks towards 2).  So a LANE OWNS A
// MAP PIXEL and walks the depth planes:
//   * (wf, hf) = Hinv_d (x, y): the lattice position that would sample exactly this pixel, and from the
//     Jacobian of Hinv_d the lattice distance within which a sample still touches the pixel (+ margin);
//   * the integer lattice points inside that box (none for 3 of 4 pixels, one for most of the rest) are
//     CANDIDATES: each is run through the FORWARD map in the forward's own fp32 op order (sweep_point_map<1>,
//     bwd_footprint) and counts only if its footprint really contains (x, y) with that tap in bounds -- the
//     set of (point, tap) pairs and the weights are exactly the forward's, whatever the inverse's rounding;
//   * a hit gathers the 32 channels of the gradient volume at (d, h, w) -- the lanes of a wave are 64
//     neighbouring pixels of a row: 16 neighbouring lattice points, one 64-byte run per channel -- into 32
//     fp32 accumulators in registers;
//   * after the last plane the pixel's 32 sums are STORED (reference layout, coalesced along x): no atomics,
//     no zero-initialised map, a deterministic summation order (plane by plane).
// Hinv_d comes from a fit: a pre-kernel evaluates the forward map at the four lattice corners per (sample,
// plane) and solves the 8 x 8 system in fp64, checks it on three interior points, and marks planes it cannot
// vouch for (a corner behind the camera or non-finite, a lattice step below ~1.1 pixels): those few planes
// are scattered with atomics by a second small kernel after the gather has written the map.
//
// Traffic: the prev half of the gradient volume is read once from HBM (a point's 2x2 pixels are two lanes of
// a wave and two rows of the same workgroup), the map is written once: 2.36 + 0.42 GB at config K.
#include <algorithm>

#include "dfm_common.h"

using namespace dfm;

namespace {

constexpr int GP_REC = 12;  // floats per (sample, plane): Hinv[9], ok, 2 spare
constexpr float GP_MAX_TOL = 1.75f;  // lattice half-width of the candidate box a plane may need (<= 4 per axis)

// ---- per (sample, plane): inverse homography of the prev half's lattice -> map correspondence ----------
__global__ __launch_bounds__(64) void gather_fit_kernel(SweepGeom g, SweepFast fast, int batch,
                                                        const float *__restrict__ depths,
                                                        const float *__restrict__ P,
                                                        const float *__restrict__ Pinv,
                                                        const float *__restrict__ Tm, float *__restrict__ planes)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= batch * g.D) return;
    const int b = i / g.D, d = i - b * g.D;
    float *rec = planes + (size_t)i * GP_REC;
    for (int k = 0; k < GP_REC; ++k) rec[k] = 0.0f;
    const float *Pb = P + b * 16, *Pib = Pinv + b * 16, *Tb = Tm + b * 16;
    const float depth = depths[d];
    const int wl = g.w_out - 1, hl = g.h_out - 1;
    if (wl < 1 || hl < 1) return;
    // four correspondences: lattice corners -> map positions, the forward's own arithmetic
    const int cw[4] = {0, wl, 0, wl}, ch[4] = {0, 0, hl, hl};
    double A[8][9];
    for (int k = 0; k < 4; ++k) {
        float sx, sy;
        sweep_point_map<1>(g, fast, Pb, Pib, Tb, depth, ch[k], cw[k], sx, sy);
        if (!(fabsf(sx) < 1.0e6f) || !(fabsf(sy) < 1.0e6f)) return;  // non-finite / far off: not vouched for
        const double u = cw[k], v = ch[k], X = sx, Y = sy;
        // X = (h0 u + h1 v + h2) / (h6 u + h7 v + 1), Y = (h3 u + h4 v + h5) / (...)
        double *r0 = A[2 * k], *r1 = A[2 * k + 1];
        r0[0] = u; r0[1] = v; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * X; r0[7] = -v * X; r0[8] = X;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = u; r1[4] = v; r1[5] = 1; r1[6] = -u * Y; r1[7] = -v * Y; r1[8] = Y;
    }
    // Gaussian elimination with partial pivoting, fp64
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        for (int r = c + 1; r < 8; ++r)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (!(fabs(A[piv][c]) > 1e-9)) return;
        if (piv != c)
            for (int j = 0; j < 9; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        const double inv = 1.0 / A[c][c];
        for (int j = c; j < 9; ++j) A[c][j] *= inv;
        for (int r = 0; r < 8; ++r)
            if (r != c) {
                const double f = A[r][c];
                if (f != 0.0)
                    for (int j = c; j < 9; ++j) A[r][j] -= f * A[c][j];
            }
    }
    const double H[9] = {A[0][8], A[1][8], A[2][8], A[3][8], A[4][8], A[5][8], A[6][8], A[7][8], 1.0};
    // inverse (adjugate), normalised to its largest entry
    double I[9] = {H[4] * H[8] - H[5] * H[7], H[2] * H[7] - H[1] * H[8], H[1] * H[5] - H[2] * H[4],
                   H[5] * H[6] - H[3] * H[8], H[0] * H[8] - H[2] * H[6], H[2] * H[3] - H[0] * H[5],
                   H[3] * H[7] - H[4] * H[6], H[1] * H[6] - H[0] * H[7], H[0] * H[4] - H[1] * H[3]};
    double mx = 0.0;
    for (int k = 0; k < 9; ++k) mx = fmax(mx, fabs(I[k]));
    if (!(mx > 1e-300) || !(mx < 1e300)) return;
    float hi[9];
    for (int k = 0; k < 9; ++k) hi[k] = (float)(I[k] / mx);
    // the fp32 inverse must bring three interior lattice points back to themselves, and the candidate box it
    // implies must stay small, at the corners and in the middle of the lattice
    const int tw[5] = {wl / 2, wl / 3, (2 * wl) / 3, 0, wl}, th[5] = {hl / 2, (2 * hl) / 3, hl / 3, 0, hl};
    for (int k = 0; k < 5; ++k) {
        float sx, sy;
        sweep_point_map<1>(g, fast, Pb, Pib, Tb, depth, th[k], tw[k], sx, sy);
        if (!(fabsf(sx) < 1.0e6f) || !(fabsf(sy) < 1.0e6f)) return;
        const float den = hi[6] * sx + hi[7] * sy + hi[8];
        const float inv = 1.0f / den;
        const float wf = (hi[0] * sx + hi[1] * sy + hi[2]) * inv, hf = (hi[3] * sx + hi[4] * sy + hi[5]) * inv;
        if (!(fabsf(wf - (float)tw[k]) < 0.02f) || !(fabsf(hf - (float)th[k]) < 0.02f)) return;
        const float tolw = (fabsf((hi[0] - wf * hi[6]) * inv) + fabsf((hi[1] - wf * hi[7]) * inv)) * 1.1f + 0.05f;
        const float tolh = (fabsf((hi[3] - hf * hi[6]) * inv) + fabsf((hi[4] - hf * hi[7]) * inv)) * 1.1f + 0.05f;
        if (!(tolw < GP_MAX_TOL) || !(tolh < GP_MAX_TOL)) return;
    }
    for (int k = 0; k < 9; ++k) rec[k] = hi[k];
    rec[9] = 1.0f;
}

// ---- the gather: lane = map pixel, 32 channels per pass -----------------------------------------------
__global__ __launch_bounds__(256) void sweep_bwd_prev_gather_kernel(
    SweepGeom g, SweepFast fast, int batch, int passes, int xtiles, const float *__restrict__ planes,
    const float *__restrict__ gout, const float *__restrict__ depths, const float *__restrict__ P,
    const float *__restrict__ Pinv, const float *__restrict__ Tm, float *__restrict__ gprev)
{
    // block id = ((ytile * xtiles + xtile) * passes + pass) * B + b: sample fastest (id % 8 == XCD keeps a
    // sample's gradient volume in one L2), then the channel passes of one pixel tile
    int t = blockIdx.x;
    const int b = t % batch;
    t /= batch;
    const int pass = t % passes;
    t /= passes;
    const int xt = t % xtiles, yt = t / xtiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int H = g.h_in, W = g.w_in;
    const int y = yt * 4 + wave, x = xt * 64 + lane;
    const bool inside = y < H && x < W;
    const float xf = (float)x, yf = (float)y;
    const int hw = g.h_out * g.w_out;
    const size_t cstride = (size_t)g.D * hw;  // elements between channels of the volume
    const float *gb = gout + ((size_t)b * 2 * g.C + g.C + (size_t)pass * 32) * cstride;
    const float *Pb = P + b * 16, *Pib = Pinv + b * 16, *Tb = Tm + b * 16;
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.0f;
    for (int d = 0; d < g.D; ++d) {
        const float *rec = planes + ((size_t)b * g.D + d) * GP_REC;  // uniform: scalar loads
        if (rec[9] == 0.0f) continue;  // a plane the fit does not vouch for: scattered by the fallback kernel
        const float den = rec[6] * xf + rec[7] * yf + rec[8];
        const float inv = 1.0f / den;
        const float wf = (rec[0] * xf + rec[1] * yf + rec[2]) * inv, hf = (rec[3] * xf + rec[4] * yf + rec[5]) * inv;
        // a sample covers this pixel when it lies within one pixel of it on both axes: in lattice units, the
        // Jacobian of the inverse times that box (+ 10 % and 0.05 for its curvature and the fp32 inverse)
        const float tolw = (fabsf((rec[0] - wf * rec[6]) * inv) + fabsf((rec[1] - wf * rec[7]) * inv)) * 1.1f + 0.05f;
        const float tolh = (fabsf((rec[3] - hf * rec[6]) * inv) + fabsf((rec[4] - hf * rec[7]) * inv)) * 1.1f + 0.05f;
        // candidate lattice box [w0, w1] x [h0, h1] (empty for most pixels); non-finite wf / hf compare false
        const float w0f = ceilf(wf - tolw), w1f = floorf(wf + tolw), h0f = ceilf(hf - tolh), h1f = floorf(hf + tolh);
        const bool some = inside && w0f <= w1f && h0f <= h1f && w1f >= 0.0f && h1f >= 0.0f &&
                          w0f <= (float)(g.w_out - 1) && h0f <= (float)(g.h_out - 1) && tolw < 2.0f && tolh < 2.0f;
        if (!__any(some)) continue;
        const int w0 = some ? max((int)w0f, 0) : 0, w1 = some ? min((int)w1f, g.w_out - 1) : -1;
        const int h0 = some ? max((int)h0f, 0) : 0, h1 = some ? min((int)h1f, g.h_out - 1) : -1;
        const float depth = depths[d];
        const float *gd = gb + (size_t)d * hw;
        for (int kh = 0; kh < 4; ++kh) {
            const int lh = h0 + kh;
            if (!__any(lh <= h1)) break;
            for (int kw = 0; kw < 4; ++kw) {
                const int lw = w0 + kw;
                const bool cand = lh <= h1 && lw <= w1;
                if (!__any(cand)) break;
                if (cand) {
                    // the forward's own arithmetic decides: does this lattice point's footprint hold (x, y)?
                    float sx, sy, fw, fn;
                    sweep_point_map<1>(g, fast, Pb, Pib, Tb, depth, lh, lw, sx, sy);
                    const uint32_t f = bwd_footprint(sx, sy, H, W, fw, fn);
                    const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
                    const int dx = x - ixw, dy = y - iyn;
                    // (a valid footprint has its taps inside the map or masked; this pixel is inside the map, so
                    //  a tap that equals it is in bounds: the ok bits agree by construction, checked anyway)
                    const bool colok = dx == 0 ? (f & (1u << 27)) != 0 : (f & (1u << 28)) != 0;
                    const bool rowok = dy == 0 ? (f & (1u << 29)) != 0 : (f & (1u << 30)) != 0;
                    if (f != 0u && (unsigned)dx <= 1u && (unsigned)dy <= 1u && colok && rowok) {
                        // ATen's weights: (row factor) * (column factor), nw = (1 - fn) * (1 - fw) ...
                        const float wgt = (dy ? fn : 1.0f - fn) * (dx ? fw : 1.0f - fw);
                        const float *gp = gd + (size_t)lh * g.w_out + lw;
                        float v[32];
#pragma unroll
                        for (int c = 0; c < 32; ++c) v[c] = gp[(size_t)c * cstride];  // (cacheable: the row below re-reads it)
#pragma unroll
                        for (int c = 0; c < 32; ++c) acc[c] += v[c] * wgt;  // (0 x Inf = NaN reaches the tap, as in ATen)
                    }
                }
            }
        }
    }
    if (inside) {
        float *o = gprev + (((size_t)b * g.C + (size_t)pass * 32) * H + y) * W + x;
        const size_t ps = (size_t)H * W;
#pragma unroll
        for (int c = 0; c < 32; ++c) o[(size_t)c * ps] = acc[c];
    }
}

// ---- planes without a vouched-for inverse: plain scatter with atomics (rare; runs after the gather) ------
__global__ __launch_bounds__(256) void sweep_bwd_prev_scatter_planes_kernel(
    SweepGeom g, SweepFast fast, int batch, const float *__restrict__ planes, const float *__restrict__ gout,
    const float *__restrict__ depths, const float *__restrict__ P, const float *__restrict__ Pinv,
    const float *__restrict__ Tm, float *__restrict__ gprev)
{
    const int i = blockIdx.x;  // (sample, plane)
    if (planes[(size_t)i * GP_REC + 9] != 0.0f) return;
    const int b = i / g.D, d = i - b * g.D;
    const int hw = g.h_out * g.w_out, H = g.h_in, W = g.w_in;
    const size_t cstride = (size_t)g.D * hw;
    const float *gd = gout + ((size_t)b * 2 * g.C + g.C) * cstride + (size_t)d * hw;
    const float depth = depths[d];
    for (int p = blockIdx.y * 256 + threadIdx.x; p < hw; p += gridDim.y * 256) {
        const int lh = p / g.w_out, lw = p - lh * g.w_out;
        float sx, sy, fw, fn;
        sweep_point_map<1>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, depth, lh, lw, sx, sy);
        const uint32_t f = bwd_footprint(sx, sy, H, W, fw, fn);
        if (f == 0u) continue;
        const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
        const bool ok[4] = {(f & (1u << 27)) && (f & (1u << 29)), (f & (1u << 28)) && (f & (1u << 29)),
                            (f & (1u << 27)) && (f & (1u << 30)), (f & (1u << 28)) && (f & (1u << 30))};
        const float wq[4] = {(1.0f - fn) * (1.0f - fw), (1.0f - fn) * fw, fn * (1.0f - fw), fn * fw};
        for (int c = 0; c < g.C; ++c) {
            const float gv = gd[(size_t)c * cstride + p];
            float *m = gprev + ((size_t)b * g.C + c) * H * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) atomicAdd(m + (size_t)(iyn + (k >> 1)) * W + ixw + (k & 1), gv * wq[k]);
        }
    }
}

}  // namespace

extern "C" {

DFM_API size_t dfm_plane_sweep_bwd_prev_gather_workspace_bytes(const dfm_sweep_desc *d)
{
    if (!d || d->batch <= 0 || d->num_depths <= 0) return 0;
    return (((size_t)d->batch * d->num_depths * GP_REC * sizeof(float)) + 255) & ~(size_t)255;
}

DFM_API int dfm_plane_sweep_bwd_prev_gather(const dfm_sweep_desc *d, const void *grad_out, const float *depths,
                                            const float *cam2img, const float *cam2img_inv, const float *cur2prev,
                                            float *grad_prev, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = sweep_check_desc(d);
    if (rc != DFM_OK) return rc;
    if (!grad_out || !depths || !cam2img || !cam2img_inv || !cur2prev || !grad_prev || !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < dfm_plane_sweep_bwd_prev_gather_workspace_bytes(d))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_plane_sweep_bwd_prev_gather_workspace_bytes");
    if (d->dtype != DFM_F32 || d->channels % 32 || d->cost_sample_factor < 1.5f || d->h_in >= 4096 || d->w_in >= 8192 ||
        d->h_out < 2 || d->w_out < 2)
        return set_error(DFM_ERR_UNSUPPORTED,
                         "prev-map gather backward: fp32, channels % 32 == 0, cost_sample_factor >= 2, lattice >= 2 x 2");
    const SweepGeom g = sweep_make_geom(d);
    const SweepFast fast = sweep_make_fast(d);
    hipStream_t st = (hipStream_t)stream;
    float *planes = (float *)workspace;
    const int np = d->batch * d->num_depths;
    hipLaunchKernelGGL(gather_fit_kernel, dim3((np + 63) / 64), dim3(64), 0, st, g, fast, d->batch, depths, cam2img,
                       cam2img_inv, cur2prev, planes);
    const int xtiles = (d->w_in + 63) / 64, ytiles = (d->h_in + 3) / 4, passes = d->channels / 32;
    const long long nb = (long long)xtiles * ytiles * passes * d->batch;
    if (nb > 2147483647ll) return set_error(DFM_ERR_UNSUPPORTED, "feature map too large");
    hipLaunchKernelGGL(sweep_bwd_prev_gather_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, fast, d->batch, passes,
                       xtiles, (const float *)planes, (const float *)grad_out, depths, cam2img, cam2img_inv, cur2prev,
                       grad_prev);
    const int ychunks = std::max(1, std::min(64, (d->h_out * d->w_out + 255) / 256));
    hipLaunchKernelGGL(sweep_bwd_prev_scatter_planes_kernel, dim3(np, ychunks), dim3(256), 0, st, g, fast, d->batch,
                       (const float *)planes, (const float *)grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_prev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    sweep_set_last_bwd_kernel(9);
    return DFM_OK;
}

}  // extern "C"

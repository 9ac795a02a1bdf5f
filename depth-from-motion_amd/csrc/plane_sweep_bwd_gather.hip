// plane_sweep_bwd_gather.hip -- backward of STRIDED fp32 sweeps (cost_sample_factor >= 2: config K), PREV map,
// as a GATHER (round 5).  Autograd of the second F.grid_sample of build_dfm_cost
// (mmdet3d/models/backbones/dfm_backbone.py:304-311).
//
// The scatter forms of this gradient all pay for the same thing: a lattice point's four taps move along
// its epipolar line from plane to plane, so every accumulation scheme that follows the lattice (LDS tiles
// over the map rows: 15 of 16 staged pixels receive nothing at csf = 4; per-wave windows flushed when the
// footprint moves; a dense pixel-major scatter -- DESIGN.md 4, profiles/archive/r04_c40..c56) re-visits the lines
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
// The same kernel serves the CUR half and gradient volumes in the NDHWC stack's own layout (bf16 or fp32,
// channels-last: a hit is ONE contiguous run of 32 channels instead of 32 loads a channel plane apart -- the
// reference-layout form is bound by the number of load instructions, not by bytes), writing the map gradient
// in the reference layout or pixel-major: dfm_plane_sweep_bwd_gather.
//
// Traffic: the prev half of the gradient volume is read once from HBM (a point's 2x2 pixels are two lanes of
// a wave and two rows of the same workgroup), the map is written once: 2.36 + 0.42 GB at config K.
#include <algorithm>
#include <cstdlib>

#include "dfm_common.h"

using namespace dfm;

namespace {

constexpr int GP_REC = 12;  // floats per (sample, plane): Hinv[9], ok, 2 spare
constexpr float GP_MAX_TOL = 1.75f;  // lattice half-width of the candidate box a plane may need (<= 4 per axis)

// ---- per (sample, plane): inverse homography of one half's lattice -> map correspondence ----------------
template <int HALF>
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
        sweep_point_map<HALF>(g, fast, Pb, Pib, Tb, depth, ch[k], cw[k], sx, sy);
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
        sweep_point_map<HALF>(g, fast, Pb, Pib, Tb, depth, th[k], tw[k], sx, sy);
        if (!(fabsf(sx) < 1.0e6f) || !(fabsf(sy) < 1.0e6f)) return;
        const float den = hi[6] * sx + hi[7] * sy + hi[8];
        const float inv = 1.0f / den;
        const float wf = (hi[0] * sx + hi[1] * sy + hi[2]) * inv, hf = (hi[3] * sx + hi[4] * sy + hi[5]) * inv;
        if (!(fabsf(wf - (float)tw[k]) < 0.02f) || !(fabsf(hf - (float)th[k]) < 0.02f)) return;
        const float tolw = (fabsf((hi[0] - wf * hi[6]) * inv) + fabsf((hi[1] - wf * hi[7]) * inv)) * 1.1f + 0.05f;
        const float tolh = (fabsf((hi[3] - hf * hi[6]) * inv) + fabsf((hi[4] - hf * hi[7]) * inv)) * 1.1f + 0.05f;
        if (!(tolw < GP_MAX_TOL) || !(tolh < GP_MAX_TOL)) return;
    }
    // ... and the box must stay small at EVERY pixel of the map, not only at those sample points (the
    // gather's loops hold 4 x 4 candidates).  With den = h6 x + h7 y + h8 the inverse's derivatives are
    //   d wf / dx = ((h0 h7 - h1 h6) y + (h0 h8 - h2 h6)) / den^2,  d wf / dy = ((h1 h6 - h0 h7) x + (h1 h8 - h2 h7)) / den^2
    // (hf: rows 3..5): affine numerators over an affine denominator of one sign -- each factor takes its extreme at
    // an end of the map's x / y range, so (max |num_x| + max |num_y|) / min den^2 bounds the gather's tolerance
    // (same 1.1 and 0.05) over the whole map.  A plane that fails it goes to the scatter kernel.
    {
        const double Wm = (double)(g.w_in - 1), Hm = (double)(g.h_in - 1);
        double dmin = 1e300;
        bool pos = false, neg = false;
        for (int k = 0; k < 4; ++k) {
            const double den = (double)hi[6] * ((k & 1) ? Wm : 0.0) + (double)hi[7] * ((k >> 1) ? Hm : 0.0) + (double)hi[8];
            pos = pos || den > 0.0;
            neg = neg || den < 0.0;
            dmin = fmin(dmin, fabs(den));
        }
        if ((pos && neg) || !(dmin > 1e-30)) return;
        for (int r = 0; r < 2; ++r) {
            const double a = hi[3 * r], bq = hi[3 * r + 1], c = hi[3 * r + 2];
            const double A1 = a * hi[7] - bq * hi[6], A0 = a * hi[8] - c * hi[6];
            const double B1 = bq * hi[6] - a * hi[7], B0 = bq * hi[8] - c * hi[7];
            const double mA = fmax(fabs(A0), fabs(A1 * Hm + A0)), mB = fmax(fabs(B0), fabs(B1 * Wm + B0));
            if (!((mA + mB) / (dmin * dmin) * 1.1 + 0.05 < (double)GP_MAX_TOL)) return;
        }
    }
    for (int k = 0; k < 9; ++k) rec[k] = hi[k];
    rec[9] = 1.0f;
}

// the 32 channels of one lattice point of the gradient volume, as fp32
//   CL == false: the reference layout (B, 2C, D, h, w): 32 loads a channel plane apart (cstride elements)
//   CL == true : channels-last (B, D, h, w, 2C) -- what the NDHWC aggregation stack's backward hands over: ONE
//                contiguous run of 32 * sizeof(T) bytes, 16-byte loads
template <typename T, bool CL>
__device__ __forceinline__ void gather_load32(const T *__restrict__ gp, size_t cstride, float (&v)[32])
{
    if constexpr (CL) {
        constexpr int VEC = 16 / (int)sizeof(T);
#pragma unroll
        for (int q = 0; q < 32 / VEC; ++q) {
            const uint4 u = *(const uint4 *)(gp + q * VEC);
            if constexpr (sizeof(T) == 4) {
                v[4 * q] = __uint_as_float(u.x); v[4 * q + 1] = __uint_as_float(u.y);
                v[4 * q + 2] = __uint_as_float(u.z); v[4 * q + 3] = __uint_as_float(u.w);
            } else {
                const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[8 * q + 2 * k] = __uint_as_float(w4[k] << 16);
                    v[8 * q + 2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u);
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) {  // (cacheable: the map row below re-reads the line)
            if constexpr (sizeof(T) == 4) v[c] = __uint_as_float(*(const uint32_t *)(gp + (size_t)c * cstride));
            else v[c] = __uint_as_float((uint32_t)(*(const uint16_t *)(gp + (size_t)c * cstride)) << 16);
        }
    }
}

// ---- pre-pass of the gather: the footprint of every (sample, plane, lattice point), 12 bytes each -------
// {bwd_footprint's packed corner + in-bounds bits, fw, fn}: the forward's own arithmetic, once per pair
template <int HALF>
__global__ __launch_bounds__(256) void gather_foot_kernel(SweepGeom g, SweepFast fast, int batch,
                                                          const float *__restrict__ planes,
                                                          const float *__restrict__ depths, const float *__restrict__ P,
                                                          const float *__restrict__ Pinv, const float *__restrict__ Tm,
                                                          uint32_t *__restrict__ foot)
{
    const int hw = g.h_out * g.w_out;
    const int bd = blockIdx.y;  // (sample, plane)
    if (planes[(size_t)bd * GP_REC + 9] == 0.0f) return;  // (the fallback scatter takes this plane)
    const int b = bd / g.D, d = bd - b * g.D;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int lh = p / g.w_out, lw = p - lh * g.w_out;
    float sx, sy, fw, fn;
    sweep_point_map<HALF>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], lh, lw, sx, sy);
    const uint32_t f = bwd_footprint(sx, sy, g.h_in, g.w_in, fw, fn);
    uint32_t *e = foot + ((size_t)bd * hw + p) * 3;
    e[0] = f; e[1] = __float_as_uint(fw); e[2] = __float_as_uint(fn);
}

// ---- the gather: lane = map pixel, 32 channels per pass -----------------------------------------------
// HALF: 0 cur map (channels [0, C) of the volume; its sample positions do not move with depth, the same
// kernel walks them anyway: the footprint flips between neighbouring pixel pairs with the rounding of each
// plane), 1 prev map.  T: dtype of the gradient volume (fp32 | bf16 bits).  CL_IN: the volume is channels-last.
// CL_OUT: the map gradient is written pixel-major (B, H, W, C) -- a lane's 32 sums are one 128-byte run --
// instead of the reference layout (B, C, H, W).
// TABLE (round 6): the footprints of all (plane, lattice point) pairs come from a table a pre-pass wrote
// (gather_foot_kernel: the forward's projection and bwd_footprint, ONCE per pair) -- a candidate costs a 12-byte load
// and a few integer compares instead of ~80 instructions of projection.  The wave executes a candidate slot whenever
// ANY of its lanes has a candidate there, so that arithmetic ran ~3 times per wave and plane whatever the hit rate:
// 1.7 of the kernel's 2.8 ms at config K (profiles/r06_c41_*).  Without the table (a workspace that is too small):
// the projection per candidate, as before.
template <int HALF, typename T, bool CL_IN, bool CL_OUT, bool TABLE>
__global__ __launch_bounds__(256) void sweep_bwd_gather_kernel(
    SweepGeom g, SweepFast fast, int batch, int passes, int xtiles, const float *__restrict__ planes,
    const T *__restrict__ gout, const float *__restrict__ depths, const float *__restrict__ P,
    const float *__restrict__ Pinv, const float *__restrict__ Tm, float *__restrict__ gmap,
    const uint32_t *__restrict__ foot)
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
    const int C2 = 2 * g.C, ch0 = HALF * g.C + pass * 32;
    const size_t cstride = (size_t)g.D * hw;  // reference layout: elements between channels of the volume
    // element (d, lh, lw) of this pass's first channel: gb + (d * hw + lh * w_out + lw) * pstride
    const T *gb = CL_IN ? gout + (size_t)b * cstride * C2 + ch0 : gout + ((size_t)b * C2 + ch0) * cstride;
    const size_t pstride = CL_IN ? (size_t)C2 : 1;
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
                          w0f <= (float)(g.w_out - 1) && h0f <= (float)(g.h_out - 1);
        if (!__any(some)) continue;
        // (clamped to the lattice in fp32 before the int conversion.)  A box holds at most 4 x 4 points: the fit kernel
        // vouches for a plane only after bounding the tolerance over EVERY pixel of the map below GP_MAX_TOL (1.75:
        // 2 tol + 1 < 4.5 points per axis) -- ADVICE round 5: it used to check five sample points only, while this
        // loop cut larger boxes silently; a plane that fails the bound is scattered by the fallback kernel instead.
        const int w0 = some ? (int)fmaxf(w0f, 0.0f) : 0, w1 = some ? (int)fminf(w1f, (float)(g.w_out - 1)) : -1;
        const int h0 = some ? (int)fmaxf(h0f, 0.0f) : 0, h1 = some ? (int)fminf(h1f, (float)(g.h_out - 1)) : -1;
        const float depth = depths[d];
        const T *gd = gb + (size_t)d * hw * pstride;
        // one candidate lattice point: the forward's own arithmetic decides whether its footprint holds (x, y)
        // (recording the hits and gathering the k-th hit of every lane together afterwards measured 18-26 % slower:
        //  the gather's loads then start after the whole candidate walk, profiles/r06_c41_*)
        auto visit = [&](int lh, int lw) {
            float fw, fn;
            uint32_t f;
            if constexpr (TABLE) {
                const uint32_t *e = foot + (((size_t)b * g.D + d) * hw + (size_t)lh * g.w_out + lw) * 3;
                f = e[0]; fw = __uint_as_float(e[1]); fn = __uint_as_float(e[2]);
            } else {
                float sx, sy;
                sweep_point_map<HALF>(g, fast, Pb, Pib, Tb, depth, lh, lw, sx, sy);
                f = bwd_footprint(sx, sy, H, W, fw, fn);
            }
            const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
            const int dx = x - ixw, dy = y - iyn;
            // (a valid footprint has its taps inside the map or masked; this pixel is inside the map, so
            //  a tap that equals it is in bounds: the ok bits agree by construction, checked anyway)
            const bool colok = dx == 0 ? (f & (1u << 27)) != 0 : (f & (1u << 28)) != 0;
            const bool rowok = dy == 0 ? (f & (1u << 29)) != 0 : (f & (1u << 30)) != 0;
            if (f != 0u && (unsigned)dx <= 1u && (unsigned)dy <= 1u && colok && rowok) {
                // ATen's weights: (row factor) * (column factor), nw = (1 - fn) * (1 - fw) ...
                const float wgt = (dy ? fn : 1.0f - fn) * (dx ? fw : 1.0f - fw);
                float v[32];
                gather_load32<T, CL_IN>(gd + ((size_t)lh * g.w_out + lw) * pstride, cstride, v);
#pragma unroll
                for (int c = 0; c < 32; ++c) acc[c] += v[c] * wgt;  // (0 x Inf = NaN reaches the tap, as in ATen)
            }
        };
        // at most 4 x 4 candidates: the fit kernel vouches for a plane only if NO pixel of the map can have a larger
        // box (gather_fit_kernel: a bound on the tolerance over the whole map, below GP_MAX_TOL)
        for (int kh = 0; kh < 4; ++kh) {
            const int lh = h0 + kh;
            if (!__any(lh <= h1)) break;
            for (int kw = 0; kw < 4; ++kw) {
                const int lw = w0 + kw;
                const bool cand = lh <= h1 && lw <= w1;
                if (!__any(cand)) break;
                if (cand) visit(lh, lw);
            }
        }
    }
    if (inside) {
        if constexpr (CL_OUT) {
            float *o = gmap + (((size_t)b * H + y) * W + x) * g.C + (size_t)pass * 32;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *(float4 *)(o + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        } else {
            float *o = gmap + (((size_t)b * g.C + (size_t)pass * 32) * H + y) * W + x;
            const size_t ps = (size_t)H * W;
#pragma unroll
            for (int c = 0; c < 32; ++c) o[(size_t)c * ps] = acc[c];
        }
    }
}

// ---- planes without a vouched-for inverse: plain scatter with atomics (rare; runs after the gather) ------
template <int HALF, typename T, bool CL_IN, bool CL_OUT>
__global__ __launch_bounds__(256) void sweep_bwd_scatter_planes_kernel(
    SweepGeom g, SweepFast fast, int batch, const float *__restrict__ planes, const T *__restrict__ gout,
    const float *__restrict__ depths, const float *__restrict__ P, const float *__restrict__ Pinv,
    const float *__restrict__ Tm, float *__restrict__ gmap)
{
    const int i = blockIdx.x;  // (sample, plane)
    if (planes[(size_t)i * GP_REC + 9] != 0.0f) return;
    const int b = i / g.D, d = i - b * g.D;
    const int hw = g.h_out * g.w_out, H = g.h_in, W = g.w_in;
    const int C2 = 2 * g.C;
    const size_t cstride = (size_t)g.D * hw;
    const T *gd = CL_IN ? gout + ((size_t)b * cstride + (size_t)d * hw) * C2 + HALF * g.C
                        : gout + ((size_t)b * C2 + HALF * g.C) * cstride + (size_t)d * hw;
    const float depth = depths[d];
    for (int p = blockIdx.y * 256 + threadIdx.x; p < hw; p += gridDim.y * 256) {
        const int lh = p / g.w_out, lw = p - lh * g.w_out;
        float sx, sy, fw, fn;
        sweep_point_map<HALF>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, depth, lh, lw, sx, sy);
        const uint32_t f = bwd_footprint(sx, sy, H, W, fw, fn);
        if (f == 0u) continue;
        const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
        const bool ok[4] = {(f & (1u << 27)) && (f & (1u << 29)), (f & (1u << 28)) && (f & (1u << 29)),
                            (f & (1u << 27)) && (f & (1u << 30)), (f & (1u << 28)) && (f & (1u << 30))};
        const float wq[4] = {(1.0f - fn) * (1.0f - fw), (1.0f - fn) * fw, fn * (1.0f - fw), fn * fw};
        for (int c = 0; c < g.C; ++c) {
            const T raw = CL_IN ? gd[(size_t)p * C2 + c] : gd[(size_t)c * cstride + p];
            float gv;
            if constexpr (sizeof(T) == 4) gv = raw;
            else gv = __uint_as_float((uint32_t)raw << 16);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) {
                    const size_t py = (size_t)(iyn + (k >> 1)), px = (size_t)(ixw + (k & 1));
                    float *m = CL_OUT ? gmap + (((size_t)b * H + py) * W + px) * g.C + c
                                      : gmap + (((size_t)b * g.C + c) * H + py) * W + px;
                    atomicAdd(m, gv * wq[k]);
                }
        }
    }
}

template <int HALF, typename T, bool CL_IN, bool CL_OUT>
int gather_launch(const dfm_sweep_desc *d, const void *grad_out, const float *depths, const float *cam2img,
                  const float *cam2img_inv, const float *cur2prev, float *grad_map, float *planes, uint32_t *foot,
                  hipStream_t st)
{
    const SweepGeom g = sweep_make_geom(d);
    const SweepFast fast = sweep_make_fast(d);
    const int np = d->batch * d->num_depths;
    hipLaunchKernelGGL(gather_fit_kernel<HALF>, dim3((np + 63) / 64), dim3(64), 0, st, g, fast, d->batch, depths,
                       cam2img, cam2img_inv, cur2prev, planes);
    const int xtiles = (d->w_in + 63) / 64, ytiles = (d->h_in + 3) / 4, passes = d->channels / 32;
    const long long nb = (long long)xtiles * ytiles * passes * d->batch;
    if (nb > 2147483647ll) return set_error(DFM_ERR_UNSUPPORTED, "feature map too large");
    if (foot) {
        const int hwl = d->h_out * d->w_out;
        hipLaunchKernelGGL(gather_foot_kernel<HALF>, dim3((hwl + 255) / 256, np), dim3(256), 0, st, g, fast, d->batch,
                           (const float *)planes, depths, cam2img, cam2img_inv, cur2prev, foot);
        hipLaunchKernelGGL((sweep_bwd_gather_kernel<HALF, T, CL_IN, CL_OUT, true>), dim3((unsigned)nb), dim3(256), 0, st,
                           g, fast, d->batch, passes, xtiles, (const float *)planes, (const T *)grad_out, depths, cam2img,
                           cam2img_inv, cur2prev, grad_map, (const uint32_t *)foot);
    } else {
        hipLaunchKernelGGL((sweep_bwd_gather_kernel<HALF, T, CL_IN, CL_OUT, false>), dim3((unsigned)nb), dim3(256), 0, st,
                           g, fast, d->batch, passes, xtiles, (const float *)planes, (const T *)grad_out, depths, cam2img,
                           cam2img_inv, cur2prev, grad_map, (const uint32_t *)nullptr);
    }
    const int ychunks = std::max(1, std::min(64, (d->h_out * d->w_out + 255) / 256));
    hipLaunchKernelGGL((sweep_bwd_scatter_planes_kernel<HALF, T, CL_IN, CL_OUT>), dim3(np, ychunks), dim3(256), 0, st, g,
                       fast, d->batch, (const float *)planes, (const T *)grad_out, depths, cam2img, cam2img_inv,
                       cur2prev, grad_map);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

}  // namespace

extern "C" {

static size_t gather_planes_bytes(const dfm_sweep_desc *d)
{
    return (((size_t)d->batch * d->num_depths * GP_REC * sizeof(float)) + 255) & ~(size_t)255;
}

// the footprint table: 12 bytes per (sample, plane, lattice point); 0 when it would not fit 2 GiB (the gather then
// projects per candidate)
static size_t gather_foot_bytes(const dfm_sweep_desc *d)
{
    if (d->h_out <= 0 || d->w_out <= 0) return 0;
    const size_t n = (size_t)d->batch * d->num_depths * d->h_out * d->w_out * 12;
    if (n > ((size_t)2 << 30) || getenv("DFM_GATHER_NO_TABLE")) return 0;
    return (n + 255) & ~(size_t)255;
}

// the plane records, then (round 6) the footprint table; a workspace of the plane records' size alone is accepted too
// (dfm_plane_sweep_bwd_gather then runs without the table)
DFM_API size_t dfm_plane_sweep_bwd_prev_gather_workspace_bytes(const dfm_sweep_desc *d)
{
    if (!d || d->batch <= 0 || d->num_depths <= 0) return 0;
    return gather_planes_bytes(d) + gather_foot_bytes(d);
}

DFM_API int dfm_plane_sweep_bwd_gather(const dfm_sweep_desc *d, int32_t half, const void *grad_out,
                                       int32_t grad_channels_last, const float *depths, const float *cam2img,
                                       const float *cam2img_inv, const float *cur2prev, float *grad_map,
                                       int32_t map_pixel_major, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = sweep_check_desc(d);
    if (rc != DFM_OK) return rc;
    if (!grad_out || !depths || !cam2img || !cam2img_inv || !cur2prev || !grad_map || !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (half != 0 && half != 1) return set_error(DFM_ERR_INVALID_ARG, "half is 0 (cur map) or 1 (prev map)");
    if (workspace_bytes < gather_planes_bytes(d))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than the plane records of dfm_plane_sweep_bwd_prev_gather_workspace_bytes");
    // (DFM_GATHER_DENSE=1: dense sweeps too -- cost_sample_factor 1, several hits per pixel and plane; experiments)
    const char *dense_e = getenv("DFM_GATHER_DENSE");
    const bool dense_ok = dense_e && dense_e[0] == '1';
    if (d->channels % 32 || (d->cost_sample_factor < 1.5f && !dense_ok) || d->h_in >= 4096 || d->w_in >= 8192 || d->h_out < 2 ||
        d->w_out < 2 || (grad_channels_last && ((uintptr_t)grad_out & 15)) || (map_pixel_major && ((uintptr_t)grad_map & 15)))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "gather backward: channels % 32 == 0, cost_sample_factor >= 2, lattice >= 2 x 2, 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float *planes = (float *)workspace;
    const size_t fbytes = gather_foot_bytes(d);
    uint32_t *foot = (fbytes && workspace_bytes >= gather_planes_bytes(d) + fbytes)
                         ? (uint32_t *)((char *)workspace + gather_planes_bytes(d)) : nullptr;
#define DFM_GL(H_, T_, CI_, CO_) \
    rc = gather_launch<H_, T_, CI_, CO_>(d, grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_map, planes, foot, st)
#define DFM_GL_T(H_, CI_, CO_)                                \
    do {                                                      \
        if (d->dtype == DFM_BF16) DFM_GL(H_, uint16_t, CI_, CO_); \
        else DFM_GL(H_, float, CI_, CO_);                     \
    } while (0)
#define DFM_GL_L(H_)                                                          \
    do {                                                                      \
        if (grad_channels_last && map_pixel_major) DFM_GL_T(H_, true, true);  \
        else if (grad_channels_last) DFM_GL_T(H_, true, false);               \
        else if (map_pixel_major) DFM_GL_T(H_, false, true);                  \
        else DFM_GL_T(H_, false, false);                                      \
    } while (0)
    if (half) DFM_GL_L(1);
    else DFM_GL_L(0);
#undef DFM_GL_L
#undef DFM_GL_T
#undef DFM_GL
    if (rc != DFM_OK) return rc;
    sweep_set_last_bwd_kernel(9);
    return DFM_OK;
}

DFM_API int dfm_plane_sweep_bwd_prev_gather(const dfm_sweep_desc *d, const void *grad_out, const float *depths,
                                            const float *cam2img, const float *cam2img_inv, const float *cur2prev,
                                            float *grad_prev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (d && d->dtype != DFM_F32)
        return set_error(DFM_ERR_UNSUPPORTED, "prev-map gather backward (reference layout): fp32 gradient volumes");
    return dfm_plane_sweep_bwd_gather(d, 1, grad_out, 0, depths, cam2img, cam2img_inv, cur2prev, grad_prev, 0, workspace,
                                      workspace_bytes, stream);
}

}  // extern "C"

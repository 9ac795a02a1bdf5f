// dfm_common.h -- shared device helpers for the gfx950 plane-sweep kernels.
//
// Numerics contract (DESIGN.md "Numerics"): all coordinate arithmetic is fp32
// with exactly one IEEE rounding per reference torch op.  This translation
// unit is compiled with -ffp-contract=off; fused multiply-adds appear ONLY
// where written as __builtin_fmaf (they restate the k-ordered fma chain of
// torch's fp32 (N,4)@(4,4) CPU matmul and of ATen's bilinear accumulation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfm_hip.h"

namespace dfm {

// records the thread-local message dfm_last_error() returns; defined in
// plane_sweep.hip, shared by every translation unit of the library
int set_error(int code, const char *msg);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): defined in
// plane_sweep.hip (mutex-protected map), used by every kernel with more than 64 KB of LDS
int ensure_dynamic_lds(const void *kern, int lds_bytes);
struct SweepGeom;
// defined in plane_sweep.hip, used by the other plane-sweep translation units
int sweep_check_desc(const dfm_sweep_desc *d);
SweepGeom sweep_make_geom(const dfm_sweep_desc *d);
void sweep_set_last_kernel(int which);  // what dfm_plane_sweep_last_kernel() reports
void sweep_set_last_bwd_kernel(int which);  // what dfm_plane_sweep_bwd_last_kernel() reports
// strided sweeps in the reference layout: pixel-major taps + LDS transpose (plane_sweep_cl.hip)
bool sweep_clt_supported(const dfm_sweep_desc *d, const void *out);
size_t sweep_clt_workspace_bytes(const dfm_sweep_desc *d);
int sweep_clt_launch(const dfm_sweep_desc *d, const void *cur, const void *prev, const float *depths,
                     const float *cam2img, const float *cam2img_inv, const float *cur2prev, void *out,
                     void *workspace, void *stream, bool nhwc = false, bool walk = true);
bool sweep_cltw_supported(const dfm_sweep_desc *d, const void *out);
// records the start (stop=false) / stop event of a timed launch when dfm_profile_begin
// is active; the start call returns whether this launch is being timed
bool profile_mark(void *stream, bool stop);

typedef unsigned short bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v)
{
    return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950: one operation; the integer
// add-and-shift it replaces cost seven); a NaN stays a quiet NaN (torch's c10::BFloat16 canonicalises
// it to 0x7fc0, the hardware keeps its sign -- NaN either way)
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    const __bf16 h = (__bf16)f;
    bf16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// two floats -> packed bf16x2 (lo = a, hi = b), round-to-nearest-even
// (v_cvt_pk_bf16_f32 on gfx950)
typedef __bf16 bf16x2_vec __attribute__((ext_vector_type(2)));
typedef float f32x2_vec __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b)
{
    f32x2_vec f = {a, b};
    bf16x2_vec h = __builtin_convertvector(f, bf16x2_vec);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

// MFMA 32x32 accumulator rows -> 16-byte stores.  A lane of the 32x32 accumulator layout holds, per pixel, four
// groups gq of 4 consecutive channels (8 gq + 4 half + 0..3): packed to bf16 that is four 8-byte pieces a pixel and
// lane, 16 bytes apart -- four store instructions whose 64 lanes write 8 bytes each (profiles/r06_c8_*: the pattern
// costs 10-18 % of a convolution).  The two halves of the wave hold complementary pieces of the SAME pixel, so they
// trade: lanes 0..31 give away groups 1 and 3 and receive the partner's 0 and 2 (v_permlane32_swap: one instruction
// per dword, no LDS), after which lane (pixel, half) owns channels [16 p + 8 half, + 8) for p = 0, 1 -- two 16-byte
// stores, the two halves together 32 contiguous bytes per pixel and instruction.
// pk[gq] = {ch 8gq+4half+0|1, ch 8gq+4half+2|3} on entry; on return q[p] = the 8 channels 16p + 8half .. + 7.
typedef uint32_t dfm_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t dfm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void acc_rows_to_16B(const dfm_u32x2 (&pk)[4], dfm_u32x4 (&q)[2])
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        // first operand: group 2p (kept by the low half), second: group 2p + 1 (kept by the high half)
        const auto x = __builtin_amdgcn_permlane32_swap(pk[2 * p].x, pk[2 * p + 1].x, false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(pk[2 * p].y, pk[2 * p + 1].y, false, false);
        // low half: {own 2p, partner's 2p}; high half: {partner's 2p + 1, own 2p + 1}
        q[p] = dfm_u32x4{x[0], y[0], x[1], y[1]};
    }
}

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int CB = 4;  // elements per 16-byte channel block
    static __device__ __forceinline__ float load(float v) { return v; }
    static __device__ __forceinline__ float store(float v) { return v; }
};
template <> struct elem<bf16_t> {
    static constexpr int CB = 8;
    static __device__ __forceinline__ float load(bf16_t v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ bf16_t store(float v) { return f32_to_bf16(v); }
};

// unpack one 16-byte channel block into CB floats
__device__ __forceinline__ void unpack16(const uint4 &q, float (&f)[4])
{
    f[0] = __uint_as_float(q.x);
    f[1] = __uint_as_float(q.y);
    f[2] = __uint_as_float(q.z);
    f[3] = __uint_as_float(q.w);
}
__device__ __forceinline__ void unpack16(const uint4 &q, float (&f)[8])
{
    f[0] = __uint_as_float(q.x << 16);
    f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16);
    f[3] = __uint_as_float(q.y & 0xffff0000u);
    f[4] = __uint_as_float(q.z << 16);
    f[5] = __uint_as_float(q.z & 0xffff0000u);
    f[6] = __uint_as_float(q.w << 16);
    f[7] = __uint_as_float(q.w & 0xffff0000u);
}

// 16-byte vector of T <-> floats (4 fp32 / 8 bf16)
template <typename T> struct vec16;
template <> struct vec16<float> { static constexpr int N = 4; };
template <> struct vec16<bf16_t> { static constexpr int N = 8; };

template <typename T>
__device__ __forceinline__ void load16(const T *p, float (&f)[vec16<T>::N])
{
    const uint4 q = *(const uint4 *)p;
    unpack16(q, f);
}

template <typename T>
__device__ __forceinline__ void store16(T *p, const float (&f)[vec16<T>::N])
{
    if constexpr (sizeof(T) == 4) {
        *(uint4 *)p = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                                 __float_as_uint(f[3]));
    } else {
        *(uint4 *)p = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                                 pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
}

// trilinear x-scale upsample with align_corners=True, ATen's index / weight arithmetic
// (nn.Upsample in DepthHead.forward, dense_heads/depth_head.py:205; shared by the depth-head
// kernels and the fused FrustumToVoxel, which must produce the same bits)
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

// exp(x) for x <= 0 (a logit minus its column maximum) -- the softmax of DepthHead and of the fused
// FrustumToVoxel share it, so the two stay bit-identical.  x * log2(e) as a two-word product, the
// integer part split off before the low word is added (so the low word survives for large |x|), the
// hardware exp2 on the fraction and a ldexp: ~9 VALU operations, within 1 ulp of libm's expf (which
// spends half of its ~15 operations on the overflow / underflow cases this argument cannot reach).
// NaN propagates; anything below -128 (-inf included) gives exp(-128) scaled out of range = 0.
__device__ __forceinline__ float exp_nonpos(float x)
{
    x = x < -128.0f ? -128.0f : x;
    const float ph = x * 1.44269504088896340736f;
    const float pl = __builtin_fmaf(x, 1.92596299112661746e-8f, __builtin_fmaf(x, 1.44269504088896340736f, -ph));
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}

__device__ __forceinline__ float lerp_fma(float w0, float a, float w1, float b)
{
    return __builtin_fmaf(w0, a, w1 * b);
}

// row of (v @ M^T): sum_k v[k]*M[k], k-ordered fma chain
// (torch fp32 mm on CPU; reference call sites utils.py:208,246 and
//  dfm_backbone.py:270)
__device__ __forceinline__ float dot4_chain(float a0, float a1, float a2, float a3,
                                            const float *__restrict__ M)
{
    float acc = a0 * M[0];
    acc = __builtin_fmaf(a1, M[1], acc);
    acc = __builtin_fmaf(a2, M[2], acc);
    acc = __builtin_fmaf(a3, M[3], acc);
    return acc;
}

// kernel-side geometry (mirror of dfm_sweep_desc, plus derived sizes)
struct SweepGeom {
    int32_t C, h_in, w_in, D, h_out, w_out;
    int32_t nblk;  // ceil(C / CB)
    int32_t flip;
    float fsf, csf, scale, crop_x, crop_y, org_w;
    long long N;  // D*h_out*w_out
};

// One lattice point -> UNNORMALISED pixel coordinates in the cur and prev
// feature maps (what F.grid_sample computes internally from the reference's
// normalised grid).  Also returns the normalised grid when `norm` != nullptr.
// Follows dfm_backbone.py:247-294 + ATen grid_sampler_unnormalize op by op.
__device__ __forceinline__ void sweep_point(const SweepGeom &g, const float *__restrict__ P,
                                            const float *__restrict__ Pinv,
                                            const float *__restrict__ Tm, float depth, int hi,
                                            int wi, float &cx, float &cy, float &px, float &py,
                                            float *norm)
{
    float x = ((float)wi * g.fsf) * g.csf;
    float y = ((float)hi * g.fsf) * g.csf;
    x = x + g.crop_x;
    y = y + g.crop_y;
    x = x / g.scale;
    y = y / g.scale;
    if (g.flip) x = g.org_w - x;
    // points_img2cam
    const float h0 = x * depth, h1 = y * depth, h2 = depth;
    const float X0 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 0);
    const float X1 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 4);
    const float X2 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 8);
    // cur: points_cam2img with the 4x4
    float a = dot4_chain(X0, X1, X2, 1.0f, P + 0);
    float b = dot4_chain(X0, X1, X2, 1.0f, P + 4);
    float c = dot4_chain(X0, X1, X2, 1.0f, P + 8);
    float cu = a / c, cv = b / c;
    // prev: cur2prev then project
    const float Y0 = dot4_chain(X0, X1, X2, 1.0f, Tm + 0);
    const float Y1 = dot4_chain(X0, X1, X2, 1.0f, Tm + 4);
    const float Y2 = dot4_chain(X0, X1, X2, 1.0f, Tm + 8);
    a = dot4_chain(Y0, Y1, Y2, 1.0f, P + 0);
    b = dot4_chain(Y0, Y1, Y2, 1.0f, P + 4);
    c = dot4_chain(Y0, Y1, Y2, 1.0f, P + 8);
    float pu = a / c, pv = b / c;
    if (g.flip) {
        cu = g.org_w - cu;
        pu = g.org_w - pu;
    }
    cu = cu * g.scale; cv = cv * g.scale;
    pu = pu * g.scale; pv = pv * g.scale;
    cu = cu - g.crop_x; cv = cv - g.crop_y;
    pu = pu - g.crop_x; pv = pv - g.crop_y;
    cu = cu / g.fsf; cv = cv / g.fsf;
    pu = pu / g.fsf; pv = pv / g.fsf;
    const float wm1 = (float)(g.w_in - 1), hm1 = (float)(g.h_in - 1);
    const float ncx = cu / wm1 * 2.0f - 1.0f;
    const float ncy = cv / hm1 * 2.0f - 1.0f;
    const float npx = pu / wm1 * 2.0f - 1.0f;
    const float npy = pv / hm1 * 2.0f - 1.0f;
    if (norm) {
        norm[0] = ncx; norm[1] = ncy; norm[2] = npx; norm[3] = npy;
    }
    // grid_sampler_unnormalize, align_corners=True
    cx = ((ncx + 1.0f) / 2.0f) * wm1;
    cy = ((ncy + 1.0f) / 2.0f) * hm1;
    px = ((npx + 1.0f) / 2.0f) * wm1;
    py = ((npy + 1.0f) / 2.0f) * hm1;
}

// Same arithmetic as sweep_point, but only the map the caller samples
// (HALF 0 = cur, 1 = prev), and with the divisions that are exact no-ops or
// exact scalings folded: x / 1.0f == x, x / 2^k == x * 2^-k (both bit-exact
// for the normal-range values this path sees; wave-uniform branches).
struct SweepFast {
    int32_t scale_is_one;  // img_scale_factor == 1.0f
    int32_t fsf_pow2;      // feat_sample_factor is a power of two
    float inv_fsf;         // 1 / fsf, exact when fsf_pow2
};

template <int HALF>
__device__ __forceinline__ void sweep_point_map(const SweepGeom &g, const SweepFast &f,
                                                const float *__restrict__ P,
                                                const float *__restrict__ Pinv,
                                                const float *__restrict__ Tm, float depth, int hi,
                                                int wi, float &ox, float &oy)
{
    float x = ((float)wi * g.fsf) * g.csf;
    float y = ((float)hi * g.fsf) * g.csf;
    x = x + g.crop_x;
    y = y + g.crop_y;
    if (!f.scale_is_one) {
        x = x / g.scale;
        y = y / g.scale;
    }
    if (g.flip) x = g.org_w - x;
    const float h0 = x * depth, h1 = y * depth, h2 = depth;
    float X0 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 0);
    float X1 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 4);
    float X2 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 8);
    if (HALF) {
        const float Y0 = dot4_chain(X0, X1, X2, 1.0f, Tm + 0);
        const float Y1 = dot4_chain(X0, X1, X2, 1.0f, Tm + 4);
        const float Y2 = dot4_chain(X0, X1, X2, 1.0f, Tm + 8);
        X0 = Y0; X1 = Y1; X2 = Y2;
    }
    const float a = dot4_chain(X0, X1, X2, 1.0f, P + 0);
    const float b = dot4_chain(X0, X1, X2, 1.0f, P + 4);
    const float c = dot4_chain(X0, X1, X2, 1.0f, P + 8);
    float u = a / c, v = b / c;
    if (g.flip) u = g.org_w - u;
    if (!f.scale_is_one) {
        u = u * g.scale;
        v = v * g.scale;
    }
    u = u - g.crop_x;
    v = v - g.crop_y;
    if (f.fsf_pow2) {
        u = u * f.inv_fsf;
        v = v * f.inv_fsf;
    } else {
        u = u / g.fsf;
        v = v / g.fsf;
    }
    const float wm1 = (float)(g.w_in - 1), hm1 = (float)(g.h_in - 1);
    const float nx = u / wm1 * 2.0f - 1.0f;
    const float ny = v / hm1 * 2.0f - 1.0f;
    ox = ((nx + 1.0f) * 0.5f) * wm1;
    oy = ((ny + 1.0f) * 0.5f) * hm1;
}

// ---- backward of the plane sweep: pieces shared by plane_sweep.hip and plane_sweep_bwd_mfma.hip ----
// packed footprint of one (plane, point): bit 31 valid, 27..30 = wok eok nok sok,
// 13..25 = ixw + 1, 0..12 = iyn + 1 (corner in [-1, W-1] x [-1, H-1])
__device__ __forceinline__ uint32_t bwd_footprint(float sx, float sy, int H, int W, float &fw, float &fn)
{
    const bool fin = (fabsf(sx) <= 3.0e38f) && (fabsf(sy) <= 3.0e38f);
    const float xw = floorf(sx), yn = floorf(sy);
    fw = sx - xw;
    fn = sy - yn;
    const bool wok = fin && xw >= 0.0f && xw <= (float)(W - 1);
    const bool eok = fin && xw >= -1.0f && xw <= (float)(W - 2);
    const bool nok = fin && yn >= 0.0f && yn <= (float)(H - 1);
    const bool sok = fin && yn >= -1.0f && yn <= (float)(H - 2);
    if (!((wok || eok) && (nok || sok))) return 0u;
    const int ixw = (int)xw, iyn = (int)yn;
    return 0x80000000u | ((uint32_t)wok << 27) | ((uint32_t)eok << 28) | ((uint32_t)nok << 29) |
           ((uint32_t)sok << 30) | ((uint32_t)(ixw + 1) << 13) | (uint32_t)(iyn + 1);
}

// First plane of sample b's sweep over map HALF from which on no lattice row / column is stretched
// beyond `zoom` map pixels per lattice point (forward motion zooms the nearest planes of the prev map;
// a plane qualifies only if every later plane does).  Judged on the sample positions of the four
// lattice corners, the extremes of a projective map.  The matrix-product backward takes the planes
// from the split on, the LDS-atomic backward the planes before it; both call THIS function with the
// same arguments, so they agree bit for bit on who owns a plane.  All threads of the workgroup call
// it (`slot` is an LDS word; two barriers inside).
template <int HALF>
__device__ __forceinline__ int sweep_zoom_split(const SweepGeom &g, const SweepFast &f, const float *__restrict__ P,
                                                const float *__restrict__ Pinv, const float *__restrict__ Tm,
                                                const float *__restrict__ depths, float zoom, int tid, int nthreads,
                                                int *slot)
{
    if (tid == 0) *slot = 0;
    __syncthreads();
    const float sx = zoom * (float)(g.w_out - 1) + 2.0f, sy = zoom * (float)(g.h_out - 1) + 2.0f;
    for (int d = tid; d < g.D; d += nthreads) {
        float cx[4], cy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            sweep_point_map<HALF>(g, f, P, Pinv, Tm, depths[d], (k >> 1) ? g.h_out - 1 : 0, (k & 1) ? g.w_out - 1 : 0,
                                  cx[k], cy[k]);
        const bool ok = fabsf(cx[1] - cx[0]) <= sx && fabsf(cx[3] - cx[2]) <= sx && fabsf(cy[2] - cy[0]) <= sy &&
                        fabsf(cy[3] - cy[1]) <= sy;  // false for NaN
        if (!ok) atomicMax(slot, d + 1);
    }
    __syncthreads();
    return *slot;
}

// host side, defined in plane_sweep.hip / plane_sweep_bwd_mfma.hip
SweepFast sweep_make_fast(const dfm_sweep_desc *d);
bool sweep_bwd_mfma_supported(const dfm_sweep_desc *d, const void *grad_out);
int sweep_bwd_mfma_launch(const dfm_sweep_desc *d, int half, const void *grad_out, const float *depths,
                          const float *P, const float *Pinv, const float *Tm, float *grad_cur, float *grad_prev,
                          void *stream);
// zoom (map pixels per lattice point) up to which the matrix-product backward takes a plane of the prev
// map: whole 32-point segments up to SWEEP_BWD_ZOOM_ONE, in two 16-point passes up to
// SWEEP_BWD_ZOOM_TWO, in four 8-point passes up to SWEEP_BWD_ZOOM_FOUR (its accumulator window is 48
// columns x 6 rows); beyond that the LDS-atomic tile kernel (the two kernels split the planes by
// sweep_zoom_split with SWEEP_BWD_ZOOM_FOUR)
constexpr float SWEEP_BWD_ZOOM_ONE = 1.4f, SWEEP_BWD_ZOOM_TWO = 2.75f, SWEEP_BWD_ZOOM_FOUR = 3.9f;

// Bilinear footprint of one sample point: top-left integer corner, the four
// corner weights (ATen compute_interp_params) and per-corner in-bounds bits.
struct Tap {
    int ix, iy;          // clamped so that (iy, ix) .. (iy+1, ix+1) are addressable
    float nw, ne, sw, se;
    uint32_t ok;         // bit0 nw, bit1 ne, bit2 sw, bit3 se
    int dx, dy;          // 0/1: offset to the east / south tap after clamping
};

__device__ __forceinline__ Tap make_tap(float x, float y, int H, int W)
{
    Tap t;
    const bool fin = (fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f);  // false for NaN/Inf
    const float xw = floorf(x), yn = floorf(y);
    // Non-finite coordinates (a projection that divides by z = 0, utils.py:209): every tap is out of bounds AND
    // every weight is zero -- Inf - floor(Inf) is NaN, and a NaN weight times a masked (zero) tap is NaN, which
    // is what F.grid_sample on PyTorch-CPU returns there; the oracle, these kernels and torch's GPU kernel
    // return 0 (tests/golden/plane_sweep_zero_depth.npz pins it).
    const float w = fin ? x - xw : 0.0f, e = 1.0f - w, n = fin ? y - yn : 0.0f, s = 1.0f - n;
    t.nw = fin ? s * e : 0.0f; t.ne = s * w; t.sw = n * e; t.se = n * w;
    // in-bounds tests in the float domain (exact for |v| < 2^24; beyond that
    // everything is out of bounds, like ATen's saturating int conversion)
    const bool wok = fin && xw >= 0.0f && xw <= (float)(W - 1);
    const bool eok = fin && xw >= -1.0f && xw <= (float)(W - 2);
    const bool nok = fin && yn >= 0.0f && yn <= (float)(H - 1);
    const bool sok = fin && yn >= -1.0f && yn <= (float)(H - 2);
    t.ok = (uint32_t)(wok && nok) | ((uint32_t)(eok && nok) << 1) | ((uint32_t)(wok && sok) << 2) |
           ((uint32_t)(eok && sok) << 3);
    // clamp the corner so every address we form is inside the plane
    float xc = fminf(fmaxf(xw, 0.0f), (float)(W - 1));
    float yc = fminf(fmaxf(yn, 0.0f), (float)(H - 1));
    if (!fin) { xc = 0.0f; yc = 0.0f; }
    t.ix = (int)xc; t.iy = (int)yc;
    t.dx = (wok && eok) ? 1 : 0;
    t.dy = (nok && sok) ? 1 : 0;
    // when only the east (south) tap is valid the clamped corner IS that tap
    // (xw == -1 -> xc == 0): its value must be read at +0, see sample().
    return t;
}

// (n, C, P) planar -> (n, P, Cp) pixel-major, Cp = C rounded up to a whole number of
// 16-byte channel blocks (zero padded): all channels of one sampled pixel / voxel are
// one contiguous run, so a tap is Cp*sizeof(T)/16 adjacent 16-byte loads instead of
// C scalar loads a whole plane apart.  64 pixels x 32 channels per workgroup through
// an LDS tile: global reads run along pixels, global writes along channels.
// grid = (ceil(P/64), ceil(Cp/32), n)
template <typename T>
__global__ __launch_bounds__(256) void pack_pixel_major_kernel(const T *__restrict__ src,
                                                               T *__restrict__ dst, int C, int Cp,
                                                               long long P)
{
    __shared__ T tile[32][64 + 1];
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const size_t n = blockIdx.z;
    {
        const int p = threadIdx.x & 63, cc = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = cc + 4 * k;
            T v = T(0);
            if (c0 + c < C && p0 + p < P) v = src[(n * C + c0 + c) * (size_t)P + p0 + p];
            tile[c][p] = v;
        }
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31, pp = threadIdx.x >> 5;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = pp + 8 * k;
            if (c0 + c < Cp && p0 + p < P) dst[(n * (size_t)P + p0 + p) * Cp + c0 + c] = tile[c][p];
        }
    }
}

// the way back for gradients accumulated pixel-major (coalesced atomics):
// dst (n, C, P) += src (n, P, C): 64 pixels x 32 channels per workgroup through LDS
template <typename F = float>
__global__ __launch_bounds__(256) void add_from_pixel_major_kernel(const float *__restrict__ src,
                                                                   float *__restrict__ dst, int C,
                                                                   long long P)
{
    __shared__ float tile[64][32 + 1];
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const size_t n = blockIdx.z;
    {
        const int c = threadIdx.x & 31, pp = threadIdx.x >> 5;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = pp + 8 * k;
            tile[p][c] = (c0 + c < C && p0 + p < P) ? src[(n * (size_t)P + p0 + p) * C + c0 + c] : 0.0f;
        }
    }
    __syncthreads();
    {
        const int p = threadIdx.x & 63, cc = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = cc + 4 * k;
            if (c0 + c < C && p0 + p < P) {
                float *q = dst + (n * C + c0 + c) * (size_t)P + p0 + p;
                *q = *q + tile[p][c];
            }
        }
    }
}

}  // namespace dfm

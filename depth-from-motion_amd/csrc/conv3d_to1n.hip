// conv3d_to1n.hip -- the prediction head's tail as ONE pass over the feature volume:
//   GroupNorm (one channel per group) + ReLU of the 32-channel volume, applied ON LOAD, followed by
//   Conv3d(32 -> 1, 3x3x3, pad 1)            mmdet3d/models/backbones/dfm_backbone.py:120-127
// (round 6; VERDICT round 5 item 1a: "normalise + ReLU inside the CONSUMING convolution").
//
// Before: gn_apply_cl_kernel read the raw convolution output and wrote the normalised volume (2 x 118 MB at
// config K), then conv3d_k3_c32_kernel<OUT_C1> ran the 32 -> 1 convolution as a 32 -> 32 one with 31 zero
// weight rows -- 54 MFMAs per 32 pixels for 3.2 GFLOP of useful work, 83 us.  Here every input pixel is
// touched ONCE:
//   * a wave loads 32 pixels x 64 B straight from global memory in MFMA B-operand order (lane = pixel,
//     8 consecutive channels per k-step half), normalises them in registers with the lane's 16 (a, b) pairs
//     -- r = v * a + b in two roundings, max(r, 0), RNE to bf16: bit for bit the values gn_apply_cl_kernel would
//     have stored -- and runs TWO v_mfma_f32_32x32x16_bf16 with A = W[tap][channel] (27 of 32 rows used):
//     T[tap][pixel] = sum_c W[tap][c] x[pixel][c], the per-tap partial products of that pixel, 27x fewer MFMAs
//     than one MFMA chain per output;
//   * T goes to LDS as [tap][pixel] (conflict-free both ways); a lane then owns ONE output pixel of the
//     8 x 32 tile and adds the 27 shifted values out[d][h][w] = sum T[kd,kh,kw][d+kd-1][h+kh-1][w+kw-1];
//   * the workgroup walks the depth axis: an input slab contributes to three output planes, whose running sums
//     live in three registers per lane -- one slab of T in LDS (double-buffered: one barrier per plane);
//   * zero padding is a padding of the NORMALISED tensor: out-of-volume pixels feed zeros to the MFMA.
// HBM-bound: the volume is read once (x 1.33 halo, mostly from L2), one bf16 per pixel is written.
#include <algorithm>

#include "dfm_common.h"

using namespace dfm;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int T1_TH = 8, T1_TW = 32;                 // output tile (rows x columns) of a workgroup
constexpr int T1_SH = T1_TH + 2, T1_SW = T1_TW + 2;  // input slab with its halo
constexpr int T1_SPX = T1_SH * T1_SW;                // 340 pixels
constexpr int T1_NG = (T1_SPX + 31) / 32;            // 11 groups of 32 pixels
constexpr int T1_ROW = T1_NG * 32;                   // 352: pixels per tap row of the LDS image
constexpr int T1_GPW = (T1_NG + 3) / 4;              // 3 groups per wave
constexpr int T1_LDS = 2 * 27 * T1_ROW * 4;          // 76032 bytes

struct T1Geom {
    int32_t N, D, H, W;
    int32_t tiles_w, tiles_h, dchunk, relu_in, relu_out;
};

template <typename TW>
__device__ __forceinline__ float t1_wload(const TW *w, int idx);
template <>
__device__ __forceinline__ float t1_wload<float>(const float *w, int idx) { return w[idx]; }
template <>
__device__ __forceinline__ float t1_wload<bf16_t>(const bf16_t *w, int idx) { return bf16_to_f32(w[idx]); }

// x: (N, D, H, W, 32) bf16, the raw output of the 32 -> 32 convolution; coef: (N, 32, 2) fp32 (a, b) of the
// normalisation y = x * a + b (dfm_group_norm_coefficients); weight: (1, 32, 3, 3, 3); out: (N, D, H, W) bf16
template <typename TW>
__global__ __launch_bounds__(256, 2) void conv3d_to1_norm_kernel(T1Geom g, const bf16_t *__restrict__ x,
                                                                 const float *__restrict__ coef,
                                                                 const TW *__restrict__ weight,
                                                                 bf16_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float tl[];  // [2][27][T1_ROW]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l32 = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x;
    const int tw = tile % g.tiles_w, th = tile / g.tiles_w;
    const int h0 = th * T1_TH, w0 = tw * T1_TW;
    const int d0 = blockIdx.y * g.dchunk, d1 = min(d0 + g.dchunk, g.D);
    const int n = blockIdx.z;
    const size_t plane = (size_t)g.H * g.W * 32;  // elements per depth plane
    const bf16_t *xn = x + (size_t)n * g.D * plane;

    // A operand: lane l holds W[tap = l & 31][channel = ks * 16 + (l >> 5) * 8 + j] (zero rows 27..31)
    bf16x8_t wf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = ks * 16 + half * 8 + 2 * j;
            const float v0 = l32 < 27 ? t1_wload<TW>(weight, c0 * 27 + l32) : 0.0f;
            const float v1 = l32 < 27 ? t1_wload<TW>(weight, (c0 + 1) * 27 + l32) : 0.0f;
            pk[j] = pack_bf16x2(v0, v1);
        }
        __builtin_memcpy(&wf[ks], pk, 16);
    }
    // the lane's normalisation coefficients: channels ks * 16 + half * 8 + j
    float ca[2][8], cb[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ks * 16 + half * 8 + j;
            ca[ks][j] = coef[((size_t)n * 32 + c) * 2];
            cb[ks][j] = coef[((size_t)n * 32 + c) * 2 + 1];
        }
    // the lane's pixels of a slab: byte offset inside a depth plane, -1 outside the volume (or past the slab)
    int poff[T1_GPW];
#pragma unroll
    for (int k = 0; k < T1_GPW; ++k) {
        const int grp = wave + 4 * k;
        const int p = grp * 32 + l32;
        const int j = p / T1_SW, i = p - j * T1_SW;
        const int h = h0 - 1 + j, w = w0 - 1 + i;
        const bool ok = grp < T1_NG && p < T1_SPX && h >= 0 && h < g.H && w >= 0 && w < g.W;
        poff[k] = ok ? ((h * g.W + w) * 32 + half * 8) * 2 : -1;
    }
    auto fetch = [&](int dz, uint4 (&q)[T1_GPW][2]) {
        const bool zok = dz >= 0 && dz < g.D;
        const unsigned char *src = (const unsigned char *)(xn + (size_t)(zok ? dz : 0) * plane);
#pragma unroll
        for (int k = 0; k < T1_GPW; ++k) {
            if (wave + 4 * k >= T1_NG) continue;  // (uniform per wave)
            // (a masked lane reads the plane's first pixel: a valid address, the value is discarded below)
            const unsigned char *s = src + ((zok && poff[k] >= 0) ? poff[k] : 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) q[k][ks] = *(const uint4 *)(s + ks * 32);
        }
    };
    // output pixel of this lane
    const int oh = tid >> 5, ow = tid & 31;
    const bool out_ok = h0 + oh < g.H && w0 + ow < g.W;
    float accA = 0.0f, accB = 0.0f;  // running sums of output planes dz - 1 and dz

    uint4 cur[T1_GPW][2], nxt[T1_GPW][2];
    fetch(d0 - 1, cur);
    for (int dz = d0 - 1; dz <= d1; ++dz) {
        const int buf = (dz - d0 + 1) & 1;
        float *tb = tl + buf * 27 * T1_ROW;
        if (dz + 1 <= d1) fetch(dz + 1, nxt);
        const bool zok = dz >= 0 && dz < g.D;  // (uniform) a slab outside the volume contributes nothing
        if (zok) {
#pragma unroll
            for (int k = 0; k < T1_GPW; ++k) {
                const int grp = wave + 4 * k;
                if (grp >= T1_NG) continue;  // (uniform per wave)
                const bool ok = poff[k] >= 0;
                f32x16_t acc;
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float f[8];
                    unpack16(cur[k][ks], f);
                    uint32_t pk[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (explicit fma: gn_apply_cl_kernel's expression, bit for bit)
                        float r0 = __builtin_fmaf(f[2 * j], ca[ks][2 * j], cb[ks][2 * j]);
                        float r1 = __builtin_fmaf(f[2 * j + 1], ca[ks][2 * j + 1], cb[ks][2 * j + 1]);
                        if (g.relu_in) { r0 = fmaxf(r0, 0.0f); r1 = fmaxf(r1, 0.0f); }
                        pk[j] = ok ? pack_bf16x2(r0, r1) : 0u;  // zero padding of the NORMALISED tensor
                    }
                    bf16x8_t xf;
                    __builtin_memcpy(&xf, pk, 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf, acc, 0, 0, 0);
                }
                // D[tap = 8 * (t / 4) + 4 * half + (t % 4)][pixel = l32] -> tl[tap][grp * 32 + l32]
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int tap = 8 * (t >> 2) + 4 * half + (t & 3);
                    if (tap < 27) tb[tap * T1_ROW + grp * 32 + l32] = acc[t];
                }
            }
        }
        __syncthreads();
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        if (zok) {
            const float *tp = tb + oh * T1_SW + ow;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int o = kh * T1_SW + kw;
                    s0 += tp[(0 * 9 + kh * 3 + kw) * T1_ROW + o];
                    s1 += tp[(1 * 9 + kh * 3 + kw) * T1_ROW + o];
                    s2 += tp[(2 * 9 + kh * 3 + kw) * T1_ROW + o];
                }
        }
        // slab dz: tap kd = 2 completes output plane dz - 1, kd = 1 adds to plane dz, kd = 0 opens plane dz + 1
        accA += s2;
        if (dz - 1 >= d0 && dz - 1 < d1 && out_ok) {
            float v = accA;
            if (g.relu_out) v = fmaxf(v, 0.0f);
            out[(((size_t)n * g.D + (dz - 1)) * g.H + h0 + oh) * g.W + w0 + ow] = f32_to_bf16(v);
        }
        accA = accB + s1;
        accB = s0;
#pragma unroll
        for (int k = 0; k < T1_GPW; ++k) { cur[k][0] = nxt[k][0]; cur[k][1] = nxt[k][1]; }
    }
}

}  // namespace

extern "C" DFM_API int dfm_conv3d_to1_norm_fwd(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                               const float *coef, const void *weight, int32_t weight_dtype,
                                               int32_t relu_in, int32_t relu_out, void *out, int32_t depth_chunk,
                                               void *stream)
{
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    if (!x || !coef || !weight || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    if ((long long)h * w * 64 >= (1ll << 31)) return set_error(DFM_ERR_UNSUPPORTED, "depth plane too large");
    if (n > 65535) return set_error(DFM_ERR_UNSUPPORTED, "batch > 65535");
    if (((uintptr_t)x & 15)) return set_error(DFM_ERR_UNSUPPORTED, "x must be 16-byte aligned");
    T1Geom g;
    g.N = n; g.D = d; g.H = h; g.W = w;
    g.tiles_w = (w + T1_TW - 1) / T1_TW;
    g.tiles_h = (h + T1_TH - 1) / T1_TH;
    g.relu_in = relu_in ? 1 : 0;
    g.relu_out = relu_out ? 1 : 0;
    // a chunk of dc output planes walks dc + 2 slabs; two workgroups per CU
    int dc = depth_chunk;
    if (dc <= 0) {
        const long long cols = (long long)g.tiles_w * g.tiles_h * n;
        double best = 1e30;
        dc = d;
        for (int c = std::min(d, 4); c <= d; ++c) {
            const long long chunks = (d + c - 1) / c;
            const long long rounds = (cols * chunks + 511) / 512;
            const double cost = (double)rounds * (c + 2.5);
            if (cost < best - 1e-9) { best = cost; dc = c; }
        }
    }
    dc = std::min(dc, d);
    g.dchunk = dc;
    const int nchunks = (d + dc - 1) / dc;
    if (nchunks > 65535) return set_error(DFM_ERR_UNSUPPORTED, "too many depth chunks");
    dim3 grid(g.tiles_w * g.tiles_h, nchunks, n);
    hipStream_t st = (hipStream_t)stream;
    if (weight_dtype == DFM_F32) {
        const int rc = ensure_dynamic_lds((const void *)conv3d_to1_norm_kernel<float>, T1_LDS);
        if (rc != DFM_OK) return rc;
        hipLaunchKernelGGL(conv3d_to1_norm_kernel<float>, grid, dim3(256), T1_LDS, st, g, (const bf16_t *)x, coef,
                           (const float *)weight, (bf16_t *)out);
    } else {
        const int rc = ensure_dynamic_lds((const void *)conv3d_to1_norm_kernel<bf16_t>, T1_LDS);
        if (rc != DFM_OK) return rc;
        hipLaunchKernelGGL(conv3d_to1_norm_kernel<bf16_t>, grid, dim3(256), T1_LDS, st, g, (const bf16_t *)x, coef,
                           (const bf16_t *)weight, (bf16_t *)out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

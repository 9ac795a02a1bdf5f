// group_norm.hip -- fused GroupNorm (+ReLU) for the cost-volume aggregation
// stacks (gfx950).
//
// Reference: every Conv3d of DfMBackbone / hourglass / FrustumToVoxel is
// followed by GroupNorm(32, C) [+ ReLU] (dfm_backbone.py:50-66,118-128,
// utils/conv_modules.py:27-43, feature_transformation.py:55-62 via mmcv's
// ConvModule).  With C = 32 that is a per-channel normalisation over the whole
// (D,H,W) volume.  torch's kernel runs it at ~0.26 TB/s
// (1.8 ms at 72x80x320, more than the bf16 convolution in front of it:
// profiles/archive/r01_miopen_conv3d_baseline.txt); it is a pure HBM-bound
// reduction + elementwise op.
//
// Layout: NC(D)HW contiguous: the elements of group g of sample n are ONE
// contiguous range of L = (C/G)*S elements.
//   gn_stats_kernel : grid (splits, N*G); each block reduces a slice with
//                     16-byte loads to (count, mean, M2) -- Welford/Chan
//                     merges, no E[x^2]-E[x]^2 cancellation -- and writes a
//                     partial.
//   gn_apply_kernel : grid (splits, N*G); merges the partials of its group
//                     (tiny), then y = (x-mean)*rstd*gamma[c]+beta[c], ReLU
//                     optional, 16-byte loads/stores.  Saves mean / rstd.
//   backward        : gn_bwd_stats (sums of dy*gamma and dy*gamma*xhat per
//                     group, dgamma/dbeta per channel) + gn_bwd_apply.
// Traffic: 2 reads + 1 write of the tensor (the stats read is L2/MALL-warm
// right after the convolution).  Bound: HBM.
#include "dfm_common.h"

#include <algorithm>

using namespace dfm;

namespace {

struct Moments {
    float n, mean, m2;
};

__device__ __forceinline__ Moments merge(const Moments &a, const Moments &b)
{
    if (b.n == 0.0f) return a;
    if (a.n == 0.0f) return b;
    Moments r;
    r.n = a.n + b.n;
    const float delta = b.mean - a.mean;
    const float f = b.n / r.n;
    r.mean = a.mean + delta * f;
    r.m2 = a.m2 + b.m2 + delta * delta * a.n * f;
    return r;
}

__device__ __forceinline__ Moments wave_merge(Moments m)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Moments other;
        other.n = __shfl_xor(m.n, o);
        other.mean = __shfl_xor(m.mean, o);
        other.m2 = __shfl_xor(m.m2, o);
        m = merge(m, other);
    }
    return m;
}

// block-level merge of per-thread moments -> thread 0 holds the result
__device__ __forceinline__ Moments block_merge(Moments m, Moments *sh)
{
    m = wave_merge(m);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        Moments r = sh[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = merge(r, sh[w]);
        sh[0] = r;
    }
    __syncthreads();
    return sh[0];
}

// slice [lo, hi) of the group's L elements handled by split `s` (multiples of VEC)
__device__ __forceinline__ void slice_of(long long L, int splits, int s, int vec, long long &lo,
                                         long long &hi)
{
    const long long nvec = (L + vec - 1) / vec;
    const long long per = (nvec + splits - 1) / splits;
    lo = min((long long)s * per * vec, L);
    hi = min(lo + per * vec, L);
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T *__restrict__ x, long long L,
                                                       int splits, float *__restrict__ partial)
{
    constexpr int VEC = vec16<T>::N;
    __shared__ Moments sh[4];
    const int grp = blockIdx.y, s = blockIdx.x;
    const T *xg = x + (size_t)grp * L;
    long long lo, hi;
    slice_of(L, splits, s, VEC, lo, hi);
    Moments m = {0.0f, 0.0f, 0.0f};
    const bool aligned = (((uintptr_t)xg) & 15) == 0;
    for (long long i = lo + (long long)threadIdx.x * VEC; i < hi; i += 256ll * VEC) {
        float f[VEC];
        int cnt = VEC;
        if (aligned && i + VEC <= hi) {
            load16<T>(xg + i, f);
        } else {
            cnt = (int)min((long long)VEC, hi - i);
            for (int k = 0; k < VEC; ++k) f[k] = k < cnt ? elem<T>::load(xg[i + k]) : 0.0f;
        }
        // moments of the (<= VEC) values, merged into the running ones
        float sm = 0.0f;
        for (int k = 0; k < cnt; ++k) sm += f[k];
        Moments v;
        v.n = (float)cnt;
        v.mean = sm / v.n;
        v.m2 = 0.0f;
        for (int k = 0; k < cnt; ++k) v.m2 += (f[k] - v.mean) * (f[k] - v.mean);
        m = merge(m, v);
    }
    m = block_merge(m, sh);
    if (threadIdx.x == 0) {
        float *p = partial + ((size_t)grp * splits + s) * 3;
        p[0] = m.n; p[1] = m.mean; p[2] = m.m2;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T *__restrict__ x, long long L,
                                                       long long spatial, int cpg, int groups,
                                                       int splits, float eps,
                                                       const float *__restrict__ partial,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, int relu,
                                                       T *__restrict__ y, float *__restrict__ mean_out,
                                                       float *__restrict__ rstd_out)
{
    constexpr int VEC = vec16<T>::N;
    const int grp = blockIdx.y, s = blockIdx.x;
    __shared__ float stat[2];
    if (threadIdx.x == 0) {
        Moments r = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < splits; ++k) {
            const float *p = partial + ((size_t)grp * splits + k) * 3;
            Moments v = {p[0], p[1], p[2]};
            r = merge(r, v);
        }
        const float var = r.m2 / r.n;  // biased, like torch
        stat[0] = r.mean;
        stat[1] = 1.0f / sqrtf(var + eps);
        if (s == 0) { mean_out[grp] = stat[0]; rstd_out[grp] = stat[1]; }
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const int c0 = (grp % groups) * cpg;
    const T *xg = x + (size_t)grp * L;
    T *yg = y + (size_t)grp * L;
    long long lo, hi;
    slice_of(L, splits, s, VEC, lo, hi);
    const bool aligned = ((((uintptr_t)xg) | ((uintptr_t)yg)) & 15) == 0 && spatial % VEC == 0;
    for (long long i = lo + (long long)threadIdx.x * VEC; i < hi; i += 256ll * VEC) {
        if (aligned && i + VEC <= hi) {
            float f[VEC];
            load16<T>(xg + i, f);
            const int c = c0 + (int)(i / spatial);  // a vector never straddles channels
            const float a = rstd * gamma[c], b = beta[c] - mean * a;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float v = f[k] * a + b;
                f[k] = relu ? fmaxf(v, 0.0f) : v;
            }
            store16<T>(yg + i, f);
        } else {
            for (long long k = i; k < min(i + VEC, hi); ++k) {
                const int c = c0 + (int)(k / spatial);
                const float a = rstd * gamma[c], b = beta[c] - mean * a;
                float v = elem<T>::load(xg[k]) * a + b;
                yg[k] = elem<T>::store(relu ? fmaxf(v, 0.0f) : v);
            }
        }
    }
}

// backward stats: per (n, channel) sums of dy' and dy'*xhat where dy' = dy (masked by y > 0
// when relu); grid (splits, N*C); partial[(nc*splits + s)*2 + {0,1}]
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T *__restrict__ dy,
                                                           const T *__restrict__ x,
                                                           const T *__restrict__ y, long long spatial,
                                                           int C, int cpg, int splits, int relu,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ rstd,
                                                           float *__restrict__ partial)
{
    __shared__ float sh[2][4];
    const int nc = blockIdx.y, s = blockIdx.x;
    const int n = nc / C, c = nc % C, grp = n * (C / cpg) + c / cpg;
    const float mu = mean[grp], rs = rstd[grp];
    const size_t base = (size_t)nc * spatial;
    const long long per = (spatial + splits - 1) / splits;
    const long long lo = min((long long)s * per, spatial), hi = min(lo + per, spatial);
    float s1 = 0.0f, s2 = 0.0f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float g = elem<T>::load(dy[base + i]);
        if (relu && !(elem<T>::load(y[base + i]) > 0.0f)) g = 0.0f;
        const float xh = (elem<T>::load(x[base + i]) - mu) * rs;
        s1 += g;
        s2 += g * xh;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *p = partial + ((size_t)nc * splits + s) * 2;
        p[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        p[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// dx = rstd * (gamma*dy' - (A + xhat*B)/L) with A = sum_group gamma*dy', B = sum_group gamma*dy'*xhat
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const T *__restrict__ dy, const T *__restrict__ x, const T *__restrict__ y, long long spatial,
    int C, int cpg, int splits, int relu, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma,
    const float *__restrict__ partial, T *__restrict__ dx, float *__restrict__ dgamma,
    float *__restrict__ dbeta)
{
    __shared__ float ab[2];
    const int nc = blockIdx.y, s = blockIdx.x;
    const int n = nc / C, c = nc % C, g0 = (c / cpg) * cpg, grp = n * (C / cpg) + c / cpg;
    if (threadIdx.x == 0) {
        float A = 0.0f, B = 0.0f;
        for (int cc = g0; cc < g0 + cpg; ++cc) {
            float a = 0.0f, b = 0.0f;
            for (int k = 0; k < splits; ++k) {
                const float *p = partial + ((size_t)(n * C + cc) * splits + k) * 2;
                a += p[0]; b += p[1];
            }
            A += gamma[cc] * a; B += gamma[cc] * b;
            if (cc == c && s == 0) { atomicAdd(dbeta + c, a); atomicAdd(dgamma + c, b); }
        }
        ab[0] = A; ab[1] = B;
    }
    __syncthreads();
    const float mu = mean[grp], rs = rstd[grp], gm = gamma[c];
    const float invL = 1.0f / ((float)cpg * (float)spatial);
    const float A = ab[0] * invL, B = ab[1] * invL;
    const size_t base = (size_t)nc * spatial;
    const long long per = (spatial + splits - 1) / splits;
    const long long lo = min((long long)s * per, spatial), hi = min(lo + per, spatial);
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float g = elem<T>::load(dy[base + i]);
        if (relu && !(elem<T>::load(y[base + i]) > 0.0f)) g = 0.0f;
        const float xh = (elem<T>::load(x[base + i]) - mu) * rs;
        dx[base + i] = elem<T>::store(rs * (gm * g - A - xh * B));
    }
}

int pick_splits(long long L)
{
    long long s = L / (256 * 16 * 8);  // >= 8 vectors per thread
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// channels-last passes: up to 2048 workgroups per sample (config K at batch 1 is ONE sample of
// 118 MB: 256 workgroups were one per CU, 3.5 TB/s); the partials are merged once by
// gn_merge_partials_kernel, not by every workgroup of the apply pass
constexpr int GN_BW_SPLITS = 1024;  // workgroups per sample of the channels-last backward's statistics pass

int pick_splits_cl(long long L)
{
    long long s = L / (256 * 16 * 4);  // >= 4 vectors per thread
    if (s < 1) s = 1;
    if (s > 2048) s = 2048;
    return (int)s;
}

// ---------------------------------------------------------------------------
// channels-last variant: x, y are (N, spatial, C) contiguous (torch channels_last_3d),
// what the NDHWC convolutions produce and consume.  A lane owns one 16-byte channel
// vector (VEC channels) of every (256 / nvb)-th voxel, nvb = C / VEC (a power of two),
// so its VEC running moments / affine coefficients live in registers.
//   gn_stats_cl : grid (splits, N): per-channel Welford over the slice, lanes holding the
//                 same channels merged through LDS, channels of a group merged -> partial
//   gn_apply_cl : grid (splits, N): merges the group's partials, y = x*a[c] + b[c] (+ReLU)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_cl_kernel(const T *__restrict__ x, long long spatial,
                                                          int C, int groups, int splits,
                                                          float *__restrict__ partial)
{
    constexpr int VEC = vec16<T>::N;
    __shared__ float sh1[4][256], sh2[4][256], shK[256];
    __shared__ Moments chm[256];
    const int n = blockIdx.y, s = blockIdx.x;
    const int nvb = C / VEC, vpi = 256 / nvb;  // voxels per iteration
    const int vb = threadIdx.x % nvb, v0 = threadIdx.x / nvb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long per = (spatial + splits - 1) / splits;
    const long long lo = min((long long)s * per, spatial), hi = min(lo + per, spatial);
    const T *xs = x + (size_t)n * spatial * C + (size_t)vb * VEC;
    // shifted sums, K = the slice's first value of the channel (the same in every lane that holds the channel: the
    // lanes' sums add up as they are, round 6 -- the first form shifted by each lane's own first value and merged
    // (count, mean, M2) triples with a division per step, 32 steps per channel: more than the loads of a slice):
    // s1 = sum(v - K), s2 = sum((v - K)^2) -- 3 VALU operations per value, as robust as Welford's update when
    // |mean| >> std; four 16-byte loads in flight per lane
    constexpr int U = 4;
    float K[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { K[k] = 0.0f; s1[k] = 0.0f; s2[k] = 0.0f; }
    if (lo < hi) load16<T>(xs + (size_t)lo * C, K);
    long long v = lo + v0;
    for (; v + (long long)(U - 1) * vpi < hi; v += (long long)U * vpi) {
        float f[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) load16<T>(xs + (size_t)(v + (long long)u * vpi) * C, f[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float d = f[u][k] - K[k];
                s1[k] += d;
                s2[k] = __builtin_fmaf(d, d, s2[k]);
            }
    }
    for (; v < hi; v += vpi) {
        float f[VEC];
        load16<T>(xs + (size_t)v * C, f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float d = f[k] - K[k];
            s1[k] += d;
            s2[k] = __builtin_fmaf(d, d, s2[k]);
        }
    }
    // the lanes of a wave that hold the same channels (lane % nvb == vb) added up, then the four waves through LDS
    for (int o = 32; o >= nvb; o >>= 1) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            s1[k] += __shfl_xor(s1[k], o);
            s2[k] += __shfl_xor(s2[k], o);
        }
    }
    if (lane < nvb) {  // (nvb <= 64; lane % nvb == vb)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            sh1[wave][vb * VEC + k] = s1[k];
            sh2[wave][vb * VEC + k] = s2[k];
        }
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) shK[vb * VEC + k] = K[k];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        float a1 = 0.0f, a2 = 0.0f;
        for (int w = 0; w < 4; ++w) { a1 += sh1[w][c]; a2 += sh2[w][c]; }
        const float cnt = (float)(hi - lo);
        Moments r = {0.0f, 0.0f, 0.0f};
        if (cnt > 0.0f) {
            const float a = a1 / cnt;
            r = Moments{cnt, shK[c] + a, fmaxf(a2 - a1 * a, 0.0f)};
        }
        chm[c] = r;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int cpg = C / groups, g = threadIdx.x;
        Moments r = {0.0f, 0.0f, 0.0f};
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) r = merge(r, chm[c]);
        float *p = partial + (((size_t)n * groups + g) * splits + s) * 3;
        p[0] = r.n; p[1] = r.mean; p[2] = r.m2;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_cl_kernel(const T *__restrict__ x, long long spatial,
                                                          int C, int groups, int splits, float eps,
                                                          const float *__restrict__ partial,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, int relu,
                                                          T *__restrict__ y, float *__restrict__ mean_out,
                                                          float *__restrict__ rstd_out, int psplits,
                                                          const T *__restrict__ res)
{
    // res: NULL, or a tensor of x's shape added after the affine map and before the ReLU
    // (the residual connections of the aggregation stacks: dfm_backbone.py:176,183,
    //  conv_modules.py:124-139) -- one pass instead of a separate elementwise add
    // psplits: partials per group to merge (0: one per workgroup of this grid, the layout
    // gn_stats_cl_kernel writes; 1: already merged by gn_merge_partials_kernel)
    constexpr int VEC = vec16<T>::N;
    __shared__ float ca[256], cb[256];
    const int n = blockIdx.y, s = blockIdx.x;
    const int cpg = C / groups;
    if ((int)threadIdx.x < groups) {
        const int g = threadIdx.x;
        Moments r = {0.0f, 0.0f, 0.0f};
        const int np_ = psplits > 0 ? psplits : splits;
        for (int k = 0; k < np_; ++k) {
            const float *p = partial + (((size_t)n * groups + g) * np_ + k) * 3;
            r = merge(r, Moments{p[0], p[1], p[2]});
        }
        const float mean = r.mean, rstd = 1.0f / sqrtf(r.m2 / r.n + eps);  // biased, like torch
        if (s == 0) { mean_out[n * groups + g] = mean; rstd_out[n * groups + g] = rstd; }
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float a = rstd * gamma[c];
            ca[c] = a;
            cb[c] = __builtin_fmaf(-mean, a, beta[c]);  // (explicit: the backward's recomputed mask and the on-load
                                                        //  normalisation of conv3d_to1n.hip use the same expression)
        }
    }
    __syncthreads();
    const int nvb = C / VEC, vpi = 256 / nvb;
    const int vb = threadIdx.x % nvb, v0 = threadIdx.x / nvb;
    float a[VEC], b[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { a[k] = ca[vb * VEC + k]; b[k] = cb[vb * VEC + k]; }
    const long long per = (spatial + splits - 1) / splits;
    const long long lo = min((long long)s * per, spatial), hi = min(lo + per, spatial);
    const size_t base = (size_t)n * spatial * C + (size_t)vb * VEC;
    constexpr int U = 4;  // vectors in flight per lane
    long long v = lo + v0;
    for (; v + (long long)(U - 1) * vpi < hi; v += (long long)U * vpi) {
        float f[U][VEC], q[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) load16<T>(x + base + (size_t)(v + (long long)u * vpi) * C, f[u]);
        if (res) {
#pragma unroll
            for (int u = 0; u < U; ++u) load16<T>(res + base + (size_t)(v + (long long)u * vpi) * C, q[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float r = __builtin_fmaf(f[u][k], a[k], b[k]);
                if (res) r += q[u][k];
                f[u][k] = relu ? fmaxf(r, 0.0f) : r;
            }
            store16<T>(y + base + (size_t)(v + (long long)u * vpi) * C, f[u]);
        }
    }
    for (; v < hi; v += vpi) {
        float f[VEC], q[VEC];
        load16<T>(x + base + (size_t)v * C, f);
        if (res) load16<T>(res + base + (size_t)v * C, q);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float r = __builtin_fmaf(f[k], a[k], b[k]);
            if (res) r += q[k];
            f[k] = relu ? fmaxf(r, 0.0f) : r;
        }
        store16<T>(y + base + (size_t)v * C, f);
    }
}

// ---- channels-last backward (the NDHWC stacks train without a layout round trip) ----------------
// pass 1: per (sample, channel) partial sums of dy' and dy' * xhat over a spatial slice
//         (dy' = dy behind the ReLU mask of y); same partial layout as gn_bwd_stats_kernel
// the ReLU mask of y = relu(x * a + b) from x (round 6): the forward's own expression -- fma, rounded through the
// storage type -- so a layer without a fused residual reads two tensors in pass 1 and three in pass 3 instead of
// three and four, and autograd keeps no second 118 MB tensor per layer alive
template <typename T>
__device__ __forceinline__ bool relu_on(float xv, float a, float b)
{
    return elem<T>::load(elem<T>::store(fmaxf(__builtin_fmaf(xv, a, b), 0.0f))) > 0.0f;
}

// XMASK: y is not read; the mask comes from x and the affine map (gamma, beta, mean, rstd)
template <typename T, bool XMASK>
__global__ __launch_bounds__(256) void gn_bwd_stats_cl_kernel(
    const T *__restrict__ dy, const T *__restrict__ x, const T *__restrict__ y, long long spatial, int C,
    int groups, int splits, int relu, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ partial)
{
    constexpr int VEC = vec16<T>::N;
    __shared__ float sh[256][2 * VEC + 1];
    const int n = blockIdx.y, s = blockIdx.x;
    const int nvb = C / VEC, vpi = 256 / nvb, cpg = C / groups;
    const int vb = threadIdx.x % nvb, v0 = threadIdx.x / nvb;
    const long long per = (spatial + splits - 1) / splits;
    const long long lo = min((long long)s * per, spatial), hi = min(lo + per, spatial);
    const size_t base = (size_t)n * spatial * C + (size_t)vb * VEC;
    float mu[VEC], rs[VEC], s1[VEC], s2[VEC];
    [[maybe_unused]] float fa[VEC], fb[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int grp = n * groups + (vb * VEC + k) / cpg;
        mu[k] = mean[grp]; rs[k] = rstd[grp]; s1[k] = 0.0f; s2[k] = 0.0f;
        if constexpr (XMASK) {
            fa[k] = rs[k] * gamma[vb * VEC + k];             // gn_apply_cl_kernel's a and b
            fb[k] = __builtin_fmaf(-mu[k], fa[k], beta[vb * VEC + k]);
        }
    }
    constexpr int U = 2;  // voxels in flight per lane
    auto take = [&](const float (&g)[VEC], const float (&xv)[VEC], const float (&yv)[VEC]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            bool on = true;
            if constexpr (XMASK) on = relu_on<T>(xv[k], fa[k], fb[k]);
            else on = !relu || yv[k] > 0.0f;
            const float gk = on ? g[k] : 0.0f;
            s1[k] += gk;
            s2[k] += gk * ((xv[k] - mu[k]) * rs[k]);
        }
    };
    long long v = lo + v0;
    for (; v + (long long)(U - 1) * vpi < hi; v += (long long)U * vpi) {
        float g[U][VEC], xv[U][VEC], yv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            load16<T>(dy + base + (size_t)(v + (long long)u * vpi) * C, g[u]);
            load16<T>(x + base + (size_t)(v + (long long)u * vpi) * C, xv[u]);
            if (!XMASK && relu) load16<T>(y + base + (size_t)(v + (long long)u * vpi) * C, yv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) take(g[u], xv[u], yv[u]);
    }
    for (; v < hi; v += vpi) {
        float g[VEC], xv[VEC], yv[VEC];
        load16<T>(dy + base + (size_t)v * C, g);
        load16<T>(x + base + (size_t)v * C, xv);
        if (!XMASK && relu) load16<T>(y + base + (size_t)v * C, yv);
        take(g, xv, yv);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) { sh[threadIdx.x][2 * k] = s1[k]; sh[threadIdx.x][2 * k + 1] = s2[k]; }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x, cvb = c / VEC, k = c % VEC;
        float a = 0.0f, b = 0.0f;
        for (int t = cvb; t < 256; t += nvb) { a += sh[t][2 * k]; b += sh[t][2 * k + 1]; }
        float *p = partial + ((size_t)(n * C + c) * splits + s) * 2;
        p[0] = a; p[1] = b;
    }
}

// pass 2 (tiny): per (sample, channel) the coefficients of dx = k1 * dy' + k2 * x + k3 and the
// parameter gradients: one workgroup per GROUP, walking the samples -- the parameter gradients are sums over
// the batch, so they are plain STORES (round 6: the per-(sample, group) workgroups added them atomically into
// buffers every caller had to zero first: 64 fill launches per training step of the stereo path, and an order
// of additions that changed from run to run)
__global__ __launch_bounds__(64) void gn_bwd_coef_kernel(const float *__restrict__ partial, int N, int C, int cpg,
                                                        int splits, long long spatial,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ rstd,
                                                        const float *__restrict__ gamma,
                                                        float *__restrict__ coef, float *__restrict__ dgamma,
                                                        float *__restrict__ dbeta)
{
    const int groups = C / cpg;
    const int grp = blockIdx.x;
    __shared__ float ab[2];
    const float invL = 1.0f / ((float)cpg * (float)spatial);
    for (int cc = grp * cpg + threadIdx.x; cc < (grp + 1) * cpg; cc += 64) { dbeta[cc] = 0.0f; dgamma[cc] = 0.0f; }
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        float A = 0.0f, B = 0.0f;
        for (int cc = grp * cpg; cc < (grp + 1) * cpg; ++cc) {
            float a = 0.0f, b = 0.0f;
            for (int k = threadIdx.x; k < splits; k += 64) {
                const float *p = partial + ((size_t)(n * C + cc) * splits + k) * 2;
                a += p[0]; b += p[1];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if (threadIdx.x == 0) { dbeta[cc] += a; dgamma[cc] += b; }   // (this workgroup owns the group's channels)
            A += gamma[cc] * a; B += gamma[cc] * b;
        }
        __syncthreads();  // (the previous sample's readers of ab are done)
        if (threadIdx.x == 0) { ab[0] = A; ab[1] = B; }
        __syncthreads();
        const int sg = n * groups + grp;
        const float mu = mean[sg], rs = rstd[sg];
        const float Am = ab[0] * invL, Bm = ab[1] * invL;
        // dx = rs * (gm * g - Am - xh * Bm), xh = (x - mu) * rs  ->  k1 g + k2 x + k3
        for (int cc = grp * cpg + threadIdx.x; cc < (grp + 1) * cpg; cc += 64) {
            float *o = coef + ((size_t)n * C + cc) * 3;
            o[0] = rs * gamma[cc];
            o[1] = -rs * rs * Bm;
            o[2] = rs * (rs * mu * Bm - Am);
        }
    }
}

// pass 3: dx = k1[c] * dy' + k2[c] * x + k3[c]; optionally the masked dy' itself (the gradient of
// a fused residual input)
template <typename T, bool XMASK>
__global__ __launch_bounds__(256) void gn_bwd_apply_cl_kernel(
    const T *__restrict__ dy, const T *__restrict__ x, const T *__restrict__ y, long long spatial, int C,
    int groups, int splits, int relu, const float *__restrict__ coef, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    T *__restrict__ dx, T *__restrict__ dres)
{
    constexpr int VEC = vec16<T>::N;
    const int n = blockIdx.y, s = blockIdx.x;
    const int nvb = C / VEC, vpi = 256 / nvb;
    const int vb = threadIdx.x % nvb, v0 = threadIdx.x / nvb;
    float k1[VEC], k2[VEC], k3[VEC];
    [[maybe_unused]] float fa[VEC], fb[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const float *o = coef + ((size_t)n * C + vb * VEC + k) * 3;
        k1[k] = o[0]; k2[k] = o[1]; k3[k] = o[2];
        if constexpr (XMASK) {
            const int grp = n * groups + (vb * VEC + k) / (C / groups);
            fa[k] = rstd[grp] * gamma[vb * VEC + k];
            fb[k] = __builtin_fmaf(-mean[grp], fa[k], beta[vb * VEC + k]);
        }
    }
    const long long per = (spatial + splits - 1) / splits;
    const long long lo = min((long long)s * per, spatial), hi = min(lo + per, spatial);
    const size_t base = (size_t)n * spatial * C + (size_t)vb * VEC;
    for (long long v = lo + v0; v < hi; v += vpi) {
        float g[VEC], xv[VEC], yv[VEC];
        load16<T>(dy + base + (size_t)v * C, g);
        load16<T>(x + base + (size_t)v * C, xv);
        if (!XMASK && relu) load16<T>(y + base + (size_t)v * C, yv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            bool on = true;
            if constexpr (XMASK) on = relu_on<T>(xv[k], fa[k], fb[k]);
            else on = !relu || yv[k] > 0.0f;
            if (!on) g[k] = 0.0f;
            xv[k] = k1[k] * g[k] + k2[k] * xv[k] + k3[k];
        }
        store16<T>(dx + base + (size_t)v * C, xv);
        if (dres) store16<T>(dres + base + (size_t)v * C, g);
    }
}

// (count, mean, M2) of `splits` moment partials p[3 k + {0, 1, 2}], in every thread of a 256-thread workgroup.  Plain
// sums around a reference mean (round 6: the convolutions' epilogues hand over 2000-7000 partials per group; the first
// form chained Chan's pairwise update -- a division per partial -- through one wave: 12-35 us a launch):
//   N = sum n_i,  S1 = sum n_i (mean_i - ref),  S2 = sum M2_i + n_i (mean_i - ref)^2
//   mean = ref + S1 / N,  M2 = S2 - S1^2 / N            (ref = the mean of the first non-empty partial)
__device__ __forceinline__ Moments merge_partials_256(const float *__restrict__ p, int splits, float (*sh)[4])
{
    const int tid = threadIdx.x;
    // the reference: the first partial with a count (the first 256 are looked at; any value serves, a mean of the data
    // keeps the shifted sums small)
    float ref = 0.0f;
    {
        float cand = 0.0f;
        int idx = 0x7fffffff;
        if (tid < splits && p[3 * tid] > 0.0f) { cand = p[3 * tid + 1]; idx = tid; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int oi = __shfl_xor(idx, o);
            const float oc = __shfl_xor(cand, o);
            if (oi < idx) { idx = oi; cand = oc; }
        }
        if ((tid & 63) == 0) { sh[tid >> 6][0] = cand; sh[tid >> 6][1] = __int_as_float(idx); }
        __syncthreads();
        int best = 0x7fffffff;
        for (int w = 0; w < 4; ++w) {
            const int wi = __float_as_int(sh[w][1]);
            if (wi < best) { best = wi; ref = sh[w][0]; }
        }
        __syncthreads();
    }
    float N = 0.0f, S1 = 0.0f, S2 = 0.0f;
    for (int k = tid; k < splits; k += 256) {
        const float n = p[3 * k], d = p[3 * k + 1] - ref, nd = n * d;
        N += n;
        S1 += nd;
        S2 += p[3 * k + 2] + nd * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        N += __shfl_xor(N, o);
        S1 += __shfl_xor(S1, o);
        S2 += __shfl_xor(S2, o);
    }
    if ((tid & 63) == 0) { sh[tid >> 6][0] = N; sh[tid >> 6][1] = S1; sh[tid >> 6][2] = S2; }
    __syncthreads();
    N = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
    S1 = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
    S2 = sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2];
    Moments r = {0.0f, 0.0f, 0.0f};
    if (N > 0.0f) {
        const float a = S1 / N;
        r = Moments{N, ref + a, fmaxf(S2 - S1 * a, 0.0f)};
    }
    return r;
}

// partials [n*groups][splits][3] -> merged [n*groups][3]; one workgroup per (sample, group)
__global__ __launch_bounds__(256) void gn_merge_partials_kernel(const float *__restrict__ partial,
                                                                int splits, float *__restrict__ merged)
{
    __shared__ float sh[4][4];
    const Moments r = merge_partials_256(partial + (size_t)blockIdx.x * splits * 3, splits, sh);
    if (threadIdx.x == 0) {
        float *o = merged + (size_t)blockIdx.x * 3;
        o[0] = r.n; o[1] = r.mean; o[2] = r.m2;
    }
}

// (a, b) of y = x * a + b per (sample, channel) from a producer's moment partials [n * groups][splits][3]: the
// merge of gn_merge_partials_kernel and the arithmetic of gn_apply_cl_kernel (mean / rstd of the group,
// a = rstd * gamma, b = beta - mean * a), for consumers that normalise ON LOAD (csrc/conv3d_to1n.hip); one wave per
// (sample, group)
__global__ __launch_bounds__(256) void gn_coefficients_kernel(const float *__restrict__ partial, int splits,
                                                              int C, int groups, float eps,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              float *__restrict__ coef)
{
    __shared__ float sh[4][4];
    const Moments r = merge_partials_256(partial + (size_t)blockIdx.x * splits * 3, splits, sh);
    const int cpg = C / groups;
    const int n = blockIdx.x / groups, gi = blockIdx.x - n * groups;
    const float mean = r.mean, rstd = 1.0f / sqrtf(r.m2 / r.n + eps);
    for (int c = gi * cpg + threadIdx.x; c < (gi + 1) * cpg; c += 256) {
        const float a = rstd * gamma[c];
        coef[((size_t)n * C + c) * 2] = a;
        coef[((size_t)n * C + c) * 2 + 1] = __builtin_fmaf(-mean, a, beta[c]);
    }
}

}  // namespace

extern "C" {

DFM_API int dfm_group_norm_coefficients(int32_t n, int32_t c, int32_t groups, float eps, const float *partials,
                                        int32_t splits, const float *gamma, const float *beta, float *coef,
                                        void *stream)
{
    if (n <= 0 || c <= 0 || groups <= 0 || c % groups || splits <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "bad sizes in dfm_group_norm_coefficients");
    if (!partials || !gamma || !beta || !coef) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    hipLaunchKernelGGL(gn_coefficients_kernel, dim3(n * groups), dim3(256), 0, (hipStream_t)stream, partials, splits,
                       c, groups, eps, gamma, beta, coef);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}


DFM_API size_t dfm_group_norm_workspace_bytes(int32_t n, int32_t c, int64_t spatial, int32_t groups)
{
    if (n <= 0 || c <= 0 || spatial <= 0 || groups <= 0 || c % groups) return 0;
    // forward partials (N*G*splits*3, splits <= 2048 -- round 6: 256 workgroups were one per CU, a slice of 28 load
    // round trips each) + the merged triples, and backward partials (N*C*splits*2, splits <= 256)
    // (+ n*c*4 floats: coefficients of the channels-last backward)
    const size_t fw = (size_t)n * groups * 2049 * 3, bw = (size_t)n * c * (GN_BW_SPLITS * 2 + 4);
    return ((fw > bw ? fw : bw) * sizeof(float) + 255) & ~(size_t)255;
}

DFM_API int dfm_group_norm_fwd(int32_t n, int32_t c, int64_t spatial, int32_t groups, float eps,
                               int32_t dtype, int32_t relu, const void *x, const float *gamma,
                               const float *beta, void *y, float *mean, float *rstd,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    if (n <= 0 || c <= 0 || spatial <= 0 || groups <= 0 || c % groups)
        return set_error(DFM_ERR_INVALID_ARG, "bad sizes in dfm_group_norm_fwd");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < dfm_group_norm_workspace_bytes(n, c, spatial, groups))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_group_norm_workspace_bytes");
    if ((long long)n * groups > 65535) return set_error(DFM_ERR_UNSUPPORTED, "n*groups > 65535");
    const int cpg = c / groups;
    const long long L = (long long)cpg * spatial;
    const int splits = pick_splits(L);
    dim3 grid(splits, n * groups);
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)workspace;
    if (dtype == DFM_F32) {
        hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, st, (const float *)x, L, splits,
                           partial);
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, st, (const float *)x, L,
                           (long long)spatial, cpg, groups, splits, eps, partial, gamma, beta, relu,
                           (float *)y, mean, rstd);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, L,
                           splits, partial);
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, L,
                           (long long)spatial, cpg, groups, splits, eps, partial, gamma, beta, relu,
                           (bf16_t *)y, mean, rstd);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_group_norm_fwd_channels_last(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                             float eps, int32_t dtype, int32_t relu, const void *x,
                                             const float *gamma, const float *beta, void *y,
                                             float *mean, float *rstd, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    return dfm_group_norm_fwd_channels_last_res(n, c, spatial, groups, eps, dtype, relu, x, gamma, beta, nullptr,
                                                y, mean, rstd, workspace, workspace_bytes, stream);
}

DFM_API int dfm_group_norm_fwd_channels_last_res(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                                 float eps, int32_t dtype, int32_t relu, const void *x,
                                                 const float *gamma, const float *beta,
                                                 const void *residual, void *y, float *mean, float *rstd,
                                                 void *workspace, size_t workspace_bytes, void *stream)
{
    if (n <= 0 || c <= 0 || spatial <= 0 || groups <= 0 || c % groups)
        return set_error(DFM_ERR_INVALID_ARG, "bad sizes in dfm_group_norm_fwd_channels_last");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < dfm_group_norm_workspace_bytes(n, c, spatial, groups))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_group_norm_workspace_bytes");
    const int vec = dtype == DFM_BF16 ? 8 : 4;
    const int nvb = c / vec;
    if (c % vec || c > 256 || (nvb & (nvb - 1)) || n > 65535 ||
        ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "channels-last GroupNorm needs C = 16-byte vectors x a power of two, C <= 256");
    const int splits = pick_splits_cl((long long)spatial * c);  // workgroups of the statistics and of the apply pass
    const int asplits = splits;
    dim3 grid(splits, n), agrid(asplits, n);
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)workspace;
    float *merged = partial + (size_t)n * groups * splits * 3;  // behind the partials
    if (dtype == DFM_F32)
        hipLaunchKernelGGL(gn_stats_cl_kernel<float>, grid, dim3(256), 0, st, (const float *)x,
                           (long long)spatial, c, groups, splits, partial);
    else
        hipLaunchKernelGGL(gn_stats_cl_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x,
                           (long long)spatial, c, groups, splits, partial);
    hipLaunchKernelGGL(gn_merge_partials_kernel, dim3(n * groups), dim3(256), 0, st, partial, splits, merged);
    if (dtype == DFM_F32)
        hipLaunchKernelGGL(gn_apply_cl_kernel<float>, agrid, dim3(256), 0, st, (const float *)x,
                           (long long)spatial, c, groups, asplits, eps, merged, gamma, beta, relu,
                           (float *)y, mean, rstd, 1, (const float *)residual);
    else
        hipLaunchKernelGGL(gn_apply_cl_kernel<bf16_t>, agrid, dim3(256), 0, st, (const bf16_t *)x,
                           (long long)spatial, c, groups, asplits, eps, merged, gamma, beta, relu,
                           (bf16_t *)y, mean, rstd, 1, (const bf16_t *)residual);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_group_norm_apply_channels_last(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                               float eps, int32_t dtype, int32_t relu, const void *x,
                                               const float *gamma, const float *beta, void *y,
                                               float *mean, float *rstd, const float *partials,
                                               int32_t splits, void *workspace,
                                               size_t workspace_bytes, void *stream)
{
    return dfm_group_norm_apply_channels_last_res(n, c, spatial, groups, eps, dtype, relu, x, gamma, beta,
                                                  nullptr, y, mean, rstd, partials, splits, workspace,
                                                  workspace_bytes, stream);
}

DFM_API int dfm_group_norm_apply_channels_last_res(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                                   float eps, int32_t dtype, int32_t relu, const void *x,
                                                   const float *gamma, const float *beta,
                                                   const void *residual, void *y, float *mean, float *rstd,
                                                   const float *partials, int32_t splits, void *workspace,
                                                   size_t workspace_bytes, void *stream)
{
    if (n <= 0 || c <= 0 || spatial <= 0 || groups <= 0 || c % groups || splits <= 0)
        return set_error(DFM_ERR_INVALID_ARG, "bad sizes in dfm_group_norm_apply_channels_last");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!x || !gamma || !beta || !y || !mean || !rstd || !partials || !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < (size_t)n * groups * 3 * sizeof(float))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than n*groups*3 floats");
    const int vec = dtype == DFM_BF16 ? 8 : 4;
    const int nvb = c / vec;
    if (c % vec || c > 256 || (nvb & (nvb - 1)) || n > 65535 ||
        ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "channels-last GroupNorm needs C = 16-byte vectors x a power of two, C <= 256");
    hipStream_t st = (hipStream_t)stream;
    float *merged = (float *)workspace;
    hipLaunchKernelGGL(gn_merge_partials_kernel, dim3(n * groups), dim3(256), 0, st, partials, splits, merged);
    const int asplits = pick_splits_cl((long long)spatial * c);  // workgroups of the apply pass
    dim3 grid(asplits, n);
    // the apply kernel slices the tensor by ITS grid and merges `1` partial per group
    if (dtype == DFM_F32)
        hipLaunchKernelGGL(gn_apply_cl_kernel<float>, grid, dim3(256), 0, st, (const float *)x,
                           (long long)spatial, c, groups, asplits, eps, merged, gamma, beta, relu,
                           (float *)y, mean, rstd, 1, (const float *)residual);
    else
        hipLaunchKernelGGL(gn_apply_cl_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x,
                           (long long)spatial, c, groups, asplits, eps, merged, gamma, beta, relu,
                           (bf16_t *)y, mean, rstd, 1, (const bf16_t *)residual);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_group_norm_bwd(int32_t n, int32_t c, int64_t spatial, int32_t groups, int32_t dtype,
                               int32_t relu, const void *grad_y, const void *x, const void *y,
                               const float *mean, const float *rstd, const float *gamma,
                               void *grad_x, float *grad_gamma, float *grad_beta, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    if (n <= 0 || c <= 0 || spatial <= 0 || groups <= 0 || c % groups)
        return set_error(DFM_ERR_INVALID_ARG, "bad sizes in dfm_group_norm_bwd");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!grad_y || !x || (relu && !y) || !mean || !rstd || !gamma || !grad_x || !grad_gamma ||
        !grad_beta || !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < dfm_group_norm_workspace_bytes(n, c, spatial, groups))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_group_norm_workspace_bytes");
    if ((long long)n * c > 65535) return set_error(DFM_ERR_UNSUPPORTED, "n*c > 65535");
    const int cpg = c / groups;
    const int splits = pick_splits(spatial);
    dim3 grid(splits, n * c);
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)workspace;
    if (dtype == DFM_F32) {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, grid, dim3(256), 0, st, (const float *)grad_y,
                           (const float *)x, (const float *)y, (long long)spatial, c, cpg, splits, relu,
                           mean, rstd, partial);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(256), 0, st, (const float *)grad_y,
                           (const float *)x, (const float *)y, (long long)spatial, c, cpg, splits, relu,
                           mean, rstd, gamma, partial, (float *)grad_x, grad_gamma, grad_beta);
    } else {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<bf16_t>, grid, dim3(256), 0, st,
                           (const bf16_t *)grad_y, (const bf16_t *)x, (const bf16_t *)y,
                           (long long)spatial, c, cpg, splits, relu, mean, rstd, partial);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, st,
                           (const bf16_t *)grad_y, (const bf16_t *)x, (const bf16_t *)y,
                           (long long)spatial, c, cpg, splits, relu, mean, rstd, gamma, partial,
                           (bf16_t *)grad_x, grad_gamma, grad_beta);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

static int gn_bwd_cl_impl(int32_t n, int32_t c, int64_t spatial, int32_t groups, int32_t dtype, int32_t relu,
                          const void *grad_y, const void *x, const void *y, const float *mean, const float *rstd,
                          const float *gamma, const float *beta, void *grad_x, void *grad_residual,
                          float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream)
{
    const bool xmask = relu && !y && beta;
    if (n <= 0 || c <= 0 || spatial <= 0 || groups <= 0 || c % groups)
        return set_error(DFM_ERR_INVALID_ARG, "bad sizes in dfm_group_norm_bwd_channels_last");
    if (dtype != DFM_F32 && dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (!grad_y || !x || (relu && !y && !beta) || !mean || !rstd || !gamma || !grad_x || !grad_gamma || !grad_beta ||
        !workspace)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < dfm_group_norm_workspace_bytes(n, c, spatial, groups))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_group_norm_workspace_bytes");
    const int vec = dtype == DFM_BF16 ? 8 : 4;
    const int nvb = c / vec;
    if (c % vec || c > 256 || (nvb & (nvb - 1)) || n > 65535 || (long long)n * groups > 2147483647ll ||
        ((uintptr_t)x & 15) || ((uintptr_t)grad_y & 15) || ((uintptr_t)grad_x & 15) || ((uintptr_t)y & 15) ||
        ((uintptr_t)grad_residual & 15))
        return set_error(DFM_ERR_UNSUPPORTED,
                         "channels-last GroupNorm needs C = 16-byte vectors x a power of two, C <= 256");
    // (round 6: up to GN_BW_SPLITS workgroups in the statistics pass -- 256 were one per CU, one voxel in flight a lane)
    const int asplits = pick_splits_cl((long long)spatial * c);
    const int splits = std::min(GN_BW_SPLITS, asplits);
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)workspace;
    float *coef = partial + (size_t)n * c * GN_BW_SPLITS * 2;
    dim3 grid(splits, n), agrid(asplits, n);
#define GN_BW(T_, X_)                                                                                             \
    do {                                                                                                          \
        hipLaunchKernelGGL((gn_bwd_stats_cl_kernel<T_, X_>), grid, dim3(256), 0, st, (const T_ *)grad_y,          \
                           (const T_ *)x, (const T_ *)y, (long long)spatial, c, groups, splits, relu, mean, rstd, \
                           gamma, beta, partial);                                                                 \
        hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(groups), dim3(64), 0, st, partial, n, c, c / groups, splits,  \
                           (long long)spatial, mean, rstd, gamma, coef, grad_gamma, grad_beta);                   \
        hipLaunchKernelGGL((gn_bwd_apply_cl_kernel<T_, X_>), agrid, dim3(256), 0, st, (const T_ *)grad_y,         \
                           (const T_ *)x, (const T_ *)y, (long long)spatial, c, groups, asplits, relu, coef, mean, \
                           rstd, gamma, beta, (T_ *)grad_x, (T_ *)grad_residual);                                 \
    } while (0)
    if (dtype == DFM_F32) { if (xmask) GN_BW(float, true); else GN_BW(float, false); }
    else { if (xmask) GN_BW(bf16_t, true); else GN_BW(bf16_t, false); }
#undef GN_BW
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_group_norm_bwd_channels_last(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                             int32_t dtype, int32_t relu, const void *grad_y, const void *x,
                                             const void *y, const float *mean, const float *rstd,
                                             const float *gamma, void *grad_x, void *grad_residual,
                                             float *grad_gamma, float *grad_beta, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    if (relu && !y) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    return gn_bwd_cl_impl(n, c, spatial, groups, dtype, relu, grad_y, x, y, mean, rstd, gamma, nullptr, grad_x,
                          grad_residual, grad_gamma, grad_beta, workspace, workspace_bytes, stream);
}

// the same for y = relu(GroupNorm(x)) WITHOUT a fused residual, y not kept: the ReLU mask is recomputed from x with
// the forward's expression (beta: the norm's bias, fp32 [c])
DFM_API int dfm_group_norm_bwd_channels_last_xmask(int32_t n, int32_t c, int64_t spatial, int32_t groups,
                                                   int32_t dtype, const void *grad_y, const void *x,
                                                   const float *mean, const float *rstd, const float *gamma,
                                                   const float *beta, void *grad_x, float *grad_gamma,
                                                   float *grad_beta, void *workspace, size_t workspace_bytes,
                                                   void *stream)
{
    if (!beta) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    return gn_bwd_cl_impl(n, c, spatial, groups, dtype, 1, grad_y, x, nullptr, mean, rstd, gamma, beta, grad_x,
                          nullptr, grad_gamma, grad_beta, workspace, workspace_bytes, stream);
}

}  // extern "C"

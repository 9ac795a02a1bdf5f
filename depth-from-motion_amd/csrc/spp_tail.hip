// spp_tail.hip -- the tail of SPPUNetNeck's pyramid-pooling branches, fused (gfx950, inference)
//
// Reference: mmdet3d/models/necks/spp_unet_neck.py:60-70 (branches: AvgPool2d(k, k) ->
// ConvModule(C_in -> C_spp, 1x1, GN, ReLU)) and :97-106 (every branch bilinearly up-sampled,
// align_corners=True, to the 1/4-resolution map and concatenated behind feats[start_level:]).
//
// As torch ops that is ~25 launches on a few hundred pixels each -- 1x1 convolutions on 5 .. 400
// pixels, GroupNorm statistics / merge / apply, ReLU, four up-samplings to (H, W), one concat --
// about 0.3 ms of device-side launch latency per neck for < 10 MFLOP and 26 MB of output.  Here:
//   spp_branch_kernel : one workgroup per (branch, sample): 1x1 convolution of the pooled pixels
//                       (fp32 accumulate, rounded to the storage type like the unfused tensor),
//                       GroupNorm with one channel per group (statistics over the branch's pixels in
//                       fp32), affine, ReLU -> a small fp32 map per branch;
//   spp_concat_kernel : one wave per output pixel: the 16-byte channel blocks of the NHWC row are
//                       either copied from the concatenated source maps or interpolated from a
//                       branch map (ATen's align_corners=True index / weight arithmetic): one coalesced
//                       row store per wave, no intermediate (H, W) tensors.
// bf16 NHWC in and out (the layout and type the channels_last neck runs in).  HBM-bound on the row
// copy: (sum C_src + C_out) * H * W * 2 bytes.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dfm_common.h"
#include "dfm_hip.h"

using namespace dfm;

namespace {

constexpr int SPP_MAX_BRANCH = 4;
constexpr int SPP_MAX_SRC = 4;

struct SppBranches {
    const bf16_t *pooled[SPP_MAX_BRANCH];  // (B, ho, wo, cin) bf16, NHWC
    const float *weight[SPP_MAX_BRANCH];   // (cspp, cin) fp32
    const float *gamma[SPP_MAX_BRANCH];    // (cspp)
    const float *beta[SPP_MAX_BRANCH];
    int ho[SPP_MAX_BRANCH], wo[SPP_MAX_BRANCH];
    int nbranch, cin, cspp, pmax;
    float eps;
};

// small[b][branch][p][c] fp32, p < ho*wo (pmax pixels reserved per branch).
// One workgroup per (branch, sample).  The weight is staged transposed in LDS ([k][c]: the 32 channel
// lanes of a pixel read consecutive floats), the pooled pixels come in tiles of 32 as fp32 ([p][k]:
// a broadcast read per pixel), thread = (pixel group, channel) accumulates 4 pixels at a time.
// (The first version read x and w straight from global memory, one scalar load pair per multiply-add
// with 32 cache lines per wave load: 0.4 ms for 2 MFLOP.)
constexpr int SPP_TP = 32;  // pixels per tile
__global__ __launch_bounds__(256) void spp_branch_kernel(SppBranches br, float *__restrict__ small)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int branch = blockIdx.x, b = blockIdx.y;
    const int P = br.ho[branch] * br.wo[branch];
    const int C = br.cspp, K = br.cin;
    float *wT = (float *)lds_raw;                 // [K][C]
    float *xs = wT + (size_t)K * C;               // [SPP_TP][K]
    bf16_t *ys = (bf16_t *)(xs + (size_t)SPP_TP * K);  // [P][C], conv output as stored (bf16)
    __shared__ float mean_s[64], rstd_s[64], red[2][256];  // [pass][pg * C + c] = one slot per thread, any C dividing 256
    const bf16_t *x = br.pooled[branch] + (size_t)b * P * K;
    const float *w = br.weight[branch];
    for (int i = threadIdx.x; i < C * K; i += 256) {
        const int c = i / K, k = i - c * K;       // coalesced read along k
        wT[k * C + c] = w[i];
    }
    const int c = threadIdx.x % C, pg = threadIdx.x / C, npg = 256 / C;  // C in {8, 16, 32, 64}: divides 256
    if (K % 16 == 0) {
        // round 6: the 1x1 convolution on the matrix cores -- D[pixel][channel] += X[pixel][k] W^T[k][channel],
        // v_mfma_f32_32x32x16_bf16.  The pooled pixels ARE bf16 (a lane's operand is one 16-byte load of 8
        // consecutive input channels of its pixel); the fp32 weights go in as hi + lo bf16 halves (two products per
        // k-step: 16 significand bits, the dropped term is 2^-17 of a product -- below the bf16 rounding of the
        // result by 2^-9).  A wave takes every fourth tile of 32 pixels.
        // (The scalar loop below read 5 LDS words per 4 multiply-adds and staged each tile through fp32 LDS behind
        // two barriers: 160 us for the 10 x 40-pixel branch of config K, on the critical path of the neck.)
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        typedef float f32x16_t __attribute__((ext_vector_type(16)));
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
        const int nks = K / 16;
        for (int n0 = 0; n0 < C; n0 += 32) {
            const int cn = n0 + l32;  // this lane's output channel (B operand column)
            for (int p0 = wave * 32; p0 < P; p0 += 4 * 32) {
                f32x16_t acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
                const int pm = p0 + l32;  // this lane's pixel (A operand row)
                for (int ks = 0; ks < nks; ++ks) {
                    uint4 xa = make_uint4(0u, 0u, 0u, 0u);
                    if (pm < P) xa = *(const uint4 *)(x + (size_t)pm * K + ks * 16 + half * 8);
                    float wf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) wf[j] = cn < C ? w[(size_t)cn * K + ks * 16 + half * 8 + j] : 0.0f;
                    bf16x8_t a, bh, bl;
                    __builtin_memcpy(&a, &xa, 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bf16_t hi = f32_to_bf16(wf[j]);
                        const bf16_t lo = f32_to_bf16(wf[j] - bf16_to_f32(hi));
                        __builtin_memcpy((char *)&bh + 2 * j, &hi, 2);
                        __builtin_memcpy((char *)&bl + 2 * j, &lo, 2);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
                }
                // acc[r]: row (pixel) 8 (r >> 2) + 4 half + (r & 3), column (channel) l32
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pr = p0 + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (pr < P && cn < C) ys[(size_t)pr * C + cn] = f32_to_bf16(acc[r]);
                }
            }
        }
    } else
    for (int p0 = 0; p0 < P; p0 += SPP_TP) {
        const int np = min(SPP_TP, P - p0);
        __syncthreads();  // wT ready / previous tile consumed
        for (int i = threadIdx.x; i < SPP_TP * K; i += 256) {
            const int p = i / K;
            xs[i] = p < np ? bf16_to_f32(x[(size_t)(p0 + p) * K + (i - p * K)]) : 0.0f;
        }
        __syncthreads();
        for (int q0 = pg; q0 < SPP_TP; q0 += 4 * npg) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int k = 0; k < K; ++k) {
                const float wv = wT[k * C + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = q0 + j * npg;
                    acc[j] = __builtin_fmaf(xs[(q < SPP_TP ? q : 0) * K + k], wv, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = q0 + j * npg;
                if (q < np) ys[(size_t)(p0 + q) * C + c] = f32_to_bf16(acc[j]);  // the unfused conv output is bf16
            }
        }
    }
    __syncthreads();
    // GroupNorm, one channel per group: biased statistics over the P pixels (two passes, fp32);
    // npg threads per channel, combined through LDS
    float s = 0.0f;
    for (int p = pg; p < P; p += npg) s += bf16_to_f32(ys[(size_t)p * C + c]);
    red[0][pg * C + c] = s;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float t = 0.0f;
        for (int g = 0; g < npg; ++g) t += red[0][g * C + threadIdx.x];
        mean_s[threadIdx.x] = t / (float)P;
    }
    __syncthreads();
    const float m = mean_s[c];
    float v = 0.0f;
    for (int p = pg; p < P; p += npg) {
        const float d = bf16_to_f32(ys[(size_t)p * C + c]) - m;
        v = __builtin_fmaf(d, d, v);
    }
    red[1][pg * C + c] = v;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float t = 0.0f;
        for (int g = 0; g < npg; ++g) t += red[1][g * C + threadIdx.x];
        rstd_s[threadIdx.x] = 1.0f / sqrtf(t / (float)P + br.eps);
    }
    __syncthreads();
    float *o = small + ((size_t)b * br.nbranch + branch) * br.pmax * C;
    const float ga = br.gamma[branch][c], be = br.beta[branch][c], rs = rstd_s[c];
    for (int p = pg; p < P; p += npg) {
        float y = (bf16_to_f32(ys[(size_t)p * C + c]) - m) * rs * ga + be;
        y = fmaxf(bf16_to_f32(f32_to_bf16(y)), 0.0f);  // GN output is a bf16 tensor, then ReLU
        o[(size_t)p * C + c] = y;
    }
}

struct SppConcat {
    const bf16_t *src[SPP_MAX_SRC];  // (B, H, W, csrc[i]) bf16 NHWC, copied in this order
    int csrc[SPP_MAX_SRC];
    int nsrc, H, W, ctot;
    int ho[SPP_MAX_BRANCH], wo[SPP_MAX_BRANCH];
    int nbranch, cspp, pmax;
};

// ATen upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
__device__ __forceinline__ void up2d(int dst, int in, int out, int &i0, int &i1, float &l1)
{
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    const float real = scale * (float)dst;
    i0 = min((int)real, in - 1);
    i1 = min(i0 + 1, in - 1);
    l1 = real - (float)i0;
}

// one wave per output pixel; lane = 16-byte block (8 channels) of its ctot-channel row
__global__ __launch_bounds__(256) void spp_concat_kernel(SppConcat cc, const float *__restrict__ small,
                                                         bf16_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (pix >= (long long)cc.H * cc.W) return;
    const int h = (int)(pix / cc.W), wq = (int)(pix - (long long)h * cc.W);
    const int nblk = cc.ctot / 8;
    uint4 *orow = (uint4 *)(out + ((size_t)b * cc.H * cc.W + pix) * cc.ctot);
    for (int blk = lane; blk < nblk; blk += 64) {
        int c0 = blk * 8;
        bool done = false;
        for (int s = 0; s < cc.nsrc; ++s) {
            if (c0 < cc.csrc[s]) {
                const uint4 *sp = (const uint4 *)(cc.src[s] + ((size_t)b * cc.H * cc.W + pix) * cc.csrc[s] + c0);
                orow[blk] = *sp;
                done = true;
                break;
            }
            c0 -= cc.csrc[s];
        }
        if (done) continue;
        const int branch = c0 / cc.cspp, cb = c0 - branch * cc.cspp;
        const int hi = cc.ho[branch], wi = cc.wo[branch];
        int y0, y1, x0, x1;
        float ly, lx;
        up2d(h, hi, cc.H, y0, y1, ly);
        up2d(wq, wi, cc.W, x0, x1, lx);
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float *m = small + ((size_t)b * cc.nbranch + branch) * cc.pmax * cc.cspp + cb;
        const float *p00 = m + (size_t)(y0 * wi + x0) * cc.cspp, *p01 = m + (size_t)(y0 * wi + x1) * cc.cspp;
        const float *p10 = m + (size_t)(y1 * wi + x0) * cc.cspp, *p11 = m + (size_t)(y1 * wi + x1) * cc.cspp;
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r[j] = hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]);  // ATen's expression
        uint4 v;
        v.x = pack_bf16x2(r[0], r[1]); v.y = pack_bf16x2(r[2], r[3]);
        v.z = pack_bf16x2(r[4], r[5]); v.w = pack_bf16x2(r[6], r[7]);
        orow[blk] = v;
    }
}

}  // namespace

extern "C" {

DFM_API size_t dfm_spp_tail_workspace_bytes(const dfm_spp_desc *d)
{
    if (!d || d->batch <= 0 || d->num_branches <= 0 || d->num_branches > SPP_MAX_BRANCH || d->spp_channels <= 0)
        return 0;
    int pmax = 0;
    for (int i = 0; i < d->num_branches; ++i) pmax = pmax > d->pooled_h[i] * d->pooled_w[i] ? pmax : d->pooled_h[i] * d->pooled_w[i];
    return ((size_t)d->batch * d->num_branches * pmax * d->spp_channels * sizeof(float) + 255) & ~(size_t)255;
}

DFM_API int dfm_spp_tail_fwd(const dfm_spp_desc *d, const void *const *pooled, const float *const *weight,
                             const float *const *gamma, const float *const *beta, const void *const *sources,
                             void *out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d || !pooled || !weight || !gamma || !beta || !sources || !out)
        return set_error(DFM_ERR_INVALID_ARG, "NULL pointer in dfm_spp_tail_fwd");
    if (d->batch <= 0 || d->batch > 65535 || d->h <= 0 || d->w <= 0 || d->num_branches <= 0 ||
        d->num_branches > SPP_MAX_BRANCH || d->num_sources < 0 || d->num_sources > SPP_MAX_SRC ||
        d->in_channels <= 0 || d->spp_channels <= 0 || d->spp_channels > 64 || d->spp_channels % 8)
        return set_error(DFM_ERR_UNSUPPORTED, "dfm_spp_tail_fwd: <= 4 branches / sources, spp_channels a multiple of 8, <= 64");
    SppBranches br{};
    SppConcat cc{};
    int pmax = 0, ctot = 0;
    for (int i = 0; i < d->num_sources; ++i) {
        if (d->source_channels[i] <= 0 || d->source_channels[i] % 8 || !sources[i] || ((uintptr_t)sources[i] & 15))
            return set_error(DFM_ERR_UNSUPPORTED, "source maps: whole 16-byte channel blocks, 16-byte aligned");
        cc.src[i] = (const bf16_t *)sources[i];
        cc.csrc[i] = d->source_channels[i];
        ctot += d->source_channels[i];
    }
    for (int i = 0; i < d->num_branches; ++i) {
        if (d->pooled_h[i] <= 0 || d->pooled_w[i] <= 0 || !pooled[i] || !weight[i] || !gamma[i] || !beta[i])
            return set_error(DFM_ERR_INVALID_ARG, "bad branch in dfm_spp_tail_fwd");
        br.pooled[i] = (const bf16_t *)pooled[i]; br.weight[i] = weight[i];
        br.gamma[i] = gamma[i]; br.beta[i] = beta[i];
        br.ho[i] = cc.ho[i] = d->pooled_h[i]; br.wo[i] = cc.wo[i] = d->pooled_w[i];
        pmax = pmax > d->pooled_h[i] * d->pooled_w[i] ? pmax : d->pooled_h[i] * d->pooled_w[i];
    }
    ctot += d->num_branches * d->spp_channels;
    if (256 % d->spp_channels || d->spp_channels < 8)
        return set_error(DFM_ERR_UNSUPPORTED, "spp_channels must be 8, 16, 32 or 64");
    const size_t lds = ((size_t)d->in_channels * d->spp_channels + (size_t)SPP_TP * d->in_channels) * sizeof(float) +
                       (size_t)pmax * d->spp_channels * 2;
    if (lds > 62 * 1024) return set_error(DFM_ERR_UNSUPPORTED, "pooled maps too large for the fused SPP tail");
    if (!workspace || workspace_bytes < dfm_spp_tail_workspace_bytes(d) || ((uintptr_t)out & 15))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_spp_tail_workspace_bytes / out misaligned");
    br.nbranch = cc.nbranch = d->num_branches; br.cin = d->in_channels; br.cspp = cc.cspp = d->spp_channels;
    br.pmax = cc.pmax = pmax; br.eps = d->eps;
    cc.nsrc = d->num_sources; cc.H = d->h; cc.W = d->w; cc.ctot = ctot;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(spp_branch_kernel, dim3(d->num_branches, d->batch), dim3(256), lds, st, br, (float *)workspace);
    const long long npix = (long long)d->h * d->w;
    hipLaunchKernelGGL(spp_concat_kernel, dim3((unsigned)((npix + 3) / 4), d->batch), dim3(256), 0, st, cc,
                       (const float *)workspace, (bf16_t *)out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

}  // extern "C"

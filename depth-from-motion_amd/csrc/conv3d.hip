// conv3d.hip -- hand-written MFMA implicit-GEMM Conv3d 3x3x3 (stride 1, pad 1) for the 3-D
// aggregation stacks of the path:
//   DfMBackbone dres0 / dres1 / pred convs   mmdet3d/models/backbones/dfm_backbone.py:50-128,175-201
//   convbn_3d                                 mmdet3d/models/utils/conv_modules.py:27-43
// Channels-last (NDHWC) bf16 activations, fp32 accumulation, C_in = C_out = 32 -- the width of
// every full-resolution convolution of config K (a 64 -> 32 dres0 runs as two 32-channel halves:
// conv(cat(cur, prev)) = conv_a(cur) + conv_b(prev), accumulated through the fp32 partial).
//
// Why not a generic implicit GEMM: with N = C_out = 32 every activation fragment feeds ONE MFMA,
// so a tiling that stages an (M x K) operand per tap moves 27x the input through LDS-DMA and is
// staging-bound at ~15 % of the MFMA peak (MIOpen: 350-370 TFLOP/s on these shapes,
// profiles/archive/r01_miopen_conv3d_baseline.txt).  This kernel instead:
//   * keeps ALL weights (27 taps x 32 x 32 bf16 = 54 fragments = 216 registers per lane) in the
//     register file of every wave for the whole launch (one wave per SIMD: 512 registers,
//     MFMA A operands straight from them) -- no LDS or cache traffic for weights at all;
//   * walks a workgroup (4 waves, 16 rows x 32 columns of output) along DEPTH with a ring of four
//     input depth slabs (18 x 34 pixels x 64 B, halo included) in LDS: every input element is
//     staged ONCE (x 1.2 for the halo) by LDS-DMA and used by all 27 taps; the slab of depth d+2
//     streams in while depth d is computed;
//   * orients the MFMA as D[cout][pixel] = W[cout][k] * X[k][pixel] (v_mfma_f32_32x32x16_bf16):
//     a wave owns 4 output rows x 32 pixels, loads the 6 input rows they touch once per
//     (kd, kw, 16-channel step) and reuses each row fragment for up to 3 kh taps
//     -> 6 LDS fragment reads per 12 MFMAs (0.5 KiB per MFMA, a quarter of the LDS peak);
//   * LDS image is XOR-swizzled at 16-byte granularity (slot ^= (pixel >> 2) & 3, applied to the
//     DMA source and to the reads) so the 64-byte pixel stride is bank-conflict free;
//   * epilogue: optional ReLU, bf16 (or fp32 partial) stores of 4 consecutive channels per lane;
//     a wave writes 32 pixels x 64 B = 2 KiB contiguous.
#include "dfm_common.h"

#include <cstdlib>

using namespace dfm;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

constexpr int CV_C = 32;                  // C_in = C_out
constexpr int CV_TH = 16, CV_TW = 32;     // output tile of a workgroup (rows x columns)
constexpr int CV_SH = CV_TH + 2, CV_SW = CV_TW + 2;
constexpr int CV_SLAB_PIX = CV_SH * CV_SW;           // 612
constexpr int CV_SLAB_BYTES = CV_SLAB_PIX * CV_C * 2;  // 39168
constexpr int CV_RING = 4;

#ifdef DFM_DEBUG_HOOKS
// s_memtime stamps of workgroup (0, 0, 0): trace[wave][plane < 16][stamp < 8] (dfm_debug_set_cv_trace, tools/conv_trace.py)
unsigned long long *g_cv_trace = nullptr;
#define CV_STAMP(i)                                                                                             \
    do {                                                                                                        \
        if (cvt && lane == 0 && d - d0 < 16) cvt[((size_t)wave * 16 + (d - d0)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CV_STAMP(i) do { } while (0)
#endif
constexpr int CV_PIECES = CV_SLAB_PIX * 4;           // 16-byte pieces per slab
constexpr int CV_ROUNDS = (CV_PIECES + 255) / 256;   // 10
constexpr int CV_NFRAG = 27 * 2;

struct ConvGeom {
    int32_t N, D, H, W;
    int32_t tiles_w, tiles_h, dchunk, relu;
    int32_t xcs;  // elements between consecutive input pixels (32, or more: a channel slice of a wider tensor)
    int32_t ycs;  // ... and between consecutive pixels of the bf16 output (32, or more: a slice of a wider tensor)
};

// weights (C_out, C_in, 3, 3, 3) -> MFMA A-operand fragments [tap*2 + ks][lane][8]:
// lane l holds W[cout = l & 31][cin = ks*16 + (l >> 5)*8 + j][tap], j = 0..7.
// transposed != 0: the fragments of the BACKWARD-DATA convolution (grad_in = conv(grad_out, W')
// with W'[o][i][tap] = W[i][cin_off + o][26 - tap]: channels swapped, taps mirrored), so the
// same kernel computes the input gradient.
template <typename TW>
__device__ __forceinline__ float conv3d_wload(const TW *w, size_t idx);
template <>
__device__ __forceinline__ float conv3d_wload<float>(const float *w, size_t idx) { return w[idx]; }
template <>
__device__ __forceinline__ float conv3d_wload<bf16_t>(const bf16_t *w, size_t idx) { return bf16_to_f32(w[idx]); }

template <typename TW>
__global__ void conv3d_pack_weights_kernel(const TW *__restrict__ w, int cin_total, int cin_off,
                                           int transposed, bf16_t *__restrict__ frag)
{
    const int f = blockIdx.x;  // fragment: tap*2 + ks
    const int l = threadIdx.x;
    const int tap = f >> 1, ks = f & 1;
    const int row = l & 31, k0 = ks * 16 + (l >> 5) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t idx = transposed
                               ? ((size_t)(k0 + j) * cin_total + cin_off + row) * 27 + (26 - tap)
                               : ((size_t)row * cin_total + cin_off + k0 + j) * 27 + tap;
        frag[((size_t)f * 64 + l) * 8 + j] = f32_to_bf16(conv3d_wload<TW>(w, idx));
    }
    // the zero page behind the fragments (out-of-bounds pixels read it): written here, by the first workgroup --
    // round 6: a hipMemsetAsync per pack was a launch of its own, ~50 per training step
    if (f == 0) {
        uint4 *z = (uint4 *)(frag + (size_t)CV_NFRAG * 64 * 8);
        for (int i = l; i < 256; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// OUT_F32: write the fp32 partial (N,D,H,W,32) instead of bf16;  ACC_IN: start from a fp32 partial
// STATS (bf16 output only): also emit, per (sample, channel, producing wave), the count / mean /
// M2 of the values it stored -- the per-channel GroupNorm statistics of the NEXT layer, so that
// layer's statistics pass over the tensor disappears (stats[((n*32 + c)*splits + s)*3 + {0,1,2}],
// s = (tile*chunks + chunk)*4 + wave, merged by dfm_group_norm_apply_channels_last).
// OUT_C1: store output channel 0 only, as a (N, D, H, W) bf16 tensor -- the 32 -> 1 prediction
// convolutions (dfm_backbone.py:120-127) run as a 32 -> 32 convolution whose weight rows 1..31 are zero
template <bool OUT_F32, bool ACC_IN, bool STATS, bool OUT_C1 = false>
__global__ __launch_bounds__(256, 1) void conv3d_k3_c32_kernel(
    ConvGeom g, const bf16_t *__restrict__ x, const uint4 *__restrict__ wfrag,
    const float *__restrict__ acc_in, void *__restrict__ yout, const uint4 *__restrict__ zero_page,
    float *__restrict__ stats
#ifdef DFM_DEBUG_HOOKS
    , unsigned long long *cv_trace
#endif
)
{
#ifdef DFM_DEBUG_HOOKS
    unsigned long long *cvt = (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) ? cv_trace : nullptr;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l32 = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x;
    const int tw = tile % g.tiles_w, th = tile / g.tiles_w;
    const int h0 = th * CV_TH, w0 = tw * CV_TW;
    const int d0 = blockIdx.y * g.dchunk, d1 = min(d0 + g.dchunk, g.D);
    const int n = blockIdx.z;
    const size_t plane = (size_t)g.H * g.W * g.xcs;  // input elements per depth plane
    const bf16_t *xn = x + (size_t)n * g.D * plane;

    // ---- weights: 54 fragments, register resident for the whole launch ------------------
    bf16x8_t wf[CV_NFRAG];
#pragma unroll
    for (int f = 0; f < CV_NFRAG; ++f) {
        const uint4 q = wfrag[f * 64 + lane];
        __builtin_memcpy(&wf[f], &q, 16);
    }

    // ---- per-lane DMA plan: byte offset inside a depth plane of each of its 16-byte pieces
    int poff[CV_ROUNDS];
#pragma unroll
    for (int k = 0; k < CV_ROUNDS; ++k) {
        const int q = k * 256 + tid;
        const int p = q >> 2, sl = (q & 3) ^ ((p >> 2) & 3);
        const int j = p / CV_SW, i = p - j * CV_SW;
        const int h = h0 - 1 + j, w = w0 - 1 + i;
        const bool ok = q < CV_PIECES && h >= 0 && h < g.H && w >= 0 && w < g.W;
        poff[k] = ok ? ((h * g.W + w) * g.xcs + sl * 8) * 2 : -1;
    }
    auto stage = [&](int dz) {  // depth dz -> ring slot (dz - d0 + 1) & 3
        const int slot = (dz - d0 + 1) & (CV_RING - 1);
        const bool zok = dz >= 0 && dz < g.D;
        const unsigned char *src = (const unsigned char *)(xn + (size_t)(zok ? dz : 0) * plane);
        unsigned char *dst = ring + slot * CV_SLAB_BYTES + wave * 1024;
#pragma unroll
        for (int k = 0; k < CV_ROUNDS; ++k) {
            if (k * 256 + tid < CV_PIECES) {
                const unsigned char *s = (zok && poff[k] >= 0) ? src + poff[k] : (const unsigned char *)zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)s,
                                                 (__attribute__((address_space(3))) void *)(dst + k * 4096),
                                                 16, 0, 0);
            }
        }
    };

    // LDS byte offset (inside a slab) of this lane's pixel in slab row (r0 + rr), before kw / ks
    const int r0 = wave * 4;
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;  // LDS offset
    // the 18 k-step-0 fragment addresses of a wave (6 slab rows x 3 column shifts) inside ring slot 0, computed
    // ONCE: a read is this + the slot's (workgroup-uniform) base, k-step 1 the same ^ 32 (slot ^ 2).  Re-deriving
    // pix / the swizzle per read cost 3 VALU operations per MFMA (round-5 counters: 3.8 VALU per MFMA in this
    // kernel, one wave per SIMD -- over the ~5 issue slots an MFMA hides)
    uint32_t fa[6][3];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int pix = (r0 + rr) * CV_SW + l32 + kw;
            fa[rr][kw] = ring_base + pix * 64 + (((half ^ ((pix >> 2) & 3))) << 4);
        }

    // prologue: slabs d0-1, d0, d0+1
    stage(d0 - 1);
    stage(d0);
    stage(d0 + 1);
    __syncthreads();

    const bool colok = w0 + l32 < g.W;
    // STATS: shifted sums of this lane's stored values per channel (16 of the 32 channels live in
    // a lane): s1 = sum(v - K), s2 = sum((v - K)^2) with K = the first value seen -- as robust as
    // Welford for |mean| >> std, 3 VALU ops per value
    float sK[16], s1[16], s2[16];
    float scnt = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { sK[t] = 0.0f; s1[t] = 0.0f; s2[t] = 0.0f; }
    for (int d = d0; d < d1; ++d) {
        CV_STAMP(0);
        if (d + 2 <= d1) stage(d + 2);  // needed by depth d+1 (<= d1-1) as its kd=2 slab
        CV_STAMP(1);
        f32x16_t acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (ACC_IN) {
                const int h = h0 + r0 + r;
                const bool ok = colok && h < g.H;
                const float *ap = acc_in + ((((size_t)n * g.D + d) * g.H + (ok ? h : 0)) * g.W + (ok ? w0 + l32 : 0)) * CV_C;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v = ok ? *(const float4 *)(ap + 8 * gq + 4 * half) : make_float4(0, 0, 0, 0);
                    acc[r][4 * gq + 0] = v.x; acc[r][4 * gq + 1] = v.y;
                    acc[r][4 * gq + 2] = v.z; acc[r][4 * gq + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[r][t] = 0.0f;
            }
        }
        // 18 steps (kd, kw, ks); the 6 row fragments of step s+1 are in flight while the 12 MFMAs
        // of step s run (inline-asm LDS reads: hipcc would otherwise drain the LDS-DMA of the
        // next slab with vmcnt(0) before the first read)
        u32x4_t q[2][6];
        auto issue = [&](int step, u32x4_t (&dst)[6]) {
            const int kd = step / 6, kw = (step / 2) % 3, ks = step & 1;
            const int slot = (d - d0 + kd) & (CV_RING - 1);  // depth d-1+kd
            const uint32_t sbase = (uint32_t)slot * CV_SLAB_BYTES;  // uniform
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) {
                const uint32_t a0 = fa[rr][kw] + sbase;
                const uint32_t a = ks ? (a0 ^ 32u) : a0;
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[rr]) : "v"(a));
            }
        };
        issue(0, q[0]);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int cb = step & 1, nb = cb ^ 1;
            if (step + 1 < 18) {
                issue(step + 1, q[nb]);
                asm volatile("s_waitcnt lgkmcnt(6)"
                             : "+v"(q[cb][0]), "+v"(q[cb][1]), "+v"(q[cb][2]), "+v"(q[cb][3]),
                               "+v"(q[cb][4]), "+v"(q[cb][5]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(q[cb][0]), "+v"(q[cb][1]), "+v"(q[cb][2]), "+v"(q[cb][3]),
                               "+v"(q[cb][4]), "+v"(q[cb][5]));
            }
            const int kd = step / 6, kw = (step / 2) % 3, ks = step & 1;
            bf16x8_t xf[6];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) __builtin_memcpy(&xf[rr], &q[cb][rr], 16);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int f = (((kd * 3 + kh) * 3 + kw) << 1) | ks;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f], xf[r + kh], acc[r], 0, 0, 0);
            }
        }
        CV_STAMP(2);
        // ---- epilogue: lane = pixel (w0 + l32), 4 groups of 4 consecutive channels
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = h0 + r0 + r;
            if (!(colok && h < g.H)) continue;
            const size_t vox = (((size_t)n * g.D + d) * g.H + h) * g.W + w0 + l32;
            if constexpr (OUT_C1) {
                if (half == 0) {  // channel 0 lives in element 0 of the lanes 0..31
                    float v = acc[r][0];
                    if (g.relu) v = fmaxf(v, 0.f);
                    ((bf16_t *)yout)[vox] = f32_to_bf16(v);
                }
                continue;
            }
            [[maybe_unused]] dfm_u32x2 pk4[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float v0 = acc[r][4 * gq], v1 = acc[r][4 * gq + 1], v2 = acc[r][4 * gq + 2], v3 = acc[r][4 * gq + 3];
                if (g.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                const int c = 8 * gq + 4 * half;
                if constexpr (OUT_F32) {
                    *(float4 *)((float *)yout + vox * CV_C + c) = make_float4(v0, v1, v2, v3);
                } else {
                    const u32x2_t pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    pk4[gq] = dfm_u32x2{pk.x, pk.y};
                    if constexpr (STATS) {
                        // the values as stored (bf16-rounded): what the normalisation will read
                        const float q[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u),
                                            __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int t = 4 * gq + j;
                            // the shift is COMMON to the 32 pixel lanes of a half (the value of its
                            // first lane in the wave's first row): the lanes' sums then add up as they
                            // are -- no count / mean / M2 merge tree with a division per step at the end
                            if (d == d0 && r == 0) sK[t] = __shfl(q[j], lane & 32);
                            const float dv = q[j] - sK[t];
                            s1[t] += dv;
                            s2[t] = __builtin_fmaf(dv, dv, s2[t]);
                        }
                    }
                }
            }
            if constexpr (!OUT_F32) {
                // two 16-byte stores per lane (the halves of the wave trade pieces of their pixel: dfm_common.h,
                // acc_rows_to_16B) instead of four 8-byte ones; a pixel's two lanes are stored or skipped together
                dfm_u32x4 q16[2];
                acc_rows_to_16B(pk4, q16);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
                    *(dfm_u32x4 *)((bf16_t *)yout + vox * g.ycs + 16 * pr + 8 * half) = q16[pr];
            }
            if constexpr (STATS && !OUT_F32) scnt += 1.0f;
        }
        CV_STAMP(3);
        __syncthreads();  // slab d+2 has landed (vmcnt drained) and slot (d-1) may be refilled
        CV_STAMP(4);
    }
    if constexpr (STATS && !OUT_F32) {
        // shifted sums of the 32 pixel lanes of each half added up (same shift K in all of them),
        // then (count, mean, M2) per channel, written by lanes 0 and 32
        const int splits = g.tiles_w * g.tiles_h * gridDim.y * 4;
        const int sidx = (blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave;
        float cn = scnt;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) cn += __shfl_xor(cn, o);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float a1 = s1[t], a2 = s2[t];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                a1 += __shfl_xor(a1, o);
                a2 += __shfl_xor(a2, o);
            }
            float mean = 0.0f, m2 = 0.0f;
            if (cn > 0.0f) {
                const float a = a1 / cn;
                mean = sK[t] + a;
                m2 = fmaxf(a2 - a1 * a, 0.0f);
            }
            if (l32 == 0) {
                const int c = (t & 3) + 8 * (t >> 2) + 4 * half;
                float *o3 = stats + (((size_t)n * CV_C + c) * splits + sidx) * 3;
                o3[0] = cn; o3[1] = mean; o3[2] = m2;
            }
        }
    }
}

}  // namespace

#ifdef DFM_DEBUG_HOOKS
#define CV_TRACE_ARG , g_cv_trace
extern "C" DFM_API void dfm_debug_set_cv_trace(void *buf) { g_cv_trace = (unsigned long long *)buf; }
#else
#define CV_TRACE_ARG
#endif

namespace {
// depth chunk: one workgroup per CU at a time (156 KB of LDS), every workgroup walks `dc` planes behind a prologue of
// three staged slabs (~1.5 plane times): the launch takes rounds(dc) x (dc + 1.5) plane times with
// rounds = ceil(columns x chunks / 256).  Rounds 1-4 took "at least 8 planes, about four rounds": config K (50 columns,
// 72 planes) got 9 chunks = 450 workgroups = 1.76 rounds, i.e. two rounds of 8 planes where ONE round of 15 does it.
int conv_depth_chunk(int n, int d, int h, int w, int depth_chunk)
{
    int dc = depth_chunk;
    if (dc <= 0) {
        const long long cols = (long long)((w + CV_TW - 1) / CV_TW) * ((h + CV_TH - 1) / CV_TH) * n;
        double best = 1e30;
        dc = d;
        static const bool old_rule = [] { const char *e = getenv("DFM_CONV_OLD_CHUNK"); return e && e[0] == '1'; }();
        if (old_rule) {  // (A/B runs: the rounds 1-4 rule)
            const long long chunks = (4 * 256 + cols - 1) / cols;
            return std::min((int)std::max<long long>(8, (d + chunks - 1) / chunks), d);
        }
        for (int c = std::min(d, 4); c <= d; ++c) {
            const long long chunks = (d + c - 1) / c;
            const long long rounds = (cols * chunks + 255) / 256;
            const double cost = (double)rounds * (c + 1.5);
            if (cost < best - 1e-9) { best = cost; dc = c; }
        }
    }
    return std::min(dc, d);
}
}  // namespace

// number of statistics partials per (sample, channel) a forward call with these sizes emits
extern "C" DFM_API int dfm_conv3d_k3_c32_stats_splits(int32_t n, int32_t d, int32_t h, int32_t w,
                                                      int32_t depth_chunk)
{
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    const int dc = conv_depth_chunk(n, d, h, w, depth_chunk);
    return ((w + CV_TW - 1) / CV_TW) * ((h + CV_TH - 1) / CV_TH) * ((d + dc - 1) / dc) * 4;
}

extern "C" DFM_API size_t dfm_conv3d_k3_c32_weight_bytes(void) { return (size_t)CV_NFRAG * 64 * 16 + 4096; }

extern "C" DFM_API int dfm_conv3d_k3_c32_pack_weights(const void *weight, int32_t weight_dtype,
                                                      int32_t cin_total, int32_t cin_offset,
                                                      int32_t transposed, void *packed, void *stream)
{
    if (!weight || !packed) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (cin_total < 32 || cin_offset < 0 || cin_offset + 32 > cin_total)
        return set_error(DFM_ERR_INVALID_ARG, "need a 32-channel slice of the input channels");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    // (the zero page behind the fragments is the pack kernel's)
    if (weight_dtype == DFM_F32)
        hipLaunchKernelGGL(conv3d_pack_weights_kernel<float>, dim3(CV_NFRAG), dim3(64), 0, st,
                           (const float *)weight, cin_total, cin_offset, transposed, (bf16_t *)packed);
    else
        hipLaunchKernelGGL(conv3d_pack_weights_kernel<bf16_t>, dim3(CV_NFRAG), dim3(64), 0, st,
                           (const bf16_t *)weight, cin_total, cin_offset, transposed, (bf16_t *)packed);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_conv3d_k3_c32_to1_fwd(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                                 const void *packed_weights, void *out, int32_t relu,
                                                 int32_t depth_chunk, void *stream)
{
    const int x_channel_stride = 32;
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    if (!x || !packed_weights || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if ((long long)h * w * x_channel_stride * 2 >= (1ll << 31)) return set_error(DFM_ERR_UNSUPPORTED, "depth plane too large");
    if (n > 65535) return set_error(DFM_ERR_UNSUPPORTED, "batch > 65535");
    ConvGeom g;
    g.N = n; g.D = d; g.H = h; g.W = w; g.xcs = x_channel_stride; g.ycs = 32;
    g.tiles_w = (w + CV_TW - 1) / CV_TW;
    g.tiles_h = (h + CV_TH - 1) / CV_TH;
    g.relu = relu ? 1 : 0;
    const int dc = conv_depth_chunk(n, d, h, w, depth_chunk);
    g.dchunk = dc;
    const int nchunks = (d + dc - 1) / dc;
    if (nchunks > 65535) return set_error(DFM_ERR_UNSUPPORTED, "too many depth chunks");
    const uint4 *wfrag = (const uint4 *)packed_weights;
    const uint4 *zero = wfrag + CV_NFRAG * 64;
    const int lds = CV_RING * CV_SLAB_BYTES;
    {
        const int rc_ = ensure_dynamic_lds((const void *)conv3d_k3_c32_kernel<false, false, false, true>, lds);
        if (rc_ != DFM_OK) return rc_;
    }
    hipLaunchKernelGGL((conv3d_k3_c32_kernel<false, false, false, true>), dim3(g.tiles_w * g.tiles_h, nchunks, n),
                       dim3(256), lds, (hipStream_t)stream, g, (const bf16_t *)x, wfrag, (const float *)nullptr,
                       out, zero, (float *)nullptr CV_TRACE_ARG);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

static int conv_c32_impl(int32_t n, int32_t d, int32_t h, int32_t w, const void *x, int32_t x_channel_stride,
                         const void *packed_weights, const float *acc_in, void *out, int32_t out_channel_stride,
                         int32_t out_f32, int32_t relu, int32_t depth_chunk, float *stats, void *stream);

extern "C" DFM_API int dfm_conv3d_k3_c32_fwd_strided(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                                     int32_t x_channel_stride, const void *packed_weights,
                                                     const float *acc_in, void *out, int32_t out_f32,
                                                     int32_t relu, int32_t depth_chunk, float *stats,
                                                     void *stream)
{
    return conv_c32_impl(n, d, h, w, x, x_channel_stride, packed_weights, acc_in, out, 32, out_f32, relu, depth_chunk,
                         stats, stream);
}

// ... whose bf16 OUTPUT is a 32-channel slice of a wider channels-last tensor too: out points at the slice's first
// channel of the first pixel, out_channel_stride (>= 32, a multiple of 8) elements lie between pixels.  The
// backward-data of a 32 k -> 32 convolution writes its k halves straight into the (N, D, H, W, 32 k) gradient
// (round 6: torch.cat of the halves and its layout conversion were two copies of a 236 MB tensor per step).
extern "C" DFM_API int dfm_conv3d_k3_c32_fwd_slices(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                                    int32_t x_channel_stride, const void *packed_weights,
                                                    void *out, int32_t out_channel_stride, int32_t relu,
                                                    int32_t depth_chunk, void *stream)
{
    if (out_channel_stride < 32 || out_channel_stride % 8 || ((uintptr_t)out & 15))
        return set_error(DFM_ERR_INVALID_ARG, "out_channel_stride must be >= 32 and a multiple of 8, out 16-byte aligned");
    if ((long long)n * d * h * w * out_channel_stride >= (1ll << 40))
        return set_error(DFM_ERR_UNSUPPORTED, "output too large");
    return conv_c32_impl(n, d, h, w, x, x_channel_stride, packed_weights, nullptr, out, out_channel_stride, 0, relu,
                         depth_chunk, nullptr, stream);
}

static int conv_c32_impl(int32_t n, int32_t d, int32_t h, int32_t w, const void *x, int32_t x_channel_stride,
                         const void *packed_weights, const float *acc_in, void *out, int32_t out_channel_stride,
                         int32_t out_f32, int32_t relu, int32_t depth_chunk, float *stats, void *stream)
{
    if (x_channel_stride < 32 || x_channel_stride % 8)
        return set_error(DFM_ERR_INVALID_ARG, "x_channel_stride must be >= 32 and a multiple of 8");
    if (stats && out_f32) return set_error(DFM_ERR_INVALID_ARG, "statistics are taken of the bf16 output");
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    if (!x || !packed_weights || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if ((long long)h * w * x_channel_stride * 2 >= (1ll << 31)) return set_error(DFM_ERR_UNSUPPORTED, "depth plane too large");
    if (n > 65535) return set_error(DFM_ERR_UNSUPPORTED, "batch > 65535");
    ConvGeom g;
    g.N = n; g.D = d; g.H = h; g.W = w; g.xcs = x_channel_stride; g.ycs = out_channel_stride;
    g.tiles_w = (w + CV_TW - 1) / CV_TW;
    g.tiles_h = (h + CV_TH - 1) / CV_TH;
    g.relu = relu ? 1 : 0;
    const int dc = conv_depth_chunk(n, d, h, w, depth_chunk);
    g.dchunk = dc;
    const int nchunks = (d + dc - 1) / dc;
    if (nchunks > 65535) return set_error(DFM_ERR_UNSUPPORTED, "too many depth chunks");
    const uint4 *wfrag = (const uint4 *)packed_weights;
    const uint4 *zero = wfrag + CV_NFRAG * 64;
    const int lds = CV_RING * CV_SLAB_BYTES;
    dim3 grid(g.tiles_w * g.tiles_h, nchunks, n);
    hipStream_t st = (hipStream_t)stream;
#define CV_LAUNCH(F32, ACC, ST)                                                                        \
    do {                                                                                           \
        const int rc_ = ensure_dynamic_lds((const void *)conv3d_k3_c32_kernel<F32, ACC, ST>, lds); \
        if (rc_ != DFM_OK) return rc_;                                                             \
        hipLaunchKernelGGL((conv3d_k3_c32_kernel<F32, ACC, ST>), grid, dim3(256), lds, st, g,      \
                           (const bf16_t *)x, wfrag, acc_in, out, zero, stats CV_TRACE_ARG);       \
    } while (0)
    if (out_f32) { if (acc_in) CV_LAUNCH(true, true, false); else CV_LAUNCH(true, false, false); }
    else if (stats) { if (acc_in) CV_LAUNCH(false, true, true); else CV_LAUNCH(false, false, true); }
    else { if (acc_in) CV_LAUNCH(false, true, false); else CV_LAUNCH(false, false, false); }
#undef CV_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_conv3d_k3_c32_fwd(int32_t n, int32_t d, int32_t h, int32_t w, const void *x,
                                             const void *packed_weights, const float *acc_in,
                                             void *out, int32_t out_f32, int32_t relu,
                                             int32_t depth_chunk, float *stats, void *stream)
{
    return dfm_conv3d_k3_c32_fwd_strided(n, d, h, w, x, 32, packed_weights, acc_in, out, out_f32, relu,
                                         depth_chunk, stats, stream);
}

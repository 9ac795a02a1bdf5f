// cost_gate.hip -- the gate that blends the stereo and the mono cost at the end of DfMBackbone.forward.
//
// Reference (mmdet3d/models/backbones/dfm_backbone.py:136-141):
//   cost   = cat(cost1, mono_cost1)                       (B, 2D, H, W)   the two 32 -> 1 predictions, squeezed
//   weight = aggregate_cost(cost).unsqueeze(1).sigmoid()  Conv2d(2D -> D, kernel 1, bias=False)
//   cost   = weight * cost1 + (1 - weight) * mono_cost1   (B, 1, D, H, W)
// As torch operations that is a concatenation, a (D x 2D) GEMM over the flattened image, a sigmoid and four
// elementwise kernels -- seven launches of 5-15 us behind the point where the two stacks of the backbone
// join, i.e. on the critical path of a forward pass that takes 2.3 ms.  Here: one launch.
//
// A workgroup takes 64 pixels of one sample; its four waves take a quarter of the D output planes each
// (<= 24 accumulators per lane).  The weights lie in LDS as fp32, [k][d] (packed once per weight version by
// dfm_cost_gate_pack_weights: staged from the (D, 2D) tensor by the kernel itself, one 2-byte load in flight per
// lane, the staging alone took longer than the seven launches it replaces), so that the weights of one input
// plane k for a wave's planes are consecutive: read as broadcast 16-byte vectors.  Per input plane a lane loads
// one value (consecutive pixels: coalesced), two batches of 48 planes in flight.  fp32 throughout, ONE rounding at the store
// (the torch sequence rounds the logit, the gate and three intermediate products to the storage type).
// Bound: launch latency (0.35 GFLOP and 11 MB at config K); inference only -- with autograd recording the
// module keeps the torch operations.
#include "dfm_common.h"

using namespace dfm;

namespace {

constexpr int CG_DQ = 24;      // output planes per wave, at most
constexpr int CG_MAX_D = 4 * CG_DQ;
constexpr int CG_UNROLL = 48;  // input planes per batch (two batches of loads in flight per lane)

template <typename T>
__device__ __forceinline__ float cg_load(const T *p)
{
    if constexpr (sizeof(T) == 2) return bf16_to_f32(*p); else return *p;
}
template <typename T>
__device__ __forceinline__ void cg_store(T *p, float v)
{
    if constexpr (sizeof(T) == 2) *p = f32_to_bf16(v); else *p = v;
}

// weight (D, 2D) row-major -> packed[k][wave * dqp + j] fp32, d = wave * dq + j; zero where d >= D
template <typename TW>
__global__ __launch_bounds__(256) void cost_gate_pack_kernel(int D, const TW *__restrict__ weight, float *__restrict__ packed)
{
    const int dq = (D + 3) / 4, pitch = 4 * ((dq + 3) / 4) * 4, dqp = pitch / 4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * D * pitch) return;
    const int k = i / pitch, r = i - k * pitch, w = r / dqp, j = r - w * dqp, d = w * dq + j;
    float v = 0.0f;
    if (j < dq && d < D) {
        if constexpr (sizeof(TW) == 2) v = bf16_to_f32(weight[(size_t)d * 2 * D + k]); else v = weight[(size_t)d * 2 * D + k];
    }
    packed[i] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void cost_gate_kernel(int D, long long HW, const T *__restrict__ stereo,
                                                        const T *__restrict__ mono, const float4 *__restrict__ packed,
                                                        T *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [2D][pitch]: input plane k, output plane d
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int dq = (D + 3) / 4;               // output planes per wave
    const int pitch = 4 * ((dq + 3) / 4) * 4;  // floats per k row: 4 waves x dq rounded up to whole vectors
    const int dqp = pitch / 4;
    // the packed weights into LDS: 16-byte vectors, several in flight per lane
    const int nvec = 2 * D * pitch / 4;
#pragma unroll 4
    for (int i = tid; i < nvec; i += 256) ((float4 *)wl)[i] = packed[i];
    __syncthreads();
    const int b = blockIdx.y;
    const long long p = (long long)blockIdx.x * 64 + lane;
    const long long pc = min(p, HW - 1);  // lanes past the image read the last pixel and store nothing
    const T *sb = stereo + (size_t)b * D * HW + pc, *mb = mono + (size_t)b * D * HW + pc;
    float acc[CG_DQ];
#pragma unroll
    for (int j = 0; j < CG_DQ; ++j) acc[j] = 0.0f;
    const float *wrow = wl + wave * dqp;
    const int nv = (dq + 3) / 4;
    // two register buffers of CG_UNROLL input planes: the loads of the next batch are in flight while this batch is
    // multiplied in (with 8 planes per batch and nothing ahead the kernel was 18 exposed load latencies long: 40 us)
    auto load_batch = [&](float (&v)[CG_UNROLL], int k0) {
#pragma unroll
        for (int u = 0; u < CG_UNROLL; ++u) {
            const int k = min(k0 + u, 2 * D - 1);
            v[u] = cg_load(k < D ? sb + (size_t)k * HW : mb + (size_t)(k - D) * HW);
        }
    };
    auto fma_batch = [&](const float (&v)[CG_UNROLL], int k0) {
#pragma unroll
        for (int u = 0; u < CG_UNROLL; ++u) {
            if (k0 + u >= 2 * D) break;
            const float4 *wv = (const float4 *)(wrow + (size_t)(k0 + u) * pitch);
#pragma unroll
            for (int q = 0; q < CG_DQ / 4; ++q) {
                if (q >= nv) break;
                const float4 w4 = wv[q];
                acc[4 * q + 0] = __builtin_fmaf(w4.x, v[u], acc[4 * q + 0]);
                acc[4 * q + 1] = __builtin_fmaf(w4.y, v[u], acc[4 * q + 1]);
                acc[4 * q + 2] = __builtin_fmaf(w4.z, v[u], acc[4 * q + 2]);
                acc[4 * q + 3] = __builtin_fmaf(w4.w, v[u], acc[4 * q + 3]);
            }
        }
    };
    float va[CG_UNROLL], vb[CG_UNROLL];
    load_batch(va, 0);
    for (int k0 = 0; k0 < 2 * D; k0 += 2 * CG_UNROLL) {
        load_batch(vb, k0 + CG_UNROLL);
        fma_batch(va, k0);
        load_batch(va, k0 + 2 * CG_UNROLL);
        fma_batch(vb, k0 + CG_UNROLL);
    }
    if (p >= HW) return;
    T *ob = out + (size_t)b * D * HW + p;
#pragma unroll
    for (int j = 0; j < CG_DQ; ++j) {
        const int d = wave * dq + j;
        if (j >= dq || d >= D) break;
        const float g = 1.0f / (1.0f + __expf(-acc[j]));
        const float s = cg_load(sb + (size_t)d * HW), m = cg_load(mb + (size_t)d * HW);
        cg_store(ob + (size_t)d * HW, g * s + (1.0f - g) * m);
    }
}

// ---- round 6: the same gate on the matrix cores (bf16 costs AND a bf16 weight: every product exact, fp32 sums) -------
// The kernel above makes each of a workgroup's four waves load all 2D input planes of its 64 pixels (each wave owns a
// quarter of the OUTPUT planes) and reads its weights as broadcast LDS vectors: 42 us at config K, alone on the main
// stream behind the join of the backbone's two stacks.  As a product out[d][p] = sum_k W[d][k] x[k][p] it is 3 x 9
// v_mfma_f32_32x32x16_bf16 per 32 pixels: a wave loads its pixels' 2D values ONCE, straight into B-operand order
// (lane = pixel, 8 consecutive planes per k-step half), the A operands are pre-packed fragments (27 KB, cache
// resident), and the epilogue (sigmoid, blend, one rounding) is the kernel's above.
static_assert(CG_MAX_D <= 96, "three row blocks of 32 output planes at most");

template <typename TW>
__global__ __launch_bounds__(64) void cost_gate_mfma_pack_kernel(int D, const TW *__restrict__ weight, bf16_t *__restrict__ frag)
{
    // blockIdx.x = mb * nks + ks; lane l: row d = mb * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + j
    const int nks = (2 * D + 15) / 16;
    const int mb = blockIdx.x / nks, ks = blockIdx.x - mb * nks, l = threadIdx.x;
    const int d = mb * 32 + (l & 31), k0 = ks * 16 + (l >> 5) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        float v = 0.0f;
        if (d < D && k < 2 * D) {
            if constexpr (sizeof(TW) == 2) v = bf16_to_f32(weight[(size_t)d * 2 * D + k]); else v = weight[(size_t)d * 2 * D + k];
        }
        frag[((size_t)blockIdx.x * 64 + l) * 8 + j] = f32_to_bf16(v);
    }
}

typedef __bf16 cg_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float cg_f32x16_t __attribute__((ext_vector_type(16)));

template <int MB>
__global__ __launch_bounds__(256) void cost_gate_mfma_kernel(int D, long long HW, const bf16_t *__restrict__ stereo,
                                                             const bf16_t *__restrict__ mono,
                                                             const uint4 *__restrict__ wfrag, bf16_t *__restrict__ out)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const long long p = ((long long)blockIdx.x * 4 + wave) * 32 + l32;
    const long long pc = min(p, HW - 1);  // lanes past the image read the last pixel and store nothing
    const bf16_t *sb = stereo + (size_t)b * D * HW + pc, *mb_ = mono + (size_t)b * D * HW + pc;
    const int nks = (2 * D + 15) / 16;
    cg_f32x16_t acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
    for (int ks = 0; ks < nks; ++ks) {
        // this lane's 8 consecutive input planes of its pixel: k = ks * 16 + half * 8 + j
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + half * 8 + j;
            const int kc = min(k, 2 * D - 1);
            const bf16_t raw = kc < D ? sb[(size_t)kc * HW] : mb_[(size_t)(kc - D) * HW];
            v[j] = k < 2 * D ? (uint32_t)raw : 0u;
        }
        const uint32_t pk[4] = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
        cg_bf16x8_t xf;
        __builtin_memcpy(&xf, pk, 16);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint4 q = wfrag[((size_t)m * nks + ks) * 64 + lane];
            cg_bf16x8_t wf;
            __builtin_memcpy(&wf, &q, 16);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[m], 0, 0, 0);
        }
    }
    if (p >= HW) return;
    bf16_t *ob = out + (size_t)b * D * HW + p;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int d = m * 32 + 8 * (i >> 2) + 4 * half + (i & 3);   // the 32x32 accumulator's row of element i
            if (d >= D) continue;
            const float g = 1.0f / (1.0f + __expf(-acc[m][i]));
            const float s = bf16_to_f32(sb[(size_t)d * HW]), mm = bf16_to_f32(mb_[(size_t)d * HW]);
            ob[(size_t)d * HW] = f32_to_bf16(g * s + (1.0f - g) * mm);
        }
}

}  // namespace

extern "C" DFM_API size_t dfm_cost_gate_mfma_weight_bytes(int32_t num_depths)
{
    if (num_depths <= 0 || num_depths > CG_MAX_D) return 0;
    return (size_t)((num_depths + 31) / 32) * ((2 * num_depths + 15) / 16) * 64 * 16;
}

extern "C" DFM_API int dfm_cost_gate_mfma_pack_weights(const void *weight, int32_t weight_dtype, int32_t num_depths,
                                                       void *packed, void *stream)
{
    if (!weight || !packed) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    if (num_depths <= 0 || num_depths > CG_MAX_D) return set_error(DFM_ERR_UNSUPPORTED, "1 .. 96 depth planes");
    if ((uintptr_t)packed & 15) return set_error(DFM_ERR_INVALID_ARG, "packed weights must be 16-byte aligned");
    const dim3 grid((unsigned)(((num_depths + 31) / 32) * ((2 * num_depths + 15) / 16)));
    if (weight_dtype == DFM_BF16)
        hipLaunchKernelGGL(cost_gate_mfma_pack_kernel<bf16_t>, grid, dim3(64), 0, (hipStream_t)stream, (int)num_depths,
                           (const bf16_t *)weight, (bf16_t *)packed);
    else
        hipLaunchKernelGGL(cost_gate_mfma_pack_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, (int)num_depths,
                           (const float *)weight, (bf16_t *)packed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_cost_gate_mfma_fwd(int32_t batch, int32_t num_depths, int64_t hw, const void *stereo,
                                              const void *mono, const void *packed_weights, void *out, void *stream)
{
    if (batch <= 0 || num_depths <= 0 || hw <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    if (!stereo || !mono || !packed_weights || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (num_depths > CG_MAX_D) return set_error(DFM_ERR_UNSUPPORTED, "more than 96 depth planes");
    if ((uintptr_t)packed_weights & 15) return set_error(DFM_ERR_INVALID_ARG, "packed weights must be 16-byte aligned");
    if (batch > 65535 || (hw + 127) / 128 > 0x7fffffffll) return set_error(DFM_ERR_UNSUPPORTED, "grid too large");
    const dim3 grid((unsigned)((hw + 127) / 128), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    const int mb = (num_depths + 31) / 32;
#define CGM_LAUNCH(MB_)                                                                                      \
    hipLaunchKernelGGL((cost_gate_mfma_kernel<MB_>), grid, dim3(256), 0, st, (int)num_depths, (long long)hw, \
                       (const bf16_t *)stereo, (const bf16_t *)mono, (const uint4 *)packed_weights, (bf16_t *)out)
    if (mb == 1) CGM_LAUNCH(1); else if (mb == 2) CGM_LAUNCH(2); else CGM_LAUNCH(3);
#undef CGM_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

static int cg_pitch(int d) { return 4 * (((d + 3) / 4 + 3) / 4) * 4; }

extern "C" DFM_API size_t dfm_cost_gate_weight_bytes(int32_t num_depths)
{
    if (num_depths <= 0 || num_depths > CG_MAX_D) return 0;
    return (size_t)2 * num_depths * cg_pitch(num_depths) * sizeof(float);
}

extern "C" DFM_API int dfm_cost_gate_pack_weights(const void *weight, int32_t weight_dtype, int32_t num_depths,
                                                  void *packed, void *stream)
{
    if (!weight || !packed) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    if (num_depths <= 0 || num_depths > CG_MAX_D) return set_error(DFM_ERR_UNSUPPORTED, "1 .. 96 depth planes");
    if ((uintptr_t)packed & 15) return set_error(DFM_ERR_INVALID_ARG, "packed weights must be 16-byte aligned");
    const int n = 2 * num_depths * cg_pitch(num_depths);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (weight_dtype == DFM_BF16)
        hipLaunchKernelGGL(cost_gate_pack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (int)num_depths,
                           (const bf16_t *)weight, (float *)packed);
    else
        hipLaunchKernelGGL(cost_gate_pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (int)num_depths,
                           (const float *)weight, (float *)packed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_cost_gate_fwd(int32_t batch, int32_t num_depths, int64_t hw, int32_t dtype,
                                         const void *stereo, const void *mono, const void *packed_weights,
                                         void *out, void *stream)
{
    if (batch <= 0 || num_depths <= 0 || hw <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
    if (!stereo || !mono || !packed_weights || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (dtype != DFM_F32 && dtype != DFM_BF16) return set_error(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16");
    if (num_depths > CG_MAX_D) return set_error(DFM_ERR_UNSUPPORTED, "more than 96 depth planes");
    if ((uintptr_t)packed_weights & 15) return set_error(DFM_ERR_INVALID_ARG, "packed weights must be 16-byte aligned");
    if (batch > 65535 || (hw + 63) / 64 > 0x7fffffffll) return set_error(DFM_ERR_UNSUPPORTED, "grid too large");
    const int lds = (int)dfm_cost_gate_weight_bytes(num_depths);
    const dim3 grid((unsigned)((hw + 63) / 64), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
#define CG_LAUNCH(T)                                                                                         \
    do {                                                                                                     \
        const int rc_ = ensure_dynamic_lds((const void *)cost_gate_kernel<T>, lds);                          \
        if (rc_ != DFM_OK) return rc_;                                                                       \
        hipLaunchKernelGGL((cost_gate_kernel<T>), grid, dim3(256), lds, st, (int)num_depths, (long long)hw,  \
                           (const T *)stereo, (const T *)mono, (const float4 *)packed_weights, (T *)out);    \
    } while (0)
    if (dtype == DFM_BF16) CG_LAUNCH(bf16_t); else CG_LAUNCH(float);
#undef CG_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

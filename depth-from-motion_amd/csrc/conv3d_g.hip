// conv3d_g.hip -- general hand-written MFMA Conv3d / ConvTranspose3d 3x3x3 for the 3-D aggregation
// stacks of the path, NDHWC bf16 activations, fp32 accumulation:
//   hourglass conv1..conv6 (stride-2, 64-channel, transposed)  mmdet3d/models/utils/conv_modules.py:73-149
//   ResModule / OutdoorImVoxelNeck Conv3d+BN3d+ReLU stacks      mmdet3d/models/necks/imvoxel_neck.py:26-55,85-117
//   DfMNeck mono / stereo stacks                                mmdet3d/models/necks/dfm_neck.py:29-95
// (the 32 -> 32 full-resolution convolutions keep their register-resident-weight kernel, conv3d.hip).
//
// Per axis the operation is either a correlation (kernel 3, stride 1 or 2, padding 0..2) or the
// x2 up-sampling transposed convolution (kernel 3, stride 2, padding 1, output_padding 1).  The
// transposed axes are evaluated by OUTPUT PARITY CLASS: an even output touches one tap, an odd
// output two, so a workgroup of class (cd, ch, cw) runs 1..8 dense taps on the low-resolution
// input -- no zero insertion, exactly the useful FLOPs.  Backward-data of every variant is another
// launch of the same kernel (stride-1: mirrored taps; stride-2 <-> transposed), only the weight
// packing differs (dfm_conv3d_g_pack_weights: swap / flip).
//
// Tiling (one workgroup = 256 lanes = 4 waves, 2 workgroups per CU):
//   * a workgroup owns a TD x TH x TW block of 128*PFW output positions and 32*CW output channels;
//     per 32-channel input chunk it stages the input halo block it touches ONCE in LDS (LDS-DMA,
//     16-byte pieces, XOR-swizzled so the 64-byte pixel stride is conflict-free for ds_read_b128)
//     and all taps read their B fragments from it: the input crosses L2 -> LDS 1.5-2.5x, not 27x;
//   * MFMA orientation D[cout][pixel] = W[cout][k] * X[k][pixel] (v_mfma_f32_32x32x16_bf16): a wave
//     owns PFW pixel fragments x CW channel fragments (16*PFW*CW accumulator registers); weight
//     fragments come pre-packed from global memory (L2-resident, one 16-byte load per lane and
//     fragment, prefetched one tap ahead) and are reused by the wave's PFW pixel fragments, each
//     activation fragment by its CW channel fragments -> 1/CW KiB of LDS and 1/PFW KiB of L2 per MFMA;
//   * epilogue in registers: per-channel scale/shift (folded BatchNorm or bias), residual add, ReLU,
//     8-byte bf16 stores of 4 consecutive channels.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "dfm_common.h"

using namespace dfm;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

constexpr int G_MAXR = 24;  // LDS-DMA rounds of 256 x 16 B: block <= 1536 pixels = 96 KiB
constexpr int G_MAX_BLOCK_PX = G_MAXR * 64;

struct GAxis {
    int32_t in, out;   // extent of the input / output along this axis
    int32_t tile;      // tile extent in "tile space" (output positions; input positions when up)
    int32_t block;     // staged input extent
    int32_t stride;    // 1 | 2 (correlation axes)
    int32_t pad;       // 0..2 (correlation axes)
    int32_t up;        // 1: x2 transposed-convolution axis
    int32_t tiles;
    int32_t k1;        // 1: the kernel has extent 1 along this axis (2-D convolutions run as depth-1
                       // volumes: kernel (1, 3, 3)); the only tap is the packed weights' centre index
};

struct GGeom {
    GAxis d, h, w;
    int32_t cin, cout, nchunk, cout_tiles, relu, nrounds, block_px;
    int32_t resident, classes;  // resident != 0: all chunks in LDS, the workgroup loops the parity classes
    int32_t ablate;      // perf experiments only (DFM_DEBUG_HOOKS builds, env DFM_CONV_ABLATE; bit 3: no output stores,
                         // bit 4: lane-contiguous output stores): bit 0 stage only
                         // the first chunk, bit 1 load weights only for the first tap, bit 2 conflict-free
                         // (wrong) LDS read addresses
    int32_t cin_stride;  // elements between consecutive input pixels (>= cin: a channel slice of a wider tensor)
    float r_bw, r_bhw;  // 1 / block.w, 1 / (block.h * block.w)
};

// generic weight packing: A-operand fragments [cout_tile][chunk][tap][ks][cw][lane][8]:
// lane l holds A[row = cw*32 + (l & 31)][k = chunk*32 + ks*16 + (l >> 5)*8 + j] of tap t, with
//   A[row][k] = swap ? W[k][row][t'] : W[row][k][t'],   t' = t with the kernel index mirrored (k -> 2 - k)
//   on the axes whose bit is set in `flip` (bit 2 = d, bit 1 = h, bit 0 = w)
// (W: the torch weight tensor, dim0 x dim1 x 27)
template <typename TW>
__device__ __forceinline__ float g_wload(const TW *w, size_t idx);
template <>
__device__ __forceinline__ float g_wload<float>(const float *w, size_t idx) { return w[idx]; }
template <>
__device__ __forceinline__ float g_wload<bf16_t>(const bf16_t *w, size_t idx) { return bf16_to_f32(w[idx]); }

// w2d != 0: W is a 2-D weight (dim0 x dim1 x 9) that sits in the CENTRE depth slice of the 27 taps (a 2-D convolution
// run as a depth-1 volume, kernel extent 1 along depth); the other two slices are packed as zeros
template <typename TW>
__global__ void conv3d_g_pack_kernel(const TW *__restrict__ w, int rows, int kk, int cw_n, int swap,
                                     int flip, bf16_t *__restrict__ frag, int w2d)
{
    // blockIdx.x = ((ct*nchunk + chunk)*27 + tap)*2 + ks, blockIdx.y = cw
    const int nchunk = kk / 32;
    int f = blockIdx.x;
    const int ks = f & 1; f >>= 1;
    const int tap = f % 27; f /= 27;
    const int chunk = f % nchunk, ct = f / nchunk;
    const int cw = blockIdx.y, l = threadIdx.x;
    const int row = (ct * cw_n + cw) * 32 + (l & 31);
    const int k0 = chunk * 32 + ks * 16 + (l >> 5) * 8;
    int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    if (flip & 4) kd = 2 - kd;
    if (flip & 2) kh = 2 - kh;
    if (flip & 1) kw = 2 - kw;
    const int t = (kd * 3 + kh) * 3 + kw;
    const size_t o = (((size_t)blockIdx.x * cw_n + cw) * 64 + l) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (w2d) {
            const size_t idx2 = (swap ? (size_t)(k0 + j) * rows + row : (size_t)row * kk + k0 + j) * 9 + (kh * 3 + kw);
            frag[o + j] = kd == 1 ? f32_to_bf16(g_wload<TW>(w, idx2)) : f32_to_bf16(0.0f);
            continue;
        }
        const size_t idx = swap ? ((size_t)(k0 + j) * rows + row) * 27 + t
                                : ((size_t)row * kk + k0 + j) * 27 + t;
        frag[o + j] = f32_to_bf16(g_wload<TW>(w, idx));
    }
    // the zero page behind the fragments, by the first workgroup (round 6: no hipMemsetAsync launch per pack)
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        uint4 *z = (uint4 *)(frag + (size_t)rows * kk * 27);
        for (int i = l; i < 256; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// F32 == true: the split-precision mode of fp32 models (dfm_conv3d_g_fwd_f32): the fp32 accumulators are
// stored as they are -- plus `residual`, then a float buffer of the output's shape -- so that the terms
// x_hi*w_hi + x_lo*w_hi + x_hi*w_lo of one convolution add up in fp32 over three launches.
// (the F32 build's `residual` -- its fp32 partial sum -- MAY be the buffer it writes, dfm_conv3d_g_fwd_f32's
// acc_in == out: those two parameters are not __restrict__ there)
typedef const bf16_t *__restrict__ g_res_restrict_t;
typedef bf16_t *__restrict__ g_out_restrict_t;
// FAST (round 5): the body for plain correlations whose w axis has the full 3-tap kernel (every convolution of the
// voxel necks and the stride-1 / stride-2 hourglass convolutions; not the transposed ones).  Same tiling, staging
// and epilogue; the tap loop is rebuilt around what the counters of round 2 said the generic loop spends its
// issue slots on (5.2 VALU + 3.4 SALU per MFMA: fragment addresses re-derived per tap, tap bookkeeping, 64-bit
// weight pointers, the copy of the next tap's weights into place):
//   * the LDS swizzle is a function of the pixel's block COLUMN, so a tap row (jd, jh) moves a fragment's
//     address by a workgroup-uniform constant: 3 addresses per pixel fragment are computed once, a read is
//     one v_add (k-step 0) or one v_xor (k-step 1);
//   * the taps are walked as rows of 3 (6 k-steps, unrolled): row offsets are a handful of scalar operations
//     per 6 * PFW * CW MFMAs, no per-tap counters or selects;
//   * weight fragments rotate through THREE register buffers with the row's period (6 k-steps = 2 x 3), loaded
//     two k-steps ahead through a uniform base + lane offset: no copies, no per-lane pointer arithmetic (round 6:
//     SIX buffers, five k-steps ahead, for waves of one pixel fragment, whose k-steps are too short to cover L2);
//   * all rows but the last run in a do-while (at least two rows), the last row is peeled and prefetches
//     nothing: no branch joins while an LDS read is in flight except the loop header's own back edge
//     (tools/verify_async_asm.py checks the build).
template <int CW, int PFW, bool F32 = false, bool FAST = false>
__global__ __launch_bounds__(256, 2) void conv3d_g_kernel(
    GGeom g, const bf16_t *__restrict__ x, const uint4 *__restrict__ wfrag,
    const float *__restrict__ scale, const float *__restrict__ shift,
    std::conditional_t<F32, const bf16_t *, g_res_restrict_t> residual,
    std::conditional_t<F32, bf16_t *, g_out_restrict_t> out,
    const uint4 *__restrict__ zero_page)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l32 = lane & 31, half = lane >> 5;

    // ---- which tile / channel tile / parity class -------------------------------------------
    int t = blockIdx.x;
    const int tiw = t % g.w.tiles; t /= g.w.tiles;
    const int tih = t % g.h.tiles;
    const int tid_ = t / g.h.tiles;
    const int ct = blockIdx.y % g.cout_tiles;
    const int n = blockIdx.z;
    // resident mode (transposed convolutions): every input chunk of the block stays in LDS and the
    // workgroup walks ALL parity classes over it -- the block is staged once, not once per class
    const int cls0 = g.resident ? 0 : blockIdx.y / g.cout_tiles;
    const int cls1 = g.resident ? g.classes : cls0 + 1;

    const int od0 = tid_ * g.d.tile, oh0 = tih * g.h.tile, ow0 = tiw * g.w.tile;  // tile space
    const int bd0 = g.d.up ? od0 : od0 * g.d.stride - g.d.pad;                     // block origin (input)
    const int bh0 = g.h.up ? oh0 : oh0 * g.h.stride - g.h.pad;
    const int bw0 = g.w.up ? ow0 : ow0 * g.w.stride - g.w.pad;
    const int BH = g.h.block, BW = g.w.block;
    const int pix_bytes = g.cin_stride * 2;
    const int BHW = BH * BW;
    const unsigned char *xs = (const unsigned char *)(x + (size_t)n * g.d.in * g.h.in * g.w.in * g.cin_stride);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)blk;
    const int chunk_bytes = g.nrounds * 4096;

    // stage the halo block of one 32-channel chunk into LDS buffer `buf`: piece q = 16 bytes, LDS
    // position q * 16 holds slot (q & 3) ^ swizzle of block pixel q >> 2 (zero page outside the volume);
    // swizzle(pixel) = (block column >> 2) & 3
    auto stage = [&](int chunk, int buf) {
        const unsigned char *src = xs + chunk * 64;
        unsigned char *dst = blk + buf * chunk_bytes + wave * 1024;
        for (int r = 0; r < g.nrounds; ++r) {
            const int q = r * 256 + tid;
            const int p = q >> 2;
            const int bd = (int)(((float)p + 0.5f) * g.r_bhw);
            const int rem = p - bd * BHW;
            const int bh = (int)(((float)rem + 0.5f) * g.r_bw);
            const int bw = rem - bh * BW;
            const int sl = (q & 3) ^ ((bw >> 2) & 3);  // swizzle by block column: tap rows shift addresses uniformly
            const int d = bd0 + bd, h = bh0 + bh, w = bw0 + bw;
            const bool ok = p < g.block_px && (unsigned)d < (unsigned)g.d.in && (unsigned)h < (unsigned)g.h.in &&
                            (unsigned)w < (unsigned)g.w.in;
            const unsigned char *s_ = ok ? src + (((d * g.h.in + h) * g.w.in + w) * pix_bytes + sl * 16)
                                         : (const unsigned char *)zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)s_,
                                             (__attribute__((address_space(3))) void *)(dst + r * 4096),
                                             16, 0, 0);
        }
    };
    if (g.resident) {
        for (int chunk = 0; chunk < g.nchunk; ++chunk) stage(chunk, chunk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- this lane's pixels: PFW fragments of 32 consecutive tile positions per wave -----------
    int base_bp[PFW], tpos[PFW];  // block pixel of tap (0,0,0); packed tile coordinates
    int bwb[PFW];                 // ... and its block column (the swizzle's argument)
    {
        const int sd = g.d.up ? 1 : g.d.stride, sh = g.h.up ? 1 : g.h.stride, sw = g.w.up ? 1 : g.w.stride;
#pragma unroll
        for (int f = 0; f < PFW; ++f) {
            const int i = (wave * PFW + f) * 32 + l32;
            const int tw = i % g.w.tile, i2 = i / g.w.tile;
            const int th = i2 % g.h.tile, td = i2 / g.h.tile;
            base_bp[f] = (td * sd * BH + th * sh) * BW + tw * sw;
            bwb[f] = tw * sw;
            tpos[f] = (td << 20) | (th << 10) | tw;  // tile extents <= 512
        }
    }
    const size_t osample = (size_t)n * g.d.out * g.h.out * g.w.out;

    for (int cls = cls0; cls < cls1; ++cls) {
        const int ncw = g.w.up ? 2 : 1, nch = g.h.up ? 2 : 1;
        const int pcw = cls % ncw, pch = (cls / ncw) % nch, pcd = cls / (ncw * nch);
        int opix[PFW];  // output voxel index inside the sample, -1: outside the volume
#pragma unroll
        for (int f = 0; f < PFW; ++f) {
            const int td = tpos[f] >> 20, th = (tpos[f] >> 10) & 1023, tw = tpos[f] & 1023;
            const int od = g.d.up ? 2 * (od0 + td) + pcd : od0 + td;
            const int oh = g.h.up ? 2 * (oh0 + th) + pch : oh0 + th;
            const int ow = g.w.up ? 2 * (ow0 + tw) + pcw : ow0 + tw;
            const bool ok = od < g.d.out && oh < g.h.out && ow < g.w.out;
            opix[f] = ok ? (od * g.h.out + oh) * g.w.out + ow : -1;
        }

        // taps of this class: counters (jd, jh, jw) -> (weight tap index, block offset in pixels)
        const int ntw = g.w.up ? 1 + pcw : (g.w.k1 ? 1 : 3), nth = g.h.up ? 1 + pch : (g.h.k1 ? 1 : 3),
                  ntd = g.d.up ? 1 + pcd : (g.d.k1 ? 1 : 3);
        const int ntaps = ntd * nth * ntw;
        int jd = 0, jh = 0, jw = 0;
        int jwc = 0;  // the current tap's column shift inside the block (its jw counter)
        auto tap_cur = [&](int &wt, int &off) {
            jwc = jw;
            // correlation axis: k = j at offset j; up axis: even class -> k 1 @ +0; odd -> k 2 @ +0, k 0 @ +1
            const int kw = g.w.up ? (pcw ? (jw ? 0 : 2) : 1) : (g.w.k1 ? 1 : jw);
            const int kh = g.h.up ? (pch ? (jh ? 0 : 2) : 1) : (g.h.k1 ? 1 : jh);
            const int kd = g.d.up ? (pcd ? (jd ? 0 : 2) : 1) : (g.d.k1 ? 1 : jd);
            wt = (kd * 3 + kh) * 3 + kw;
            off = (jd * BH + jh) * BW + jw;
        };
        auto tap_adv = [&]() {
            if (++jw == ntw) {
                jw = 0;
                if (++jh == nth) { jh = 0; ++jd; }
            }
        };

        f32x16_t acc[PFW][CW];
#pragma unroll
        for (int f = 0; f < PFW; ++f)
#pragma unroll
            for (int c = 0; c < CW; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[f][c][i] = 0.0f;

#ifdef DFM_DEBUG_HOOKS
        if (g.ablate & 64) goto taps_done;  // what everything around the tap loops costs
#endif
        if constexpr (FAST) {
            // ---- the row-walking tap loop (see the template's comment) ----------------------------------
            uint32_t A[PFW][3];  // LDS address of the fragment's pixel at column shift jw, row (0, 0), k-step 0
#pragma unroll
            for (int f = 0; f < PFW; ++f)
#pragma unroll
                for (int j3 = 0; j3 < 3; ++j3)
                    A[f][j3] = lds0 + (uint32_t)(base_bp[f] + j3) * 64u +
                               ((((uint32_t)half) ^ (((uint32_t)(bwb[f] + j3) >> 2) & 3u)) << 4);
#ifdef DFM_DEBUG_HOOKS
            if (g.ablate & 4) {  // conflict-free (wrong) read addresses: what the LDS bank conflicts cost
#pragma unroll
                for (int f = 0; f < PFW; ++f)
#pragma unroll
                    for (int j3 = 0; j3 < 3; ++j3) A[f][j3] = lds0 + (uint32_t)lane * 16u + (uint32_t)(f * 3 + j3) * 64u;
            }
#endif
            const int nth_f = g.h.k1 ? 1 : 3;
            const int nrows = (g.d.k1 ? 1 : 3) * nth_f;  // 9 (3-D) or 3 (a 2-D convolution as a depth-1 volume)
            const uint32_t lane16 = (uint32_t)lane * 16u;
            constexpr int KSB = CW * 1024;  // bytes of one (tap, k-step): CW fragments of 64 lanes x 16 B
            // LDS byte offset / weight byte offset of tap row r = (jd, jh)
            auto row_lds = [&](int r) -> uint32_t {
                const int rd = r / nth_f, rh = r - rd * nth_f;
                return (uint32_t)((rd * BH + rh) * BW) * 64u;
            };
            auto row_wt = [&](int r) -> uint32_t {
                const int rd = r / nth_f, rh = r - rd * nth_f;
                const int kd = g.d.k1 ? 1 : rd, kh = g.h.k1 ? 1 : rh;
                return (uint32_t)((kd * 3 + kh) * 3) * 2u * KSB;
            };
            for (int chunk = 0; chunk < g.nchunk; ++chunk) {
                if (chunk > 0) __syncthreads();  // every wave is done reading the previous chunk's block
#ifdef DFM_DEBUG_HOOKS
                if (!(g.ablate & 1) || chunk == 0)
#endif
                stage(chunk, 0);
                const char *wchunk = (const char *)(wfrag + (size_t)(ct * g.nchunk + chunk) * 27 * 2 * CW * 64);
                // weight fragments travel WD k-steps ahead of their MFMAs through a ring of WR register buffers (WR divides
                // the row's 6 k-steps: the ring's phase is the same in every row).  A k-step of a one-fragment wave is
                // CW MFMAs = 64 clocks: two steps ahead (round 5) is 128 clocks against an L2 round trip of ~700 -- the
                // small hourglass layers (250 workgroups, PFW = 1) stalled on every step (profiles/r06_c8_*: conv1 102 ->
                // 72 us with the weight loads ablated).  Round 6: five steps ahead for PFW = 1 (48 registers at CW = 2).
#ifdef DFM_WRING3   // (A/B builds: the round-5 ring)
                constexpr int WR = 3, WD = WR - 1;
#else
#ifndef DFM_WRING6_MAXPFW
#define DFM_WRING6_MAXPFW 3
#endif
                constexpr int WR = PFW <= DFM_WRING6_MAXPFW ? 6 : 3, WD = WR - 1;
#endif
                bf16x8_t w[WR][CW];  // weight fragments of k-steps s .. s + WD (mod WR)
                u32x4_t q[2][PFW];   // activation fragments of k-steps s, s + 1 (mod 2)
                uint32_t addr[PFW];  // the k-step-0 addresses of the tap being read (k-step 1 = ^ 32)
                auto wld = [&](const char *wrow, int t, bf16x8_t (&dst)[CW]) {  // k-step t of a row: tap t >> 1
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const uint4 v = *(const uint4 *)(wrow + (t * CW + c) * 1024 + lane16);
                        __builtin_memcpy(&dst[c], &v, 16);
                    }
                };
                auto rd0 = [&](int j3, uint32_t lrow, u32x4_t (&dst)[PFW]) {
#pragma unroll
                    for (int f = 0; f < PFW; ++f) {
                        addr[f] = A[f][j3] + lrow;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[f]) : "v"(addr[f]));
                    }
                };
                auto rd1 = [&](u32x4_t (&dst)[PFW]) {
#pragma unroll
                    for (int f = 0; f < PFW; ++f) {
                        const uint32_t a = addr[f] ^ 32u;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[f]) : "v"(a));
                    }
                };
#define G_WAIT(Q, N)                                                                                         \
                do {                                                                                         \
                    if constexpr (PFW == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]));           \
                    if constexpr (PFW == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]), "+v"(Q[1])); \
                    if constexpr (PFW == 3) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2])); \
                    if constexpr (PFW == 4) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3])); \
                } while (0)
#define G_WAIT_PFW(Q)                                                                                        \
                do {                                                                                         \
                    if constexpr (PFW == 1) G_WAIT(Q, 1);                                                    \
                    if constexpr (PFW == 2) G_WAIT(Q, 2);                                                    \
                    if constexpr (PFW == 3) G_WAIT(Q, 3);                                                    \
                    if constexpr (PFW == 4) G_WAIT(Q, 4);                                                    \
                } while (0)
                // one tap row: 6 k-steps.  Before step s runs, the reads of step s + 1 and the weights of
                // step s + 2 are requested (the next row's first steps at the end of a row, unless LAST)
                auto row = [&](auto last_c, uint32_t lcur, uint32_t lnext, const char *wcur, const char *wnext) {
                    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                    for (int st = 0; st < 6; ++st) {
                        asm volatile("" ::: "memory");  // keep the weight loads where they are written
                        bool reads_ahead = true;
                        if (st + 1 < 6) {
                            if ((st + 1) & 1) rd1(q[(st + 1) & 1]);
                            else rd0((st + 1) >> 1, lcur, q[(st + 1) & 1]);
                        } else if constexpr (!LAST) {
                            rd0(0, lnext, q[0]);
                        } else {
                            reads_ahead = false;
                        }
#ifdef DFM_DEBUG_HOOKS
                        if (!(g.ablate & 2)) {
#endif
                        if (st + WD < 6) wld(wcur, st + WD, w[(st + WD) % WR]);
                        else if constexpr (!LAST) wld(wnext, st + WD - 6, w[(st + WD) % WR]);
#ifdef DFM_DEBUG_HOOKS
                        }
#endif
                        if (reads_ahead) G_WAIT_PFW(q[st & 1]);
                        else G_WAIT(q[st & 1], 0);
#pragma unroll
                        for (int f = 0; f < PFW; ++f) {
                            bf16x8_t xf;
                            __builtin_memcpy(&xf, &q[st & 1][f], 16);
#pragma unroll
                            for (int c = 0; c < CW; ++c)
                                acc[f][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[st % WR][c], xf, acc[f][c], 0, 0, 0);
                        }
                    }
                };
                // weights of the first WD k-steps travel while the block lands
                uint32_t lcur = row_lds(0);
                const char *wcur = wchunk + row_wt(0);
#pragma unroll
                for (int t = 0; t < WD; ++t) wld(wcur, t, w[t]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // the block has landed
                rd0(0, lcur, q[0]);
                int r = 0;
                do {  // rows 0 .. nrows - 2 (nrows >= 3)
                    const uint32_t lnext = row_lds(r + 1);
                    const char *wnext = wchunk + row_wt(r + 1);
                    row(std::false_type{}, lcur, lnext, wcur, wnext);
                    lcur = lnext;
                    wcur = wnext;
                } while (++r < nrows - 1);
                row(std::true_type{}, lcur, 0u, wcur, wcur);
#undef G_WAIT
#undef G_WAIT_PFW
            }
        } else {
        for (int chunk = 0; chunk < g.nchunk; ++chunk) {
            const uint32_t ldsb = lds0 + (g.resident ? chunk * chunk_bytes : 0);
            // one round of ds_read_b128: the PFW B fragments of a tap.  k-step 0 computes the swizzled
            // addresses; k-step 1 is the same pixel's slot ^ 2, i.e. address ^ 32 (the dynamic LDS
            // base and the chunk stride are multiples of 64: no static __shared__ in this kernel)
            uint32_t addr[PFW];
            auto issue0 = [&](int off, u32x4_t (&dst)[PFW]) {
#pragma unroll
                for (int f = 0; f < PFW; ++f) {
                    const int bp = base_bp[f] + off;
                    uint32_t a = ldsb + (uint32_t)bp * 64u +
                                 ((((uint32_t)half) ^ (((uint32_t)(bwb[f] + jwc) >> 2) & 3u)) << 4);
#ifdef DFM_DEBUG_HOOKS
                    if (g.ablate & 4) a = ldsb + ((uint32_t)(off & 15) << 10) + lane * 16;
#endif
                    addr[f] = a;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[f]) : "v"(a));
                }
            };
            auto issue1 = [&](u32x4_t (&dst)[PFW]) {
#pragma unroll
                for (int f = 0; f < PFW; ++f) {
                    const uint32_t a = addr[f] ^ 32u;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[f]) : "v"(a));
                }
            };
            if (!g.resident) {
                if (chunk > 0) __syncthreads();  // every wave is done reading the previous chunk's block
#ifdef DFM_DEBUG_HOOKS
                if (!(g.ablate & 1) || chunk == 0)
#endif
                stage(chunk, 0);
            }
            // weights of the first tap travel while the block lands
            const uint4 *wp = wfrag + ((size_t)(ct * g.nchunk + chunk) * 27 * 2 * CW) * 64 + lane;
            auto wload = [&](int wt, bf16x8_t (&dst)[2][CW]) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const uint4 q = wp[((wt * 2 + ks) * CW + c) * 64];
                        __builtin_memcpy(&dst[ks][c], &q, 16);
                    }
            };
            bf16x8_t wa[2][CW], wb[2][CW];  // weights of the current / next tap, ping-pong (no copies)
            int wt, off, offn;
            jd = jh = jw = 0;
            tap_cur(wt, off);
            wload(wt, wa);
            if (!g.resident) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // the block has landed
            }

            u32x4_t q0[PFW], q1[PFW];
            issue0(off, q0);
            // q (issued one round earlier) is complete when at most PFW newer reads are outstanding
#define G_WAIT(Q, N)                                                                                         \
            do {                                                                                             \
                if constexpr (PFW == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]));               \
                if constexpr (PFW == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]), "+v"(Q[1]));  \
                if constexpr (PFW == 3) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2])); \
                if constexpr (PFW == 4) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3])); \
            } while (0)
#define G_WAIT_PFW(Q)                                                                                        \
            do {                                                                                             \
                if constexpr (PFW == 1) G_WAIT(Q, 1);                                                        \
                if constexpr (PFW == 2) G_WAIT(Q, 2);                                                        \
                if constexpr (PFW == 3) G_WAIT(Q, 3);                                                        \
                if constexpr (PFW == 4) G_WAIT(Q, 4);                                                        \
            } while (0)
            auto mfmas = [&](bf16x8_t (&wk)[CW], u32x4_t (&q)[PFW]) {
#pragma unroll
                for (int f = 0; f < PFW; ++f) {
                    bf16x8_t xf;
                    __builtin_memcpy(&xf, &q[f], 16);
#pragma unroll
                    for (int c = 0; c < CW; ++c)
                        acc[f][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[c], xf, acc[f][c], 0, 0, 0);
                }
            };
            // one tap: request the next tap's weights into `wnext`, run this tap's two k-steps from `wcur`
            auto step = [&](bf16x8_t (&wcur)[2][CW], bf16x8_t (&wnext)[2][CW]) {
                asm volatile("" ::: "memory");  // keep the next step's weight loads from being hoisted here
                tap_adv();
                tap_cur(wt, offn);
#ifdef DFM_DEBUG_HOOKS
                if (!(g.ablate & 2))
#endif
                wload(wt, wnext);
                issue1(q1);
                G_WAIT_PFW(q0);
                mfmas(wcur[0], q0);
                issue0(offn, q0);
                G_WAIT_PFW(q1);
                mfmas(wcur[1], q1);
            };
            auto last = [&](bf16x8_t (&wcur)[2][CW]) {
                issue1(q1);
                G_WAIT_PFW(q0);
                mfmas(wcur[0], q0);
                G_WAIT(q1, 0);
                mfmas(wcur[1], q1);
            };
            for (int j = 0; j + 1 < ntaps; ++j) {
                step(wa, wb);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int cw = 0; cw < CW; ++cw) wa[ks][cw] = wb[ks][cw];
            }
            last(wa);
#undef G_WAIT
#undef G_WAIT_PFW
        }

        }
#ifdef DFM_DEBUG_HOOKS
    taps_done:
#endif
        // ---- epilogue: lane = pixel (l32) x 4 groups of 4 consecutive channels per channel fragment ----
        auto epilogue = [&](auto has_scale, auto has_res) {
#pragma unroll
            for (int f = 0; f < PFW; ++f) {
                if (opix[f] < 0) continue;
                const size_t vox = osample + (size_t)opix[f];
                u32x2_t rr[CW][4];
                if constexpr (decltype(has_res)::value) {
#pragma unroll
                    for (int c = 0; c < CW; ++c)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
                            rr[c][gq] = *(const u32x2_t *)(residual + vox * g.cout + (ct * CW + c) * 32 + 8 * gq + 4 * half);
                }
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    dfm_u32x2 pk[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int ch = (ct * CW + c) * 32 + 8 * gq + 4 * half;
                        float v0 = acc[f][c][4 * gq], v1 = acc[f][c][4 * gq + 1], v2 = acc[f][c][4 * gq + 2],
                              v3 = acc[f][c][4 * gq + 3];
                        if constexpr (decltype(has_scale)::value) {
                            const float4 s4 = *(const float4 *)(scale + ch), b4 = *(const float4 *)(shift + ch);
                            v0 = __builtin_fmaf(v0, s4.x, b4.x); v1 = __builtin_fmaf(v1, s4.y, b4.y);
                            v2 = __builtin_fmaf(v2, s4.z, b4.z); v3 = __builtin_fmaf(v3, s4.w, b4.w);
                        }
                        if constexpr (decltype(has_res)::value) {
                            v0 += __uint_as_float(rr[c][gq].x << 16); v1 += __uint_as_float(rr[c][gq].x & 0xffff0000u);
                            v2 += __uint_as_float(rr[c][gq].y << 16); v3 += __uint_as_float(rr[c][gq].y & 0xffff0000u);
                        }
                        if (g.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        pk[gq] = dfm_u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    }
#ifdef DFM_DEBUG_HOOKS
                    if ((g.ablate & 8) && pk[0].x != 0x12345678u) continue;  // what the output stores cost
                    if (g.ablate & 16) {  // ... and what their pattern costs: lane-contiguous 8-byte stores
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
                            *(dfm_u32x2 *)(out + ((size_t)blockIdx.x * 256 + tid) * 4 + (size_t)(c * 4 + gq) * 1024) = pk[gq];
                        continue;
                    }
                    if (g.ablate & 32) {  // the round-5 form: four 8-byte stores per pixel and lane
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
                            *(dfm_u32x2 *)(out + vox * g.cout + (ct * CW + c) * 32 + 8 * gq + 4 * half) = pk[gq];
                        continue;
                    }
#endif
                    // the two halves of the wave trade pieces of their pixel: two 16-byte stores per lane, 32 contiguous
                    // bytes per pixel and instruction (dfm_common.h: acc_rows_to_16B), instead of four 8-byte ones
                    // (both lanes of a pixel are active or skipped together: opix depends on the pixel only)
                    dfm_u32x4 q16[2];
                    acc_rows_to_16B(pk, q16);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
                        *(dfm_u32x4 *)(out + vox * g.cout + (ct * CW + c) * 32 + 16 * pr + 8 * half) = q16[pr];
                }
            }
        };
        if constexpr (F32) {
            const float *acc_in = (const float *)residual;
            float *out32 = (float *)out;
#pragma unroll
            for (int f = 0; f < PFW; ++f) {
                if (opix[f] < 0) continue;
                const size_t vox = osample + (size_t)opix[f];
#pragma unroll
                for (int c = 0; c < CW; ++c)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int ch = (ct * CW + c) * 32 + 8 * gq + 4 * half;
                        float4 v = make_float4(acc[f][c][4 * gq], acc[f][c][4 * gq + 1], acc[f][c][4 * gq + 2],
                                               acc[f][c][4 * gq + 3]);
                        if (acc_in) {
                            const float4 a = *(const float4 *)(acc_in + vox * g.cout + ch);
                            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                        }
                        *(float4 *)(out32 + vox * g.cout + ch) = v;
                    }
            }
        } else if (scale) {
            if (residual) epilogue(std::true_type{}, std::true_type{});
            else epilogue(std::true_type{}, std::false_type{});
        } else {
            if (residual) epilogue(std::false_type{}, std::true_type{});
            else epilogue(std::false_type{}, std::false_type{});
        }
    }
}

// ---- host: tile selection ------------------------------------------------------------------------
struct GPlan {
    GGeom g;
    int pfw, cw, classes;
    size_t lds;
};

void axis_fill(GAxis &a, int tile)
{
    a.tile = tile;
    const int space = a.up ? a.in : a.out;
    a.tiles = (space + tile - 1) / tile;
    a.block = a.up ? tile + 1 : (tile - 1) * a.stride + (a.k1 ? 1 : 3);
}

// picks (PFW, TD, TH, TW): minimum estimated time = rounds of workgroups over the chip x per-workgroup
// cost (MFMA work ~ positions, staging ~ block pixels)
bool g_plan(const dfm_conv3d_desc *d, GPlan &pl)
{
    GGeom g{};
    GAxis *ax[3] = {&g.d, &g.h, &g.w};
    int classes = 1;
    for (int i = 0; i < 3; ++i) {
        ax[i]->in = d->in_size[i]; ax[i]->out = d->out_size[i];
        ax[i]->stride = d->stride[i]; ax[i]->pad = d->padding[i]; ax[i]->up = d->transposed[i] ? 1 : 0;
        ax[i]->k1 = d->kernel1[i] ? 1 : 0;
        if (ax[i]->up) classes *= 2;
    }
    double ntap_all = 1.0;
    for (int i = 0; i < 3; ++i) ntap_all *= d->kernel1[i] ? 1.0 : 3.0;
    g.cin = d->cin; g.cout = d->cout; g.nchunk = d->cin / 32;
#ifdef DFM_DEBUG_HOOKS
    {
        const char *ab = getenv("DFM_CONV_ABLATE");  // perf experiments only
        g.ablate = ab ? atoi(ab) : 0;
    }
#endif
    g.cin_stride = d->in_channel_stride > 0 ? d->in_channel_stride : d->cin;
    const int cw = d->cout % 64 == 0 ? 2 : 1;
    g.cout_tiles = d->cout / (32 * cw);
    g.relu = d->relu ? 1 : 0;
    double best = 1e300;
    bool found = false;
    const int sp[3] = {g.d.up ? g.d.in : g.d.out, g.h.up ? g.h.in : g.h.out, g.w.up ? g.w.in : g.w.out};
    // pass 0 keeps tiles inside (twice) the volume; pass 1 (tiny volumes) takes any factorisation
    // DFM_CONV_G_PLAN="pfw,td,th,tw": perf experiments only -- the planner considers this tiling alone
    int force[4] = {0, 0, 0, 0};
    if (const char *fp = getenv("DFM_CONV_G_PLAN")) {
        if (sscanf(fp, "%d,%d,%d,%d", &force[0], &force[1], &force[2], &force[3]) != 4) force[0] = 0;
    }
    for (int pass = 0; pass < 2 && !found; ++pass)
    for (int pfw = 4; pfw >= 1; --pfw) {
        const int P = 128 * pfw;
        if (force[0] && pfw != force[0]) continue;
        // x2 transposed on all three axes (8 parity classes, the generic tap loop): two pixel fragments a wave at most --
        // measured, hourglass conv6 64 -> 32 at (36, 40, 160): pfw 2 (4, 8, 8) 0.071 ms, pfw 4 (8, 8, 8) 0.079 ms
        // (profiles/r06_c39_conv_g_plan_sweep_hourglass_layers.txt)
        if (!force[0] && classes == 8 && pfw > 2) continue;
        for (int td = 1; td <= P; ++td) {
            if (P % td) continue;
            if (pass == 0 && td > sp[0] && td > 1) continue;
            if (force[0] && td != force[1]) continue;
            for (int th = 1; th <= P / td; ++th) {
                if ((P / td) % th) continue;
                const int tw = P / td / th;
                if (pass == 0 && ((th > 2 * sp[1] && th > 1) || (tw > 2 * sp[2] && tw > 1))) continue;
                if (force[0] && (th != force[2] || tw != force[3])) continue;
                GGeom c = g;
                axis_fill(c.d, td); axis_fill(c.h, th); axis_fill(c.w, tw);
                const long long bpx = (long long)c.d.block * c.h.block * c.w.block;
                if (bpx > G_MAX_BLOCK_PX) continue;
                const int rounds = (int)((bpx * 4 + 255) / 256);
                // transposed convolutions keep every chunk of the block in LDS and walk the parity
                // classes inside the workgroup (staged once instead of once per class) when it fits
                const bool resident = classes > 1 && (size_t)rounds * 4096 * g.nchunk <= 160 * 1024;
                const size_t lds = (size_t)rounds * 4096 * (resident ? g.nchunk : 1);
                const long long wgs = (long long)c.d.tiles * c.h.tiles * c.w.tiles * c.cout_tiles *
                                      (resident ? 1 : classes) * d->n;
                // workgroups that actually share a CU: at most 2 (registers), what the LDS allows,
                // and no more than there are workgroups per CU
                const int wg_per_cu = (int)std::max<long long>(
                    1, std::min<long long>(std::min(2, (int)(160 * 1024 / lds)), (wgs + 255) / 256));
                const long long waves = (wgs + 256 * wg_per_cu - 1) / (256 * wg_per_cu);
                // per-workgroup cost in clocks.  One tap = 2 k-steps: 2*pfw*cw MFMAs of 32 clocks per
                // wave, against 4 waves x 2*cw KiB of weight fragments from L2 (~40 B/clk/CU); resident
                // workgroups share both the MFMA pipes and the L2 port.  Staging ~ 0.1 clk/B.
                const double taps = resident ? ntap_all : ntap_all / classes;
                const double mf_tap = (double)pfw * cw * 64.0 * wg_per_cu;
                const double wt_tap = 8192.0 * cw / 40.0 * wg_per_cu;
                const double mf = std::max(mf_tap, wt_tap) * taps * g.nchunk;
                const double st = ((double)bpx * 64 * 0.1 + 1500.0) * g.nchunk;
                const double cost = (double)waves * (mf + st);
                // (a tie between two tilings of the same block size goes to the one that is deeper along d: measured,
                //  256 -> 256 at (220, 300, 3): (16, 8, 3) 0.648 ms, (8, 16, 3) 0.691 ms -- profiles/r06_c39_*)
                if (cost < best || (found && cost == best && tw > 1 && td > pl.g.d.tile)) {   // (tw = 1: no preference measured)
                    best = cost; found = true;
                    c.block_px = (int)bpx; c.nrounds = rounds;
                    c.r_bw = 1.0f / (float)c.w.block; c.r_bhw = 1.0f / (float)(c.h.block * c.w.block);
                    c.resident = resident ? 1 : 0; c.classes = classes;
                    pl.g = c; pl.pfw = pfw; pl.cw = cw; pl.classes = resident ? 1 : classes; pl.lds = lds;
                }
            }
        }
    }
    return found;
}

int g_check(const dfm_conv3d_desc *d)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "NULL conv descriptor");
    if (d->n <= 0 || d->n > 65535) return set_error(DFM_ERR_INVALID_ARG, "batch must be 1..65535");
    if (d->cin <= 0 || d->cin % 32 || d->cout <= 0 || d->cout % 32)
        return set_error(DFM_ERR_UNSUPPORTED, "channel counts must be multiples of 32");
    long long in_px = 1, out_px = 1;
    for (int i = 0; i < 3; ++i) {
        if (d->in_size[i] <= 0 || d->out_size[i] <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
        if (d->kernel1[i]) {
            if (d->transposed[i] || d->padding[i] != 0 || d->stride[i] < 1 || d->stride[i] > 2 ||
                d->out_size[i] != (d->in_size[i] - 1) / d->stride[i] + 1)
                return set_error(DFM_ERR_INVALID_ARG, "kernel-extent-1 axes: not transposed, padding 0, out = (in - 1) / stride + 1");
        } else if (d->transposed[i]) {
            if (d->out_size[i] != 2 * d->in_size[i])
                return set_error(DFM_ERR_UNSUPPORTED, "transposed axes are kernel 3, stride 2, padding 1, output_padding 1 (out = 2 in)");
        } else {
            if (d->stride[i] < 1 || d->stride[i] > 2 || d->padding[i] < 0 || d->padding[i] > 2)
                return set_error(DFM_ERR_UNSUPPORTED, "stride must be 1 or 2, padding 0..2");
            const int o = (d->in_size[i] + 2 * d->padding[i] - 3) / d->stride[i] + 1;
            if (d->in_size[i] + 2 * d->padding[i] < 3 || o != d->out_size[i])
                return set_error(DFM_ERR_INVALID_ARG, "out_size does not match in_size / stride / padding");
        }
        in_px *= d->in_size[i]; out_px *= d->out_size[i];
    }
    if (d->in_channel_stride != 0 && (d->in_channel_stride < d->cin || d->in_channel_stride % 8))
        return set_error(DFM_ERR_INVALID_ARG, "in_channel_stride must be 0 or >= cin and a multiple of 8");
    if (in_px * std::max(d->cin, d->in_channel_stride) * 2 >= (1ll << 31) || out_px >= (1ll << 31))
        return set_error(DFM_ERR_UNSUPPORTED, "sample too large for 32-bit offsets");
    return DFM_OK;
}

}  // namespace

extern "C" DFM_API size_t dfm_conv3d_g_weight_bytes(int32_t cin, int32_t cout)
{
    if (cin <= 0 || cout <= 0 || cin % 32 || cout % 32) return 0;
    return (size_t)cin * cout * 27 * 2 + 4096;  // fragments + the zero page
}

static int g_pack_impl(const void *weight, int32_t weight_dtype, int32_t cin, int32_t cout, int32_t swap, int32_t flip,
                       void *packed, void *stream, int w2d);

extern "C" DFM_API int dfm_conv3d_g_pack_weights(const void *weight, int32_t weight_dtype, int32_t cin,
                                                 int32_t cout, int32_t swap, int32_t flip, void *packed,
                                                 void *stream)
{
    return g_pack_impl(weight, weight_dtype, cin, cout, swap, flip, packed, stream, 0);
}

// the same for a 2-D weight (dim0, dim1, 3, 3): its taps in the centre depth slice of the 27, zeros in the other two --
// what a 2-D convolution run as a depth-1 volume (dfm_conv3d_desc.kernel1[0] = 1) reads; flip bits 2 (h), 1 (w)
extern "C" DFM_API int dfm_conv3d_g_pack_weights_2d(const void *weight, int32_t weight_dtype, int32_t cin,
                                                    int32_t cout, int32_t swap, int32_t flip, void *packed,
                                                    void *stream)
{
    return g_pack_impl(weight, weight_dtype, cin, cout, swap, flip & 3, packed, stream, 1);
}

static int g_pack_impl(const void *weight, int32_t weight_dtype, int32_t cin, int32_t cout, int32_t swap, int32_t flip,
                       void *packed, void *stream, int w2d)
{
    if (!weight || !packed) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (cin <= 0 || cout <= 0 || cin % 32 || cout % 32)
        return set_error(DFM_ERR_UNSUPPORTED, "channel counts must be multiples of 32");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;   // (the zero page behind the cin * cout * 27 fragments' elements is the pack kernel's)
    const int cw = cout % 64 == 0 ? 2 : 1;
    const int cts = cout / (32 * cw), nchunk = cin / 32;
    dim3 grid(cts * nchunk * 27 * 2, cw);
    if (weight_dtype == DFM_F32)
        hipLaunchKernelGGL(conv3d_g_pack_kernel<float>, grid, dim3(64), 0, st, (const float *)weight, cout, cin,
                           cw, swap, flip, (bf16_t *)packed, w2d);
    else
        hipLaunchKernelGGL(conv3d_g_pack_kernel<bf16_t>, grid, dim3(64), 0, st, (const bf16_t *)weight, cout,
                           cin, cw, swap, flip, (bf16_t *)packed, w2d);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

static int conv3d_g_launch(const dfm_conv3d_desc *desc, const void *x, const void *packed_weights,
                           const float *scale, const float *shift, const void *residual, void *out,
                           void *stream, bool f32);

extern "C" DFM_API int dfm_conv3d_g_fwd(const dfm_conv3d_desc *desc, const void *x, const void *packed_weights,
                                        const float *scale, const float *shift, const void *residual,
                                        void *out, void *stream)
{
    return conv3d_g_launch(desc, x, packed_weights, scale, shift, residual, out, stream, false);
}

extern "C" DFM_API int dfm_conv3d_g_fwd_f32(const dfm_conv3d_desc *desc, const void *x, const void *packed_weights,
                                            const float *acc_in, float *out, void *stream)
{
    if (desc && desc->relu) return set_error(DFM_ERR_INVALID_ARG, "the fp32-accumulating form has no epilogue");
    if ((((uintptr_t)out) | ((uintptr_t)acc_in)) & 15) return set_error(DFM_ERR_INVALID_ARG, "fp32 buffers must be 16-byte aligned");
    return conv3d_g_launch(desc, x, packed_weights, nullptr, nullptr, acc_in, out, stream, true);
}

static int conv3d_g_launch(const dfm_conv3d_desc *desc, const void *x, const void *packed_weights,
                           const float *scale, const float *shift, const void *residual, void *out,
                           void *stream, bool f32)
{
    const int rc = g_check(desc);
    if (rc != DFM_OK) return rc;
    if (!x || !packed_weights || !out) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if ((scale == nullptr) != (shift == nullptr))
        return set_error(DFM_ERR_INVALID_ARG, "scale and shift come together");
    GPlan pl;
    if (!g_plan(desc, pl)) return set_error(DFM_ERR_UNSUPPORTED, "no tiling fits the LDS budget");
    const long long tiles = (long long)pl.g.d.tiles * pl.g.h.tiles * pl.g.w.tiles;
    if (tiles >= (1ll << 31) || pl.g.cout_tiles * pl.classes > 65535)
        return set_error(DFM_ERR_UNSUPPORTED, "grid too large");
    const uint4 *wfrag = (const uint4 *)packed_weights;
    const uint4 *zero = (const uint4 *)((const char *)packed_weights + (size_t)desc->cin * desc->cout * 27 * 2);
    dim3 grid((unsigned)tiles, pl.g.cout_tiles * pl.classes, desc->n);
    hipStream_t st = (hipStream_t)stream;
    const int lds = (int)pl.lds;
    // the row-walking body: plain correlations (no transposed axis) with the 3-tap kernel along w and at least
    // along one of d, h (DFM_CONV_GENERIC=1 pins the generic body: A/B runs)
    static const bool pin_generic = getenv("DFM_CONV_GENERIC") != nullptr;
    const bool fast = !pin_generic && !pl.g.resident && !pl.g.d.up && !pl.g.h.up && !pl.g.w.up && !pl.g.w.k1 &&
                      !(pl.g.d.k1 && pl.g.h.k1);
#define G_LAUNCH1(CW_, PFW_, F32_, FAST_)                                                            \
    do {                                                                                             \
        const int rc_ = ensure_dynamic_lds((const void *)conv3d_g_kernel<CW_, PFW_, F32_, FAST_>, 160 * 1024); \
        if (rc_ != DFM_OK) return rc_;                                                               \
        hipLaunchKernelGGL((conv3d_g_kernel<CW_, PFW_, F32_, FAST_>), grid, dim3(256), lds, st, pl.g, \
                           (const bf16_t *)x, wfrag, scale, shift, (const bf16_t *)residual,         \
                           (bf16_t *)out, zero);                                                     \
    } while (0)
#define G_LAUNCH(CW_, PFW_)                                                                          \
    do {                                                                                             \
        if (f32 && fast && CW_ * PFW_ < 8) G_LAUNCH1(CW_, PFW_, true, true); /* (the 128-accumulator fp32 form spills) */ \
        else if (f32) G_LAUNCH1(CW_, PFW_, true, false);                                             \
        else if (fast) G_LAUNCH1(CW_, PFW_, false, true);                                            \
        else G_LAUNCH1(CW_, PFW_, false, false);                                                     \
    } while (0)
#define G_LAUNCH_OLD(CW_, PFW_)                                                                      \
    do {                                                                                             \
        if (f32) {                                                                                   \
            const int rc_ = ensure_dynamic_lds((const void *)conv3d_g_kernel<CW_, PFW_, true>, 160 * 1024); \
            if (rc_ != DFM_OK) return rc_;                                                           \
            hipLaunchKernelGGL((conv3d_g_kernel<CW_, PFW_, true>), grid, dim3(256), lds, st, pl.g,   \
                               (const bf16_t *)x, wfrag, scale, shift, (const bf16_t *)residual,     \
                               (bf16_t *)out, zero);                                                 \
        } else {                                                                                     \
            const int rc_ = ensure_dynamic_lds((const void *)conv3d_g_kernel<CW_, PFW_>, 160 * 1024); \
            if (rc_ != DFM_OK) return rc_;                                                           \
            hipLaunchKernelGGL((conv3d_g_kernel<CW_, PFW_>), grid, dim3(256), lds, st, pl.g,         \
                               (const bf16_t *)x, wfrag, scale, shift, (const bf16_t *)residual,     \
                               (bf16_t *)out, zero);                                                 \
        }                                                                                            \
    } while (0)
    if (pl.cw == 2) {
        switch (pl.pfw) {
        case 1: G_LAUNCH(2, 1); break;
        case 2: G_LAUNCH(2, 2); break;
        case 3: G_LAUNCH(2, 3); break;
        default: G_LAUNCH(2, 4); break;
        }
    } else {
        switch (pl.pfw) {
        case 1: G_LAUNCH(1, 1); break;
        case 2: G_LAUNCH(1, 2); break;
        case 3: G_LAUNCH(1, 3); break;
        default: G_LAUNCH(1, 4); break;
        }
    }
#undef G_LAUNCH
#undef G_LAUNCH1
#undef G_LAUNCH_OLD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

// the tiling the launch will use: {PFW, CW, TD, TH, TW, block pixels, LDS bytes, workgroups}
extern "C" DFM_API int dfm_conv3d_g_plan(const dfm_conv3d_desc *desc, int64_t *plan8)
{
    const int rc = g_check(desc);
    if (rc != DFM_OK) return rc;
    if (!plan8) return set_error(DFM_ERR_INVALID_ARG, "NULL plan");
    GPlan pl;
    if (!g_plan(desc, pl)) return set_error(DFM_ERR_UNSUPPORTED, "no tiling fits the LDS budget");
    plan8[0] = pl.pfw; plan8[1] = pl.cw; plan8[2] = pl.g.d.tile; plan8[3] = pl.g.h.tile; plan8[4] = pl.g.w.tile;
    plan8[5] = pl.g.block_px; plan8[6] = (int64_t)pl.lds;
    plan8[7] = (int64_t)pl.g.d.tiles * pl.g.h.tiles * pl.g.w.tiles * pl.g.cout_tiles * pl.classes * desc->n;
    return DFM_OK;
}

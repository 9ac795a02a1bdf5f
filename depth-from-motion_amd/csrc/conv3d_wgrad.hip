// conv3d_wgrad.hip -- hand-written MFMA weight gradient of the 3x3x3 convolutions of the path
// (backward-weight of Conv3d / ConvTranspose3d in the aggregation stacks and voxel necks:
//  mmdet3d/models/backbones/dfm_backbone.py:50-128, models/utils/conv_modules.py:27-43,73-149,
//  models/necks/imvoxel_neck.py:26-55,85-117, models/necks/dfm_neck.py:29-95):
//
//   out[a][b][kd][kh][kw] = sum over output positions o of  g[o][a] * x[o * stride - pad + k][b]
//
// g: (N, Do, Ho, Wo, A), x: (N, Di, Hi, Wi, B) bf16 channels-last, out fp32 (A, B, 27).
//   nn.Conv3d:          g = grad_output, x = input          -> grad_weight (C_out, C_in, 3, 3, 3)
//   nn.ConvTranspose3d: g = input, x = grad_output, stride 2, pad 1 -> grad_weight (C_in, C_out, 3, 3, 3)
// (MIOpen's untuned NDHWC bf16 kernels for these shapes take 84 ms .. 1.26 s per convolution.)
//
// The contraction runs over PIXELS, the one dimension that is not contiguous in a channels-last
// tensor, so both MFMA operands (D[a][b] += A[a][k] * B[k][b], k = 16 consecutive output positions
// of one row) are transposed on their way into LDS:
//   * a workgroup (3 waves) owns a 32 x 32 block of (a, b) channels and walks tiles of 2 output rows
//     x 64 (32 when the row stride is 2) output positions; per tile it loads the rows of g and the
//     input rows of x the 27 taps touch as 16-byte pieces (8 channels of one pixel), four adjacent
//     pixels per lane, transposes the 4 x 8 block in registers (v_perm) and writes 8-byte runs of 4
//     pixels per channel: LDS holds g^T[row][a][w] and x^T[slice][row][phase][b][w];
//   * wave z takes kernel depth slice kd = z: 9 accumulator tiles (kh, kw), 144 registers;
//     per (output row, 16-position step, kh) it reads ONE aligned 16-byte vector of x^T per phase
//     (+ one neighbouring dword) and derives the three kw operands from it -- stride 1: funnel
//     shifts (v_alignbit) by one element left / right; stride 2: the row is staged de-interleaved
//     (even / odd phases), kw = 0, 1 are the two phases and kw = 2 the even phase shifted by one;
//   * partial sums of a workgroup's tiles stay in its accumulators; at the end every workgroup
//     writes its 27 x 32 x 32 partial to scratch and a small kernel sums them (deterministic, no
//     atomics).
// The kernel's row axis can be the tensor's H or W axis (the host swaps strides / extents and the
// tap index on the way out): narrow volumes (W = 12, 6, 3 in the voxel necks) contract along H.
#include <algorithm>
#include <cstdlib>

#include "dfm_common.h"

using namespace dfm;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

constexpr int WG_TH = 2;       // output rows per tile
constexpr int WG_THREADS = 192;
constexpr int WG_BATCH = 3;     // staging items a lane has in flight (4 x 16 bytes each)
constexpr int WG_JMAX = 9;      // staging items per lane and tile, at most (the plan checks)
constexpr int WG_IS_G = 1 << 13, WG_DEAD = 1 << 14;

#ifdef DFM_DEBUG_HOOKS
// s_memtime stamps of workgroup (0, 0): trace[wave][tile < 16][stamp < 8] (dfm_debug_set_wg_trace, tools/wgrad_trace.py)
unsigned long long *g_wg_trace = nullptr;
#define WG_STAMP(i)                                                                                       \
    do {                                                                                                  \
        if (trace && lane == 0 && tcount < 16) trace[((size_t)wave * 16 + tcount) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#endif

struct WGeom {
    int32_t N, Do, Ho, Wo, Di, Hi, Wi;
    int32_t sd, sh, pd, ph, pw;            // (the row stride sw is a template parameter)
    int64_t gsN, gsD, gsH, gsW;            // element strides of g (channels contiguous)
    int64_t xsN, xsD, xsH, xsW;            // element strides of x
    int32_t tiles_h, tiles_w, ntiles;      // tiles per (n, od) plane; N * Do * tiles_h * tiles_w
    int32_t wgs_per_pair, b_tiles;
    int32_t dchunk, nchunks, nitems;       // COL: output planes per work item, chunks per column, work items
};

// 4 pixels x 8 channels (q[e] = the 8 bf16 channels of pixel e) -> per channel the 4 pixels
template <typename V4>
__device__ __forceinline__ void transpose4x8(const V4 (&q)[4], u32x2_t (&t)[8])
{
    const uint32_t d[4][4] = {{q[0].x, q[0].y, q[0].z, q[0].w}, {q[1].x, q[1].y, q[1].z, q[1].w},
                              {q[2].x, q[2].y, q[2].z, q[2].w}, {q[3].x, q[3].y, q[3].z, q[3].w}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // dword i holds channels 2i (low half) and 2i + 1 (high half)
        t[2 * i].x = __builtin_amdgcn_perm(d[1][i], d[0][i], 0x05040100u);
        t[2 * i].y = __builtin_amdgcn_perm(d[3][i], d[2][i], 0x05040100u);
        t[2 * i + 1].x = __builtin_amdgcn_perm(d[1][i], d[0][i], 0x07060302u);
        t[2 * i + 1].y = __builtin_amdgcn_perm(d[3][i], d[2][i], 0x07060302u);
    }
}

// FLAT: the volume has depth 1 and depth padding 1 (a 2-D convolution run as a depth-1 volume: the
// training path of the 2-D necks).  Only the centre depth slice of the 27 taps can be non-zero, so one
// slice is staged instead of three and the three waves take the three kernel ROWS (kh) instead of the
// three depth slices: a third of the staging and of the MFMAs of the general form.
// COL (row stride 1, depth stride 1, a volume): a workgroup's consecutive tiles are consecutive output planes of ONE
// (h, w) window -- a column, cut into chunks of planes.  The three input slices then live in a ring of the three LDS
// slots (slice id in slot id mod 3): only the first tile of a chunk stages all three, every other tile ONE new slice
// (and its g rows) -- a third of the loads, transposes and LDS writes, which are 2/3 of a tile (profiles/archive/r05_c44_*).
template <int SW, bool FLAT = false, bool COL = false>
__global__ __launch_bounds__(WG_THREADS, 2) void conv3d_wgrad_kernel(WGeom g, const bf16_t *__restrict__ G,
                                                                 const bf16_t *__restrict__ X,
                                                                 float *__restrict__ part
#ifdef DFM_DEBUG_HOOKS
                                                                 , unsigned long long *trace_buf
#endif
)
{
    constexpr int TW = SW == 1 ? 64 : 32;   // output positions of a tile row
    constexpr int EP = TW + 16;             // staged elements per (row, phase, channel): 8 + TW + 8
    // LDS pitches of a channel's row in x^T / g^T: 4 x an ODD number of dwords (44 / 28 and 36 / 20).  The MFMA phase
    // reads 16 bytes per lane with lane = channel: at pitch 40 dwords (EP) 16 lanes cover 8 of the 16 four-bank groups
    // twice, at pitch 32 (TW) two of them eight times -- the LDS, not the matrix core, was what a tile took
    constexpr int XP = TW + 24, GP = TW + 8;
    constexpr int NPH = SW;                 // phases (stride 2: even / odd input positions)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int RH = (WG_TH - 1) * g.sh + 3;  // input rows per depth slice
    constexpr int NZ = FLAT ? 1 : 3;        // staged depth slices
    bf16_t *XS = (bf16_t *)smem;                                  // [NZ][RH][NPH][32][XP]
    bf16_t *GT = XS + (size_t)NZ * RH * NPH * 32 * XP;            // [WG_TH][32][GP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l32 = lane & 31, half = lane >> 5;
    const int pair = blockIdx.y;
    const int a0 = (pair / g.b_tiles) * 32, b0 = (pair % g.b_tiles) * 32;

    f32x16_t acc[NZ][3];  // FLAT: acc[0][kw] of this wave's kernel row
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- this lane's staging items (tile-invariant): packed coordinates and LDS offset ----
    // item: depth slice z (bits 0-1) | row ihr / ohr (2-4) | channel block (5-6) | phase (7) | position group (8-12) |
    // WG_IS_G | WG_DEAD; ilds: element offset from XS of the item's first 8-byte run
    const int xitems = NZ * RH * NPH * 4 * (EP / 4);
    const int gitems = WG_TH * 4 * (TW / 4);
    const int nitems = xitems + gitems;
    int item[WG_JMAX], ilds[WG_JMAX];
#pragma unroll
    for (int j = 0; j < WG_JMAX; ++j) {
        const int it = tid + j * WG_THREADS;
        if (it < xitems) {
            // (channel block fastest: four consecutive lanes load the 64 contiguous bytes of one pixel -- with the
            //  position group fastest every lane of a load sat in a line of its own, 256 bytes from its neighbour's,
            //  and ISSUING a batch of 12 loads took 1900 cycles: profiles/archive/r05_c45_*)
            int q = it;
            const int cb = q & 3; q >>= 2;
            const int mg = q % (EP / 4); q /= (EP / 4);
            const int ph = q % NPH; q /= NPH;
            const int ihr = q % RH;
            const int z = q / RH;
            item[j] = z | (ihr << 2) | (cb << 5) | (ph << 7) | (mg << 8);
            ilds[j] = ((((COL ? 0 : z) * RH + ihr) * NPH + ph) * 32 + cb * 8) * XP + mg * 4;  // (COL: + the slice's slot, per tile)
        } else if (it < nitems) {
            int q = it - xitems;
            const int cb = q & 3; q >>= 2;
            const int mg = q % (TW / 4);
            const int ohr = q / (TW / 4);
            item[j] = (ohr << 2) | (cb << 5) | (mg << 8) | WG_IS_G;
            ilds[j] = NZ * RH * NPH * 32 * XP + (ohr * 32 + cb * 8) * GP + mg * 4;
        } else {
            item[j] = WG_DEAD;
            ilds[j] = 0;
        }
    }
#ifdef DFM_DEBUG_HOOKS
    unsigned long long *trace = (blockIdx.x == 0 && blockIdx.y == 0) ? trace_buf : nullptr;
    int tcount = -1;
#endif
    for (int t = blockIdx.x; t < (COL ? g.nitems : g.ntiles); t += g.wgs_per_pair) {
      int r = t;
      const int twb = r % g.tiles_w; r /= g.tiles_w;
      const int thb = r % g.tiles_h; r /= g.tiles_h;
      const int odq = COL ? r % g.nchunks : r % g.Do;
      const int n = COL ? r / g.nchunks : r / g.Do;
      const int od_lo = COL ? odq * g.dchunk : odq, od_hi = COL ? min(od_lo + g.dchunk, g.Do) : odq + 1;
      for (int od = od_lo; od < od_hi; ++od) {
        const bool first = !COL || od == od_lo;
#ifdef DFM_DEBUG_HOOKS
        ++tcount;
#endif
        WG_STAMP(0);
        const int ow0 = twb * TW, oh0 = thb * WG_TH;
        // COL: the ring slots of the slices od - pd + {0, 1, 2}
        const int s0 = COL ? (od * g.sd - g.pd + 3) % 3 : 0, s1 = COL ? (s0 + 1) % 3 : 1, s2 = COL ? (s0 + 2) % 3 : 2;

        // ---- stage x^T (rows (z, ihr), phases, 4 channel blocks, EP / 4 groups of 4 positions) and g^T ----
        // One list of items (x items first, then g items), a lane's items are `tid + j * WG_THREADS`; their
        // coordinates inside the tile do not depend on the tile and were decoded ONCE above (item[], ilds[]): per
        // tile an item costs a few compares, clamps and 32-bit multiply-adds -- decoded per tile (divisions by
        // run-time extents, 64-bit products) the address arithmetic of a batch was 2500 cycles and a tile's staging
        // 10 000 of its 15 000 (round 5 trace: profiles/archive/r05_c44_*).  WG_BATCH items at a time: ALL their 16-byte loads
        // are issued first -- unconditionally, from a clamped address, through global-address-space pointers -- and
        // only then selected against the bounds, transposed and written to LDS.  (Round 1-4: a `valid ? load : 0`
        // per piece, compiled to a flat_load followed by s_waitcnt vmcnt(0) lgkmcnt(0): 20 to 32 dependent round
        // trips to memory per tile.  Raw buffer loads with out-of-range offsets instead of clamp + select: the
        // stride-2 form 177 -> 213 us, not kept.)
        {
            typedef const __attribute__((address_space(1))) u32x4_t *gv_t;
            const bf16_t *xb = X + (size_t)n * g.xsN + b0;
            const bf16_t *gb = G + (size_t)n * g.gsN + (size_t)od * g.gsD + a0;
            const int id0 = od * g.sd - g.pd, ih0 = oh0 * g.sh - g.ph;
            const int iw0 = SW * (ow0 - 8) - g.pw + (SW == 1 ? 1 : 0);
            const int xsD = (int)g.xsD, xsH = (int)g.xsH, xsW = (int)g.xsW, gsH = (int)g.gsH, gsW = (int)g.gsW;
#pragma unroll
            for (int bi = 0; bi < WG_JMAX / WG_BATCH; ++bi) {
                if (bi * WG_BATCH * WG_THREADS >= nitems) break;
                // COL, not the chunk's first tile: only slice 2 is new, and with row stride 1 its 320 items and the
                // 128 g items are exactly the lanes' items 3 .. 5 (the second batch)
                if (COL && !first && bi == 0) continue;
                u32x4_t qv[WG_BATCH][4];
                unsigned okm[WG_BATCH];
#pragma unroll
                for (int k = 0; k < WG_BATCH; ++k) {
                    const int j = bi * WG_BATCH + k;
                    int ge = item[j];
                    asm volatile("" : "+v"(ge));  // (unpacked here, per tile: hoisted, the fields of 9 items are 60 registers)
                    if (COL && !first && !(ge & WG_IS_G) && (ge & 3) != 2) ge |= WG_DEAD;
                    const int mg4 = ((ge >> 8) & 31) * 4, cb8 = ((ge >> 5) & 3) * 8;
                    okm[k] = 0u;
                    if (!(ge & WG_IS_G)) {
                        const int id = id0 + (FLAT ? 1 : (ge & 3)), ih = ih0 + ((ge >> 2) & 7);
                        const bool rok = !(ge & WG_DEAD) && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi;
                        const int rowo = min(max(id, 0), g.Di - 1) * xsD + min(max(ih, 0), g.Hi - 1) * xsH + cb8;
                        const int iwb = iw0 + SW * mg4 + ((ge >> 7) & 1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int iw = iwb + SW * e;
                            if (rok && (unsigned)iw < (unsigned)g.Wi) okm[k] |= 1u << e;
                            qv[k][e] = *(gv_t)(xb + (rowo + min(max(iw, 0), g.Wi - 1) * xsW));
                        }
                    } else {
                        const int oh = oh0 + ((ge >> 2) & 7);
                        const bool rok = !(ge & WG_DEAD) && oh < g.Ho;
                        const int rowo = min(oh, g.Ho - 1) * gsH + cb8;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ow = ow0 + mg4 + e;
                            if (rok && ow < g.Wo) okm[k] |= 1u << e;
                            qv[k][e] = *(gv_t)(gb + (rowo + min(ow, g.Wo - 1) * gsW));
                        }
                    }
                }
                if (bi == 0) WG_STAMP(1);   // first batch: loads issued
#ifdef DFM_DEBUG_HOOKS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                if (bi == 0) WG_STAMP(2);   // ... arrived
#pragma unroll
                for (int k = 0; k < WG_BATCH; ++k) {
                    const int j = bi * WG_BATCH + k;
                    int ge = item[j], lo = ilds[j];
                    asm volatile("" : "+v"(ge), "+v"(lo));
                    if (COL && !first && !(ge & WG_IS_G) && (ge & 3) != 2) ge |= WG_DEAD;
                    if (ge & WG_DEAD) continue;
                    if (COL && !(ge & WG_IS_G)) {
                        const int zz = ge & 3;
                        lo += (zz == 0 ? s0 : zz == 1 ? s1 : s2) * (RH * NPH * 32 * XP);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!((okm[k] >> e) & 1u)) qv[k][e] = u32x4_t{0u, 0u, 0u, 0u};
                    u32x2_t tt[8];
                    transpose4x8(qv[k], tt);
                    bf16_t *dst = XS + lo;
                    const int pitch = (ge & WG_IS_G) ? GP : XP;
#pragma unroll
                    for (int c = 0; c < 8; ++c) *(u32x2_t *)(dst + (size_t)c * pitch) = tt[c];
                }
            }
        }
        WG_STAMP(3);   // staged
        __syncthreads();
        WG_STAMP(4);

        // ---- MFMAs: wave = kernel depth slice (FLAT: kernel row of the one slice) ----
        const int z = FLAT ? 0 : COL ? (wave == 0 ? s0 : wave == 1 ? s1 : s2) : wave;
#pragma unroll
        for (int ohr = 0; ohr < WG_TH; ++ohr) {
#pragma unroll
            for (int ks = 0; ks < TW / 16; ++ks) {
                bf16x8_t af;
                {
                    const u32x4_t v = *(const u32x4_t *)(GT + ((size_t)ohr * 32 + l32) * GP + ks * 16 + half * 8);
                    __builtin_memcpy(&af, &v, 16);
                }
                const int m0 = ks * 16 + half * 8 + 8;
#pragma unroll
                for (int khi = 0; khi < NZ; ++khi) {
                    const int kh = FLAT ? wave : khi;
                    const int ihr = ohr * g.sh + kh;
                    const bf16_t *base = XS + ((size_t)(z * RH + ihr) * NPH * 32 + l32) * XP;
                    u32x4_t f0, f1, f2;
                    if constexpr (SW == 1) {
                        const u32x4_t cur = *(const u32x4_t *)(base + m0);
                        const uint32_t pv = *(const uint32_t *)(base + m0 - 2);
                        const uint32_t nx = *(const uint32_t *)(base + m0 + 8);
                        f1 = cur;
                        f0 = u32x4_t{__builtin_amdgcn_alignbit(cur.x, pv, 16), __builtin_amdgcn_alignbit(cur.y, cur.x, 16),
                                     __builtin_amdgcn_alignbit(cur.z, cur.y, 16), __builtin_amdgcn_alignbit(cur.w, cur.z, 16)};
                        f2 = u32x4_t{__builtin_amdgcn_alignbit(cur.y, cur.x, 16), __builtin_amdgcn_alignbit(cur.z, cur.y, 16),
                                     __builtin_amdgcn_alignbit(cur.w, cur.z, 16), __builtin_amdgcn_alignbit(nx, cur.w, 16)};
                    } else {
                        const u32x4_t c0 = *(const u32x4_t *)(base + m0);
                        const u32x4_t c1 = *(const u32x4_t *)(base + (size_t)32 * XP + m0);
                        const uint32_t nx = *(const uint32_t *)(base + m0 + 8);
                        f0 = c0;
                        f1 = c1;
                        f2 = u32x4_t{__builtin_amdgcn_alignbit(c0.y, c0.x, 16), __builtin_amdgcn_alignbit(c0.z, c0.y, 16),
                                     __builtin_amdgcn_alignbit(c0.w, c0.z, 16), __builtin_amdgcn_alignbit(nx, c0.w, 16)};
                    }
                    bf16x8_t b0f, b1f, b2f;
                    __builtin_memcpy(&b0f, &f0, 16);
                    __builtin_memcpy(&b1f, &f1, 16);
                    __builtin_memcpy(&b2f, &f2, 16);
                    acc[khi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0f, acc[khi][0], 0, 0, 0);
                    acc[khi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1f, acc[khi][1], 0, 0, 0);
                    acc[khi][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2f, acc[khi][2], 0, 0, 0);
                }
            }
        }
        WG_STAMP(5);   // multiplied
        __syncthreads();
        WG_STAMP(6);
      }
    }

    // ---- this workgroup's partial: part[pair][wg][tap][a][b] ----
    float *pp = part + ((size_t)pair * g.wgs_per_pair + blockIdx.x) * (27 * 1024);
#pragma unroll
    for (int khi = 0; khi < NZ; ++khi)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            // FLAT: the centre slice's row `wave` (the reduce pass writes zeros for the other two slices)
            const int tap = FLAT ? (3 + wave) * 3 + kw : (wave * 3 + khi) * 3 + kw;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int a = 8 * (i >> 2) + 4 * half + (i & 3);
                pp[((size_t)tap * 32 + a) * 32 + l32] = acc[khi][kw][i];
            }
        }
}

// out[(a0 + a) * B + b0 + b][tap'] = sum over the pair's workgroups; tap' undoes the h/w swap.
// 64 consecutive elements per block, the workgroup partials split over 4 thread groups (the sum of
// up to 512 partials per element is latency-bound when one thread walks them all)
// TO: the gradient's type (fp32, or bf16 -- the parameter's own type: the sums rounded once, no conversion launch)
template <typename TO>
__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float *__restrict__ part, int wgs_per_pair,
                                                                  int b_tiles, int B, int swap_hw, int flat,
                                                                  TO *__restrict__ out)
{
    __shared__ float sh[4][64];
    const int pair = blockIdx.y;
    const int e = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;  // tap * 1024 + a * 32 + b  (27 * 1024 is a multiple of 64)
    const float *p = part + (size_t)pair * wgs_per_pair * (27 * 1024) + idx;
    float s = 0.0f;
    // flat (depth-1 volume): only the centre depth slice (taps 9..17) was computed and written
    const bool live = !flat || ((idx >> 10) >= 9 && (idx >> 10) < 18);  // (uniform per block: 1024 % 64 == 0)
    if (live)
        for (int w = sub; w < wgs_per_pair; w += 4) s += p[(size_t)w * (27 * 1024)];
    sh[sub][e] = s;
    __syncthreads();
    if (sub) return;
    s = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
    const int tap = idx >> 10, a = (idx >> 5) & 31, b = idx & 31;
    const int kd = tap / 9, k1 = (tap / 3) % 3, k2 = tap % 3;
    const int tp = swap_hw ? (kd * 3 + k2) * 3 + k1 : tap;
    const int a0 = (pair / b_tiles) * 32, b0 = (pair % b_tiles) * 32;
    out[((size_t)(a0 + a) * B + b0 + b) * 27 + tp] = elem<TO>::store(s);
}

struct WPlan {
    WGeom g;
    int sw, swap, pairs, flat, col;
    size_t lds, scratch;
};

int wgrad_plan(const dfm_conv3d_wgrad_desc *d, WPlan &pl)
{
    if (!d) return set_error(DFM_ERR_INVALID_ARG, "NULL wgrad descriptor");
    if (d->n <= 0 || d->a <= 0 || d->b <= 0 || d->a % 32 || d->b % 32)
        return set_error(DFM_ERR_UNSUPPORTED, "channel counts must be positive multiples of 32");
    for (int i = 0; i < 3; ++i) {
        if (d->g_size[i] <= 0 || d->x_size[i] <= 0) return set_error(DFM_ERR_INVALID_ARG, "non-positive size");
        if (d->stride[i] < 1 || d->stride[i] > 2 || d->padding[i] < 0 || d->padding[i] > 2)
            return set_error(DFM_ERR_UNSUPPORTED, "stride must be 1 or 2, padding 0..2");
        // the first tap of the last output position lies inside the padded input
        if ((d->g_size[i] - 1) * d->stride[i] - d->padding[i] >= d->x_size[i] + d->padding[i])
            return set_error(DFM_ERR_INVALID_ARG, "g_size does not fit x_size / stride / padding");
    }
    for (int i = 0; i < 4; ++i)
        if (d->g_stride[i] <= 0 || d->x_stride[i] <= 0 || (i > 0 && (d->g_stride[i] % 8 || d->x_stride[i] % 8)))
            return set_error(DFM_ERR_INVALID_ARG, "element strides must be positive multiples of 8");
    WGeom &g = pl.g;
    // contract along the longer of the two in-plane axes
    const bool swap = d->g_size[1] > d->g_size[2];
    const int hi = swap ? 2 : 1, wi = swap ? 1 : 2;
    pl.swap = swap ? 1 : 0;
    g.N = d->n; g.Do = d->g_size[0]; g.Ho = d->g_size[hi]; g.Wo = d->g_size[wi];
    g.Di = d->x_size[0]; g.Hi = d->x_size[hi]; g.Wi = d->x_size[wi];
    g.sd = d->stride[0]; g.sh = d->stride[hi]; pl.sw = d->stride[wi];
    g.pd = d->padding[0]; g.ph = d->padding[hi]; g.pw = d->padding[wi];
    g.gsN = d->g_stride[0]; g.gsD = d->g_stride[1]; g.gsH = d->g_stride[1 + hi]; g.gsW = d->g_stride[1 + wi];
    g.xsN = d->x_stride[0]; g.xsD = d->x_stride[1]; g.xsH = d->x_stride[1 + hi]; g.xsW = d->x_stride[1 + wi];
    const int TW = pl.sw == 1 ? 64 : 32;
    g.tiles_w = (g.Wo + TW - 1) / TW;
    g.tiles_h = (g.Ho + WG_TH - 1) / WG_TH;
    const long long nt = (long long)g.N * g.Do * g.tiles_h * g.tiles_w;
    if (nt >= (1ll << 31)) return set_error(DFM_ERR_UNSUPPORTED, "too many tiles");
    g.ntiles = (int)nt;
    g.b_tiles = d->b / 32;
    pl.pairs = (d->a / 32) * g.b_tiles;
    if (pl.pairs > 65535) return set_error(DFM_ERR_UNSUPPORTED, "too many channel tiles");
    // ~2 resident workgroups per CU over the whole launch, at least 8 tiles per workgroup
    int wpp = (int)std::max<long long>(1, std::min<long long>((512 + pl.pairs - 1) / pl.pairs, (nt + 7) / 8));
    g.wgs_per_pair = wpp;
    const int RH = (WG_TH - 1) * g.sh + 3, XP = TW + 24, GP = TW + 8;
    // a depth-1 volume with depth padding 1 (a 2-D convolution): the centre depth slice only
    pl.flat = (g.Di == 1 && g.Do == 1 && g.pd == 1 && g.sd == 1) ? 1 : 0;
    pl.lds = ((size_t)(pl.flat ? 1 : 3) * RH * pl.sw * 32 * XP + (size_t)WG_TH * 32 * GP) * 2;
    {
        // the kernel's staging: at most WG_JMAX items per lane, 32-bit element offsets inside a sample
        const int NZ = pl.flat ? 1 : 3, EPx = TW + 16;
        const long long items = (long long)NZ * RH * pl.sw * 4 * (EPx / 4) + (long long)WG_TH * 4 * (TW / 4);
        if (items > (long long)WG_JMAX * WG_THREADS) return set_error(DFM_ERR_UNSUPPORTED, "tile stages too many pieces");
        if ((long long)g.Di * g.xsD >= (1ll << 31) || (long long)g.Hi * g.xsH >= (1ll << 31) ||
            (long long)g.Wi * g.xsW >= (1ll << 31) || (long long)g.Ho * g.gsH >= (1ll << 31) ||
            (long long)g.Wo * g.gsW >= (1ll << 31) ||
            (long long)g.Di * g.xsD + (long long)g.Hi * g.xsH + (long long)g.Wi * g.xsW >= (1ll << 31) ||
            (long long)g.Ho * g.gsH + (long long)g.Wo * g.gsW >= (1ll << 31))
            return set_error(DFM_ERR_UNSUPPORTED, "sample larger than 2^31 elements");
    }
    // column mode (COL): row and depth stride 1 (and h stride 1: the item order the kernel relies on), a volume; work
    // items = columns x chunks of output planes (a chunk's first tile stages all three slices).  DFM_WGRAD_COL=0 keeps a tile per step (A/B runs), DFM_WGRAD_CHUNK=<planes> (tests)
    pl.col = 0;
    g.dchunk = 1; g.nchunks = g.Do; g.nitems = g.ntiles;
    {
        const char *e = getenv("DFM_WGRAD_COL");
        // (the kernel takes "the second batch of a lane's items = the new slice + the g rows" from the item order:
        //  3 slices x 4 rows x 4 channel blocks x 20 position groups = 5 items per lane exactly)
        const bool order_ok = 3 * RH * 4 * ((TW + 16) / 4) == 5 * WG_THREADS && WG_BATCH == 3;
        if (!(e && e[0] == '0') && order_ok && pl.sw == 1 && g.sd == 1 && g.sh == 1 && !pl.flat && g.Do > 1) {
            const long long cols = (long long)g.N * g.tiles_h * g.tiles_w;
            const long long wgs = std::max<long long>(1, 512 / pl.pairs);
            // planes per chunk: the launch is rounds x (planes + the first tile's two extra slices, ~0.7 of a tile)
            // long, rounds = items per workgroup, rounded UP ("about three items, at least 6 planes" gave config K
            // 1600 items on 512 workgroups: 3.1 -> 4 rounds of 9 planes where 2 rounds of 15 do it)
            int dc = g.Do;
            double best = 1e30;
            for (int c = std::min(g.Do, 3); c <= g.Do; ++c) {
                const long long items = cols * ((g.Do + c - 1) / c);
                const long long rounds = (items + wgs - 1) / wgs;
                const double cost = (double)rounds * (c + 0.7);
                if (cost < best - 1e-9) { best = cost; dc = c; }
            }
            if (const char *c = getenv("DFM_WGRAD_CHUNK")) dc = std::max(1, atoi(c));
            dc = std::min(dc, g.Do);
            g.dchunk = dc;
            g.nchunks = (g.Do + dc - 1) / dc;
            const long long items = cols * g.nchunks;
            if (items < (1ll << 31)) {
                g.nitems = (int)items;
                wpp = (int)std::max<long long>(1, std::min<long long>(wgs, items));
                g.wgs_per_pair = wpp;
                pl.col = 1;
            }
        }
    }
    pl.scratch = (size_t)pl.pairs * wpp * 27 * 1024 * sizeof(float);
    return DFM_OK;
}

}  // namespace

extern "C" DFM_API size_t dfm_conv3d_wgrad_workspace_bytes(const dfm_conv3d_wgrad_desc *desc)
{
    WPlan pl;
    if (wgrad_plan(desc, pl) != DFM_OK) return 0;
    return pl.scratch;
}

static int wgrad_impl(const dfm_conv3d_wgrad_desc *desc, const void *g, const void *x, void *out, int out_dtype,
                      void *workspace, size_t workspace_bytes, void *stream);

extern "C" DFM_API int dfm_conv3d_wgrad(const dfm_conv3d_wgrad_desc *desc, const void *g, const void *x,
                                        float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return wgrad_impl(desc, g, x, out, DFM_F32, workspace, workspace_bytes, stream);
}

// the same with the gradient written in out_dtype (DFM_F32 | DFM_BF16): a bf16 parameter's gradient leaves the
// reduction kernel in the parameter's type -- the fp32 sums rounded once -- instead of through a conversion launch
extern "C" DFM_API int dfm_conv3d_wgrad_to(const dfm_conv3d_wgrad_desc *desc, const void *g, const void *x,
                                           void *out, int32_t out_dtype, void *workspace, size_t workspace_bytes,
                                           void *stream)
{
    if (out_dtype != DFM_F32 && out_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "gradient dtype must be DFM_F32 or DFM_BF16");
    return wgrad_impl(desc, g, x, out, out_dtype, workspace, workspace_bytes, stream);
}

static int wgrad_impl(const dfm_conv3d_wgrad_desc *desc, const void *g, const void *x, void *out, int out_dtype,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    WPlan pl;
    int rc = wgrad_plan(desc, pl);
    if (rc != DFM_OK) return rc;
    if (!g || !x || !out || !workspace) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (workspace_bytes < pl.scratch) return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_conv3d_wgrad_workspace_bytes");
    if (((uintptr_t)g & 15) || ((uintptr_t)x & 15)) return set_error(DFM_ERR_INVALID_ARG, "g and x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(pl.g.wgs_per_pair, pl.pairs);
    if (pl.lds > 160 * 1024) return set_error(DFM_ERR_UNSUPPORTED, "tile does not fit the LDS");
#ifdef DFM_DEBUG_HOOKS
#define WG_TRACE_ARG , g_wg_trace
#else
#define WG_TRACE_ARG
#endif
#define W_LAUNCH(SW_, FLAT_)                                                                              \
    do {                                                                                               \
        rc = ensure_dynamic_lds((const void *)conv3d_wgrad_kernel<SW_, FLAT_>, 160 * 1024);            \
        if (rc != DFM_OK) return rc;                                                                   \
        hipLaunchKernelGGL((conv3d_wgrad_kernel<SW_, FLAT_>), grid, dim3(WG_THREADS), pl.lds, st, pl.g, \
                           (const bf16_t *)g, (const bf16_t *)x, (float *)workspace WG_TRACE_ARG);     \
    } while (0)
    if (pl.col) {
        rc = ensure_dynamic_lds((const void *)conv3d_wgrad_kernel<1, false, true>, 160 * 1024);
        if (rc != DFM_OK) return rc;
        hipLaunchKernelGGL((conv3d_wgrad_kernel<1, false, true>), grid, dim3(WG_THREADS), pl.lds, st, pl.g,
                           (const bf16_t *)g, (const bf16_t *)x, (float *)workspace WG_TRACE_ARG);
    } else if (pl.sw == 1) {
        if (pl.flat) W_LAUNCH(1, true); else W_LAUNCH(1, false);
    } else {
        if (pl.flat) W_LAUNCH(2, true); else W_LAUNCH(2, false);
    }
#undef W_LAUNCH
    if (out_dtype == DFM_BF16)
        hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel<bf16_t>, dim3(27 * 1024 / 64, pl.pairs), dim3(256), 0, st,
                           (const float *)workspace, pl.g.wgs_per_pair, pl.g.b_tiles, desc->b, pl.swap, pl.flat,
                           (bf16_t *)out);
    else
        hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel<float>, dim3(27 * 1024 / 64, pl.pairs), dim3(256), 0, st,
                           (const float *)workspace, pl.g.wgs_per_pair, pl.g.b_tiles, desc->b, pl.swap, pl.flat,
                           (float *)out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

#ifdef DFM_DEBUG_HOOKS
extern "C" DFM_API void dfm_debug_set_wg_trace(void *buf) { g_wg_trace = (unsigned long long *)buf; }
#endif

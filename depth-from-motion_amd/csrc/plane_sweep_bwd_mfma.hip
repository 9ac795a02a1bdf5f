// plane_sweep_bwd_mfma.hip -- backward of dense bf16 plane sweeps as a banded matrix product.
//
// Replaces (for cost_sample_factor == 1, bf16 gradients) the scatter-add of
//   grad_cur/prev[c, y, x] += weight(d, h, w; y, x) * grad_out[c, d, h, w]
// i.e. autograd of F.grid_sample in build_dfm_cost (reference dfm_backbone.py:296-311), which
// sweep_bwd_tile_kernel (plane_sweep.hip) does with four 64-bit fixed-point LDS atomics per value.
//
// The bilinear weights of the 32 lattice points of one lattice-row segment at one depth plane form a
// banded 32 x (pixels) matrix with two non-zeros per point and image row; it does not depend on the
// channel.  So per (plane, lattice row) the contribution to one image row is
//   out[c][x] = sum_k g[c][k] * Wr[k][x]          (k = lattice point, x = map pixel)
// -- a 16(ch) x 32(points) x 16(pixels) v_mfma_f32_16x16x32_bf16 per 16 channels and 16 pixels, whose
// A operand is the gradient exactly as it lies in memory (8 consecutive points of one channel = one
// 16-byte load) and whose B operand is built ONCE per (tile, plane) for all channels: every lattice
// point writes its four bf16 weights into a zeroed fragment image in LDS (and clears them again two
// planes later), the waves read ready-made B fragments with one ds_read_b128 each.  Sums are combined
// by the matrix product, so there are no atomics, no fixed point and no float->fixed conversions on
// the way; the accumulators (a window of 6 image rows x 48 columns x 32 channels per wave) stay in
// registers across depth planes for as long as the tile's footprints stay inside the window -- the
// whole depth range for the cur map (its sample positions do not move with depth), tens of planes
// for the far planes of the prev map -- and are added to the gradient map with coalesced global
// atomics when the window has to move.
//
// Forward motion zooms the nearest planes of the prev map to several map pixels per lattice point: a
// 32-point segment then no longer fits the 48-column window.  Such planes take two steps of 16 points
// or four of 8 (same loop, a point mask); beyond SWEEP_BWD_ZOOM_FOUR the plane stays with
// sweep_bwd_tile_kernel.  Both kernels derive the split plane from the same device function on the same
// inputs (sweep_zoom_split: the sample positions of the four lattice corners), so every (plane, point)
// is handled by exactly one of them.
//
// Numerics: the fp32 bilinear weight w enters the matrix product as TWO bf16 terms, hi = bf16(w) and
// lo = bf16(w - hi) (w - hi is exact in fp32), each in its own fragment image: |w - hi - lo| <= 2^-18 |w|,
// the gradients are bf16 already, products are exact and sums fp32 in the MFMA -- i.e. fp32 arithmetic
// on (to 2^-18) fp32 weights, like the reference's autograd of grid_sample; a single bf16 term
// (round 3) was a 2^-9 relative error per term.  Costs a second fragment read + MFMA per fragment on a
// pipe that was a quarter busy.  A tile whose window sum turns non-finite is redone by a per-value path
// with plain float atomics, so Inf / NaN gradients propagate to their four taps exactly as in torch (a
// matrix product would smear 0 * Inf over the window).
#include <algorithm>
#include <type_traits>

#include "dfm_common.h"

namespace dfm {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM_TW = 32;   // lattice points per row segment (the K of one MFMA)
constexpr int BM_TH = 2;    // lattice rows per tile
constexpr int BM_R = 6;     // accumulator window: image rows
constexpr int BM_NT = 3;    //                     16-column blocks
constexpr int BM_MT = 2;    // 16-channel blocks per wave
constexpr int BM_SLOTS = 3; // image rows one lattice row may touch in one plane
constexpr int BM_FRAG = 1024;                                  // bytes of one B fragment (64 lanes x 16)
constexpr int BM_IMG1 = BM_TH * BM_SLOTS * BM_NT * BM_FRAG;    // fragment image of one plane and one weight term: 18 KB
constexpr int BM_IMG = 2 * BM_IMG1;                            // [hi | lo] terms of the weights
constexpr int BM_WAVES = 4;                                    // waves per tile, at most
constexpr int BM_RING = 3;                                     // planes of footprints ahead of the image
// LDS of one tile (a workgroup holds two neighbouring tiles)
constexpr int BM_OFF_TAPS = 2 * BM_IMG;                        // [2][64][4] u16 element offsets of the written weights
constexpr int BM_OFF_TABLE = BM_OFF_TAPS + 2 * 64 * 4 * 2;     // [3][BM_RING][64] packed footprint, fw, fn
constexpr int BM_OFF_EXT = BM_OFF_TABLE + 3 * BM_RING * 64 * 4;  // [BM_RING][8] extents
constexpr int BM_OFF_META = BM_OFF_EXT + BM_RING * 8 * 4;      // [32] ints
constexpr int BM_TILE_LDS = BM_OFF_META + 32 * 4;
constexpr int BM_LDS = 2 * BM_TILE_LDS;

enum { BM_EMPTY = 1, BM_SLOW = 2, BM_FLUSH = 4 };
// meta: [buf * 8 + ..]: 0 flags, 1 xbase, 2 ybase, 3 / 4 touched (row, block) bits of lattice row 0 / 1,
//       5 / 6 first image row of lattice row 0 / 1 relative to ybase;   16.. window state: have, xbase, ybase
enum { BM_M_FLAGS = 0, BM_M_XB = 1, BM_M_YB = 2, BM_M_BITS = 3, BM_M_ROW = 5, BM_M_STATE = 16 };

struct BmGrid {
    int batch, pairs_w, tiles_w, tiles_h, cblocks, waves;
    int per_xcd, total;
    int ablate;  // debug builds: 1 no gradient loads, 2 no MFMA, 4 no flush, 8 no producer
    unsigned long long *trace;  // debug builds (dfm_debug_set_bm_trace): cycles per phase of workgroup 0's waves
};

#ifdef DFM_BM_ABLATE  // experiments at release speed: -DDFM_BM_ABLATE=<mask> (1 no gradient loads, 2 no MFMA, 4 no flush, 8 no producer)
#define BM_AB(bit) (((DFM_BM_ABLATE) & (bit)) != 0)
#define BM_STAMP(i) do { } while (0)
#elif defined(DFM_DEBUG_HOOKS)
#define BM_AB(bit) ((tg.ablate & (bit)) != 0)
// cycles since the previous stamp are added to phase i of this wave
#define BM_STAMP(i)                                                                       \
    do {                                                                                   \
        if (tg.trace && work == 0) {                                                       \
            const unsigned long long t_ = __builtin_readcyclecounter();                    \
            tsum[i] += t_ - tlast;                                                         \
            tlast = t_;                                                                    \
        }                                                                                  \
    } while (0)
unsigned long long *g_bm_trace = nullptr;
#else
#define BM_AB(bit) false
#define BM_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void bm_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
}

// wave-wide reductions on the DPP network (no LDS round trips).  The result is valid in lane 63
// (wave) / lanes 31 and 63 (half-waves).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);  // invalid source lane: keep v
}
template <typename Op>
__device__ __forceinline__ int half_wave_reduce(int v, Op op)
{
    v = op(v, dpp_mov<0xb1>(v));         // quad_perm [1,0,3,2]
    v = op(v, dpp_mov<0x4e>(v));         // quad_perm [2,3,0,1]
    v = op(v, dpp_mov<0x114>(v));        // row_shr:4
    v = op(v, dpp_mov<0x118>(v));        // row_shr:8
    v = op(v, dpp_mov<0x142, 0xa>(v));   // row_bcast:15 into rows 1 and 3
    return v;                            // lanes 31 / 63: rows 0-1 / rows 2-3
}
// the lane id through an opaque move: what a rarely-run block derives from it is computed inside the
// block instead of being hoisted out of the plane loop into registers that the accumulators need
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
struct OpMin { __device__ int operator()(int a, int b) const { return min(a, b); } };
struct OpMax { __device__ int operator()(int a, int b) const { return max(a, b); } };
struct OpOr { __device__ int operator()(int a, int b) const { return a | b; } };

// raw gradient words of one A fragment: the 16 bytes at the 4-byte-aligned address at or below the
// lane's first element, and the lane's eighth element on its own (rows of an odd width start on odd
// elements, where the run of 8 straddles five words)
struct ARaw {
    u32x4_t q;
    uint32_t last;
};

__device__ __forceinline__ ARaw a_load(const bf16_t *p)
{
    // global address space spelled out: through a generic pointer (an integer cast loses the address
    // space) this is a flat_load, which also counts on lgkmcnt -- every LDS wait and every barrier
    // would then wait for the gradient words in flight
    typedef const __attribute__((address_space(1))) u32x4_t *gq_t;
    typedef const __attribute__((address_space(1))) bf16_t *gs_t;
    ARaw r;
    const uintptr_t a = (uintptr_t)p;
    r.q = *(gq_t)(a & ~(uintptr_t)3);
    r.last = ((gs_t)a)[7];
    return r;
}

__device__ __forceinline__ bf16x8_t a_frag(const ARaw &r, bool odd, bool zero)
{
    const uint32_t sh = odd ? 16u : 0u;
    u32x4_t o;
    o.x = __builtin_amdgcn_alignbit(r.q.y, r.q.x, sh);
    o.y = __builtin_amdgcn_alignbit(r.q.z, r.q.y, sh);
    o.z = __builtin_amdgcn_alignbit(r.q.w, r.q.z, sh);
    o.w = __builtin_amdgcn_alignbit(r.last, r.q.w, sh);
    if (zero) o = u32x4_t{0u, 0u, 0u, 0u};
    bf16x8_t f;
    __builtin_memcpy(&f, &o, 16);
    return f;
}

// Workgroup = two neighbouring 2 x 32 lattice tiles (a 128-byte line of a gradient row holds 64
// points: fetched by one CU, both of its halves are used) x up to 4 waves per tile of 32 channels
// each.  The two tiles run the same loop side by side with one barrier per plane.
template <int HALF>
__global__ __launch_bounds__(2 * 64 * BM_WAVES, 2) void sweep_bwd_mfma_kernel(
    SweepGeom g, SweepFast fast, BmGrid tg, const bf16_t *__restrict__ gout, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    float *__restrict__ gcur, float *__restrict__ gprev)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const int tid = threadIdx.x, lane = tid & 63, nw = tg.waves;
    const int sel = (tid >> 6) / nw, wave = (tid >> 6) - sel * nw;  // tile of the pair, wave of the tile
    unsigned char *lds = lds_all + sel * BM_TILE_LDS;
    unsigned char *img = lds;                                       // [2][BM_IMG]
    unsigned short *taps = (unsigned short *)(lds + BM_OFF_TAPS);
    uint32_t *tab_f = (uint32_t *)(lds + BM_OFF_TABLE);
    float *tab_w = (float *)(tab_f + BM_RING * 64), *tab_n = tab_w + BM_RING * 64;
    int *ext = (int *)(lds + BM_OFF_EXT);
    int *meta = (int *)(lds + BM_OFF_META);

    // consecutive work items (neighbouring tile pairs of one lattice row pair) on one XCD
    const int work = (blockIdx.x & 7) * tg.per_xcd + (blockIdx.x >> 3);
    if (work >= tg.total) return;
#if defined(DFM_DEBUG_HOOKS) && !defined(DFM_BM_ABLATE)
    unsigned long long tsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    int t = work;
    const int cb = t % tg.cblocks;
    t /= tg.cblocks;
    const int pair = t % tg.pairs_w;
    t /= tg.pairs_w;
    const int th = t % tg.tiles_h;
    const int b = t / tg.tiles_h;
    const int tw = 2 * pair + sel;
    const bool tile_live = tw < tg.tiles_w;  // an odd number of tiles per row: the last pair is half empty
    // the last tile of a row / column is shifted inside the lattice (its loads stay whole); the
    // points it shares with its neighbour belong to the neighbour
    const int own_w = tw * BM_TW, own_h = th * BM_TH;
    const int w0 = min(own_w, g.w_out - BM_TW), h0 = min(own_h, g.h_out - BM_TH);
    const int W = g.w_in, H = g.h_in, HW = H * W, hw = g.h_out * g.w_out;
    const float *Pb = P + b * 16, *Pib = Pinv + b * 16, *Tb = Tm + b * 16;
    float *gf = (HALF ? gprev : gcur) + (size_t)b * g.C * HW;
    const int c0 = (cb * nw + wave) * (16 * BM_MT);   // this wave's first channel
    const bool wave_live = tile_live && c0 < g.C;

    for (int i = wave * 64 + lane; i < 2 * BM_IMG / 16; i += 64 * nw) ((uint4 *)img)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = wave * 64 + lane; i < 2 * 64 * 4; i += 64 * nw) taps[i] = 0xffffu;
    if (wave == 0 && lane < 32) meta[lane] = 0;
    __syncthreads();
    // planes [0, d_start): the tile kernel (zoom beyond SWEEP_BWD_ZOOM_FOUR); [d_start, d_four): four
    // steps each, 8 of a segment's 32 points at a time; [d_four, d_two): two steps of 16 points (a whole
    // segment zoomed beyond SWEEP_BWD_ZOOM_ONE is wider than the accumulator window); the rest: one step
    int d_start = 0, d_four = 0, d_two = 0;
    if (HALF) {
        int *slot = (int *)(lds_all + BM_OFF_META) + 31;
        d_start = sweep_zoom_split<HALF>(g, fast, Pb, Pib, Tb, depths, SWEEP_BWD_ZOOM_FOUR, tid,
                                         2 * 64 * nw, slot);
        if (d_start >= g.D) return;
        __syncthreads();
        d_four = max(d_start, sweep_zoom_split<HALF>(g, fast, Pb, Pib, Tb, depths, SWEEP_BWD_ZOOM_TWO,
                                                     tid, 2 * 64 * nw, slot));
        __syncthreads();
        d_two = max(d_four, sweep_zoom_split<HALF>(g, fast, Pb, Pib, Tb, depths, SWEEP_BWD_ZOOM_ONE,
                                                   tid, 2 * 64 * nw, slot));
    }
    const int n4 = d_four - d_start, n2 = d_two - d_four, nsteps = 4 * n4 + 2 * n2 + (g.D - d_two);
    auto step_plane = [&](int t) -> int {
        return t < 4 * n4 ? d_start + (t >> 2) : t < 4 * n4 + 2 * n2 ? d_four + ((t - 4 * n4) >> 1) : d_two + (t - 4 * n4 - 2 * n2);
    };
    // which of a segment's four 8-point groups step t takes: bit i = group i
    auto step_groups = [&](int t) -> unsigned {
        return t < 4 * n4 ? 1u << (t & 3) : t < 4 * n4 + 2 * n2 ? 3u << (2 * ((t - 4 * n4) & 1)) : 15u;
    };

    // ---- footprints of the tile's 64 points (one per lane) for step t, two steps ahead of the image
    //      that needs them (ring of BM_RING steps); extents of the in-bounds taps next to them ------
    auto fill_step = [&](int t) {
        const int lane = opaque(tid) & 63, phh = lane >> 5, pk = lane & 31, ph = h0 + phh, pw = w0 + pk;
        const bool p_own = tile_live && ph >= own_h && pw >= own_w && ((step_groups(t) >> (pk >> 3)) & 1u);
        const int p = t % BM_RING;
        float sx, sy, fw = 0.0f, fn = 0.0f;
        uint32_t f = 0u;
        if (p_own) {
            sweep_point_map<HALF>(g, fast, Pb, Pib, Tb, depths[step_plane(t)], ph, pw, sx, sy);
            f = bwd_footprint(sx, sy, H, W, fw, fn);
        }
        tab_f[p * 64 + lane] = f;
        tab_w[p * 64 + lane] = fw;
        tab_n[p * 64 + lane] = fn;
        const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
        const bool wok = f & (1u << 27), eok = f & (1u << 28), nok = f & (1u << 29), sok = f & (1u << 30);
        const int big = 0x3fffffff;
        const int xlo = half_wave_reduce(f ? (wok ? ixw : ixw + 1) : big, OpMin());
        const int xhi = half_wave_reduce(f ? (eok ? ixw + 1 : ixw) : -big, OpMax());
        const int ylo = half_wave_reduce(f ? (nok ? iyn : iyn + 1) : big, OpMin());
        const int yhi = half_wave_reduce(f ? (sok ? iyn + 1 : iyn) : -big, OpMax());
        // lanes 31 / 63 hold the extents of lattice row 0 / 1
        if ((lane & 31) == 31) {
            int *e = ext + p * 8 + 4 * phh;
            e[0] = xlo;
            e[1] = xhi;
            e[2] = ylo;
            e[3] = yhi;
        }
    };

    // ---- the fragment image of step t from its footprints: one lattice point per lane ----------
    auto produce = [&](int t, int buf) {
        const int lane = opaque(tid) & 63, phh = lane >> 5, pk = lane & 31;
        unsigned short *img16 = (unsigned short *)(img + buf * BM_IMG);
        unsigned short *tl = taps + (buf * 64 + lane) * 4;
        const int p = t % BM_RING;
        const uint32_t f = tab_f[p * 64 + lane];
        const float fw = tab_w[p * 64 + lane], fn = tab_n[p * 64 + lane];
        const int *e = ext + p * 8;
        const int x0lo = e[0], x0hi = e[1], y0lo = e[2], y0hi = e[3], x1lo = e[4], x1hi = e[5], y1lo = e[6], y1hi = e[7];
        int have = meta[BM_M_STATE], xb = meta[BM_M_STATE + 1], yb = meta[BM_M_STATE + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off = tl[q];
            if (off != 0xffffu) { img16[off] = 0; img16[off + BM_IMG1 / 2] = 0; }
        }
        const int xmin = min(x0lo, x1lo), xmax = max(x0hi, x1hi), ymin = min(y0lo, y1lo), ymax = max(y0hi, y1hi);
        int flags = 0;
        if (xmax < xmin) {
            flags = BM_EMPTY;
        } else if (xmax - xmin + 1 > 16 * BM_NT || ymax - ymin + 1 > BM_R ||
                   (y0hi >= y0lo && y0hi - y0lo + 1 > BM_SLOTS) || (y1hi >= y1lo && y1hi - y1lo + 1 > BM_SLOTS)) {
            flags = BM_SLOW | (have ? BM_FLUSH : 0);  // a plane of its own between two windows
            have = 0;
        } else if (!have || xmin < xb || xmax >= xb + 16 * BM_NT || ymin < yb || ymax >= yb + BM_R) {
            // move the window; the slack goes to the side the footprints left it on
            const int sl_x = 16 * BM_NT - (xmax - xmin + 1), sl_y = BM_R - (ymax - ymin + 1);
            const int nxb = !have ? xmin - sl_x / 2 : xmax >= xb + 16 * BM_NT ? xmin : xmin < xb ? xmin - sl_x : xmin - sl_x / 2;
            const int nyb = !have ? ymin - sl_y / 2 : ymax >= yb + BM_R ? ymin : ymin < yb ? ymin - sl_y : ymin - sl_y / 2;
            flags = have ? BM_FLUSH : 0;
            xb = nxb;
            yb = nyb;
            have = 1;
        }
        // first image row of each lattice row, relative to the window (0 when the row has no taps)
        const int row0 = y0hi >= y0lo ? y0lo - yb : 0, row1 = y1hi >= y1lo ? y1lo - yb : 0;
        // this lane's four weights into the image; touched (window row, 16-column block) pairs of its
        // lattice row: bit r * BM_NT + nt
        unsigned bits = 0u;
        {
            const bool wok = f & (1u << 27), eok = f & (1u << 28), nok = f & (1u << 29), sok = f & (1u << 30);
            const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
            const bool on = !(flags & (BM_EMPTY | BM_SLOW)) && f != 0u;
            const float cwt = wok ? 1.0f - fw : 0.0f, cet = eok ? fw : 0.0f;
            const float rnt = nok ? 1.0f - fn : 0.0f, rst = sok ? fn : 0.0f;
            const float wq[4] = {rnt * cwt, rnt * cet, rst * cwt, rst * cet};
            const bool okq[4] = {wok && nok, eok && nok, wok && sok, eok && sok};
            const int rbase = yb + (phh ? row1 : row0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned off = 0xffffu;
                if (on && okq[q]) {
                    const int y = iyn + (q >> 1), n = ixw + (q & 1) - xb;
                    const int fr = (phh * BM_SLOTS + (y - rbase)) * BM_NT + (n >> 4);
                    off = (unsigned)(fr * (BM_FRAG / 2) + ((pk >> 3) * 16 + (n & 15)) * 8 + (pk & 7));
                    const bf16_t hi = f32_to_bf16(wq[q]);
                    img16[off] = hi;
                    img16[off + BM_IMG1 / 2] = f32_to_bf16(wq[q] - bf16_to_f32(hi));
                    bits |= 1u << ((y - yb) * BM_NT + (n >> 4));
                }
                tl[q] = (unsigned short)off;
            }
        }
        const int br = half_wave_reduce((int)bits, OpOr());  // a half-wave = one lattice row
        if ((lane & 31) == 31) meta[buf * 8 + BM_M_BITS + phh] = br;
        if (lane == 0) {
            meta[buf * 8 + BM_M_FLAGS] = flags;
            meta[buf * 8 + BM_M_XB] = xb;
            meta[buf * 8 + BM_M_YB] = yb;
            meta[buf * 8 + BM_M_ROW] = row0;
            meta[buf * 8 + BM_M_ROW + 1] = row1;
            meta[BM_M_STATE] = have;
            meta[BM_M_STATE + 1] = xb;
            meta[BM_M_STATE + 2] = yb;
        }
    };

    // ---- this wave's share of the gradient volume ---------------------------------------------
    // lane = (channel of the 16-channel block, 8-point group); the addresses are derived from an
    // opaque copy of the lane id where they are used: kept across the plane loop they cost the
    // registers the accumulators need
    const bf16_t *gbase = gout + (size_t)b * 2 * g.C * g.N + (size_t)HALF * g.C * g.N;
    auto a_addr = [&](int l_, int d, int mt, int hh) -> const bf16_t * {
        const int c = min(c0 + mt * 16 + (l_ & 15), g.C - 1);
        return gbase + (size_t)c * g.N + (size_t)d * hw + (size_t)((h0 + hh) * g.w_out + w0 + (l_ >> 4) * 8);
    };
    auto load_plane = [&](int d, ARaw (&raw)[BM_MT][BM_TH]) {
        if (BM_AB(1)) return;
        const int dd = min(d, g.D - 1), l_ = opaque(lane);
#pragma unroll
        for (int mt = 0; mt < BM_MT; ++mt)
#pragma unroll
            for (int hh = 0; hh < BM_TH; ++hh) raw[mt][hh] = a_load(a_addr(l_, dd, mt, hh));
    };

    f32x4_t acc[BM_MT][BM_R][BM_NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < BM_MT; ++mt)
#pragma unroll
            for (int r = 0; r < BM_R; ++r)
#pragma unroll
                for (int nt = 0; nt < BM_NT; ++nt) acc[mt][r][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // per-value path (plain float atomics, torch's semantics for Inf / NaN): steps [ta, tb) of this
    // wave's channels
    auto slow_steps = [&](int ta, int tb) {
        if (!wave_live) return;
        const int l_ = opaque(lane), ac = l_ & 15, akg = l_ >> 4;
        for (int t = ta; t < tb; ++t) {
            const int d = step_plane(t);
            if (!((step_groups(t) >> akg) & 1u)) continue;  // this lane's 8 points belong to another step
            const float depth = depths[d];
            for (int hh = 0; hh < BM_TH; ++hh)
                for (int j = 0; j < 8; ++j) {
                    const int h = h0 + hh, w = w0 + akg * 8 + j;
                    if (h < own_h || w < own_w) continue;
                    float sx, sy, fw, fn;
                    sweep_point_map<HALF>(g, fast, Pb, Pib, Tb, depth, h, w, sx, sy);
                    const uint32_t f = bwd_footprint(sx, sy, H, W, fw, fn);
                    if (!f) continue;
                    const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
                    const bool wok = f & (1u << 27), eok = f & (1u << 28), nok = f & (1u << 29), sok = f & (1u << 30);
                    const float cwt = 1.0f - fw, cet = fw, rnt = 1.0f - fn, rst = fn;
                    const float wq[4] = {rnt * cwt, rnt * cet, rst * cwt, rst * cet};
                    // every in-bounds tap gets weight * value, a zero weight included (ATen's
                    // grid_sampler_2d_backward: 0 * Inf = NaN reaches the tap)
                    const bool okq[4] = {wok && nok, eok && nok, wok && sok, eok && sok};
#pragma unroll
                    for (int mt = 0; mt < BM_MT; ++mt) {
                        const int c = c0 + mt * 16 + ac;
                        if (c >= g.C) continue;
                        const float gv = bf16_to_f32(gbase[(size_t)c * g.N + (size_t)d * hw + (size_t)h * g.w_out + w]);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (okq[q])
                                atomicAdd(gf + (size_t)c * HW + (size_t)(iyn + (q >> 1)) * W + ixw + (q & 1), gv * wq[q]);
                    }
                }
        }
    };

    unsigned touched = 0u;  // (row, column block) pairs written since the last flush
    int epoch_start = 0;  // first step accumulated since the last flush
    auto flush = [&](int xb, int yb, int t_end) {
        if (!wave_live || !touched) { touched = 0u; epoch_start = t_end; return; }
        bool bad = false;
#pragma unroll
        for (int mt = 0; mt < BM_MT; ++mt)
#pragma unroll
            for (int r = 0; r < BM_R; ++r)
#pragma unroll
                for (int nt = 0; nt < BM_NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bad |= (__float_as_uint(acc[mt][r][nt][j]) & 0x7f800000u) == 0x7f800000u;
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
            slow_steps(epoch_start, t_end);
        } else if (!BM_AB(4)) {
            const int l_ = opaque(lane), ac = l_ & 15, akg = l_ >> 4;
#pragma unroll
            for (int r = 0; r < BM_R; ++r)
#pragma unroll
                for (int nt = 0; nt < BM_NT; ++nt) {
                    if (!((touched >> (r * BM_NT + nt)) & 1u)) continue;
                    const int y = yb + r, x = xb + nt * 16 + ac;
                    const bool in = y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
                    for (int mt = 0; mt < BM_MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = c0 + mt * 16 + 4 * akg + j;
                            const float v = acc[mt][r][nt][j];
                            if (in && c < g.C && v != 0.0f) atomicAdd(gf + (size_t)c * HW + (size_t)y * W + x, v);
                        }
                }
        }
        zero_acc();
        touched = 0u;
        epoch_start = t_end;
    };

    // prologue: footprints of the first two steps, the first step's image and gradient words
    for (int t = wave; t < min(2, nsteps); t += nw) fill_step(t);
    ARaw raw[BM_MT][BM_TH];
#pragma unroll
    for (int mt = 0; mt < BM_MT; ++mt)
#pragma unroll
        for (int hh = 0; hh < BM_TH; ++hh) raw[mt][hh] = ARaw{u32x4_t{0u, 0u, 0u, 0u}, 0u};
    if (wave_live) load_plane(step_plane(0), raw);
    bm_barrier();
    if (wave == 0) produce(0, 0);
    bm_barrier();
    int xb = 0, yb = 0;
    for (int t = 0; t < nsteps; ++t) {
        const int buf = t & 1, d = step_plane(t);
        BM_STAMP(0);  // loop overhead
        // one wave: footprints of step t + 2; another: the image of step t + 1 (their registers come
        // and go before the consumer's fragments are live)
        if (t + 2 < nsteps && wave == ((t + 2) % nw) && !BM_AB(8)) fill_step(t + 2);
        BM_STAMP(8);  // footprints
        if (t + 1 < nsteps && wave == ((t + 1) % nw) && !BM_AB(8)) produce(t + 1, buf ^ 1);
        BM_STAMP(1);  // image
        // this step's A fragments out of the raw words, then the next step's loads into them
        bf16x8_t a[BM_MT][BM_TH];
        if (wave_live) {
            const int l_ = opaque(lane);
#pragma unroll
            for (int mt = 0; mt < BM_MT; ++mt)
#pragma unroll
                for (int hh = 0; hh < BM_TH; ++hh)
                    a[mt][hh] = a_frag(raw[mt][hh], ((uintptr_t)a_addr(l_, d, mt, hh) & 2) != 0,
                                       c0 + mt * 16 + (l_ & 15) >= g.C);
            load_plane(step_plane(min(t + 1, nsteps - 1)), raw);
        }
        BM_STAMP(2);  // wait for this step's gradient words, issue the next step's
        const int flags = __builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_FLAGS]);
        const int nxb = __builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_XB]);
        const int nyb = __builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_YB]);
        const unsigned bits0 = (unsigned)__builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_BITS]);
        const unsigned bits1 = (unsigned)__builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_BITS + 1]);
        const int row0 = __builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_ROW]);
        const int row1 = __builtin_amdgcn_readfirstlane(meta[buf * 8 + BM_M_ROW + 1]);
        BM_STAMP(3);  // meta
        if (flags & BM_FLUSH) flush(xb, yb, t);
        BM_STAMP(4);  // flush
        xb = nxb;
        yb = nyb;
        if (flags & BM_SLOW) {
            slow_steps(t, t + 1);
            epoch_start = t + 1;
        } else if (!(flags & BM_EMPTY) && wave_live && !BM_AB(2)) {
            const unsigned char *im = img + buf * BM_IMG + opaque(lane) * 16;
#pragma unroll
            for (int hh = 0; hh < BM_TH; ++hh) {
                const unsigned bits = hh ? bits1 : bits0;
                // image row slot of window row r: r - (first window row of this lattice row)
                const unsigned char *imh = im + (hh * BM_SLOTS - (hh ? row1 : row0)) * (BM_NT * BM_FRAG);
#pragma unroll
                for (int r = 0; r < BM_R; ++r) {
                    if (!((bits >> (r * BM_NT)) & 7u)) continue;
#pragma unroll
                    for (int term = 0; term < 2; ++term) {  // hi, then lo
                        bf16x8_t bfr[BM_NT];
#pragma unroll
                        for (int nt = 0; nt < BM_NT; ++nt)
                            bfr[nt] = *(const bf16x8_t *)(imh + term * BM_IMG1 + (r * BM_NT + nt) * BM_FRAG);
#pragma unroll
                        for (int nt = 0; nt < BM_NT; ++nt)
#pragma unroll
                            for (int mt = 0; mt < BM_MT; ++mt)
                                acc[mt][r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][hh], bfr[nt], acc[mt][r][nt], 0, 0, 0);
                    }
                }
            }
            touched |= bits0 | bits1;
        }
        BM_STAMP(5);  // fragments + MFMA
        bm_barrier();
        BM_STAMP(6);  // barrier
    }
    flush(xb, yb, nsteps);
    BM_STAMP(7);
#if defined(DFM_DEBUG_HOOKS) && !defined(DFM_BM_ABLATE)
    if (tg.trace && work == 0 && lane == 0 && sel == 0)
        for (int i = 0; i < 12; ++i) tg.trace[wave * 12 + i] = tsum[i];
#endif
}

}  // namespace

// does the matrix-product backward take this problem?  (dense sweep, bf16, a tile fits the lattice)
bool sweep_bwd_mfma_supported(const dfm_sweep_desc *d, const void *grad_out)
{
    return d->dtype == DFM_BF16 && d->cost_sample_factor == 1.0f && d->w_out >= BM_TW && d->h_out >= BM_TH &&
           ((uintptr_t)grad_out & 3) == 0 && d->h_in < 4096 && d->w_in < 8192 &&
           (long long)d->num_depths * d->h_out * d->w_out < (1ll << 31);
}

int sweep_bwd_mfma_launch(const dfm_sweep_desc *d, int half, const void *grad_out, const float *depths,
                          const float *P, const float *Pinv, const float *Tm, float *grad_cur, float *grad_prev,
                          void *stream)
{
    const SweepGeom g = sweep_make_geom(d);
    SweepFast fast = sweep_make_fast(d);
    BmGrid tg;
    tg.batch = d->batch;
    tg.tiles_w = (g.w_out + BM_TW - 1) / BM_TW;
    tg.pairs_w = (tg.tiles_w + 1) / 2;
    tg.tiles_h = (g.h_out + BM_TH - 1) / BM_TH;
    const int cgroups = (g.C + 16 * BM_MT - 1) / (16 * BM_MT);  // 32-channel groups = waves needed per tile
    tg.waves = std::min(BM_WAVES, cgroups);
    tg.cblocks = (cgroups + tg.waves - 1) / tg.waves;
    const long long total = (long long)tg.batch * tg.tiles_h * tg.pairs_w * tg.cblocks;
    if (total > (1ll << 30)) return set_error(DFM_ERR_UNSUPPORTED, "too many lattice tiles");
    tg.total = (int)total;
    tg.per_xcd = (tg.total + 7) / 8;
    tg.ablate = 0;
    tg.trace = nullptr;
#ifdef DFM_DEBUG_HOOKS
    if (const char *e = getenv("DFM_BWD_ABLATE")) tg.ablate = atoi(e) >> 4;  // bits 16, 32, 64, 128
    tg.trace = g_bm_trace ? g_bm_trace + half * 64 : nullptr;  // 4 waves x 12 phases (+ slack)
#endif
    const void *kern = half ? (const void *)sweep_bwd_mfma_kernel<1> : (const void *)sweep_bwd_mfma_kernel<0>;
    int rc = ensure_dynamic_lds(kern, BM_LDS);
    if (rc != DFM_OK) return rc;
    const dim3 grid((unsigned)(tg.per_xcd * 8)), block(2 * 64 * tg.waves);
    if (half)
        hipLaunchKernelGGL(sweep_bwd_mfma_kernel<1>, grid, block, BM_LDS, (hipStream_t)stream, g, fast, tg,
                           (const bf16_t *)grad_out, depths, P, Pinv, Tm, grad_cur, grad_prev);
    else
        hipLaunchKernelGGL(sweep_bwd_mfma_kernel<0>, grid, block, BM_LDS, (hipStream_t)stream, g, fast, tg,
                           (const bf16_t *)grad_out, depths, P, Pinv, Tm, grad_cur, grad_prev);
    if (hipGetLastError() != hipSuccess) return set_error(DFM_ERR_HIP, "sweep_bwd_mfma_kernel launch failed");
    return DFM_OK;
}

}  // namespace dfm
#ifdef DFM_DEBUG_HOOKS
extern "C" DFM_API void dfm_debug_set_bm_trace(void *buf) { dfm::g_bm_trace = (unsigned long long *)buf; }
#endif

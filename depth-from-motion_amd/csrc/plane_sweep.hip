// plane_sweep.hip -- gfx950 kernels + C ABI for build_dfm_cost
// (reference: mmdet3d/models/backbones/dfm_backbone.py:217-314).
//
// Data layout in HBM
//   cur/prev  : caller tensors (B, C, H, W), T in {f32, bf16}
//   workspace : the same two maps re-blocked to [B][nblk][H][W][CB] with
//               CB*sizeof(T) == 16 B, so ONE 16-byte load fetches CB channels
//               of a pixel and a row of pixels of one channel block is one
//               contiguous run (coalesced by a wave, linear for LDS staging).
//   out       : (B, 2C, D, h_out, w_out) written exactly once.
//
// Kernels
//   pack_blocked_kernel : NCHW -> blocked copy (reads+writes 2*|feats|, <1 %
//                         of the volume bytes).
//   sweep_gather_kernel : one lane = one lattice point; loops channel blocks,
//                         four 16-B taps per map straight from L2/L1.
//   sweep_bwd_tile_kernel: backward, gradients accumulated in fixed-point LDS rows.
//   sweep_bwd_kernel    : backward fallback (maps wider than the LDS rows), lane-per-point
//                         scatter-add with global atomics.
//   sweep_grid_kernel   : parity aid, dumps the normalised grids.
#include "dfm_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

using namespace dfm;

namespace {

thread_local char g_err[512] = "";
thread_local int g_last_kernel = 0;
std::atomic<int> g_last_bwd_kernel{0};  // process-wide: autograd runs the backward on its own thread
#ifdef DFM_DEBUG_HOOKS
unsigned long long *g_trace = nullptr;  // debug builds only: see dfm_debug_set_trace
#endif

// The library keeps no mutable launch state: the shape of a forward launch comes from the
// caller's dfm_sweep_opts (per call) or from the tuned-schedule cache below (per device and
// problem shape, filled by dfm_plane_sweep_autotune, guarded by a mutex).
struct Launch {
    int kernel;      // 0 auto, 1 gather, 2 LDS tiles, 3 direct tiles, 4 pixel-major taps + LDS transpose
    int lanes;       // lanes per workgroup of the tile kernels
    int lds_kib;     // dynamic LDS per workgroup
    int bpg;         // channel blocks per workgroup (all by default)
    int planes;      // depth planes per workgroup
    int band_chunk;  // adjacent bands scheduled back to back
    int ppl;         // lattice points per lane
    int align;       // tile boundaries: multiples of this many points (8, 16, 32 or 64)
    int pair;        // 4 points per lane: 16-byte stores through a lane-pair exchange (1) or 8-byte stores (0)
    int unpack;      // bf16 taps: 1 unpacked by the matrix core, 0 by VALU shifts
    int pipe;        // LDS tile kernel body: 1 serial (stage | barrier | blend + store | barrier),
                     // 2 pipelined (two buffers, one barrier per block, stores never waited for)
};

struct TuneKey {
    int v[9];
    bool operator<(const TuneKey &o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
std::mutex g_tune_mu;
std::map<TuneKey, dfm_sweep_opts> g_tuned;

// optional per-launch timing of the dominant (volume-writing) kernel with HIP
// events on the caller's stream (bench.py's roofline leg)
struct Profiler {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> ev;  // pairs
    int used = 0;
} g_prof;

int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
}  // namespace

int dfm::set_error(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of on
// every launch.  The attribute is per DEVICE: every translation unit of the library goes through
// this one (device, kernel) map, so a second GPU driven from the same process gets its own call.
int dfm::ensure_dynamic_lds(const void *kern, int lds_bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> seen;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(mu);
    int &have = seen[std::make_pair(dev, kern)];
    if (lds_bytes > have) {
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
        have = lds_bytes;
    }
    return DFM_OK;
}

namespace {

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(DFM_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------
// NCHW -> [B][nblk][H][W][CB]
// ---------------------------------------------------------------------------
// Each lane re-blocks PACK_PPL pixels (256 apart).  One pixel per lane measured fastest in
// isolation (0.083 ms for the N* maps, 5.8 TB/s read+write; 4 pixels per lane: 0.20 ms --
// tools/pack_microbench.hip, profiles/archive/r02_c3_pack_microbench.txt).  Inside the pipeline its
// rocprofv3 duration is ~0.19 ms either way: it starts while the previous launch's 26.8 GB of
// volume writes are still draining to HBM.
constexpr int PACK_PPL = 1;
// flags[(map * batch + sample) * H + row] (zeroed by sweep_reset_kernel before the launch) gets bit 0 when a
// bf16 feature value of that map ROW (any channel) is not a finite, normal number or +0: Inf / NaN, a denormal,
// or -0.  A tile whose staged rows include a flagged row unpacks with VALU shifts instead of the matrix core
// (see compute_store): a product 0 x Inf inside the selecting MFMA would be NaN for the block's other seven
// channels, and the matrix core's sum of zeros loses the sign of -0 (and may flush a denormal) -- values a
// trained network rarely produces, but the results stay the reference's bit for bit.  (Rounds 3-4 kept ONE
// flag per launch: a single -0 anywhere in the batch sent every tile of the launch to the second-chance pass.)
// the two spill counters and the pack pass's row flags (n_flags words)
__global__ void sweep_reset_kernel(int *flags, int n_flags, int *b, int *c)
{
    for (int i = threadIdx.x; i < n_flags; i += blockDim.x) flags[i] = 0;
    if (threadIdx.x == 0) { *b = 0; *c = 0; }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_blocked_kernel(const T *__restrict__ src0,
                                                           const T *__restrict__ src1,
                                                           uint4 *__restrict__ dst0,
                                                           uint4 *__restrict__ dst1, int batch,
                                                           int C, int HW, int nblk, int *__restrict__ flags,
                                                           int W)
{
    constexpr int CB = elem<T>::CB;
    const int pix0 = blockIdx.x * (256 * PACK_PPL) + threadIdx.x;
    const int blk = blockIdx.y;
    const bool second = (int)blockIdx.z >= batch;  // z = [cur samples | prev samples]
    const int b = second ? blockIdx.z - batch : blockIdx.z;
    const T *__restrict__ src = (second ? src1 : src0) + ((size_t)b * C + (size_t)blk * CB) * HW;
    uint4 *__restrict__ dst = (second ? dst1 : dst0) + ((size_t)b * nblk + blk) * HW;
    T v[PACK_PPL][CB];
#pragma unroll
    for (int i = 0; i < PACK_PPL; ++i) {
        const int pix = pix0 + i * 256;
#pragma unroll
        for (int j = 0; j < CB; ++j)
            v[i][j] = (blk * CB + j < C && pix < HW) ? src[(size_t)j * HW + pix] : T(0);
    }
#pragma unroll
    for (int i = 0; i < PACK_PPL; ++i) {
        const int pix = pix0 + i * 256;
        if (pix < HW) {
            uint4 q;
            memcpy(&q, v[i], 16);
            dst[pix] = q;
            if constexpr (sizeof(T) == 2) {
                bool odd_value = false;
                const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int hbit = 0; hbit < 32; hbit += 16) {
                        const uint32_t e = (w4[k] >> hbit) & 0xffffu, ex = e & 0x7f80u;
                        odd_value |= ex == 0x7f80u || (ex == 0u && e != 0u);
                    }
                // (rare: a plain load first, so that a map full of -0 does not queue 10^6 atomics on one word)
                if (flags && odd_value) {
                    int *f = flags + (size_t)blockIdx.z * (HW / W) + pix / W;
                    if (__builtin_nontemporal_load(f) == 0) atomicOr(f, 1);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// bilinear blend of four 16-byte channel blocks, ATen order:
// fma(se_v, se, fma(sw_v, sw, fma(ne_v, ne, nw_v * nw)))
// ---------------------------------------------------------------------------
template <int CB>
__device__ __forceinline__ void blend(const Tap &t, const uint4 &qnw, const uint4 &qne,
                                      const uint4 &qsw, const uint4 &qse, float (&r)[CB])
{
    float a[CB], b[CB], c[CB], d[CB];
    unpack16(qnw, a);
    unpack16(qne, b);
    unpack16(qsw, c);
    unpack16(qse, d);
    const bool k0 = t.ok & 1u, k1 = t.ok & 2u, k2 = t.ok & 4u, k3 = t.ok & 8u;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const float vnw = k0 ? a[j] : 0.0f;
        const float vne = k1 ? b[j] : 0.0f;
        const float vsw = k2 ? c[j] : 0.0f;
        const float vse = k3 ? d[j] : 0.0f;
        float acc = vnw * t.nw;
        acc = __builtin_fmaf(vne, t.ne, acc);
        acc = __builtin_fmaf(vsw, t.sw, acc);
        acc = __builtin_fmaf(vse, t.se, acc);
        r[j] = acc;
    }
}


template <int CB>
__device__ __forceinline__ void blend_nomask(const Tap &t, const uint4 &qnw, const uint4 &qne,
                                             const uint4 &qsw, const uint4 &qse, float (&r)[CB])
{
    float a[CB], b[CB], c[CB], d[CB];
    unpack16(qnw, a);
    unpack16(qne, b);
    unpack16(qsw, c);
    unpack16(qse, d);
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        float acc = a[j] * t.nw;
        acc = __builtin_fmaf(b[j], t.ne, acc);
        acc = __builtin_fmaf(c[j], t.sw, acc);
        acc = __builtin_fmaf(d[j], t.se, acc);
        r[j] = acc;
    }
}

// ---------------------------------------------------------------------------
// direct-gather forward: grid (ceil(N/256), B)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sweep_gather_kernel(
    SweepGeom g, const uint4 *__restrict__ cur_blk, const uint4 *__restrict__ prev_blk,
    const float *__restrict__ depths, const float *__restrict__ P, const float *__restrict__ Pinv,
    const float *__restrict__ Tm, T *__restrict__ out)
{
    constexpr int CB = elem<T>::CB;
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= g.N) return;
    const int hw = g.h_out * g.w_out;
    const int d = (int)(n / hw);
    const int rem = (int)(n - (long long)d * hw);
    const int hi = rem / g.w_out;
    const int wi = rem - hi * g.w_out;

    float cx, cy, px, py;
    sweep_point(g, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, cx, cy, px, py,
                nullptr);
    const Tap tc = make_tap(cx, cy, g.h_in, g.w_in);
    const Tap tp = make_tap(px, py, g.h_in, g.w_in);

    const int HW = g.h_in * g.w_in;
    const int c00 = tc.iy * g.w_in + tc.ix, c01 = c00 + tc.dx;
    const int c10 = c00 + tc.dy * g.w_in, c11 = c10 + tc.dx;
    const int p00 = tp.iy * g.w_in + tp.ix, p01 = p00 + tp.dx;
    const int p10 = p00 + tp.dy * g.w_in, p11 = p10 + tp.dx;

    const uint4 *cb = cur_blk + (size_t)b * g.nblk * HW;
    const uint4 *pb = prev_blk + (size_t)b * g.nblk * HW;
    T *ocur = out + ((size_t)b * 2 * g.C) * g.N + n;
    T *oprev = ocur + (size_t)g.C * g.N;

    for (int blk = 0; blk < g.nblk; ++blk) {
        const uint4 qc0 = cb[c00], qc1 = cb[c01], qc2 = cb[c10], qc3 = cb[c11];
        const uint4 qp0 = pb[p00], qp1 = pb[p01], qp2 = pb[p10], qp3 = pb[p11];
        float rc[CB], rp[CB];
        blend<CB>(tc, qc0, qc1, qc2, qc3, rc);
        blend<CB>(tp, qp0, qp1, qp2, qp3, rp);
        const int cbase = blk * CB;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            if (cbase + j < g.C) {
                ocur[(size_t)(cbase + j) * g.N] = elem<T>::store(rc[j]);
                oprev[(size_t)(cbase + j) * g.N] = elem<T>::store(rp[j]);
            }
        }
        cb += HW;
        pb += HW;
    }
}

// ---------------------------------------------------------------------------
// LDS-staged forward.
//
// One workgroup = NT lanes = (a tile of NT*V consecutive lattice points of one
// sample) x (ONE of the two maps: cur or prev).  V = 16 B / sizeof(T): the
// points whose values of one channel make one aligned 16-byte store.
// Per lane: the V bilinear footprints are computed ONCE (coordinates do not
// depend on the channel) and kept in registers; then for every 16-byte
// channel block
//   1. the workgroup DMAs (global_load_lds, 16 B/lane, no VGPR round trip) the
//      rows of the map its tile touches -- full rows, one contiguous run of
//      the blocked layout -- into LDS, XOR-swizzled at 16-byte granularity,
//   2. each lane blends its V points x CB channels from LDS taps
//      (ds_read_b128: one tap = CB channels of one pixel),
//   3. and writes, per channel, ONE aligned 16-byte vector (V points): a wave
//      stores 1 KiB contiguous per channel plane (non-temporal).
// HBM traffic: the volume is written once; feature rows are re-read from
// L2/MALL (block id % batch == sample keeps a sample's maps on one XCD's L2
// when batch == 8).
// LDS swizzle: lanes of a wave read pixels 8 apart (V = 8 points per lane), a
// 128-byte stride that would put a 16-lane ds_read_b128 group on two bank
// quads (8-way conflict).  slot(q) = q ^ ((q >> 4) & 7) spreads them over all
// 16 quads; the DMA writes LDS linearly, so the swizzle is applied to the
// SOURCE pixel each lane fetches (an involution) and again on the tap reads.
// If the rows of a tile do not fit the LDS budget (extreme poses) the tile is
// flagged and sweep_spill_kernel redoes it with direct global taps.
// ---------------------------------------------------------------------------
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// How the volume's 16-byte vectors leave the CU.  Default: non-temporal (global_store_dwordx4 ... nt).
// -DDFM_STORE_FLAVOUR=1 plain, 2 sc1, 3 sc0 sc1 (write-through), 4 sc1 nt: experiment builds
// (build.build_variant), measured in profiles/archive/r04_c6_*.
#ifndef DFM_STORE_FLAVOUR
#define DFM_STORE_FLAVOUR 0
#endif
__device__ __forceinline__ void vol_store16(const u32x4_t &v, void *p)
{
#if DFM_STORE_FLAVOUR == 0
    __builtin_nontemporal_store(v, (u32x4_t *)p);
#elif DFM_STORE_FLAVOUR == 1
    *(u32x4_t *)p = v;
#elif DFM_STORE_FLAVOUR == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif DFM_STORE_FLAVOUR == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
#endif
}

// s_waitcnt immediate (gfx9 encoding) that waits only on vmcnt <= n
__device__ __forceinline__ constexpr int waitcnt_vm(int n)
{
    return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14);
}

__device__ __forceinline__ int swz(int q) { return q ^ ((q >> 4) & 7); }

// footprint of one point in one map: rows (clamped into the map), west column
// in [-1, W-1] (east = west + 1), fractions and the 4 in-bounds bits
__device__ __forceinline__ uint32_t footprint(float x, float y, int H, int W, int &rN, int &rS,
                                              int &ix, float &fw, float &fn)
{
    const Tap t = make_tap(x, y, H, W);
    ix = (t.ok & 5u) ? t.ix : t.ix - ((t.ok & 10u) ? 1 : 0);
    rN = t.iy;
    rS = t.iy + t.dy;  // a lone south tap (yn == -1): rN == rS == 0, north masked
    // (non-finite coordinates: fractions 0, so the weights are 1, 0, 0, 0 on four masked -- zero -- taps: the
    // point is 0, not NaN x 0; see make_tap)
    const bool fin = (fabsf(x) <= 3.0e38f) && (fabsf(y) <= 3.0e38f);
    fw = fin ? x - floorf(x) : 0.0f;
    fn = fin ? y - floorf(y) : 0.0f;
    return t.ok;
}

struct TileGrid {
    int batch;
    int bands;             // tiles per depth plane
    int band_pts;          // points per tile and plane (multiple of 8, <= NT*V/planes)
    int planes;            // depth planes per workgroup (divides NT/64)
    int dgroups;           // ceil(D / planes)
    int blocks_per_group;  // channel blocks one workgroup sweeps
    int band_chunk;        // adjacent bands scheduled together (see the block id map)
    const int *flags;      // flags[(map * batch + b) * H + row] != 0: that row of that map holds a non-finite /
                           // denormal / -0 bf16 value (pack_blocked_kernel)
    int mfma_unpack;       // bf16: unpack the taps with the matrix core (when flags allow) instead of VALU shifts
    int pair_stores;       // 4 points per lane (bf16): lane pairs trade halves and store 16-byte vectors
    int align;             // tile boundaries are multiples of this many points of the flat (d,h,w) index
                           // (a power of two >= 8): 32 points = 64 bytes of bf16, a whole memory-side write
    unsigned long long *trace;  // perf experiments only: per-phase s_memtime stamps
    int ablate;            // perf experiments only (DFM_ABLATE): 1 no staging,
                           // 2 no volume stores, 4 no taps/blend; results are wrong
};

// LDS == true : the staged kernel described above.
// LDS == false: same lane/point/store structure, taps straight from the blocked
//               map in global memory; runs only the tiles flagged by the LDS
//               kernel (spill list) -- or, as sweep_tile_kernel<.., false>, every tile.
// buffer stride of the pipelined (PIPE) body, in 16-byte slots: two buffers + the 8 scratch slots
// make 78.4 KiB, so two workgroups share a CU's 160 KiB with room for the allocation granule; the
// stride is a compile-time constant
// because the second buffer is addressed through the 16-bit immediate offset of ds_read_b128
constexpr int PIPE_BUF_SLOTS = 2504;  // 8 rows of 311 pixels + pad
constexpr int PIPE_LDS_BYTES = (8 + 2 * PIPE_BUF_SLOTS) * 16;
// experiments (build_variant): how the matrix-core unpack is written
#ifndef DFM_MX_MODE
#define DFM_MX_MODE 1   // 1 products and chain steps in a pinned order, 2 left to hipcc
#endif
template <int N> struct IntC { static constexpr int value = N; };

template <typename T, int NT, bool LDS, int V, bool PIPE = false, bool MXK = false>
__device__ __forceinline__ int tile_body(
    const int bid, const SweepGeom &g, const SweepFast &fast, const TileGrid &tg, int lds_slots,
    const uint4 *__restrict__ cur_blk, const uint4 *__restrict__ prev_blk,
    const float *__restrict__ depths, const float *__restrict__ P,
    const float *__restrict__ Pinv, const float *__restrict__ Tm, T *__restrict__ out,
    int *__restrict__ spill_list, int blk_lo_ovr = -1, int blk_hi_ovr = -1, bool retry_inline = false)
{
    // returns (workgroup-uniform), when retry_inline is set: 1 = the tile's rows exceed THIS body's LDS budget
    // (the caller runs the serial body over the same dynamic LDS as ONE buffer, then direct taps), 2 = a staged
    // row holds a value the matrix-core unpack is not exact for (the caller runs the VALU body); 0 otherwise
    constexpr int CB = elem<T>::CB;
    // V = points per lane.  V == CB: one 16-byte store per channel (the default).  V == 4 with
    // bf16: 8-byte stores, half the footprint / output registers per lane, so twice as many
    // waves fit a CU for the same tile (the wave's run per channel is 512 B instead of 1 KiB).
    static_assert(V == 4 || V == 8, "points per lane");
    static_assert(V * sizeof(T) == 16 || V * sizeof(T) == 8, "8- or 16-byte stores");
    static_assert(LDS || !PIPE, "the pipelined body is an LDS-staged body");
    static_assert(!MXK || (LDS && sizeof(T) == 2 && V == 8), "matrix-core unpack: bf16, LDS taps, 8 points per lane");
    constexpr int VW = V * (int)sizeof(T) / 4;  // dwords per channel vector
    constexpr int PAD = 8; // slots in front of the rows (keeps q = p + PAD >= 7)
    constexpr int SLAB = 8;  // slab starts at a multiple of 8 so the swizzle stays inside it
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    int *bb = (int *)lds;  // slot 0: {ymin, ymax}

    const int tid = threadIdx.x;
    // Tiles: every depth plane (hw points) is cut into `bands` pieces of
    // `band_pts` points whose boundaries are multiples of 8 in the flat
    // (d,h,w) index, so every lane's V points form one aligned 16-byte store.
    // block id = ((((group*chunks + chunk)*2 + half)*dgroups + dgroup)*band_chunk + band_in)*batch + b
    //   band = chunk*band_chunk + band_in; band_chunk = 1 by default
    //   sample fastest -> id % 8 == XCD keeps one sample's maps in one L2 (batch 8)
    //   depth next     -> the workgroups resident on an XCD sweep the SAME band
    //                     of the SAME map over consecutive depth planes: they
    //                     stage the same (cur) or neighbouring (prev) feature
    //                     rows, which therefore stay in that XCD's L2
    // A workgroup = `planes` consecutive depth planes x one band: wave w sweeps
    // plane (w / waves_per_plane).  All of them sample the same band of the map,
    // so one staged slab serves `planes` times as many volume bytes (the cur
    // rows are identical for every plane, the prev rows shift slowly with depth).
    const int batch = tg.batch;
    const int b = bid % batch;
    int th = bid / batch;
    const int band_in = th % tg.band_chunk;
    th /= tg.band_chunk;
    const int dgroup = th % tg.dgroups;
    th /= tg.dgroups;
    const int half = th & 1;
    th >>= 1;
    const int nchunks = (tg.bands + tg.band_chunk - 1) / tg.band_chunk;
    const int band = (th % nchunks) * tg.band_chunk + band_in;
    const int group = th / nchunks;
    if (band >= tg.bands) return 0;
    // (the second-chance pass hands a flagged tile to several workgroups, each with its own
    // range of channel blocks)
    const int blk_lo = blk_lo_ovr >= 0 ? blk_lo_ovr : group * tg.blocks_per_group;
    const int blk_hi = blk_lo_ovr >= 0 ? min(blk_hi_ovr, g.nblk) : min(blk_lo + tg.blocks_per_group, g.nblk);
    const int lanes_per_plane = NT / tg.planes;
    const int d_tile = dgroup * tg.planes + tid / lanes_per_plane;
    const int tid_p = tid % lanes_per_plane;
    const long long hw_ll = (long long)g.h_out * g.w_out;
    // A plane's tiles cover [a0, a1): from the 16-byte vector that holds the plane's first point to
    // the one that holds the next plane's.  The cuts BETWEEN the bands of a plane are multiples of
    // tg.align points of the flat index (band_pts is one), so that two workgroups never share a
    // memory-side write: with 16-byte-aligned cuts every band boundary of every channel plane left
    // two partial 64-byte writes behind, and the store stream alone ran 27 % slower
    // (profiles/archive/r04_c2_*: 5.52 -> 4.35 ms with nothing but the stores left in the kernel).
    const long long a0 = (d_tile * hw_ll) & ~7ll;
    const long long a1 = d_tile >= g.D - 1 ? g.N : (((d_tile + 1) * hw_ll) & ~7ll);
    const long long base = a0 & ~(long long)(tg.align - 1);
    const long long t_end = d_tile < g.D ? min(base + (long long)(band + 1) * tg.band_pts, a1) : 0;
    const long long n0 = base + (long long)band * tg.band_pts + (long long)tid_p * V;
    const bool active = n0 >= a0 && n0 < t_end;
    const int W = g.w_in, H = g.h_in;
    const int HW = H * W;

    if (LDS) {
        // slot 0: bbox scratch
        if (tid == 0) {
            bb[0] = 0x7fffffff;
            bb[1] = -1;
            // zero corner of each buffer: slot 0 of the slab (q < PAD is never a pixel, and the
            // swizzle keeps slots 0..7 among themselves); the DMA never writes it
            lds[SLAB] = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PIPE) lds[SLAB + PIPE_BUF_SLOTS] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
    }

    // ---- per-lane footprints: pixel index relative to row 0 of the map ----------
    int qN[V], qS[V];
    float fw[V], fn[V];
    uint32_t okbits = 0;  // 4 bits per point
    int ymin = 0x7fffffff, ymax = -1;
    if (active) {
        const int hw = g.h_out * g.w_out;
        int d = (int)(n0 / hw);
        int rem = (int)(n0 - (long long)d * hw);
        int hi = rem / g.w_out;
        int wi = rem - hi * g.w_out;
        const float *Pb = P + b * 16, *Pib = Pinv + b * 16, *Tb = Tm + b * 16;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float sx, sy;
            if (half) sweep_point_map<1>(g, fast, Pb, Pib, Tb, depths[d], hi, wi, sx, sy);
            else sweep_point_map<0>(g, fast, Pb, Pib, Tb, depths[d], hi, wi, sx, sy);
            int rN, rS, ix;
            uint32_t ok = footprint(sx, sy, H, W, rN, rS, ix, fw[j], fn[j]);
            // A plane's first tile starts at a multiple of 8 in the flat index, so its first vector can
            // hold up to 7 points of the PREVIOUS depth plane (last image row) next to
            // points of the first rows of this one: staging both would take the whole
            // map.  The LDS pass writes zeros there; sweep_patch_kernel fills them in.
            if (LDS && d != d_tile) ok = 0;
            qN[j] = rN * W + ix;
            qS[j] = rS * W + ix;
            okbits |= ok << (4 * j);
            if (ok) { ymin = min(ymin, rN); ymax = max(ymax, rS); }
            if (++wi == g.w_out) { wi = 0; if (++hi == g.h_out) { hi = 0; ++d; } }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    T *o = out + ((size_t)b * 2 * g.C + (size_t)half * g.C) * g.N + n0;
    const uint32_t full = (V == 8) ? 0xffffffffu : 0xffffu;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const bool all_in = __all(!active || okbits == full);

    int cnt = 0, nslots = 0;
    int ta[V][4];  // LDS path: byte address of the nw / ne / sw / se corner
    const uint4 *src = (half ? prev_blk : cur_blk) + ((size_t)b * g.nblk + blk_lo) * HW;
    if (LDS) {
        // workgroup bounding rows
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            ymin = min(ymin, __shfl_xor(ymin, s));
            ymax = max(ymax, __shfl_xor(ymax, s));
        }
        if ((tid & 63) == 0) { atomicMin(&bb[0], ymin); atomicMax(&bb[1], ymax); }
        __syncthreads();
        const int y0 = bb[0], y1 = bb[1];
        if (y1 < y0) {
            // no point of this tile lands inside the map: the volume is zero here
            if (active) {
                for (int c = blk_lo * CB; c < min(blk_hi * CB, g.C); ++c) {
                    if constexpr (VW == 4) {
                        const u32x4_t z = {0u, 0u, 0u, 0u};
                        __builtin_nontemporal_store(z, (u32x4_t *)(o + (size_t)c * g.N));
                    } else {
                        const u32x2_t z = {0u, 0u};
                        __builtin_nontemporal_store(z, (u32x2_t *)(o + (size_t)c * g.N));
                    }
                }
            }
            return 0;
        }
        cnt = (y1 - y0 + 1) * W;  // pixels (16-B slots) to stage per block
        nslots = (PAD + cnt + 1 + 7) & ~7;
        const bool fits = PIPE ? nslots <= min(PIPE_BUF_SLOTS, (lds_slots - SLAB) / 2)
                               : SLAB + nslots <= lds_slots;
        if (!fits) {
            // rows beyond the LDS budget: the caller retries in place with a larger one, or the tile is
            // queued for the next pass (once per tile)
            if (retry_inline) return 1;
            if (tid == 0 && blk_lo_ovr <= 0) spill_list[1 + atomicAdd(&spill_list[0], 1)] = bid;
            return 0;
        }
        if constexpr (MXK) {
            // a staged row with a non-finite / denormal / -0 value (pack_blocked_kernel's row flags): this body
            // has no VALU unpack -- the caller runs the one that has (same rows for every wave: uniform)
            // (`fits` bounds staged PIXELS, not rows: a narrow map, W <= 38, stages more than 64 rows in one
            //  pipelined buffer -- every row is looked at, 64 per trip)
            const int nrows = y1 - y0 + 1;
            const int *rf = tg.flags + ((size_t)half * batch + b) * H + y0;
            bool odd_row = false;
            for (int r = tid & 63; r < nrows; r += 64) odd_row |= rf[r] != 0;
            if (__any(odd_row)) {
                if (retry_inline) return 2;
                if (tid == 0 && blk_lo_ovr <= 0) spill_list[1 + atomicAdd(&spill_list[0], 1)] = bid;
                return 0;
            }
        }
        // LDS byte address of each corner: swizzled slot of pixel q (= index in
        // the staged rows + PAD), or the zero slot for an out-of-bounds corner.
        // Nothing in the channel loop depends on the in-bounds bits any more.
        const int off = PAD - y0 * W;
        // zero corner: slab slot 0, zeroed with the scratch slot above
        constexpr int ZERO = SLAB << 4;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const uint32_t ok = (okbits >> (4 * j)) & 15u;
            const int a = qN[j] + off, c = qS[j] + off;
            ta[j][0] = (ok & 1u) ? (SLAB + swz(a)) << 4 : ZERO;
            ta[j][1] = (ok & 2u) ? (SLAB + swz(a + 1)) << 4 : ZERO;
            ta[j][2] = (ok & 4u) ? (SLAB + swz(c)) << 4 : ZERO;
            ta[j][3] = (ok & 8u) ? (SLAB + swz(c + 1)) << 4 : ZERO;
        }
        src += (size_t)y0 * W;
    } else {
        if (!active) return 0;
        // direct taps: keep every index inside the map (masked taps are discarded)
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const bool any = (okbits >> (4 * j)) & 15u;
            qN[j] = any ? qN[j] : 0;
            qS[j] = any ? qS[j] : 0;
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
    // the selecting A operand of the matrix-core unpack (see compute_store): in each 4-lane block of
    // v_mfma_f32_4x4x4_16b_bf16, lane i holds row i of the 4x4 identity
    typedef __bf16 mxs_bf16x4 __attribute__((ext_vector_type(4)));
    mxs_bf16x4 mx_sel;
    if constexpr (MXK) {
        const int i = lane & 3;
        const uint32_t s2[2] = {i == 0 ? 0x3f80u : i == 1 ? 0x3f800000u : 0u,
                                i == 2 ? 0x3f80u : i == 3 ? 0x3f800000u : 0u};
        __builtin_memcpy(&mx_sel, s2, 8);
    }
    // debug trace: wave 0 of every 509th workgroup stamps its phases
#ifdef DFM_DEBUG_HOOKS
    unsigned long long *tr = (LDS && tg.trace && (bid % 509) == 0 && tid == 0)
                                 ? tg.trace + (size_t)(bid / 509) * 64 : nullptr;
    int tri = 0;
#endif
#ifndef DFM_DEBUG_HOOKS
#define TRACE_STAMP() do {} while (0)
#define ABLATE(bit) false
#else
#define ABLATE(bit) ((tg.ablate & (bit)) != 0)
#define TRACE_STAMP()                                                                       \
    do {                                                                                    \
        if (tr && tri < 64) {                                                               \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
            tr[tri++] = t_;                                                                 \
        }                                                                                   \
    } while (0)
#endif
    TRACE_STAMP();

    // blend the V points x CB channels of one channel block and store them
    // (bufc: which LDS buffer the taps come from -- the pipelined body's second buffer is reached
    // through the instruction's immediate offset, so the tap addresses never change)
    // mxc (bf16, LDS taps): unpack with the matrix core.  v_mfma_f32_4x4x4_16b_bf16 multiplies, in each of its
    // 16 four-lane blocks, a 4x4 A (lane i: row i) by a 4x4 B (lane i: column i = four bf16 of ITS OWN tap) and
    // leaves column i of the product in lane i: against the identity that is the lane's four channels in
    // fp32 (1.0 x v is exact, the other three terms are +0).  Two such products replace a tap's eight VALU
    // unpacks (v_and_b32 2.5 clocks, v_lshlrev_b32 4.0: 26 of the ~54 clocks a point's tap costs this
    // wave, profiles/archive/r04_c6_valu_microbench.txt) on a pipe this kernel does not use otherwise, 8 clocks each.
    // Exact for finite normal values and +0 -- pack_blocked_kernel flags anything else (0 x Inf would be
    // NaN for the lane's other channels; -0 and denormals) and the VALU path runs.
    // pk: per channel, one 16- (or 8-) byte vector of V points
    auto compute_block = [&](int blk, const uint4 *gsrc, auto bufc, auto mxc, uint32_t (&pk)[CB][VW]) {
        constexpr int BOFF = decltype(bufc)::value * PIPE_BUF_SLOTS * 16;
        constexpr bool MX = decltype(mxc)::value != 0 && MXK;
        static_assert(BOFF < 65536, "ds_read_b128 immediate offset");
        if (ABLATE(4)) {
#pragma unroll
            for (int k = 0; k < CB; ++k)
#pragma unroll
                for (int j = 0; j < VW; ++j) pk[k][j] = (uint32_t)qN[j] + k + blk;
        } else if constexpr (MX) {
            typedef __bf16 mx_bf16x4 __attribute__((ext_vector_type(4)));
            typedef float mx_f32x4 __attribute__((ext_vector_type(4)));
#define DFM_TAP_READ(dst, addr) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(BOFF))
            const mx_f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            // one tap -> its CB channels in fp32 (two 4x4x4 products, 4 channels each)
            auto cvt = [&](const u32x4_t &qq, float (&f)[CB]) {
                mx_bf16x4 b0, b1;
                const uint32_t h0[2] = {qq.x, qq.y}, h1[2] = {qq.z, qq.w};
                __builtin_memcpy(&b0, h0, 8);
                __builtin_memcpy(&b1, h1, 8);
                const mx_f32x4 lo = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(mx_sel, b0, z4, 0, 0, 0);
                f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
                if constexpr (CB == 8) {  // (the branch only runs for bf16: CB == 8)
                    const mx_f32x4 hi = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(mx_sel, b1, z4, 0, 0, 0);
                    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
                }
            };
            (void)cvt;
            // half a tap (4 channels): ONE 4x4x4 product
            auto cvt_half = [&](uint32_t lo, uint32_t hi, float *f) {
                mx_bf16x4 bfrag;
                const uint32_t h[2] = {lo, hi};
                __builtin_memcpy(&bfrag, h, 8);
                const mx_f32x4 d4 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(mx_sel, bfrag, z4, 0, 0, 0);
                f[0] = d4[0]; f[1] = d4[1]; f[2] = d4[2]; f[3] = d4[3];
            };
            // reads run one point ahead of the conversions (LDS returns in order: lgkmcnt(4) == "point j has
            // landed"); a point's taps are converted and folded into the chain one at a time, so only two
            // taps' fp32 channels are live beside the accumulators
            u32x4_t q[2][4];
            float keep[CB];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) DFM_TAP_READ(q[0][c4], ta[0][c4]);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int cb = j & 1, nb = cb ^ 1;
                if (j + 1 < V) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) DFM_TAP_READ(q[nb][c4], ta[j + 1][c4]);
                    asm volatile("s_waitcnt lgkmcnt(4)"
                                 : "+v"(q[cb][0]), "+v"(q[cb][1]), "+v"(q[cb][2]), "+v"(q[cb][3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(q[cb][0]), "+v"(q[cb][1]), "+v"(q[cb][2]), "+v"(q[cb][3]));
                }
                // (opaque copies: hipcc otherwise hoists the four weights of all 8 points out of the channel-
                // block loop -- 32 registers this body does not have)
                float w = fw[j], n = fn[j];
                asm volatile("" : "+v"(w), "+v"(n));
                // One product, then four VALU operations, strictly alternating, each product a whole tap ahead
                // of the chain step that consumes it: two 4x4x4 products back to back hold the wave's issue for
                // the matrix pipe (8 clocks), and products placed right in front of their consumers wait out
                // the latency.  hipcc's own placement varies from build to build between 5.16 and 5.7 ms
                // (profiles/archive/r04_c21/c22/c23), so the order is pinned with scheduling fences.
#if DFM_MX_MODE == 2
                const float e = 1.0f - w, s2 = 1.0f - n;
                const float wt[4] = {s2 * e, s2 * w, n * e, n * w};  // nw ne sw se: blend_nomask's chain
                float r[CB];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    float f[CB];
                    cvt(q[cb][c4], f);
#pragma unroll
                    for (int k = 0; k < CB; ++k) r[k] = c4 == 0 ? f[k] * wt[0] : __builtin_fmaf(f[k], wt[c4], r[k]);
                }
#else
                float fa[CB], fb[CB], r[CB];
#define DFM_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DFM_CHAIN(dst, src, wgt, lo, first)                                                          \
    _Pragma("unroll") for (int k = lo; k < lo + 4; ++k) dst[k] = first ? src[k] * wgt : __builtin_fmaf(src[k], wgt, dst[k])
                cvt_half(q[cb][0].x, q[cb][0].y, &fa[0]);
                DFM_FENCE();
                const float e = 1.0f - w, s2 = 1.0f - n;
                DFM_FENCE();
                cvt_half(q[cb][0].z, q[cb][0].w, &fa[4]);
                DFM_FENCE();
                const float wt[4] = {s2 * e, s2 * w, n * e, n * w};  // nw ne sw se: blend_nomask's chain
                DFM_FENCE();
                cvt_half(q[cb][1].x, q[cb][1].y, &fb[0]);
                DFM_FENCE();
                DFM_CHAIN(r, fa, wt[0], 0, true);
                DFM_FENCE();
                cvt_half(q[cb][1].z, q[cb][1].w, &fb[4]);
                DFM_FENCE();
                DFM_CHAIN(r, fa, wt[0], 4, true);
                DFM_FENCE();
                cvt_half(q[cb][2].x, q[cb][2].y, &fa[0]);
                DFM_FENCE();
                DFM_CHAIN(r, fb, wt[1], 0, false);
                DFM_FENCE();
                cvt_half(q[cb][2].z, q[cb][2].w, &fa[4]);
                DFM_FENCE();
                DFM_CHAIN(r, fb, wt[1], 4, false);
                DFM_FENCE();
                cvt_half(q[cb][3].x, q[cb][3].y, &fb[0]);
                DFM_FENCE();
                DFM_CHAIN(r, fa, wt[2], 0, false);
                DFM_FENCE();
                cvt_half(q[cb][3].z, q[cb][3].w, &fb[4]);
                DFM_FENCE();
                DFM_CHAIN(r, fa, wt[2], 4, false);
                DFM_CHAIN(r, fb, wt[3], 0, false);
                DFM_CHAIN(r, fb, wt[3], 4, false);
#undef DFM_CHAIN
#undef DFM_FENCE
#endif
                // (pins the chain here: the vectoriser otherwise builds one tree from the 16-byte store
                // vectors down through all 8 points and sinks every chain below the last conversion)
                if constexpr (CB == 8)
                    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),
                                      "+v"(r[6]), "+v"(r[7]));
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    if constexpr (sizeof(T) == 4) pk[k][j] = __float_as_uint(r[k]);
                    else if (j & 1) pk[k][j >> 1] = pack_bf16x2(keep[k], r[k]);
                    else keep[k] = r[k];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (LDS) {
            // Taps come from LDS through inline-asm ds_read_b128 (hipcc would put
            // a vmcnt(0) in front of the first LDS read after an LDS-DMA was
            // issued, and serialises reads against blends).  Software pipeline:
            // the 4 corner reads of point j+1 are in flight while point j blends;
            // LDS returns in order, so lgkmcnt(4) == "point j has landed".
            u32x4_t q[2][4];
            float keep[CB];  // even point of a pair, waiting for its odd partner
            DFM_TAP_READ(q[0][0], ta[0][0]);
            DFM_TAP_READ(q[0][1], ta[0][1]);
            DFM_TAP_READ(q[0][2], ta[0][2]);
            DFM_TAP_READ(q[0][3], ta[0][3]);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int cb = j & 1, nb = cb ^ 1;
                if (j + 1 < V) {
                    DFM_TAP_READ(q[nb][0], ta[j + 1][0]);
                    DFM_TAP_READ(q[nb][1], ta[j + 1][1]);
                    DFM_TAP_READ(q[nb][2], ta[j + 1][2]);
                    DFM_TAP_READ(q[nb][3], ta[j + 1][3]);
                    asm volatile("s_waitcnt lgkmcnt(4)"
                                 : "+v"(q[cb][0]), "+v"(q[cb][1]), "+v"(q[cb][2]), "+v"(q[cb][3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(q[cb][0]), "+v"(q[cb][1]), "+v"(q[cb][2]), "+v"(q[cb][3]));
                }
                Tap t;
                const float w = fw[j], n = fn[j], e = 1.0f - w, s2 = 1.0f - n;
                t.nw = s2 * e; t.ne = s2 * w; t.sw = n * e; t.se = n * w;
                float r[CB];
                blend_nomask<CB>(t, make_uint4(q[cb][0].x, q[cb][0].y, q[cb][0].z, q[cb][0].w),
                                 make_uint4(q[cb][1].x, q[cb][1].y, q[cb][1].z, q[cb][1].w),
                                 make_uint4(q[cb][2].x, q[cb][2].y, q[cb][2].z, q[cb][2].w),
                                 make_uint4(q[cb][3].x, q[cb][3].y, q[cb][3].z, q[cb][3].w), r);
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    if constexpr (sizeof(T) == 4) {
                        pk[k][j] = __float_as_uint(r[k]);
                    } else {
                        if (j & 1) pk[k][j >> 1] = pack_bf16x2(keep[k], r[k]);
                        else keep[k] = r[k];
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                Tap t;
                const float w = fw[j], n = fn[j], e = 1.0f - w, s2 = 1.0f - n;
                t.nw = s2 * e; t.ne = s2 * w; t.sw = n * e; t.se = n * w;
                const int a = qN[j], c = qS[j];
                const uint4 qnw = gsrc[max(a, 0)], qne = gsrc[min(a + 1, HW - 1)];
                const uint4 qsw = gsrc[max(c, 0)], qse = gsrc[min(c + 1, HW - 1)];
                t.ok = (okbits >> (4 * j)) & 15u;
                float r[CB];
                if (all_in) blend_nomask<CB>(t, qnw, qne, qsw, qse, r);
                else blend<CB>(t, qnw, qne, qsw, qse, r);
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    if constexpr (sizeof(T) == 4) pk[k][j] = __float_as_uint(r[k]);
                    else if (j & 1) pk[k][j >> 1] |= (uint32_t)f32_to_bf16(r[k]) << 16;
                    else pk[k][j >> 1] = (uint32_t)f32_to_bf16(r[k]);
                }
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        TRACE_STAMP();
    };
    auto store_block = [&](int blk, const uint32_t (&pk)[CB][VW]) {
        const int cbase = blk * CB;
        if constexpr (VW == 2) {
            // 8-byte halves of a 16-byte vector lie in neighbouring lanes (4 points per lane): the pair
            // trades halves -- the even lane takes channel 2i of both, the odd lane channel 2i + 1 -- and
            // each stores ONE 16-byte vector per channel pair instead of two 8-byte ones (half the store
            // instructions of the workgroup, whole 16-byte lane accesses; 4 v_cndmask_b32_dpp per pair).
            // The tile's cuts are multiples of 8 points and an even lane's first point is one, so the two
            // lanes of a pair are active together.
            if (tg.pair_stores) {
                const bool odd = tid & 1;
#pragma unroll
                for (int kp = 0; kp < CB / 2; ++kp) {
                    const int ka = 2 * kp, kb = ka + 1;
                    if (cbase + kb < g.C) {
                        // quad_perm [1,0,3,2]: the neighbour's value
                        const uint32_t na0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk[ka][0], 0xb1, 0xf, 0xf, false);
                        const uint32_t na1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk[ka][1], 0xb1, 0xf, 0xf, false);
                        const uint32_t nb0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk[kb][0], 0xb1, 0xf, 0xf, false);
                        const uint32_t nb1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk[kb][1], 0xb1, 0xf, 0xf, false);
                        // even: channel ka = {own points 0-3, neighbour's 4-7}; odd: channel kb = {neighbour's 0-3, own 4-7}
                        const u32x4_t v = {odd ? nb0 : pk[ka][0], odd ? nb1 : pk[ka][1],
                                           odd ? pk[kb][0] : na0, odd ? pk[kb][1] : na1};
                        T *dst = o + (size_t)(cbase + ka + (odd ? 1 : 0)) * g.N - (odd ? 4 : 0);
                        if (!ABLATE(2) || v.x == 0x12345u) vol_store16(v, dst);
                    } else if (cbase + ka < g.C && (!ABLATE(2) || pk[ka][0] == 0x12345u)) {
                        u32x2_t v = {pk[ka][0], pk[ka][1]};  // an odd channel count's last channel
                        __builtin_nontemporal_store(v, (u32x2_t *)(o + (size_t)(cbase + ka) * g.N));
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (cbase + k < g.C && (!ABLATE(2) || pk[k][0] == 0x12345u)) {
                if constexpr (VW == 4) {
                    u32x4_t v = {pk[k][0], pk[k][1], pk[k][2], pk[k][3]};
                    vol_store16(v, o + (size_t)(cbase + k) * g.N);
                } else {
                    u32x2_t v = {pk[k][0], pk[k][1]};
                    __builtin_nontemporal_store(v, (u32x2_t *)(o + (size_t)(cbase + k) * g.N));
                }
            }
        }
    };
    auto compute_store = [&](int blk, const uint4 *gsrc, auto bufc, auto mxc) {
        uint32_t pk[CB][VW];
        compute_block(blk, gsrc, bufc, mxc, pk);
        store_block(blk, pk);
    };

    if constexpr (!LDS) {
        for (int blk = blk_lo; blk < blk_hi; ++blk) {
            compute_store(blk, src, IntC<0>{}, IntC<0>{});
            src += HW;
        }
    } else {
        // stage one channel block's rows into LDS, pixel p -> slot swz(p + PAD), by
        // LDS-DMA (global_load_lds: no VGPR round trip; the DMA writes LDS linearly,
        // so the swizzle goes on the SOURCE pixel).  A global_load -> ds_write
        // variant measured 2x slower and cost 20 VGPRs (r01 profiles).
        auto stage = [&](int buf_slot0, const uint4 *gsrc) {
            if (ABLATE(1)) return;
            for (int s0 = wave * 64; s0 < nslots; s0 += NT) {
                const int p = swz(s0 + lane) - PAD;
                if (p >= 0 && p < cnt)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(gsrc + p),
                        (__attribute__((address_space(3))) void *)(lds + buf_slot0 + s0), 16, 0, 0);
            }
        };
        // stage -> barrier -> blend+store -> barrier.  (A double-buffered variant
        // with counted vmcnt measured no faster on MI355X and doubled the LDS per
        // workgroup; profiles/archive/r01_v6_double_buffer_variants.txt.)
        if constexpr (!PIPE) {
            for (int blk = blk_lo; blk < blk_hi; ++blk) {
                stage(SLAB, src);
                TRACE_STAMP();
                __syncthreads();  // drains the DMA (vmcnt) and makes the rows visible
                TRACE_STAMP();
                if (active) compute_store(blk, src, IntC<0>{}, IntC<MXK ? 1 : 0>{});
                TRACE_STAMP();
                src += HW;
                __syncthreads();  // everyone is done with the rows before the refill
                TRACE_STAMP();
            }
        } else {
            // Pipelined body: two LDS buffers, ONE barrier per channel block, and no wait for the
            // volume stores.  Per block k:   wait for DMA(k) | barrier | issue DMA(k+1) into the
            // other buffer | blend + store block k.
            // vmcnt counts loads and stores of a wave in issue order (gfx9: one in-order counter), so
            // "DMA(k) has landed" is "everything but the newest stores(k-1) has retired": a counted
            // vmcnt(CB) (one store per channel of the block and lane).  The serial body above waits vmcnt(0) before
            // its barrier -- for the DMA AND for the acknowledgement of the previous block's stores,
            // which on parts with a slow HBM write path is what the kernel then runs at
            // (profiles/archive/r03_c53_*: the stores add 1.2 ms on a fast part, 2.7 ms on a slow one, to a
            // sampling skeleton of the same 4.5 ms).  Here a store has a whole block (~3 us) to be
            // acknowledged, the DMA of the next rows flies under the blend, and the barrier that
            // protected the single buffer against its refill is gone.
            // A wave with no lattice point issues no stores: it waits vmcnt(0).
            const bool wave_stores = __any(active) != 0;  // wave-uniform
            const bool paired = VW == 2 && tg.pair_stores;  // one 16-byte store per channel PAIR
            auto wait_rows = [&]() {
                if (wave_stores && !ABLATE(2) && !paired)
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(CB) : "memory");
                else if (wave_stores && !ABLATE(2))
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(CB / 2) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            };
            stage(SLAB, src);
            TRACE_STAMP();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // (Tried and dropped, profiles/archive/r04_c21..c26: issuing the stores of block k AFTER its barrier -- a
            // store then has two block periods to be acknowledged -- 8 % slower on a box where it mattered:
            // the four waves store in one synchronised burst; the block in two half passes, channels 0-3,
            // their stores, channels 4-7 -- no faster; starting every second workgroup of a CU half a
            // block period late -- no effect.)
            for (int blk = blk_lo;;) {
                if (blk + 1 < blk_hi) stage(SLAB + PIPE_BUF_SLOTS, src + HW);
                TRACE_STAMP();
                if (active) compute_store(blk, src, IntC<0>{}, IntC<MXK ? 1 : 0>{});
                TRACE_STAMP();
                src += HW;
                if (++blk >= blk_hi) break;
                wait_rows();
                TRACE_STAMP();
                if (blk + 1 < blk_hi) stage(SLAB, src + HW);
                TRACE_STAMP();
                if (active) compute_store(blk, src, IntC<1>{}, IntC<MXK ? 1 : 0>{});
                TRACE_STAMP();
                src += HW;
                if (++blk >= blk_hi) break;
                wait_rows();
                TRACE_STAMP();
            }
        }
    }
#undef DFM_TAP_READ
    return 0;
}

#ifndef DFM_TILE_WAVES
#define DFM_TILE_WAVES 1  // min waves/SIMD the LDS tile kernel is compiled for (1 = compiler's choice)
#endif
// one workgroup per tile; LDS == false is the "direct taps for every tile" mode.
// V * sizeof(T) == 8 (bf16, 4 points per lane): compiled for 4 waves per SIMD (<= 128 VGPRs),
// i.e. two 512-lane workgroups or one 1024-lane workgroup per CU.
template <typename T, int NT, bool LDS, int V, bool PIPE = false, bool MXK = false>
__global__ __launch_bounds__(NT, (LDS ? (V * sizeof(T) == 8 ? 4 : (MXK ? 2 : DFM_TILE_WAVES)) : 1)) void sweep_tile_kernel(
    SweepGeom g, SweepFast fast, TileGrid tg, int lds_slots, const uint4 *__restrict__ cur_blk,
    const uint4 *__restrict__ prev_blk, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    T *__restrict__ out, int *__restrict__ spill_list)
{
    int st = tile_body<T, NT, LDS, V, PIPE, MXK>(blockIdx.x, g, fast, tg, lds_slots, cur_blk, prev_blk, depths, P,
                                                 Pinv, Tm, out, spill_list, -1, -1, PIPE);
    if constexpr (PIPE) {
        // Second chances IN PLACE (round 5).  A tile whose rows exceed one of the pipelined body's two buffers
        // (8 rows of 311 pixels; about 1 % of the tiles at N*: the nearest planes of the prev map) runs the
        // serial body -- stage, barrier, blend + store, barrier -- with the two buffers as one (16 rows); what
        // does not fit that either takes direct taps.  The separate second-chance launch (sweep_respill_kernel,
        // 144 KiB of LDS, one workgroup per CU) cost 0.145 ms per N* launch for 1 % of the tiles -- a latency
        // chain on a mostly idle chip; here those tiles cost their workgroup about 1.5 block periods among 512
        // resident ones, and a dense pipelined launch queues nothing: the second-chance and direct kernels are
        // not launched at all.  A tile that stages a row with a value the matrix-core unpack is not exact for
        // (pack_blocked_kernel's row flags) runs the pipelined body with the VALU unpack: per TILE, where rounds
        // 3-4 sent the whole launch to the second-chance pass.
        if constexpr (MXK) {
            if (st == 2) {
                __syncthreads();  // (every lane has read the bounding rows of the previous attempt)
                st = tile_body<T, NT, true, V, true, false>(blockIdx.x, g, fast, tg, lds_slots, cur_blk, prev_blk,
                                                            depths, P, Pinv, Tm, out, spill_list, -1, -1, true);
            }
        }
        if (st == 1) {
            __syncthreads();
            // 16-byte stores (the default shapes): direct taps in place as the last resort; the 8-byte-store
            // shapes (128 registers per lane) keep the queue and the direct-tap launch for it
            constexpr bool INPLACE3 = V * sizeof(T) == 16;
            st = tile_body<T, NT, true, V, false, false>(blockIdx.x, g, fast, tg, lds_slots, cur_blk, prev_blk, depths,
                                                         P, Pinv, Tm, out, spill_list, -1, -1, INPLACE3);
            if constexpr (INPLACE3) {
                if (st == 1) {
                    __syncthreads();
                    tile_body<T, NT, false, V>(blockIdx.x, g, fast, tg, 0, cur_blk, prev_blk, depths, P, Pinv, Tm,
                                               out, nullptr);
                }
            }
        }
    }
}

// the tiles the LDS pass queued (spill_list[0] = count), a fixed small grid strides over them:
// launching one mostly-empty workgroup per tile would cost ~0.5 ms for 26 k tiles
template <typename T, int NT, int V>
__global__ __launch_bounds__(NT) void sweep_spill_kernel(
    SweepGeom g, SweepFast fast, TileGrid tg, const uint4 *__restrict__ cur_blk,
    const uint4 *__restrict__ prev_blk, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    T *__restrict__ out, int *__restrict__ spill_list)
{
    const int count = spill_list[0];
    for (int i = blockIdx.x; i < count; i += gridDim.x)
        tile_body<T, NT, false, V>(spill_list[1 + i], g, fast, tg, 0, cur_blk, prev_blk, depths, P,
                                   Pinv, Tm, out, nullptr);
}

// Second chance for the flagged tiles (about 1 % at N*: the nearest planes of the prev map, whose
// 3x zoom needs a few rows more than the budget that lets two workgroups share a CU): the same
// LDS-staged body with (almost) the whole LDS of a CU, each tile split over `groups` workgroups
// by channel block so the few hundred tiles still fill the chip.  What does not fit even then
// goes to list_out and the direct-tap pass.  (The direct pass alone cost 0.12-0.17 ms per N* launch.)
template <typename T, int NT, int V>
__global__ __launch_bounds__(NT, (V * sizeof(T) == 8 ? 4 : DFM_TILE_WAVES)) void sweep_respill_kernel(
    SweepGeom g, SweepFast fast, TileGrid tg, int lds_slots, int groups, int blocks_per_group,
    const uint4 *__restrict__ cur_blk, const uint4 *__restrict__ prev_blk,
    const float *__restrict__ depths, const float *__restrict__ P, const float *__restrict__ Pinv,
    const float *__restrict__ Tm, T *__restrict__ out, const int *__restrict__ list_in,
    int *__restrict__ list_out)
{
    const int count = list_in[0];
    for (int i = blockIdx.x; i < count * groups; i += gridDim.x) {
        const int grp = i % groups;
        tile_body<T, NT, true, V>(list_in[1 + i / groups], g, fast, tg, lds_slots, cur_blk, prev_blk,
                                  depths, P, Pinv, Tm, out, list_out, grp * blocks_per_group,
                                  (grp + 1) * blocks_per_group);
        __syncthreads();  // the next tile re-initialises the LDS scratch
    }
}

// The < align lattice points in front of every depth-plane boundary that the LDS pass
// masked (see sweep_tile_kernel): direct taps, scalar stores.  grid = (D-1, 2, B),
// block = 8 points x 32 channel-block lanes, looping over the boundary's points.
template <typename T>
__global__ __launch_bounds__(256) void sweep_patch_kernel(
    SweepGeom g, SweepFast fast, int align, const uint4 *__restrict__ cur_blk,
    const uint4 *__restrict__ prev_blk, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    T *__restrict__ out)
{
    constexpr int CB = elem<T>::CB;
    const int d_next = blockIdx.x + 1;  // boundary in front of plane d_next
    const int half = blockIdx.y, b = blockIdx.z;
    const long long hw = (long long)g.h_out * g.w_out;
    const long long edge = d_next * hw;
    const int d = d_next - 1;
    const int HW = g.h_in * g.w_in;
    const uint4 *mb = (half ? prev_blk : cur_blk) + (size_t)b * g.nblk * HW;
    (void)align;
    for (long long n = (edge & ~7ll) + (threadIdx.x >> 5); n < edge; n += 8) {
        const int rem = (int)(n - (long long)d * hw);
        const int hi = rem / g.w_out, wi = rem - hi * g.w_out;
        float sx, sy;
        if (half) sweep_point_map<1>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, sx, sy);
        else sweep_point_map<0>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, sx, sy);
        const Tap t = make_tap(sx, sy, g.h_in, g.w_in);
        const int i00 = t.iy * g.w_in + t.ix, i01 = i00 + t.dx;
        const int i10 = i00 + t.dy * g.w_in, i11 = i10 + t.dx;
        T *o = out + ((size_t)b * 2 * g.C + (size_t)half * g.C) * g.N + n;
        for (int blk = threadIdx.x & 31; blk < g.nblk; blk += 32) {
            const uint4 *q = mb + (size_t)blk * HW;
            float r[CB];
            blend<CB>(t, q[i00], q[i01], q[i10], q[i11], r);
#pragma unroll
            for (int k = 0; k < CB; ++k)
                if (blk * CB + k < g.C) o[(size_t)(blk * CB + k) * g.N] = elem<T>::store(r[k]);
        }
    }
}

// ---------------------------------------------------------------------------
// backward: grad feats += scatter(grad_out * weights); grid (ceil(N/256), B)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sweep_bwd_kernel(
    SweepGeom g, const T *__restrict__ gout, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    float *__restrict__ gcur, float *__restrict__ gprev)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= g.N) return;
    const int hw = g.h_out * g.w_out;
    const int d = (int)(n / hw);
    const int rem = (int)(n - (long long)d * hw);
    const int hi = rem / g.w_out;
    const int wi = rem - hi * g.w_out;
    float cx, cy, px, py;
    sweep_point(g, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, cx, cy, px, py,
                nullptr);
    const Tap tc = make_tap(cx, cy, g.h_in, g.w_in);
    const Tap tp = make_tap(px, py, g.h_in, g.w_in);
    const int HW = g.h_in * g.w_in;
    const int c00 = tc.iy * g.w_in + tc.ix, c01 = c00 + tc.dx;
    const int c10 = c00 + tc.dy * g.w_in, c11 = c10 + tc.dx;
    const int p00 = tp.iy * g.w_in + tp.ix, p01 = p00 + tp.dx;
    const int p10 = p00 + tp.dy * g.w_in, p11 = p10 + tp.dx;
    const T *gc = gout + ((size_t)b * 2 * g.C) * g.N + n;
    const T *gp = gc + (size_t)g.C * g.N;
    float *dc = gcur + (size_t)b * g.C * HW;
    float *dp = gprev + (size_t)b * g.C * HW;
    for (int c = 0; c < g.C; ++c) {
        const float go_c = elem<T>::load(gc[(size_t)c * g.N]);
        const float go_p = elem<T>::load(gp[(size_t)c * g.N]);
        if (tc.ok & 1u) atomicAdd(dc + c00, go_c * tc.nw);
        if (tc.ok & 2u) atomicAdd(dc + c01, go_c * tc.ne);
        if (tc.ok & 4u) atomicAdd(dc + c10, go_c * tc.sw);
        if (tc.ok & 8u) atomicAdd(dc + c11, go_c * tc.se);
        if (tp.ok & 1u) atomicAdd(dp + p00, go_p * tp.nw);
        if (tp.ok & 2u) atomicAdd(dp + p01, go_p * tp.ne);
        if (tp.ok & 4u) atomicAdd(dp + p10, go_p * tp.sw);
        if (tp.ok & 8u) atomicAdd(dp + p11, go_p * tp.se);
        dc += HW;
        dp += HW;
    }
}

// ---------------------------------------------------------------------------
// backward: LDS-accumulating tiles.
// A workgroup = one band of 256 lattice points (same (h, w) for every plane) of ONE map x G plane
// groups (1024 lanes, G = 4 for the prev map; 512 lanes, G = 2 for the cur map), over a chunk of
// <= BWD_MAXP depth planes: lane (tid & 255) is the point, (tid >> 8) takes the planes
// p = group, group + G, ...
//   * Footprints once: a prologue computes every (plane, point) sampling position with the
//     forward kernel's own sweep_point_map and keeps it in an LDS table (corner + in-bounds bits
//     and the two fractions: 12 bytes per entry), so the C/CW channel passes that follow never
//     touch the geometry again.  (Round 1 re-derived the position per plane per pass and was
//     VALU-bound on exactly that: 34 of 39 ms at N*.)
//   * Per pass of CW channels the taps' gradients are added into a slab of feature rows in LDS
//     ([channel][row][x], 64-bit two's-complement fixed point: integer LDS atomics run at 10-14
//     lanes per clock, ds_add_f32 at 0.33 -- profiles/archive/r01_atomic_microbench.txt); the slab goes
//     to the global gradient with ONE coalesced fp32 atomic per touched pixel at the end of each
//     slab window (the longest run of planes whose rows fit; the chunk, unless the footprint
//     drifts far).
//       cur map : x = w +- 1e-5, the footprint stays inside a 3x3 block anchored at the lane's
//                 smallest corner: gradients are summed over the lane's planes in registers and
//                 scattered once per window (a lane whose footprint leaves its block -- general
//                 poses -- scatters that plane's taps straight to memory);
//       prev map: the footprint drifts with depth; the four taps are scattered per plane.
//   * The fixed-point scale is LOCAL: the prev kernel scans the pass's gradient values of its own
//     (chunk, band) first (they are re-read from L2 right after), the cur kernel takes the maximum
//     of its register sums at scatter time -- no pass over the whole gradient volume
//     (absmax_bits_kernel: 5 ms of the 39 at N*), no device allocation.  An all-zero pass is
//     skipped; a pass holding Inf / NaN takes plain float atomics so they propagate like torch's.
// ---------------------------------------------------------------------------
constexpr int BWD_MAXP = 32;  // depth planes per workgroup, at most
constexpr int BWD_PTS = 256;  // lattice points per band
// plane groups per workgroup: 4 for the prev map (70 VGPRs, LDS-atomic latency wants many waves),
// 2 for the cur map (its 3x3 register block needs more than the 128 VGPRs a 1024-lane group allows)
__host__ __device__ constexpr int bwd_groups(int half) { return half ? 4 : 2; }

struct BwdGrid {
    int batch, bands, band_pts, planes, dchunks, rows;
    int ablate;  // debug builds only (DFM_BWD_ABLATE): 1 no gradient loads, 2 no slab atomics, 4 no flush
    int grad_cl;  // 1: the gradient volume is stored channels-last, (B, D, h, w, 2C) (torch channels_last_3d)
    int row_tiles;  // 0: bands are runs of band_pts points of the flat (h, w) index;
                    // > 0 (strided sweeps): that many bands per lattice row, none crossing rows --
                    // consecutive lattice rows sample feature rows `cost_sample_factor` apart,
                    // which one slab window cannot hold
    int split;      // 1: only the planes before sweep_zoom_split (the matrix-product backward takes the rest)
};

// float -> 64-bit two's-complement fixed point (|x| < 2^61 after scaling): high word =
// floor(x / 2^32), low word = x - high * 2^32, which the fma delivers exactly except for a negative
// x of tiny magnitude, whose 2^32 - |x| rounds to 2^32 and saturates the conversion -- one unit of
// 2^-50 of the local maximum.  Five VALU operations; the pair is assembled from the two converted
// words (the first version went through a float -> u64 conversion: 12 operations per add).
__device__ __forceinline__ unsigned long long bwd_to_fixed(float x)
{
    const float hif = floorf(x * 2.3283064365386963e-10f);
    const float lof = __builtin_fmaf(hif, -4294967296.0f, x);  // in [0, 2^32]
    unsigned lo;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(lo) : "v"(lof));  // saturating
    const unsigned hi = (unsigned)(int)hif;
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}

// scale 2^sh with max|x| * 2^sh < 2^50 from the raw bits of max|x| (finite, non-zero)
__device__ __forceinline__ void bwd_scale(unsigned mb, float &fx_scale, float &fx_inv)
{
    const int sh = min(120, max(-100, 50 - ((int)(mb >> 23) - 127 + 1)));
    fx_scale = __uint_as_float((unsigned)(sh + 127) << 23);
    fx_inv = __uint_as_float((unsigned)(127 - sh) << 23);
}

#ifdef DFM_DEBUG_HOOKS
#define BWD_ABLATE(bit) ((tg.ablate & (bit)) != 0)
#else
#define BWD_ABLATE(bit) false
#endif
template <typename T, int CW, int HALF>
__global__ __launch_bounds__(BWD_PTS * bwd_groups(HALF)) void sweep_bwd_tile_kernel(
    SweepGeom g, SweepFast fast, BwdGrid tg, const T *__restrict__ gout,
    const float *__restrict__ depths, const float *__restrict__ P, const float *__restrict__ Pinv,
    const float *__restrict__ Tm, float *__restrict__ gcur, float *__restrict__ gprev)
{
    constexpr int BWD_GROUPS = bwd_groups(HALF);
    constexpr int NT = BWD_PTS * BWD_GROUPS;
    constexpr int VB = 8;  // plane slots whose gradient values are fetched together (one latency)
    extern __shared__ __attribute__((aligned(16))) unsigned long long slab[];
    __shared__ int yr[2 * BWD_MAXP];
    __shared__ unsigned wgm[3];  // rotating slots of the workgroup-wide maximum (see wg_max)
    __shared__ int wins[4 * BWD_MAXP + 1];  // slab windows of the chunk: count, then {first, last plane, y0, top}
    // block id = (band*dchunks + dchunk)*batch + b
    int th = blockIdx.x;
    const int b = th % tg.batch;
    th /= tg.batch;
    const int dchunk = th % tg.dchunks;
    const int band = th / tg.dchunks;
    const int tid = threadIdx.x, pt = tid & (BWD_PTS - 1), grp = tid >> 8;
    const int hw = g.h_out * g.w_out;
    const int W = g.w_in, H = g.h_in, HW = H * W;
    int p_lo = band * tg.band_pts, p_hi = min(p_lo + tg.band_pts, hw);
    if (tg.row_tiles > 0) {
        const int row = band / tg.row_tiles, t = band - row * tg.row_tiles;
        p_lo = row * g.w_out + t * tg.band_pts;
        p_hi = min(p_lo + tg.band_pts, (row + 1) * g.w_out);
    }
    const int d_lo = dchunk * tg.planes;
    int d_hi = min(d_lo + tg.planes, g.D);
    const float *Pb = P + b * 16, *Pib = Pinv + b * 16, *Tb = Tm + b * 16;
    if (tg.split) {  // workgroup-uniform
        d_hi = min(d_hi, sweep_zoom_split<HALF>(g, fast, Pb, Pib, Tb, depths, SWEEP_BWD_ZOOM_FOUR, threadIdx.x,
                                                BWD_PTS * BWD_GROUPS, &yr[0]));
        if (d_hi <= d_lo) return;
        __syncthreads();  // yr is initialised below
    }
    const int np = d_hi - d_lo;
    const int rows = tg.rows, slab_c = rows * W;
    // footprint table behind the slab: [plane][point] x {packed, fw, fn}
    uint32_t *fpT = (uint32_t *)(slab + (size_t)CW * slab_c);
    float *fwT = (float *)(fpT + tg.planes * BWD_PTS);
    float *fnT = fwT + tg.planes * BWD_PTS;
    const int idx = p_lo + pt;
    const bool live = idx < p_hi;
    const int hi = idx / g.w_out, wi = idx - hi * g.w_out;

    for (int i = tid; i < 2 * BWD_MAXP; i += NT) yr[i] = (i & 1) ? -1 : 0x7fffffff;
    for (int i = tid; i < CW * slab_c; i += NT) slab[i] = 0ull;
    if (tid < 3) wgm[tid] = 0u;
    __syncthreads();
    // workgroup-wide maximum with ONE barrier per call: round k accumulates into slot k % 3 and
    // clears slot (k + 1) % 3, which nobody reads (round k - 1 reads slot (k - 1) % 3) or
    // writes (round k + 1 starts after this barrier) meanwhile
    int mround = 0;
    auto wg_max = [&](unsigned m) -> unsigned {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        const int slot = mround % 3;
        if ((tid & 63) == 0 && m) atomicMax(&wgm[slot], m);
        if (tid == 0) wgm[(mround + 1) % 3] = 0u;
        __syncthreads();
        ++mround;
        return wgm[slot];
    };

    // ---- footprints of the band's points in every plane of the chunk -> LDS table ------------
    int bx = 0x7fffffff, by = 0x7fffffff;  // cur map: block anchor = this lane's smallest corner
    for (int p = grp; p < np; p += BWD_GROUPS) {
        int ymin = 0x7fffffff, ymax = -1;
        uint32_t f = 0u;
        float fw = 0.0f, fn = 0.0f;
        if (live) {
            float sx, sy;
            sweep_point_map<HALF>(g, fast, Pb, Pib, Tb, depths[d_lo + p], hi, wi, sx, sy);
            f = bwd_footprint(sx, sy, H, W, fw, fn);
            if (f) {
                const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
                if (f & (1u << 29)) { ymin = min(ymin, iyn); ymax = max(ymax, iyn); }
                if (f & (1u << 30)) { ymin = min(ymin, iyn + 1); ymax = max(ymax, iyn + 1); }
                bx = min(bx, ixw);
                by = min(by, iyn);
            }
        }
        fpT[p * BWD_PTS + pt] = f;
        fwT[p * BWD_PTS + pt] = fw;
        fnT[p * BWD_PTS + pt] = fn;
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) {
            ymin = min(ymin, __shfl_xor(ymin, s2));
            ymax = max(ymax, __shfl_xor(ymax, s2));
        }
        if ((tid & 63) == 0 && ymax >= 0) {
            atomicMin(&yr[2 * p], ymin);
            atomicMax(&yr[2 * p + 1], ymax);
        }
    }
    __syncthreads();
    // slab windows: maximal runs of planes whose rows fit `rows` slab rows -- the same for every
    // channel pass, so they are laid out once
    if (tid == 0) {
        int nw = 0, p = 0;
        while (p < np) {
            while (p < np && yr[2 * p + 1] < yr[2 * p]) ++p;  // planes that miss the map
            if (p >= np) break;
            int umin = yr[2 * p], umax = yr[2 * p + 1], e = p;
            while (e + 1 < np) {
                const int l2 = yr[2 * (e + 1)], u2 = yr[2 * (e + 1) + 1];
                if (u2 >= l2) {
                    if (max(umax, u2) - min(umin, l2) + 1 > rows) break;
                    umin = min(umin, l2);
                    umax = max(umax, u2);
                }
                ++e;
            }
            wins[1 + 4 * nw] = p;
            wins[2 + 4 * nw] = e;
            wins[3 + 4 * nw] = umin;
            wins[4 + 4 * nw] = min(umax, umin + rows - 1);
            ++nw;
            p = e + 1;
        }
        wins[0] = nw;
    }
    __syncthreads();
    const int nwin = __builtin_amdgcn_readfirstlane(wins[0]);

    // element strides of the gradient volume: the reference layout (B, 2C, D, h, w), or channels-last
    // (B, D, h, w, 2C) -- what the NDHWC aggregation stack's backward hands over (read in place: the
    // 236 MB layout conversion at config K cost 2.2 ms per training step, and a lane's CW channels of a
    // (plane, point) are then adjacent)
    const size_t s_chan = tg.grad_cl ? (size_t)1 : (size_t)g.N;
    const size_t s_plane = tg.grad_cl ? (size_t)hw * 2 * g.C : (size_t)hw;
    const size_t s_point = tg.grad_cl ? (size_t)2 * g.C : (size_t)1;
    const T *go = gout + (size_t)b * 2 * g.C * g.N + (size_t)HALF * g.C * s_chan + (size_t)d_lo * s_plane +
                  (size_t)min(idx, p_hi - 1) * s_point;
    float *gf = (HALF ? gprev : gcur) + (size_t)b * g.C * HW;

    for (int c0 = 0; c0 < g.C; c0 += CW) {
        const int nc = min(CW, g.C - c0);
        const T *gp = go + (size_t)c0 * s_chan;
        float fx_scale = 1.0f, fx_inv = 1.0f;
        bool plain = false;  // Inf / NaN in this pass (window): plain float atomics
        int y0 = -1, top = -1;
        auto flush = [&]() {
            if (BWD_ABLATE(4)) return;
            const int cnt = (top - y0 + 1) * W;
            for (int c = 0; c < nc; ++c) {
                unsigned long long *sl = slab + c * slab_c;
                float *dst = gf + (size_t)(c0 + c) * HW + (size_t)y0 * W;
                for (int r = tid; r < cnt; r += NT) {
                    const unsigned long long v = sl[r];
                    if (v != 0ull) {
                        const float f = __builtin_fmaf((float)(int)(v >> 32), 4294967296.0f, (float)(unsigned)v);
                        atomicAdd(dst + r, f * fx_inv);
                        sl[r] = 0ull;
                    }
                }
            }
        };
        // gradient values of up to VB of this lane's planes (slots k0 .. k0+VB-1; slot k is plane
        // grp + k*G), all loads in flight together; planes outside the chunk / map give 0
        auto load_block = [&](int k0, T (&gv)[VB][CW]) {
            // every load is unconditional (clamped, always-valid address) and the masking happens on
            // the values afterwards: a load under a runtime condition makes hipcc branch around it
            // and wait for each one separately (cdna_hip_programming.md, ".s-level traps" (c))
#pragma unroll
            for (int k = 0; k < VB; ++k) {
                const int p = min(grp + (k0 + k) * BWD_GROUPS, np - 1);
#pragma unroll
                for (int c = 0; c < CW; ++c) gv[k][c] = gp[(size_t)p * s_plane + (size_t)min(c, nc - 1) * s_chan];
            }
#pragma unroll
            for (int k = 0; k < VB; ++k) {
                const int p = grp + (k0 + k) * BWD_GROUPS;
                const bool on = p < np && fpT[min(p, np - 1) * BWD_PTS + pt] != 0u && !BWD_ABLATE(1);
#pragma unroll
                for (int c = 0; c < CW; ++c) gv[k][c] = (on && c < nc) ? gv[k][c] : T(0);
            }
        };

        if constexpr (HALF == 1) {
            static_assert(BWD_MAXP / bwd_groups(1) <= VB, "one block holds all planes of a lane");
            T gv[VB][CW];
            load_block(0, gv);
            // ---- scale of this pass: max |grad| over the workgroup's values ----------------
            unsigned m = 0u;
#pragma unroll
            for (int k = 0; k < VB; ++k)
#pragma unroll
                for (int c = 0; c < CW; ++c) m = max(m, __float_as_uint(elem<T>::load(gv[k][c])) & 0x7fffffffu);
            const unsigned mb = wg_max(m);
            if (mb == 0u) continue;  // nothing to add in this pass
            plain = (mb >> 23) == 0xffu;
            if (!plain) bwd_scale(mb, fx_scale, fx_inv);
            // ---- windows of planes; the four taps of every plane go into the slab ----------
            for (int wi_ = 0; wi_ < nwin; ++wi_) {
                const int pw = __builtin_amdgcn_readfirstlane(wins[1 + 4 * wi_]);
                const int pe = __builtin_amdgcn_readfirstlane(wins[2 + 4 * wi_]);
                y0 = __builtin_amdgcn_readfirstlane(wins[3 + 4 * wi_]);
                top = __builtin_amdgcn_readfirstlane(wins[4 + 4 * wi_]);
#pragma unroll
                for (int k = 0; k < VB; ++k) {
                    const int p = grp + k * BWD_GROUPS;
                    if (p < pw || p > pe) continue;
                    const uint32_t f = fpT[p * BWD_PTS + pt];
                    if (!f) continue;
                    const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
                    const float fw = fwT[p * BWD_PTS + pt], fn = fnT[p * BWD_PTS + pt];
                    const float cwt = (f & (1u << 27)) ? 1.0f - fw : 0.0f, cet = (f & (1u << 28)) ? fw : 0.0f;
                    const float rnt = (f & (1u << 29)) ? 1.0f - fn : 0.0f, rst = (f & (1u << 30)) ? fn : 0.0f;
                    const float wq[4] = {rnt * cwt, rnt * cet, rst * cwt, rst * cet};
                    float gvf[CW];
#pragma unroll
                    for (int c = 0; c < CW; ++c) gvf[c] = elem<T>::load(gv[k][c]);
                    const int rr0 = iyn - y0;  // >= 0 for an in-bounds north row: y0 <= the window's first row
                    if ((f & 0x78000000u) == 0x78000000u && rr0 + 1 < rows && !plain && !BWD_ABLATE(2)) {
                        // interior footprint inside the slab window (nearly every point): four taps x CW
                        // channels without a branch; channels past nc carry 0 into slab rows nobody flushes
                        unsigned long long *l = slab + rr0 * W + ixw;
                        const float w0 = wq[0] * fx_scale, w1 = wq[1] * fx_scale;
                        const float w2 = wq[2] * fx_scale, w3 = wq[3] * fx_scale;
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            unsigned long long *lc = l + c * slab_c;
                            atomicAdd(lc, bwd_to_fixed(gvf[c] * w0));
                            atomicAdd(lc + 1, bwd_to_fixed(gvf[c] * w1));
                            atomicAdd(lc + W, bwd_to_fixed(gvf[c] * w2));
                            atomicAdd(lc + W + 1, bwd_to_fixed(gvf[c] * w3));
                        }
                        continue;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (wq[q] == 0.0f) continue;  // out-of-bounds (or weightless) tap
                        const int py = iyn + (q >> 1), px = ixw + (q & 1);
                        const int rr = py - y0;  // >= 0: y0 <= the window's first row
                        if (BWD_ABLATE(2)) continue;
                        if (rr < rows && !plain) {
                            unsigned long long *l = slab + rr * W + px;
                            const float ws = wq[q] * fx_scale;
#pragma unroll
                            for (int c = 0; c < CW; ++c) atomicAdd(l + c * slab_c, bwd_to_fixed(gvf[c] * ws));
                        } else {  // taller than the slab window (rare), or a non-finite pass
                            float *gl = gf + (size_t)c0 * HW + (size_t)py * W + px;
#pragma unroll
                            for (int c = 0; c < CW; ++c)
                                if (c < nc) atomicAdd(gl + (size_t)c * HW, gvf[c] * wq[q]);
                        }
                    }
                }
                __syncthreads();
                flush();
                __syncthreads();
            }
        } else {
            // ---- cur map: per window, sum over this lane's planes in registers, one scatter -
            for (int wi_ = 0; wi_ < nwin; ++wi_) {
                const int pw = __builtin_amdgcn_readfirstlane(wins[1 + 4 * wi_]);
                const int pe = __builtin_amdgcn_readfirstlane(wins[2 + 4 * wi_]);
                y0 = __builtin_amdgcn_readfirstlane(wins[3 + 4 * wi_]);
                top = __builtin_amdgcn_readfirstlane(wins[4 + 4 * wi_]);
                float acc[9][CW];
#pragma unroll
                for (int cell = 0; cell < 9; ++cell)
#pragma unroll
                    for (int c = 0; c < CW; ++c) acc[cell][c] = 0.0f;
                for (int k0 = 0; k0 < BWD_MAXP / BWD_GROUPS; k0 += VB) {
                    if (grp + k0 * BWD_GROUPS > pe) break;  // uniform per wave (grp is)
                    T gv[VB][CW];
                    load_block(k0, gv);
#pragma unroll
                    for (int k = 0; k < VB; ++k) {
                        const int p = grp + (k0 + k) * BWD_GROUPS;
                        if (p < pw || p > pe) continue;
                        const uint32_t f = fpT[p * BWD_PTS + pt];
                        if (!f) continue;
                        float gvf[CW];
#pragma unroll
                        for (int c = 0; c < CW; ++c) gvf[c] = elem<T>::load(gv[k][c]);
                        const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
                        const float fw = fwT[p * BWD_PTS + pt], fn = fnT[p * BWD_PTS + pt];
                        const float cw = (f & (1u << 27)) ? 1.0f - fw : 0.0f, ce = (f & (1u << 28)) ? fw : 0.0f;
                        const float rn = (f & (1u << 29)) ? 1.0f - fn : 0.0f, rs = (f & (1u << 30)) ? fn : 0.0f;
                        const int ox = ixw - bx, oy = iyn - by;  // 0 or 1 while the footprint stays in the block
                        if ((unsigned)ox <= 1u && (unsigned)oy <= 1u) {
                            const bool xlo = ox == 0, ylo = oy == 0;
                            const float cx[3] = {xlo ? cw : 0.0f, xlo ? ce : cw, xlo ? 0.0f : ce};
                            const float ry[3] = {ylo ? rn : 0.0f, ylo ? rs : rn, ylo ? 0.0f : rs};
#pragma unroll
                            for (int r = 0; r < 3; ++r)
#pragma unroll
                                for (int q = 0; q < 3; ++q) {
                                    const float wgt = ry[r] * cx[q];
#pragma unroll
                                    for (int c = 0; c < CW; ++c)
                                        acc[3 * r + q][c] = __builtin_fmaf(gvf[c], wgt, acc[3 * r + q][c]);
                                }
                        } else {  // the footprint left the lane's block (general poses): straight to memory
                            const float wq[4] = {rn * cw, rn * ce, rs * cw, rs * ce};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (wq[q] == 0.0f) continue;
                                float *gl = gf + (size_t)c0 * HW + (size_t)(iyn + (q >> 1)) * W + ixw + (q & 1);
#pragma unroll
                                for (int c = 0; c < CW; ++c)
                                    if (c < nc) atomicAdd(gl + (size_t)c * HW, gvf[c] * wq[q]);
                            }
                        }
                    }
                }
                // scale from the register sums themselves
                unsigned m = 0u;
#pragma unroll
                for (int cell = 0; cell < 9; ++cell)
#pragma unroll
                    for (int c = 0; c < CW; ++c) m = max(m, __float_as_uint(acc[cell][c]) & 0x7fffffffu);
                const unsigned mb = wg_max(m);
                if (mb != 0u) {
                    const bool pl = (mb >> 23) == 0xffu;
                    if (!pl) bwd_scale(mb, fx_scale, fx_inv);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int py = by + r, rr = py - y0;
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const int px = bx + q;
#pragma unroll
                            for (int c = 0; c < CW; ++c) {
                                const float v = acc[3 * r + q][c];
                                if (c >= nc || v == 0.0f || BWD_ABLATE(2)) continue;  // (out-of-bounds cells carry 0)
                                if (!pl && rr >= 0 && rr < rows)
                                    atomicAdd(slab + c * slab_c + rr * W + px, bwd_to_fixed(v * fx_scale));
                                else
                                    atomicAdd(gf + (size_t)(c0 + c) * HW + (size_t)py * W + px, v);
                            }
                        }
                    }
                    __syncthreads();
                    if (!pl) flush();
                    __syncthreads();
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// parity aid: normalised grids of sample b
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_grid_kernel(SweepGeom g, int b,
                                                         const float *__restrict__ depths,
                                                         const float *__restrict__ P,
                                                         const float *__restrict__ Pinv,
                                                         const float *__restrict__ Tm,
                                                         float *__restrict__ cur_grid,
                                                         float *__restrict__ prev_grid)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= g.N) return;
    const int hw = g.h_out * g.w_out;
    const int d = (int)(n / hw);
    const int rem = (int)(n - (long long)d * hw);
    const int hi = rem / g.w_out;
    const int wi = rem - hi * g.w_out;
    float cx, cy, px, py, norm[4];
    sweep_point(g, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, cx, cy, px, py, norm);
    cur_grid[2 * n] = norm[0];
    cur_grid[2 * n + 1] = norm[1];
    prev_grid[2 * n] = norm[2];
    prev_grid[2 * n + 1] = norm[3];
}

// ---------------------------------------------------------------------------
// camera matrices on the device: pad cam2img to the 4x4 points_img2cam / points_cam2img build
// (utils.py:199-203, 239-240) and invert it in fp32 (Gauss-Jordan with partial pivoting), one
// lane per sample.  Replaces a host round trip per build_dfm_cost call when the intrinsics are
// device tensors (dfm_backbone.py:151-154).
// ---------------------------------------------------------------------------
__global__ void camera_prepare_kernel(const float *__restrict__ cam2img, int rows, int cols,
                                      int batch, float *__restrict__ P, float *__restrict__ Pinv)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    float m[4][8];
    const float *src = cam2img + (size_t)b * rows * cols;
    const int use_rows = rows == 4 ? 3 : rows;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = i == j ? 1.0f : 0.0f;
            if (i < use_rows && j < cols) v = src[i * cols + j];
            m[i][j] = v;
            m[i][j + 4] = i == j ? 1.0f : 0.0f;
            P[b * 16 + i * 4 + j] = v;
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        float best = fabsf(m[c][c]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > c && fabsf(m[r][c]) > best) { best = fabsf(m[r][c]); piv = r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r == piv && r != c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = m[c][j]; m[c][j] = m[r][j]; m[r][j] = t; }
            }
        const float inv = 1.0f / m[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[c][j] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const float f = m[r][c];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[r][j] = m[r][j] - f * m[c][j];
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Pinv[b * 16 + i * 4 + j] = m[i][j + 4];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int check_desc(const dfm_sweep_desc *d)
{
    if (!d) return fail(DFM_ERR_INVALID_ARG, "desc is NULL%s");
    if (d->batch <= 0 || d->channels <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->num_depths <= 0 ||
        d->h_out <= 0 || d->w_out <= 0)
        return fail(DFM_ERR_INVALID_ARG, "non-positive size in dfm_sweep_desc%s");
    if (d->dtype != DFM_F32 && d->dtype != DFM_BF16)
        return fail(DFM_ERR_UNSUPPORTED, "dtype must be DFM_F32 or DFM_BF16%s");
    if ((long long)d->h_in * d->w_in >= (1ll << 31) / 64)
        return fail(DFM_ERR_UNSUPPORTED, "feature map too large for 32-bit tap offsets%s");
    if (d->batch > 32767) return fail(DFM_ERR_UNSUPPORTED, "batch > 32767%s");
    return DFM_OK;
}

SweepGeom make_geom(const dfm_sweep_desc *d)
{
    SweepGeom g;
    g.C = d->channels;
    g.h_in = d->h_in;
    g.w_in = d->w_in;
    g.D = d->num_depths;
    g.h_out = d->h_out;
    g.w_out = d->w_out;
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    g.nblk = (d->channels + CB - 1) / CB;
    g.flip = d->flip;
    g.fsf = d->feat_sample_factor;
    g.csf = d->cost_sample_factor;
    g.scale = d->img_scale_factor;
    g.crop_x = d->crop_x;
    g.crop_y = d->crop_y;
    g.org_w = d->org_w;
    g.N = (long long)d->num_depths * d->h_out * d->w_out;
    return g;
}

size_t blocked_bytes(const dfm_sweep_desc *d)
{
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    const size_t nblk = (d->channels + CB - 1) / CB;
    size_t one = (size_t)d->batch * nblk * d->h_in * d->w_in * 16;
    return (one + 255) & ~(size_t)255;
}

// one spill flag per (tile, sample) of the LDS kernel at its smallest tile
// the pack pass's flags: one word per (map, sample, feature row)
size_t row_flag_bytes(const dfm_sweep_desc *d)
{
    return (((size_t)2 * d->batch * d->h_in * 4) + 255) & ~(size_t)255;
}

size_t flag_bytes(const dfm_sweep_desc *d)
{
    const int V = d->dtype == DFM_BF16 ? 8 : 4;
    const long long hw = (long long)d->h_out * d->w_out;
    const long long bands = (hw + 7 + 63 + 64ll * V - 1) / (64ll * V);  // smallest tile (one wave per plane), coarsest alignment
    const long long nblk = (d->channels + V - 1) / V;  // worst case: one block per group
    // int32 counter + one int32 tile id per (worst-case) tile
    return ((size_t)(1 + bands * d->num_depths * 2 * d->batch * nblk) * 4 + 255) & ~(size_t)255;
}

SweepFast make_fast(const dfm_sweep_desc *d)
{
    SweepFast fast;
    fast.scale_is_one = d->img_scale_factor == 1.0f;
    int e = 0;
    const float m = frexpf(d->feat_sample_factor, &e);
    fast.fsf_pow2 = (m == 0.5f) && e > -60 && e < 60;
    fast.inv_fsf = 1.0f / d->feat_sample_factor;
    return fast;
}

// caller options (0 = default) -> a complete launch description
int resolve(const dfm_sweep_desc *d, const dfm_sweep_opts *o, Launch &L)
{
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    L.kernel = o ? o->kernel : 0;
    L.lanes = o && o->lanes_per_workgroup ? o->lanes_per_workgroup : 256;
    L.lds_kib = o && o->lds_kib ? o->lds_kib : 0;  // 0: resolved below (depends on the body)
    L.bpg = o && o->blocks_per_group ? o->blocks_per_group : (1 << 20);
    L.planes = o && o->planes_per_workgroup ? o->planes_per_workgroup : 2;
    L.band_chunk = o && o->bands_per_chunk ? o->bands_per_chunk : 1;
    L.ppl = o && o->points_per_lane ? o->points_per_lane : CB;
    L.align = o && o->store_align_points ? o->store_align_points : 64;
    if (L.align != 8 && L.align != 16 && L.align != 32 && L.align != 64)
        return fail(DFM_ERR_INVALID_ARG, "opts: store_align_points in {8,16,32,64}%s");
    L.pair = o && o->pair_stores ? (o->pair_stores == 1) : 1;  // 0 default (on), 1 on, 2 off
    // 0 default (matrix core), 1 matrix core, 2 VALU
    if (o && (o->unpack < 0 || o->unpack > 2)) return fail(DFM_ERR_INVALID_ARG, "opts: unpack must be 0, 1 or 2%s");
    L.unpack = o && o->unpack ? (o->unpack == 1) : 1;
    L.pipe = o && o->pipeline ? o->pipeline : 2;
    if (L.pipe != 1 && L.pipe != 2) return fail(DFM_ERR_INVALID_ARG, "opts: pipeline must be 0, 1 or 2%s");
    if (!L.lds_kib) L.lds_kib = L.pipe >= 2 ? 80 : 52;
    if (L.kernel < 0 || L.kernel > 5) return fail(DFM_ERR_INVALID_ARG, "opts: kernel must be 0..5%s");
    if (L.lanes != 128 && L.lanes != 256 && L.lanes != 512 && L.lanes != 1024)
        return fail(DFM_ERR_INVALID_ARG, "opts: lanes_per_workgroup in {128,256,512,1024}%s");
    if (L.lds_kib < 4 || L.lds_kib > 160 || L.bpg < 1 || L.planes < 1 || L.band_chunk < 1)
        return fail(DFM_ERR_INVALID_ARG,
                    "opts: 4 <= lds_kib <= 160, blocks_per_group / planes / bands_per_chunk >= 1%s");
    if (L.ppl != CB && !(L.ppl == 4 && CB == 8 && L.lanes >= 256))
        return fail(DFM_ERR_INVALID_ARG,
                    "opts: points_per_lane is 16/sizeof(T), or 4 for bf16 with >= 256 lanes%s");
    if (L.lanes == 1024 && L.ppl != 4)
        return fail(DFM_ERR_INVALID_ARG, "opts: 1024 lanes need points_per_lane = 4%s");
    return DFM_OK;
}

template <typename T, int NT, int V>
int launch_tiles(int which, const dfm_sweep_desc *d, const SweepGeom &g, const Launch &L,
                 const uint4 *cur_blk, const uint4 *prev_blk, const float *depths, const float *P,
                 const float *Pinv, const float *Tm, T *out, int *spill_list, hipStream_t st)
{
    const int lds_bytes = L.lds_kib * 1024;
    const int bpg = which == 2 ? std::min(L.bpg, g.nblk) : g.nblk;
    const int groups = (g.nblk + bpg - 1) / bpg;
    const long long hw = (long long)g.h_out * g.w_out;
    TileGrid tg;
    tg.batch = d->batch;
    tg.planes = std::max(1, std::min(L.planes, NT / 64));
    while ((NT / 64) % tg.planes) --tg.planes;  // whole waves per plane
    tg.dgroups = (g.D + tg.planes - 1) / tg.planes;
    const long long per_plane = (long long)(NT / tg.planes) * V;  // points per plane and tile
    // tile boundaries at multiples of L.align points where the tile shape allows it (a tile holds
    // per_plane points of a plane; per_plane is a multiple of 256 except for test-sized workgroups)
    tg.align = (int)std::min<long long>(L.align, per_plane & -per_plane);
    const long long am = tg.align - 1;
    // a plane's tiles span at most hw + 7 points (its first vector may start 7 points early) plus
    // the am points between the aligned base of the cuts and that vector
    tg.bands = (int)((hw + 7 + am + per_plane - 1) / per_plane);
    tg.band_pts = (int)((((hw + 7 + am + tg.bands - 1) / tg.bands) + am) & ~am);
    tg.blocks_per_group = bpg;
    tg.band_chunk = std::max(1, std::min(L.band_chunk, tg.bands));
    tg.trace = nullptr;
    tg.ablate = 0;
    tg.pair_stores = L.pair;
    tg.mfma_unpack = L.unpack;
    tg.flags = (const int *)((const char *)spill_list + 2 * flag_bytes(d));  // set by pack_blocked_kernel
#ifdef DFM_DEBUG_HOOKS
    tg.trace = g_trace;
    {
        const char *ab = getenv("DFM_ABLATE");  // perf experiments only
        tg.ablate = ab ? atoi(ab) : 0;
    }
#endif
    const long long nchunks = (tg.bands + tg.band_chunk - 1) / tg.band_chunk;
    const long long nb = nchunks * tg.band_chunk * tg.dgroups * 2 * d->batch * groups;
    if (nb > 2147483647ll) return fail(DFM_ERR_UNSUPPORTED, "too many lattice points%s");
    if ((size_t)(nb + 1) * 4 > flag_bytes(d)) return fail(DFM_ERR_WORKSPACE, "spill list too small%s");
    int rc = DFM_OK;
    const SweepFast fast = make_fast(d);
    if (which == 2) {
        int *spill2 = (int *)((char *)spill_list + flag_bytes(d));  // (both counters: sweep_reset_kernel, launch_fwd)
        constexpr bool CAN_MX = sizeof(T) == 2 && V == 8;
        if (L.pipe >= 2 && CAN_MX && L.unpack) {
            // bf16, 8 points per lane: the build that unpacks the taps with the matrix core
            if constexpr (CAN_MX) {
                const void *kern = (const void *)sweep_tile_kernel<T, NT, true, V, true, true>;
                rc = ensure_dynamic_lds(kern, PIPE_LDS_BYTES);
                if (rc != DFM_OK) return rc;
                hipLaunchKernelGGL((sweep_tile_kernel<T, NT, true, V, true, true>), dim3((unsigned)nb), dim3(NT),
                                   PIPE_LDS_BYTES, st, g, fast, tg, std::min(lds_bytes, PIPE_LDS_BYTES) / 16,
                                   cur_blk, prev_blk, depths, P, Pinv, Tm, out, spill_list);
            }
        } else if (L.pipe >= 2) {
            // fixed 78 KiB (two buffers at a compile-time stride); lds_kib only lowers the
            // budget a tile's rows are checked against (tests force spills that way)
            const void *kern = (const void *)sweep_tile_kernel<T, NT, true, V, true>;
            rc = ensure_dynamic_lds(kern, PIPE_LDS_BYTES);
            if (rc != DFM_OK) return rc;
            hipLaunchKernelGGL((sweep_tile_kernel<T, NT, true, V, true>), dim3((unsigned)nb), dim3(NT),
                               PIPE_LDS_BYTES, st, g, fast, tg, std::min(lds_bytes, PIPE_LDS_BYTES) / 16,
                               cur_blk, prev_blk, depths, P, Pinv, Tm, out, spill_list);
        } else {
            const void *kern = (const void *)sweep_tile_kernel<T, NT, true, V>;
            rc = ensure_dynamic_lds(kern, lds_bytes);
            if (rc != DFM_OK) return rc;
            hipLaunchKernelGGL((sweep_tile_kernel<T, NT, true, V>), dim3((unsigned)nb), dim3(NT),
                               lds_bytes, st, g, fast, tg, lds_bytes / 16, cur_blk, prev_blk, depths, P,
                               Pinv, Tm, out, spill_list);
        }
        // The pipelined kernels with 16-byte stores take every second chance in place (serial body over both
        // buffers, then direct taps): nothing is ever queued and the two launches below would be empty.
        const bool queues = !(L.pipe >= 2 && V * sizeof(T) == 16);
        // tiles whose rows exceeded the LDS budget: once more with 144 KiB of
        // LDS, split by channel block; then direct taps for what is left
        if (queues) {
            // (a test-sized budget below 16 KiB keeps its size, so that the direct pass stays covered)
            const int BIG = L.lds_kib < 16 ? lds_bytes : 144 * 1024;
            const void *rk = (const void *)sweep_respill_kernel<T, NT, V>;
            rc = ensure_dynamic_lds(rk, BIG);
            if (rc != DFM_OK) return rc;
            const int rgroups = std::min(8, g.nblk);
            const int rbpg = (g.nblk + rgroups - 1) / rgroups;
            hipLaunchKernelGGL((sweep_respill_kernel<T, NT, V>), dim3(1024), dim3(NT), BIG, st, g, fast,
                               tg, BIG / 16, rgroups, rbpg, cur_blk, prev_blk, depths, P, Pinv, Tm, out,
                               (const int *)spill_list, spill2);
        }
        if (queues)
            hipLaunchKernelGGL((sweep_spill_kernel<T, NT, V>), dim3(512), dim3(NT), 16, st, g, fast, tg,
                               cur_blk, prev_blk, depths, P, Pinv, Tm, out, spill2);
        if (g.D > 1 && hw % 8 != 0)
            hipLaunchKernelGGL(sweep_patch_kernel<T>, dim3(g.D - 1, 2, d->batch), dim3(256), 0, st,
                               g, fast, tg.align, cur_blk, prev_blk, depths, P, Pinv, Tm, out);
    } else {
        hipLaunchKernelGGL((sweep_tile_kernel<T, NT, false, V>), dim3((unsigned)nb), dim3(NT), 16,
                           st, g, fast, tg, 0, cur_blk, prev_blk, depths, P, Pinv, Tm, out,
                           (int *)nullptr);
    }
    return DFM_OK;
}

template <typename T>
int launch_fwd(const dfm_sweep_desc *d, const Launch &L, const void *cur, const void *prev,
               const float *depths, const float *P, const float *Pinv, const float *Tm, void *out,
               void *ws, hipStream_t st)
{
    const SweepGeom g = make_geom(d);
    const int HW = d->h_in * d->w_in;
    // strided sweeps (cost_sample_factor >= 2: config K) in the reference layout: pixel-major taps
    // + an LDS transpose (kernel 4, plane_sweep_cl.hip) instead of the direct tile kernel (3)
    // (kernel 5: the same with the depth axis walked per wave -- what 0 picks where it applies; 4 pins the
    // per-plane kernel)
    if ((L.kernel == 4 || L.kernel == 5 || (L.kernel == 0 && d->cost_sample_factor >= 1.5f)) &&
        sweep_clt_supported(d, out)) {
        // (sweep_clt_launch reports which body it launched: 5 walking, 4 per plane)
        return sweep_clt_launch(d, cur, prev, depths, P, Pinv, Tm, out, ws, (void *)st, false, L.kernel != 4);
    }
    uint4 *cur_blk = (uint4 *)ws;
    uint4 *prev_blk = (uint4 *)((char *)ws + blocked_bytes(d));
    dim3 pg((HW + 256 * PACK_PPL - 1) / (256 * PACK_PPL), g.nblk, 2 * d->batch);
    int *feat_flags = (int *)((char *)ws + 2 * blocked_bytes(d) + 2 * flag_bytes(d));
    // the whole build is timed (dfm_profile_begin): pack pass, tile kernel, second-chance / direct / patch passes
    const bool timed = profile_mark(st, false);
    // the pack pass's row flags and the two spill counters: one launch (three hipMemsetAsync were three)
    hipLaunchKernelGGL(sweep_reset_kernel, dim3(1), dim3(256), 0, st, feat_flags, 2 * d->batch * d->h_in,
                       (int *)((char *)ws + 2 * blocked_bytes(d)),
                       (int *)((char *)ws + 2 * blocked_bytes(d) + flag_bytes(d)));
    hipLaunchKernelGGL(pack_blocked_kernel<T>, pg, dim3(256), 0, st, (const T *)cur, (const T *)prev,
                       cur_blk, prev_blk, d->batch, g.C, HW, g.nblk, feat_flags, d->w_in);
    constexpr int CB = elem<T>::CB;
    // the tile kernels store one aligned vector of V points per channel: every channel
    // plane (N elements) has to start 16-byte aligned.
    // Dense sampling (about one feature pixel per lattice step) reuses staged rows
    // well; a strided sweep (cost_sample_factor >= 2) would stage mostly unused
    // pixels, so it takes the same tile kernel with direct taps (3).
    const bool vec_ok = g.N % CB == 0 && ((uintptr_t)out & 15) == 0;
    int which = !vec_ok ? 1 : (d->cost_sample_factor < 1.5f ? 2 : 3);
    if (L.kernel == 1 || (L.kernel == 3 && vec_ok)) which = L.kernel;
    if (L.kernel == 2 && vec_ok) which = 2;
    int rc = DFM_OK;
    if (which == 1) {
        const long long nb = (g.N + 255) / 256;
        if (nb > 2147483647ll) return fail(DFM_ERR_UNSUPPORTED, "too many lattice points%s");
        dim3 grid((unsigned)nb, d->batch);
        hipLaunchKernelGGL(sweep_gather_kernel<T>, grid, dim3(256), 0, st, g, cur_blk, prev_blk,
                           depths, P, Pinv, Tm, (T *)out);
    } else {
        int *spill_list = (int *)((char *)ws + 2 * blocked_bytes(d));
#define DFM_TILES(NT, V)                                                                            \
    rc = launch_tiles<T, NT, V>(which, d, g, L, cur_blk, prev_blk, depths, P, Pinv, Tm, (T *)out,   \
                                spill_list, st)
        if (L.ppl == CB) {
            if (L.lanes == 128) DFM_TILES(128, CB);
            else if (L.lanes == 256) DFM_TILES(256, CB);
            else if (L.lanes == 512) DFM_TILES(512, CB);
            else if constexpr (CB == 4) DFM_TILES(1024, 4);
        } else {
            if constexpr (CB == 8) {
                if (L.lanes == 256) DFM_TILES(256, 4);
                else if (L.lanes == 512) DFM_TILES(512, 4);
                else DFM_TILES(1024, 4);
            }
        }
#undef DFM_TILES
    }
    if (timed) profile_mark(st, true);
    if (rc != DFM_OK) return rc;
    g_last_kernel = which;
    HIP_TRY(hipGetLastError());
    return DFM_OK;
}

int run_fwd(const dfm_sweep_desc *desc, const Launch &L, const void *cur, const void *prev,
            const float *depths, const float *cam2img, const float *cam2img_inv,
            const float *cur2prev, void *out, void *workspace, hipStream_t st)
{
    if (desc->dtype == DFM_F32)
        return launch_fwd<float>(desc, L, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out,
                                 workspace, st);
    return launch_fwd<bf16_t>(desc, L, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out,
                              workspace, st);
}

// blocked maps + spill lists (tile kernels), or the pixel-major maps of the strided-sweep kernel
size_t fwd_workspace_bytes(const dfm_sweep_desc *desc)
{
    // blocked maps + the two spill lists + the feature-flag word of the pack pass
    return std::max(2 * blocked_bytes(desc) + 2 * flag_bytes(desc) + row_flag_bytes(desc),
                    sweep_clt_workspace_bytes(desc));
}

int check_fwd_args(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                   const float *depths, const float *cam2img, const float *cam2img_inv,
                   const float *cur2prev, void *out, void *workspace, size_t workspace_bytes)
{
    int rc = check_desc(desc);
    if (rc != DFM_OK) return rc;
    if (!cur || !prev || !depths || !cam2img || !cam2img_inv || !cur2prev || !out)
        return fail(DFM_ERR_INVALID_ARG, "NULL device pointer%s");
    if (!workspace || workspace_bytes < fwd_workspace_bytes(desc))
        return fail(DFM_ERR_WORKSPACE, "workspace smaller than dfm_plane_sweep_workspace_bytes%s");
    if (((uintptr_t)workspace & 15) || ((uintptr_t)out & 1))
        return fail(DFM_ERR_INVALID_ARG, "workspace must be 16-byte aligned%s");
    return DFM_OK;
}

int tune_key(const dfm_sweep_desc *d, TuneKey &k)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const int v[9] = {dev, d->batch, d->channels, d->h_in, d->w_in, d->num_depths, d->h_out,
                      d->w_out, d->dtype};
    memcpy(k.v, v, sizeof(v));
    return DFM_OK;
}

// would the default dispatch take the LDS tile kernel (the only one with schedules to tune)?
bool takes_lds_tiles(const dfm_sweep_desc *d, const void *out)
{
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    const long long N = (long long)d->num_depths * d->h_out * d->w_out;
    return N % CB == 0 && ((uintptr_t)out & 15) == 0 && d->cost_sample_factor < 1.5f;
}

}  // namespace

extern "C" {

DFM_API int dfm_version(void) { return 3; }
DFM_API const char *dfm_last_error(void) { return g_err; }
DFM_API int dfm_plane_sweep_last_kernel(void) { return g_last_kernel; }
DFM_API int dfm_plane_sweep_bwd_last_kernel(void) { return g_last_bwd_kernel.load(); }
#ifdef DFM_DEBUG_HOOKS
// debug builds only (not in dfm_hip.h): device buffer of 64 x u64 per traced workgroup
DFM_API void dfm_debug_set_trace(void *buf) { g_trace = (unsigned long long *)buf; }
#endif

DFM_API int dfm_profile_begin(int max_launches)
{
    if (max_launches <= 0 || max_launches > 65536)
        return fail(DFM_ERR_INVALID_ARG, "max_launches out of range%s");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.assign(2 * (size_t)max_launches, nullptr);
    for (auto &e : g_prof.ev) HIP_TRY(hipEventCreate(&e));
    g_prof.used = 0;
    g_prof.on = true;
    return DFM_OK;
}

DFM_API int dfm_profile_end(double *total_ms, int *launches)
{
    if (!total_ms || !launches) return fail(DFM_ERR_INVALID_ARG, "NULL output%s");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    double sum = 0.0;
    for (int i = 0; i + 1 < g_prof.used; i += 2) {
        HIP_TRY(hipEventSynchronize(g_prof.ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = g_prof.used / 2;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.used = 0;
    return DFM_OK;
}

}  // extern "C"

// shared with the other translation units
int dfm::sweep_check_desc(const dfm_sweep_desc *d) { return check_desc(d); }
dfm::SweepGeom dfm::sweep_make_geom(const dfm_sweep_desc *d) { return make_geom(d); }
dfm::SweepFast dfm::sweep_make_fast(const dfm_sweep_desc *d) { return make_fast(d); }
void dfm::sweep_set_last_kernel(int which) { g_last_kernel = which; }
void dfm::sweep_set_last_bwd_kernel(int which) { g_last_bwd_kernel.store(which); }
bool dfm::profile_mark(void *stream, bool stop)
{
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!stop) {
        if (!(g_prof.on && g_prof.used + 2 <= (int)g_prof.ev.size())) return false;
        (void)hipEventRecord(g_prof.ev[g_prof.used], st);
        return true;
    }
    if (!(g_prof.on && g_prof.used + 2 <= (int)g_prof.ev.size())) return false;
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
    g_prof.used += 2;
    return true;
}

extern "C" {

DFM_API int dfm_camera_prepare(const float *cam2img, int32_t rows, int32_t cols, int32_t batch,
                               float *cam2img_4x4, float *cam2img_inv, void *stream)
{
    if (!cam2img || !cam2img_4x4 || !cam2img_inv) return fail(DFM_ERR_INVALID_ARG, "NULL device pointer%s");
    if ((rows != 3 && rows != 4) || (cols != 3 && cols != 4) || batch <= 0)
        return fail(DFM_ERR_INVALID_ARG, "cam2img must be (B,3|4,3|4) with B >= 1%s");
    hipLaunchKernelGGL(camera_prepare_kernel, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       cam2img, rows, cols, batch, cam2img_4x4, cam2img_inv);
    HIP_TRY(hipGetLastError());
    return DFM_OK;
}

DFM_API size_t dfm_plane_sweep_workspace_bytes(const dfm_sweep_desc *desc)
{
    if (check_desc(desc) != DFM_OK) return 0;
    return fwd_workspace_bytes(desc);
}

DFM_API int dfm_plane_sweep_fwd_opts(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream,
                                     const dfm_sweep_opts *opts)
{
    int rc = check_fwd_args(desc, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace,
                            workspace_bytes);
    if (rc != DFM_OK) return rc;
    Launch L;
    if (opts) {
        rc = resolve(desc, opts, L);
    } else {
        // the schedule dfm_plane_sweep_autotune cached for this device and shape, else defaults
        TuneKey k;
        rc = tune_key(desc, k);
        if (rc != DFM_OK) return rc;
        dfm_sweep_opts tuned;
        bool have = false;
        {
            std::lock_guard<std::mutex> lk(g_tune_mu);
            auto it = g_tuned.find(k);
            if (it != g_tuned.end()) { tuned = it->second; have = true; }
        }
        if (!have && takes_lds_tiles(desc, out)) {
            // First launch of a large volume on this device: time the candidate schedules once
            // (DFM_AUTOTUNE=0 disables; never inside a stream capture).  `out` ends up valid.
            const int esz = desc->dtype == DFM_BF16 ? 2 : 4;
            const double vol = 2.0 * desc->batch * desc->channels * (double)desc->num_depths *
                               desc->h_out * desc->w_out * esz;
            const char *env = getenv("DFM_AUTOTUNE");
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing((hipStream_t)stream, &cs);
            if (vol >= 1.0e9 && !(env && env[0] == '0') && cs == hipStreamCaptureStatusNone)
                return dfm_plane_sweep_autotune(desc, cur, prev, depths, cam2img, cam2img_inv,
                                                cur2prev, out, workspace, workspace_bytes, stream,
                                                nullptr);
        }
        rc = resolve(desc, have ? &tuned : nullptr, L);
    }
    if (rc != DFM_OK) return rc;
    return run_fwd(desc, L, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace,
                   (hipStream_t)stream);
}

DFM_API int dfm_plane_sweep_fwd(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                const float *depths, const float *cam2img,
                                const float *cam2img_inv, const float *cur2prev, void *out,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    return dfm_plane_sweep_fwd_opts(desc, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out,
                                    workspace, workspace_bytes, stream, nullptr);
}

DFM_API int dfm_plane_sweep_autotune(const dfm_sweep_desc *desc, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream,
                                     dfm_sweep_opts *chosen)
{
    // The workgroup order trades HBM write locality against L2 reuse of the staged rows, and the
    // tile shape trades waves per CU against registers per lane; which side wins depends on the
    // part the process landed on (profiles/archive/r01_store_microbench3.txt, r02_*): time the candidates
    // on the caller's own tensors and remember the fastest for this (device, shape).  Synchronous.
    int rc = check_fwd_args(desc, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace,
                            workspace_bytes);
    if (rc != DFM_OK) return rc;
    TuneKey key;
    rc = tune_key(desc, key);
    if (rc != DFM_OK) return rc;
    // candidates: 8 points per lane x 256 lanes, 4 points per lane x 512 lanes (bf16, lane pairs storing
    // 16-byte vectors); both with the pipelined body, band cuts on 128-byte boundaries and
    // bands_per_chunk 1.  Since round 4 the two are within 0-4 % of each other on every part seen
    // (profiles/archive/r04_c2..c4_*): with whole 64-byte writes the parts that ran round 3's binary at 0.48 of
    // the roofline run either shape at 0.59-0.60.  The re-fetching schedules are no candidates:
    // bands_per_chunk 29 moved 41 GB and 15 moves 37 GB of HBM traffic per N* launch against 28 GB
    // (profiles/archive/r02_nstar_traffic.json), and a short timing window once picked 15 on a part where it
    // then ran 20 % slower in steady state (profiles/archive/r02_c43_bench_default_mispick.json); with aligned
    // cuts chunks of 2 / 4 measure 3-8 % slower (r04_c4).  bands_per_chunk stays in dfm_sweep_opts.
    std::vector<dfm_sweep_opts> cand;
    {
        dfm_sweep_opts o;
        memset(&o, 0, sizeof(o));
        o.bands_per_chunk = 1;
        cand.push_back(o);
        if (desc->dtype == DFM_BF16) {
            o.lanes_per_workgroup = 512;
            o.points_per_lane = 4;
            cand.push_back(o);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    dfm_sweep_opts best = cand[0];
    if (takes_lds_tiles(desc, out)) {
        // Timing window: warm every candidate up (code objects, clocks), then ROUNDS round-robin passes
        // of PER launches per candidate and the MEDIAN round -- 8 timed launches per candidate, ~0.1 s
        // for both at N*.  (History: one cold launch per candidate picked a schedule 20 % slower than
        // the best; 5 x 3 launches with the minimum still mis-picked once between candidates 5 % apart;
        // round 3 spent 8 x 4 launches per candidate, 0.4 s of first-call latency inside a thin ABI.
        // Now the candidates are within a few per cent on every part seen, a wrong pick costs that, and
        // the window is sized to that.)  One-time cost per (device, shape).
        constexpr int ROUNDS = 4, PER = 2;
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        std::vector<Launch> Ls(cand.size());
        for (size_t i = 0; i < cand.size() && rc == DFM_OK; ++i) rc = resolve(desc, &cand[i], Ls[i]);
        for (size_t i = 0; i < cand.size() && rc == DFM_OK; ++i)  // loads every code object, warms up
            rc = run_fwd(desc, Ls[i], cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace, st);
        for (int w = 0; w < 2 && rc == DFM_OK; ++w)
            rc = run_fwd(desc, Ls[0], cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace, st);
        std::vector<float> tmin(cand.size(), 3.0e38f);
        std::vector<std::vector<float>> rounds(cand.size());
        for (int r = 0; r < ROUNDS && rc == DFM_OK; ++r)
            for (size_t i = 0; i < cand.size() && rc == DFM_OK; ++i) {
                (void)hipEventRecord(e0, st);
                for (int k = 0; k < PER && rc == DFM_OK; ++k)
                    rc = run_fwd(desc, Ls[i], cur, prev, depths, cam2img, cam2img_inv, cur2prev, out,
                                 workspace, st);
                (void)hipEventRecord(e1, st);
                if (rc == DFM_OK && hipEventSynchronize(e1) != hipSuccess)
                    rc = fail(DFM_ERR_HIP, "autotune: event sync failed%s");
                float ms = 0.0f;
                if (rc == DFM_OK) (void)hipEventElapsedTime(&ms, e0, e1);
                if (rc == DFM_OK) rounds[i].push_back(ms);
            }
        for (size_t i = 0; i < cand.size() && rc == DFM_OK; ++i) {
            std::sort(rounds[i].begin(), rounds[i].end());
            tmin[i] = rounds[i][rounds[i].size() / 2];  // (the name is historical: the median round)
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (rc != DFM_OK) return rc;
        // a schedule with bands_per_chunk > 1 re-fetches staged rows (1.3x the HBM traffic): it has
        // to win by more than timing noise to be taken
        size_t bi = 0;
        for (size_t i = 1; i < cand.size(); ++i) {
            const float margin = cand[i].bands_per_chunk > 1 && cand[bi].bands_per_chunk <= 1 ? 0.985f : 1.0f;
            if (tmin[i] < tmin[bi] * margin) bi = i;
        }
        best = cand[bi];
    } else {
        Launch L;
        rc = resolve(desc, nullptr, L);
        if (rc == DFM_OK)
            rc = run_fwd(desc, L, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace, st);
        if (rc != DFM_OK) return rc;
    }
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tuned[key] = best;
    }
    if (chosen) *chosen = best;
    return DFM_OK;
}

DFM_API int dfm_plane_sweep_tuning(const dfm_sweep_desc *desc, dfm_sweep_opts *opts)
{
    int rc = check_desc(desc);
    if (rc != DFM_OK) return rc;
    if (!opts) return fail(DFM_ERR_INVALID_ARG, "NULL output%s");
    TuneKey k;
    rc = tune_key(desc, k);
    if (rc != DFM_OK) return rc;
    memset(opts, 0, sizeof(*opts));
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(k);
    if (it == g_tuned.end()) return 0;
    *opts = it->second;
    return 1;
}

// Which part did the process land on?  The tile kernel's store stream -- every workgroup writes a 4 KiB
// run into each of the 2C channel planes of a sample, planes D*h*w elements apart -- sustains 5.1-5.3 TB/s
// on most MI355X parts and ~3.9-4.1 TB/s on others (same binary, same clocks; a linear fill runs at
// 6.8 TB/s on both: profiles/archive/r03_c17_*).  This probe writes zeros in exactly that pattern so that a
// bench line can say which kind of part produced it.  `out` is overwritten with zeros.
__global__ __launch_bounds__(256) void store_probe_kernel(uint4 *__restrict__ out, long long plane_vec,
                                                          long long runs_per_plane, int planes, int pieces, int group)
{
    // block = (run, plane group, sample), run fastest: `pieces` x (256 lanes x 16 B = 4 KiB) contiguous per
    // plane; walks the `group` planes of its plane group (group == planes: the tile kernel's pattern)
    typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));
    const long long run = blockIdx.x % runs_per_plane;
    const int pg = (int)(blockIdx.x / runs_per_plane);
    probe_u32x4 *p = (probe_u32x4 *)out + ((size_t)blockIdx.y * planes + (size_t)pg * group) * plane_vec +
                     run * 256 * pieces + threadIdx.x;
    const probe_u32x4 z = {0u, 0u, 0u, 0u};
    const int n = min(group, planes - pg * group);
    for (int c = 0; c < n; ++c)
        for (int k = 0; k < pieces; ++k) __builtin_nontemporal_store(z, p + (size_t)c * plane_vec + k * 256);
}

DFM_API int dfm_store_probe(void *out, int32_t batch, int32_t planes, int64_t plane_bytes, int32_t run_bytes,
                            int32_t planes_per_workgroup, void *stream)
{
    if (!out || batch <= 0 || planes <= 0 || plane_bytes < 4096 || ((uintptr_t)out & 15) || (plane_bytes & 15))
        return fail(DFM_ERR_INVALID_ARG, "store probe: aligned buffer of batch x planes x plane_bytes%s");
    const int pieces = run_bytes > 0 ? run_bytes / 4096 : 1;  // 0: the tile kernel's 4 KiB runs
    if (pieces < 1 || pieces * 4096 != (run_bytes > 0 ? run_bytes : 4096) || plane_bytes < 4096ll * pieces)
        return fail(DFM_ERR_INVALID_ARG, "store probe: run_bytes must be a multiple of 4096%s");
    const int group = planes_per_workgroup > 0 ? std::min(planes_per_workgroup, planes) : planes;  // 0: all planes
    const long long plane_vec = plane_bytes / 16, runs = plane_vec / (256 * pieces);  // (a plane's tail is skipped)
    const long long nblk = runs * ((planes + group - 1) / group);
    if (nblk > 2147483647ll || batch > 65535) return fail(DFM_ERR_UNSUPPORTED, "store probe: grid too large%s");
    hipLaunchKernelGGL(store_probe_kernel, dim3((unsigned)nblk, batch), dim3(256), 0, (hipStream_t)stream,
                       (uint4 *)out, plane_vec, runs, planes, pieces, group);
    HIP_TRY(hipGetLastError());
    return DFM_OK;
}

// the shader clock this part sustains with every CU busy: cycles (s_memtime) and 100 MHz reference
// ticks (s_memrealtime) across `iterations` dependent FMAs per lane, written by workgroup 0
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *out, int iterations, float seed)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = seed + threadIdx.x, x1 = seed * 2.0f, x2 = seed * 3.0f, x3 = seed * 4.0f;
    for (int i = 0; i < iterations; ++i) {
        x0 = __builtin_fmaf(x0, 0.999f, 0.5f);
        x1 = __builtin_fmaf(x1, 0.998f, 0.25f);
        x2 = __builtin_fmaf(x2, 0.997f, 0.125f);
        x3 = __builtin_fmaf(x3, 0.996f, 0.0625f);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
    if (x0 + x1 + x2 + x3 == 12345.678f) out[2] = 1;  // keeps the loop
}

DFM_API int dfm_clock_probe(void *out3, int32_t iterations, void *stream)
{
    if (!out3 || iterations <= 0 || ((uintptr_t)out3 & 7)) return fail(DFM_ERR_INVALID_ARG, "clock probe: 3 x u64 device buffer%s");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (unsigned long long *)out3,
                       iterations, 1.0f);
    HIP_TRY(hipGetLastError());
    return DFM_OK;
}

DFM_API void dfm_plane_sweep_reset_tuning(void)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned.clear();
}

DFM_API int dfm_plane_sweep_bwd(const dfm_sweep_desc *desc, const void *grad_out,
                                const float *depths, const float *cam2img,
                                const float *cam2img_inv, const float *cur2prev, float *grad_cur,
                                float *grad_prev, void *stream)
{
    return dfm_plane_sweep_bwd_opts(desc, grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_cur,
                                    grad_prev, stream, nullptr);
}

}  // extern "C"
namespace {
int sweep_bwd_impl(const dfm_sweep_desc *desc, const void *grad_out, const float *depths, const float *cam2img,
                   const float *cam2img_inv, const float *cur2prev, float *grad_cur, float *grad_prev, void *stream,
                   const dfm_sweep_opts *opts, bool grad_cl);

// (n, P, C) pixel-major -> (n, C, P) planar: 64 pixels x 8 16-byte channel pieces per workgroup through an
// LDS tile; 16-byte loads along the channels of a pixel, 16-byte stores along the pixels of a channel.
// C and P are whole 16-byte runs (checked by the caller).  grid = (ceil(P / 64), ceil(C / (8 * VEC)), n)
template <typename T>
__global__ __launch_bounds__(256) void unpack_pixel_major_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                                 int C, long long P)
{
    constexpr int VEC = 16 / sizeof(T);   // elements per 16-byte piece
    constexpr int TC = 8 * VEC;           // channels per tile
    constexpr int PITCH = 64 + VEC;       // tile row: 64 pixels (+ one piece: rows stay 16-byte aligned)
    __shared__ __attribute__((aligned(16))) T tile[TC * PITCH];
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * TC;
    const size_t n = blockIdx.z;
    const T *s = src + n * (size_t)P * C;
    T *d = dst + n * (size_t)C * P;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = threadIdx.x + 256 * k;          // piece: pixel q / 8, channel piece q % 8
        const int p = q >> 3, cp = q & 7;
        if (p0 + p < P && c0 + cp * VEC < C) {
            const uint4 v = *(const uint4 *)(s + (size_t)(p0 + p) * C + c0 + cp * VEC);
            T e[VEC];
            __builtin_memcpy(e, &v, 16);
#pragma unroll
            for (int j = 0; j < VEC; ++j) tile[(cp * VEC + j) * PITCH + p] = e[j];
        }
    }
    __syncthreads();
    constexpr int PPR = 64 / VEC;  // 16-byte pieces per tile row
#pragma unroll
    for (int k = 0; k < (TC * PPR) / 256; ++k) {
        const int q = threadIdx.x + 256 * k;
        const int c = q / PPR, pp = (q % PPR) * VEC;
        if (c0 + c < C && p0 + pp < P)
            *(uint4 *)(d + (size_t)(c0 + c) * P + p0 + pp) = *(const uint4 *)(tile + c * PITCH + pp);
    }
}
}
extern "C" {
DFM_API int dfm_plane_sweep_bwd_opts(const dfm_sweep_desc *desc, const void *grad_out,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev,
                                     float *grad_cur, float *grad_prev, void *stream,
                                     const dfm_sweep_opts *opts)
{
    return sweep_bwd_impl(desc, grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_cur, grad_prev, stream, opts,
                          false);
}

DFM_API int dfm_plane_sweep_bwd_channels_last(const dfm_sweep_desc *desc, const void *grad_out,
                                              const float *depths, const float *cam2img,
                                              const float *cam2img_inv, const float *cur2prev,
                                              float *grad_cur, float *grad_prev, void *workspace,
                                              size_t workspace_bytes, void *stream)
{
    int rc = check_desc(desc);
    if (rc != DFM_OK) return rc;
    if (!grad_out) return fail(DFM_ERR_INVALID_ARG, "NULL device pointer%s");
    const size_t esz = desc->dtype == DFM_BF16 ? 2 : 4;
    const long long P = (long long)desc->num_depths * desc->h_out * desc->w_out;
    const int C2 = 2 * desc->channels, vec = (int)(16 / esz);
    const size_t vol = (size_t)desc->batch * C2 * P * esz;
    if (workspace && workspace_bytes >= vol && C2 % vec == 0 && P % vec == 0 && !((uintptr_t)grad_out & 15) &&
        !((uintptr_t)workspace & 15) && desc->batch <= 65535) {
        // (B, P, 2C) -> (B, 2C, P) through an LDS tile at copy speed, then the backward on the reference
        // layout: its lanes are consecutive lattice points, which the planar layout serves with one
        // coalesced load per (plane, channel); read in place, a wave's 2-byte loads land 4C bytes apart
        // and every channel pass re-fetches the lines (config K: 2.2 ms instead of 1.2 ms)
        const dim3 grid((unsigned)((P + 63) / 64), (unsigned)((C2 + 8 * vec - 1) / (8 * vec)), desc->batch);
        if (desc->dtype == DFM_BF16)
            hipLaunchKernelGGL(unpack_pixel_major_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t *)grad_out, (bf16_t *)workspace, C2, P);
        else
            hipLaunchKernelGGL(unpack_pixel_major_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                               (const float *)grad_out, (float *)workspace, C2, P);
        HIP_TRY(hipGetLastError());
        return sweep_bwd_impl(desc, workspace, depths, cam2img, cam2img_inv, cur2prev, grad_cur, grad_prev, stream,
                              nullptr, false);
    }
    return sweep_bwd_impl(desc, grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_cur, grad_prev, stream,
                          nullptr, true);
}
}  // extern "C"
namespace {
int sweep_bwd_impl(const dfm_sweep_desc *desc, const void *grad_out, const float *depths, const float *cam2img,
                   const float *cam2img_inv, const float *cur2prev, float *grad_cur, float *grad_prev, void *stream,
                   const dfm_sweep_opts *opts, bool grad_cl)
{
    const bool force_scatter = opts && opts->kernel == 1;
    // 8: the tile kernel for the prev map only (the caller has the cur map from dfm_plane_sweep_bwd_cur_nhwc)
    const bool skip_cur = opts && opts->kernel == 8;
    int rc = check_desc(desc);
    if (rc != DFM_OK) return rc;
    if (!grad_out || !depths || !cam2img || !cam2img_inv || !cur2prev || !grad_cur || !grad_prev)
        return fail(DFM_ERR_INVALID_ARG, "NULL device pointer%s");
    const SweepGeom g = make_geom(desc);
    hipStream_t st = (hipStream_t)stream;
    // dense sweeps whose feature rows fit the LDS: accumulate there (see sweep_bwd_tile_kernel).
    // One 1024-lane workgroup per CU: the LDS holds the footprint table of the chunk
    // (planes x 256 points x 12 B) and the slab of 64-bit accumulators (CW channels x rows x W).
    int planes = std::max(1, std::min(BWD_MAXP, (g.D + 3) / 4));
    int cw_max = 4;
#ifdef DFM_DEBUG_HOOKS
    if (const char *e = getenv("DFM_BWD_PLANES")) planes = std::max(1, std::min(BWD_MAXP, atoi(e)));
    if (const char *e = getenv("DFM_BWD_CW")) cw_max = atoi(e);
#endif
    const int table_bytes = planes * BWD_PTS * 12;
    const int budget = 160 * 1024 - 1024 - table_bytes;  // 1 KiB for the static LDS
    // channels per pass: as many (of 4) as leave >= 4 rows in the budget
    auto pick_cw = [&](int cw) {
        while (cw > 2 && (long long)budget / ((long long)cw * desc->w_in * 8) < 4) cw >>= 1;
        return cw;
    };
    const int cw_cur = pick_cw(cw_max), cw_prev = cw_cur;
    int row_cap = 8;
#ifdef DFM_DEBUG_HOOKS
    if (const char *e = getenv("DFM_BWD_ROWCAP")) row_cap = atoi(e);
#endif
    int rows_cur = (int)std::min<long long>(std::min(desc->h_in, row_cap),
                                            (long long)budget / ((long long)cw_cur * desc->w_in * 8));
#ifdef DFM_DEBUG_HOOKS
    if (const char *e = getenv("DFM_BWD_ROWS")) rows_cur = std::min(rows_cur, std::max(4, atoi(e)));
#endif
    const int rows_prev = rows_cur;
    const long long hw = (long long)g.h_out * g.w_out;
    if (!force_scatter && rows_cur >= 4 && desc->h_in < 4096 && desc->w_in < 8192) {
        BwdGrid tg;
        tg.batch = desc->batch;
        tg.band_pts = BWD_PTS;
        tg.row_tiles = desc->cost_sample_factor < 1.5f ? 0 : (g.w_out + tg.band_pts - 1) / tg.band_pts;
        tg.bands = tg.row_tiles ? g.h_out * tg.row_tiles : (int)((hw + tg.band_pts - 1) / tg.band_pts);
        tg.planes = planes;
        tg.ablate = 0;
        tg.grad_cl = grad_cl ? 1 : 0;
#ifdef DFM_DEBUG_HOOKS
        {
            const char *ab = getenv("DFM_BWD_ABLATE");
            tg.ablate = ab ? atoi(ab) : 0;
        }
#endif
        tg.dchunks = (g.D + tg.planes - 1) / tg.planes;
        const long long nb = (long long)tg.bands * tg.dchunks * desc->batch;
        if (nb > 2147483647ll) return fail(DFM_ERR_UNSUPPORTED, "too many lattice points%s");
        const SweepFast fast = make_fast(desc);
        // dense bf16 sweeps: the matrix-product backward (plane_sweep_bwd_mfma.hip) takes the cur map and
        // the prev map's planes up to a zoom of SWEEP_BWD_ZOOM_FOUR map pixels per lattice point; this
        // kernel keeps the (nearest) planes beyond that (opts->kernel: 5 = never, 6 = whenever it applies
        // [the default])
        tg.split = 0;
        const bool mfma = !(opts && (opts->kernel == 5 || skip_cur)) && !grad_cl && sweep_bwd_mfma_supported(desc, grad_out);
        if (mfma) {
            tg.split = 1;
            rc = sweep_bwd_mfma_launch(desc, 0, grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_cur, grad_prev,
                                       stream);
            if (rc != DFM_OK) return rc;
            rc = sweep_bwd_mfma_launch(desc, 1, grad_out, depths, cam2img, cam2img_inv, cur2prev, grad_cur, grad_prev,
                                       stream);
            if (rc != DFM_OK) return rc;
        }
#define DFM_BWD_LAUNCH(T, CW, HALF, ROWS)                                                            \
    do {                                                                                             \
        tg.rows = (ROWS);                                                                            \
        const int lds_bytes = (CW) * (ROWS) * desc->w_in * 8 + table_bytes;                          \
        rc = ensure_dynamic_lds((const void *)sweep_bwd_tile_kernel<T, CW, HALF>, lds_bytes);        \
        if (rc != DFM_OK) return rc;                                                                 \
        hipLaunchKernelGGL((sweep_bwd_tile_kernel<T, CW, HALF>), dim3((unsigned)nb),                 \
                           dim3(BWD_PTS * bwd_groups(HALF)),                                         \
                           lds_bytes, st, g, fast, tg, (const T *)grad_out, depths, cam2img,         \
                           cam2img_inv, cur2prev, grad_cur, grad_prev);                              \
    } while (0)
#define DFM_BWD_HALF(T, HALF, CWV, ROWS)                                                             \
    do {                                                                                             \
        if ((CWV) == 8) DFM_BWD_LAUNCH(T, 8, HALF, ROWS);                                            \
        else if ((CWV) == 4) DFM_BWD_LAUNCH(T, 4, HALF, ROWS);                                       \
        else DFM_BWD_LAUNCH(T, 2, HALF, ROWS);                                                       \
    } while (0)
        if (desc->dtype == DFM_F32) {
            if (!skip_cur) DFM_BWD_HALF(float, 0, cw_cur, rows_cur);
            DFM_BWD_HALF(float, 1, cw_prev, rows_prev);
        } else {
            if (!mfma && !skip_cur) DFM_BWD_HALF(bf16_t, 0, cw_cur, rows_cur);
            DFM_BWD_HALF(bf16_t, 1, cw_prev, rows_prev);
        }
#undef DFM_BWD_HALF
#undef DFM_BWD_LAUNCH
        HIP_TRY(hipGetLastError());
        g_last_bwd_kernel.store(mfma ? 6 : 5);
        return DFM_OK;
    }
    if (grad_cl)  // the scatter kernel reads the reference layout only: the caller converts and calls dfm_plane_sweep_bwd
        return fail(DFM_ERR_UNSUPPORTED, "channels-last gradient: the LDS-tile backward does not take this shape%s");
    if (skip_cur) return fail(DFM_ERR_UNSUPPORTED, "prev-only backward: the LDS-tile kernel does not take this shape%s");
    const long long nb = (g.N + 255) / 256;
    if (nb > 2147483647ll) return fail(DFM_ERR_UNSUPPORTED, "too many lattice points%s");
    dim3 grid((unsigned)nb, desc->batch);
    if (desc->dtype == DFM_F32)
        hipLaunchKernelGGL(sweep_bwd_kernel<float>, grid, dim3(256), 0, st, g,
                           (const float *)grad_out, depths, cam2img, cam2img_inv, cur2prev,
                           grad_cur, grad_prev);
    else
        hipLaunchKernelGGL(sweep_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, g,
                           (const bf16_t *)grad_out, depths, cam2img, cam2img_inv, cur2prev,
                           grad_cur, grad_prev);
    HIP_TRY(hipGetLastError());
    g_last_bwd_kernel.store(1);
    return DFM_OK;
}
}  // namespace
extern "C" {

DFM_API int dfm_plane_sweep_grid(const dfm_sweep_desc *desc, int32_t b, const float *depths,
                                 const float *cam2img, const float *cam2img_inv,
                                 const float *cur2prev, float *cur_grid, float *prev_grid,
                                 void *stream)
{
    int rc = check_desc(desc);
    if (rc != DFM_OK) return rc;
    if (b < 0 || b >= desc->batch) return fail(DFM_ERR_INVALID_ARG, "sample index out of range%s");
    if (!depths || !cam2img || !cam2img_inv || !cur2prev || !cur_grid || !prev_grid)
        return fail(DFM_ERR_INVALID_ARG, "NULL device pointer%s");
    const SweepGeom g = make_geom(desc);
    const long long nb = (g.N + 255) / 256;
    if (nb > 2147483647ll) return fail(DFM_ERR_UNSUPPORTED, "too many lattice points%s");
    hipLaunchKernelGGL(sweep_grid_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g,
                       b, depths, cam2img, cam2img_inv, cur2prev, cur_grid, prev_grid);
    HIP_TRY(hipGetLastError());
    return DFM_OK;
}

}  // extern "C"

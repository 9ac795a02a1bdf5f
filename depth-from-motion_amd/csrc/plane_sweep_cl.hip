// plane_sweep_cl.hip -- plane-sweep cost volume written channels-last (gfx950)
//
// Same function as dfm_plane_sweep_fwd (build_dfm_cost, reference
// mmdet3d/models/backbones/dfm_backbone.py:217-314), same arithmetic, same
// values bit for bit; only the memory layout of the result differs:
//   out[b][d][h][w][2C]   (torch: shape (B,2C,D,H,W), memory_format channels_last_3d)
// which is the layout the consumer wants (MIOpen's bf16 Conv3d runs NDHWC, and an
// implicit-GEMM conv reads K = channels contiguously) and the layout the HBM wants:
// a lattice point's 2C values are ONE contiguous run (1 KiB at C=256 bf16), a wave
// writes whole KiBs back to back and a workgroup one contiguous 256 KiB region --
// the streaming-fill pattern (6.2-6.8 TB/s on this part) instead of 2 KiB pieces
// spread over 512 channel planes (4.6-5.4 TB/s, profiles/archive/r01_store_microbench*.txt).
//
// Layout in HBM
//   workspace : [zero pixel][cur maps][prev maps], pixel-major [b][h][w][C]: a tap is
//               C*sizeof(T) contiguous bytes; out-of-bounds taps read the zero pixel
//               (grid_sample's zeros padding with no masking in the blend)
// Kernel: workgroup = 256 consecutive lattice points of one depth plane.
//   phase 1  lane = point: sampling positions of both maps in the reference's fp32 op
//            order, 4 tap slots + 4 weights per map -> LDS (64 B per point)
//   phase 2  a wave walks its 64 points; the 64 lanes are the 2 x (C/CB) 16-byte
//            channel blocks of ONE point (cur | prev): 4 coalesced 16-byte tap loads
//            (L1/L2 hits: neighbouring points share taps, all depth planes of the cur
//            map share them), blend in ATen's order, one 16-byte nt store -- a wave
//            stores 1 KiB contiguous per point.
// Bound: HBM write.  No LDS staging: the taps of a point are whole cache lines.
#include "dfm_common.h"

#include <stdio.h>
#include <stdlib.h>

using namespace dfm;

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4w_t __attribute__((ext_vector_type(4)));

struct ClGrid {
    int batch, tiles;     // tiles of 256 points per depth plane
    int lpi_shift;        // log2(lanes per (point, map) item)
    unsigned zero_slot;   // 16-byte slot of the zero pixel
    unsigned cur_slot, prev_slot;  // first slot of sample 0's cur / prev map
};

struct Foot {
    unsigned slot[4];  // 16-byte slot of the nw / ne / sw / se pixel (block 0)
    float w[4];
};

template <int CB>
__device__ __forceinline__ void blend4(const float (&w)[4], const uint4 &qnw, const uint4 &qne,
                                       const uint4 &qsw, const uint4 &qse, float (&r)[CB])
{
    float a[CB], b[CB], c[CB], d[CB];
    unpack16(qnw, a);
    unpack16(qne, b);
    unpack16(qsw, c);
    unpack16(qse, d);
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        float acc = a[j] * w[0];
        acc = __builtin_fmaf(b[j], w[1], acc);
        acc = __builtin_fmaf(c[j], w[2], acc);
        acc = __builtin_fmaf(d[j], w[3], acc);
        r[j] = acc;
    }
}

__device__ __forceinline__ void make_foot(const Tap &t, unsigned map_slot, unsigned zero_slot, int W,
                                          int nblk, Foot &f)
{
    const int i00 = t.iy * W + t.ix, i01 = i00 + t.dx;
    const int i10 = i00 + t.dy * W, i11 = i10 + t.dx;
    f.slot[0] = (t.ok & 1u) ? map_slot + (unsigned)i00 * nblk : zero_slot;
    f.slot[1] = (t.ok & 2u) ? map_slot + (unsigned)i01 * nblk : zero_slot;
    f.slot[2] = (t.ok & 4u) ? map_slot + (unsigned)i10 * nblk : zero_slot;
    f.slot[3] = (t.ok & 8u) ? map_slot + (unsigned)i11 * nblk : zero_slot;
    f.w[0] = t.nw; f.w[1] = t.ne; f.w[2] = t.sw; f.w[3] = t.se;
}

template <typename T>
__global__ __launch_bounds__(256) void sweep_cl_kernel(
    SweepGeom g, ClGrid tg, const uint4 *__restrict__ ws, const uint4 *__restrict__ cur_maps,
    const uint4 *__restrict__ prev_maps, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    uint4 *__restrict__ out)
{
    // cur_maps / prev_maps: the pixel-major maps (B, H, W, C) -- the packed copies inside `ws`, or the
    // caller's own channels-last (NHWC) feature maps sampled where they lie.  Tap slots are relative
    // to the lane's map; an out-of-bounds corner is the sentinel slot and reads the zero pixel at ws.
    constexpr int CB = elem<T>::CB;
    constexpr unsigned ZERO = 0xffffffffu;
    __shared__ Foot foot[256][2];
    const int tid = threadIdx.x;
    // block id = ((tile*D + d)*B + b): sample fastest -> id % 8 == XCD keeps a sample's
    // maps in one L2; depth next -> the workgroups resident on an XCD read the same
    // (cur) or neighbouring (prev) pixels
    int th = blockIdx.x;
    const int b = th % tg.batch;
    th /= tg.batch;
    const int d = th % g.D;
    const int tile = th / g.D;
    const int hw = g.h_out * g.w_out;
    const int p0 = tile * 256;
    const int npts = min(256, hw - p0);
    const int HW = g.h_in * g.w_in;

    if (tid < npts) {
        const int p = p0 + tid;
        const int hi = p / g.w_out, wi = p - hi * g.w_out;
        float cx, cy, px, py;
        sweep_point(g, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, cx, cy, px, py,
                    nullptr);
        const Tap tc = make_tap(cx, cy, g.h_in, g.w_in);
        const Tap tp = make_tap(px, py, g.h_in, g.w_in);
        make_foot(tc, (unsigned)b * HW * g.nblk, ZERO, g.w_in, g.nblk, foot[tid][0]);
        make_foot(tp, (unsigned)b * HW * g.nblk, ZERO, g.w_in, g.nblk, foot[tid][1]);
    }
    __syncthreads();

    // lanes of a wave: [point in iteration][map][channel block]
    const int wave = tid >> 6, lane = tid & 63;
    const int lpi = 1 << tg.lpi_shift;
    const int sub = lane & (lpi - 1);
    const int half = (lane >> tg.lpi_shift) & 1;
    const int pin = lane >> (tg.lpi_shift + 1);
    const int ppi = 64 >> (tg.lpi_shift + 1);  // points per iteration
    const uint4 *mp = half ? prev_maps : cur_maps;
    // out slot of (point n, map, block): ((b*N + n)*2 + map)*nblk + block
    const size_t plane0 = ((size_t)b * g.N + (size_t)d * hw + p0) * 2;
    // U points in flight per lane: the loop body is one dependent chain (LDS read -> 4 tap
    // loads -> blend -> store), so the loads of U points are issued before the first blend
    constexpr int U = 4;  // measured: 2 -> 1187, 4 -> 1200, 8 -> 1146 vol/s on one box
    const int qlast = npts - 1;
    for (int blk = sub; blk < g.nblk; blk += lpi) {
        for (int it = 0; it < 64; it += U * ppi) {
            uint4 tap[U][4];
            float wgt[U][4];
            int qq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                qq[u] = wave * 64 + it + u * ppi + pin;
                const Foot f = foot[min(qq[u], qlast)][half];  // past the tile end: reload, never stored
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 *src = f.slot[k] == ZERO ? ws : mp + f.slot[k];
                    tap[u][k] = src[blk];
                    wgt[u][k] = f.w[k];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float r[CB];
                blend4<CB>(wgt[u], tap[u][0], tap[u][1], tap[u][2], tap[u][3], r);
                u32x4_t v;
                if constexpr (sizeof(T) == 4) {
                    v = u32x4_t{__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]),
                                __float_as_uint(r[3])};
                } else {
                    v = u32x4_t{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]),
                                pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7])};
                }
                if (qq[u] < npts && it + u * ppi < 64)
                    __builtin_nontemporal_store(
                        v, (u32x4_t *)(out + (plane0 + (size_t)qq[u] * 2 + half) * g.nblk + blk));
            }
        }
    }
}

// Same sampling (pixel-major maps, footprints once per point, coalesced 16-byte tap loads), result in
// the REFERENCE layout (B, 2C, D, h, w): the workgroup's 256 points x CP channels are transposed
// through an LDS tile and leave as 16-byte stores of consecutive points per channel plane (a wave
// writes one contiguous 1 KiB run per channel).  For strided sweeps (cost_sample_factor >= 2,
// config K) where staging whole feature rows would fetch mostly unused pixels and the direct tile
// kernel pays a scattered 16-byte tap per channel block.
template <typename T, int CP>
__global__ __launch_bounds__(256) void sweep_clt_kernel(
    SweepGeom g, ClGrid tg, const uint4 *__restrict__ ws, const uint4 *__restrict__ cur_maps,
    const uint4 *__restrict__ prev_maps, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    T *__restrict__ out)
{
    // cur_maps / prev_maps, ZERO: as in sweep_cl_kernel (packed copies in ws, or the caller's NHWC maps)
    constexpr unsigned ZERO = 0xffffffffu;
    constexpr int CB = elem<T>::CB;
    constexpr int BPP = CP / CB;          // 16-byte channel blocks per pass (8)
    constexpr int VEC = 16 / sizeof(T);   // points per 16-byte store
    constexpr int PITCH = 256 + VEC;      // elements per tile row (16-byte aligned rows)
    constexpr int PPI = 64 / BPP;         // points per wave iteration
    __shared__ Foot foot[256][2];
    __shared__ __attribute__((aligned(16))) T tile[CP * PITCH];
    const int tid = threadIdx.x;
    int th = blockIdx.x;
    const int b = th % tg.batch;
    th /= tg.batch;
    const int d = th % g.D;
    const int tl = th / g.D;
    const int hw = g.h_out * g.w_out;
    const int p0 = tl * 256;
    const int npts = min(256, hw - p0);
    const int HW = g.h_in * g.w_in;

    if (tid < npts) {
        const int p = p0 + tid;
        const int hi = p / g.w_out, wi = p - hi * g.w_out;
        float cx, cy, px, py;
        sweep_point(g, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[d], hi, wi, cx, cy, px, py,
                    nullptr);
        const Tap tc = make_tap(cx, cy, g.h_in, g.w_in);
        const Tap tp = make_tap(px, py, g.h_in, g.w_in);
        make_foot(tc, (unsigned)b * HW * g.nblk, ZERO, g.w_in, g.nblk, foot[tid][0]);
        make_foot(tp, (unsigned)b * HW * g.nblk, ZERO, g.w_in, g.nblk, foot[tid][1]);
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane & (BPP - 1), pin = lane / BPP;
    const int qlast = npts - 1;
    constexpr int U = 4;
    const int npass = (g.nblk + BPP - 1) / BPP;
#ifndef DFM_CLT_ABLATE  // experiments at release speed (build_variant): 1 no cur gather, 2 no prev gather, 4 no stores
#define DFM_CLT_ABLATE 0
#endif
    for (int map = 0; map < 2; ++map) {
        const uint4 *mp = map ? prev_maps : cur_maps;
        for (int pass = 0; pass < npass; ++pass) {
            const int blk = pass * BPP + sub;
            if (blk < g.nblk && !((DFM_CLT_ABLATE >> map) & 1)) {
                for (int it = 0; it < 64; it += U * PPI) {
                    uint4 tap[U][4];
                    float wgt[U][4];
                    int qq[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        qq[u] = wave * 64 + it + u * PPI + pin;
                        const Foot f = foot[min(qq[u], qlast)][map];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint4 *src = f.slot[k] == ZERO ? ws : mp + f.slot[k];
                            tap[u][k] = src[blk];
                            wgt[u][k] = f.w[k];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float r[CB];
                        blend4<CB>(wgt[u], tap[u][0], tap[u][1], tap[u][2], tap[u][3], r);
                        if (it + u * PPI < 64) {
#pragma unroll
                            for (int j = 0; j < CB; ++j) tile[(sub * CB + j) * PITCH + qq[u]] = elem<T>::store(r[j]);
                        }
                    }
                }
            }
            __syncthreads();
            // flush: CP rows (channels) x 256 points, one 16-byte vector of VEC points per lane and step
            const int crem = min(CP, g.C - pass * CP);
            constexpr int VPR = 256 / VEC;  // vectors per row
            for (int idx = tid; idx < crem * VPR; idx += 256) {
                const int row = idx / VPR, v = idx - row * VPR;
                if (v * VEC < npts && !(DFM_CLT_ABLATE & 4)) {
                    const u32x4_t val = *(const u32x4_t *)(tile + row * PITCH + v * VEC);
                    const size_t ch = (size_t)b * 2 * g.C + (size_t)map * g.C + pass * CP + row;
                    __builtin_nontemporal_store(val, (u32x4_t *)(out + (ch * g.D + d) * hw + p0 + v * VEC));
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// Strided sweeps, reference layout, WALKING the depth axis (fp32, config K).
//
// sweep_clt_kernel fetches the four taps of every (point, plane) anew: 8 x 128 bytes per point and
// plane for both maps, 15 GB of L2 -> L1 traffic for a 3.8 GB volume, and its time is the sum of its
// store time and its gather time (profiles/archive/r03_c39_*).  But a point's footprint barely moves from one
// plane to the next: the cur position is the lattice pixel up to rounding noise (the footprint flips
// between two neighbouring pixel pairs), the prev position slides along its epipolar line by less
// than a pixel per plane beyond ~20 m: over the planes of config K 60 % of the footprints equal the
// previous plane's.
//
// So a WAVE owns 32 consecutive lattice points of one map and a run of 24 depth planes, keeps each
// point's 2x2 footprint (the lane's 16-byte channel block of it) in registers, and per plane
//   * loads only the taps whose offset differs from the previous plane's (hand-masked buffer loads; an
//     out-of-bounds corner is an offset beyond the buffer, which the hardware answers with zeros),
//   * blends in ATen's order, transposes the 32 points x 32 channels through a wave-private LDS tile
//     and stores one 128-byte run per channel,
// with the loads of plane d + 1 issued BEFORE the stores of plane d (a counted vmcnt then waits for the
// loads without waiting for a store acknowledgement).
// Lanes: (sub = lane & 7: channel block, pin = lane >> 3: point of a group of 8); 4 groups = 32
// points.  Footprints are computed lane = point for TWO planes at a time (lanes 32-63: the next plane)
// in the reference's op order (sweep_point_map) and handed over through a ring of three planes in LDS,
// which is also where the previous plane's offsets and the weights are re-read from (registers go to
// the taps: 114 VGPRs, four waves per SIMD).  No workgroup barrier: the four waves of a workgroup (four
// neighbouring 32-point tiles) only share the launch.
// Measured (profiles/archive/r04_c28..c34): 1.22 ms against the per-plane kernel's 1.36-1.47 ms on the same
// boxes (0.47 of 8 TB/s against 0.39-0.43).  Not more, because with the volume's stores streaming
// through the L2 a tap re-read a few planes (10+ us) later misses it: the L2 -> L1 requests drop
// three-fold, the fabric reads grow from 0.87 to 2.4 GB, and with both loads and stores in flight the
// kernel takes 1.3 ms where either alone takes 0.56 / 0.69 ms over a 0.43 ms skeleton
// (r04_c32_walk_kernel_ablation.txt).  Permuting the cached taps when a footprint moves by one pixel
// (25-35 % of the planes) saves another third of the loads but costs more issue slots than it saves
// (first version, r04_c28).
// ---------------------------------------------------------------------------
#ifndef DFM_WALK_ABLATE  // experiments at release speed (build_variant): 1 no tap loads, 2 no stores
#define DFM_WALK_ABLATE 0
#endif
#ifndef DFM_WALK_WAVES
#define DFM_WALK_WAVES 4  // waves per SIMD the walking kernel is compiled for (5 spills, 3 is 8 % slower: profiles/archive/r04_c33_*)
#endif
struct WalkGrid {
    int batch, tiles, chunks, chunk_planes, passes;
};

// one tap, loaded only by the lanes whose cached offset differs: exec masking by hand (a branch per tap
// costs more than the load), through the maps' raw buffer descriptor
__device__ __forceinline__ void walk_load_if(u32x4_t &v, unsigned long long need, unsigned off, const u32x4_t &rs)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "buffer_load_dwordx4 %[v], %[off], %[rs], 0 offen\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [v] "+v"(v), [sv] "=&s"(save)
                 : [m] "s"(need), [off] "v"(off), [rs] "s"(rs)
                 : "memory", "scc");  // (s_and_saveexec writes SCC)
}

template <int HALF>
__device__ __forceinline__ void walk_body(const SweepGeom &g, const SweepFast &fast, const int b, const int tile,
                                          const int d0, const int d1, const int pass, const int lane,
                                          Foot (*foot_s)[32], float *tile_s, const uint4 *__restrict__ mp,
                                          const unsigned map_bytes_all, const float *__restrict__ depths,
                                          const float *__restrict__ P, const float *__restrict__ Pinv,
                                          const float *__restrict__ Tm, float *__restrict__ out)
{
    // Taps come through a raw buffer descriptor of the maps: an out-of-bounds corner is an offset beyond
    // the buffer, which the hardware answers with zeros (grid_sample's zeros padding) -- no zero pixel, no
    // 64-bit address arithmetic, no select per tap.
    constexpr unsigned OOB = 0xf0000000u;
    constexpr int G = 4, PTS = 8 * G;
    constexpr int PITCH = PTS + 4;  // floats per tile row: rows stay 16-byte aligned
    u32x4_t rs;
    {
        const unsigned long long base = (unsigned long long)mp;
        rs.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        rs.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));  // stride 0: raw
        rs.z = (unsigned)__builtin_amdgcn_readfirstlane((int)map_bytes_all);
        rs.w = 0x00020000u;
    }
    const int sub = lane & 7, pin = lane >> 3;
    const int hw = g.h_out * g.w_out;
    const int HW = g.h_in * g.w_in;
    const int p0 = tile * PTS;
    const unsigned boff = (unsigned)(pass * 8 + sub) * 16u;
    // this lane's lattice point in the footprint phase
    const int fp = p0 + (lane & 31);
    const int fhi = fp / g.w_out, fwi = fp - fhi * g.w_out;
    const unsigned pix_bytes = (unsigned)g.nblk * 16u;
    const unsigned smp_off = (unsigned)b * (unsigned)HW * pix_bytes;
    u32x4_t tap[G][4];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int k = 0; k < 4; ++k) tap[gi][k] = u32x4_t{0, 0, 0, 0};
    // footprints of planes d, d + 1 (lanes 32-63: the second plane), lane = point, reference op order, into
    // a ring of three planes: the offsets of plane d - 1 are what the cache holds when plane d arrives
    auto footprints = [&](int d) {
        const int dd = d + (lane >> 5);
        // (both depths through the scalar cache: a per-lane load would put a vmcnt(0) -- a wait for the
        // previous plane's store acknowledgements -- into the loop)
        const float dep0 = depths[d], dep1 = depths[min(d + 1, d1 - 1)];
        if (dd < d1) {
            float x, y;
            sweep_point_map<HALF>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, (lane >> 5) ? dep1 : dep0, fhi,
                                  fwi, x, y);
            const Tap t = make_tap(x, y, g.h_in, g.w_in);
            const int i00 = t.iy * g.w_in + t.ix, i01 = i00 + t.dx;
            const int i10 = i00 + t.dy * g.w_in, i11 = i10 + t.dx;
            Foot f;
            f.slot[0] = (t.ok & 1u) ? smp_off + (unsigned)i00 * pix_bytes : OOB;
            f.slot[1] = (t.ok & 2u) ? smp_off + (unsigned)i01 * pix_bytes : OOB;
            f.slot[2] = (t.ok & 4u) ? smp_off + (unsigned)i10 * pix_bytes : OOB;
            f.slot[3] = (t.ok & 8u) ? smp_off + (unsigned)i11 * pix_bytes : OOB;
            f.w[0] = t.nw; f.w[1] = t.ne; f.w[2] = t.sw; f.w[3] = t.se;
            foot_s[(dd - d0) % 3][lane & 31] = f;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // bring the cache to plane d: load the taps whose offset differs from plane d - 1's
    auto update = [&](int d, bool first) {
        const Foot *fnew = foot_s[(d - d0) % 3], *fold = foot_s[(d - d0 + 2) % 3];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const u32x4_t sn = *(const u32x4_t *)fnew[gi * 8 + pin].slot;
            const u32x4_t so = *(const u32x4_t *)fold[gi * 8 + pin].slot;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long need =
                    (DFM_WALK_ABLATE & 1) ? 0ull : __builtin_amdgcn_ballot_w64(first || sn[k] != so[k]);
                walk_load_if(tap[gi][k], need, sn[k] + boff, rs);
            }
        }
    };
    // the four channel rows this lane flushes: row i * 8 + (lane >> 3), points (lane & 7) * 4 ..
    float *orow = out + (((size_t)b * 2 * g.C + (size_t)HALF * g.C + pass * 32 + (lane >> 3)) * g.D + d0) * hw + p0 +
                  (lane & 7) * 4;
    const size_t row8 = (size_t)8 * g.D * hw;
    footprints(d0);
    update(d0, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Per plane: blend d | footprints / loads for d + 1 | transpose + store d.  The loads of d + 1 are OLDER
    // than the stores of d, so the wait for them (vmcnt(4): everything but the four newest operations, the
    // stores) does not wait for the acknowledgement of those stores.
    for (int d = d0; d < d1; ++d) {
        asm volatile("s_waitcnt vmcnt(4)"
                     : "+v"(tap[0][0]), "+v"(tap[0][1]), "+v"(tap[0][2]), "+v"(tap[0][3]), "+v"(tap[1][0]),
                       "+v"(tap[1][1]), "+v"(tap[1][2]), "+v"(tap[1][3]), "+v"(tap[2][0]), "+v"(tap[2][1]),
                       "+v"(tap[2][2]), "+v"(tap[2][3]), "+v"(tap[3][0]), "+v"(tap[3][1]), "+v"(tap[3][2]),
                       "+v"(tap[3][3])
                     :
                     : "memory");
        const Foot *fcur = foot_s[(d - d0) % 3];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            float r[4], w4[4];
            const f32x4w_t wv = *(const f32x4w_t *)fcur[gi * 8 + pin].w;
            w4[0] = wv.x; w4[1] = wv.y; w4[2] = wv.z; w4[3] = wv.w;
            blend4<4>(w4, make_uint4(tap[gi][0].x, tap[gi][0].y, tap[gi][0].z, tap[gi][0].w),
                      make_uint4(tap[gi][1].x, tap[gi][1].y, tap[gi][1].z, tap[gi][1].w),
                      make_uint4(tap[gi][2].x, tap[gi][2].y, tap[gi][2].z, tap[gi][2].w),
                      make_uint4(tap[gi][3].x, tap[gi][3].y, tap[gi][3].z, tap[gi][3].w), r);
#pragma unroll
            for (int j = 0; j < 4; ++j) tile_s[(sub * 4 + j) * PITCH + gi * 8 + pin] = r[j];
        }
        __builtin_amdgcn_wave_barrier();
        if (d + 1 < d1) {
            if (((d + 1 - d0) & 1) == 0) footprints(d + 1);
            update(d + 1, false);
        }
        // 32 channels x 32 points: 8 lanes per channel row, one 128-byte run each
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4_t val = *(const u32x4_t *)(tile_s + (i * 8 + (lane >> 3)) * PITCH + (lane & 7) * 4);
            if (!(DFM_WALK_ABLATE & 2) || val.x == 0x12345u) __builtin_nontemporal_store(val, (u32x4_t *)(orow + i * row8));
        }
        orow += hw;
        __builtin_amdgcn_wave_barrier();
    }
    // nothing is in flight into a tap register past this point (the last plane issues no loads; the build's
    // disassembly check -- tools/verify_walk_asm.py -- follows every path from a masked load to a wait)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256, DFM_WALK_WAVES) void sweep_cltw_kernel(
    SweepGeom g, SweepFast fast, WalkGrid tg, const uint4 *__restrict__ cur_maps, const uint4 *__restrict__ prev_maps,
    unsigned map_bytes_all, const float *__restrict__ depths, const float *__restrict__ P,
    const float *__restrict__ Pinv, const float *__restrict__ Tm, float *__restrict__ out)
{
    __shared__ Foot foot_s[4][3][32];
    __shared__ __attribute__((aligned(16))) float tile_s[4][32 * 36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // block id = (((tile4 * 2 + map) * passes + pass) * chunks + chunk) * B + b: sample fastest (id % 8 == XCD
    // keeps a sample's maps in one L2), then the depth chunks of one tile (the same cur pixels)
    int th = blockIdx.x;
    const int b = th % tg.batch;
    th /= tg.batch;
    const int chunk = th % tg.chunks;
    th /= tg.chunks;
    const int pass = th % tg.passes;
    th /= tg.passes;
    const int map = th & 1;
    const int tile = (th >> 1) * 4 + wave;
    if (tile >= tg.tiles) return;
    const int d0 = chunk * tg.chunk_planes, d1 = min(g.D, d0 + tg.chunk_planes);
    if (map == 0)
        walk_body<0>(g, fast, b, tile, d0, d1, pass, lane, foot_s[wave], tile_s[wave], cur_maps, map_bytes_all,
                     depths, P, Pinv, Tm, out);
    else
        walk_body<1>(g, fast, b, tile, d0, d1, pass, lane, foot_s[wave], tile_s[wave], prev_maps, map_bytes_all,
                     depths, P, Pinv, Tm, out);
}

// ---------------------------------------------------------------------------
// Backward of strided fp32 sweeps, CUR map (autograd of F.grid_sample at dfm_backbone.py:296-311 with
// cost_sample_factor >= 2: config K).
//
// The LDS-atomic tile kernel (plane_sweep.hip) stages whole feature rows per depth chunk: for a
// strided sweep 15 of 16 staged pixels receive nothing, and config K takes 10.7 ms, half of it for the
// cur map -- whose sample position does not move with depth at all: it is the lattice pixel up to
// rounding noise, so over ALL planes a point's taps stay inside one 3x3 pixel window.  A WAVE owns 16
// lattice points and the whole depth run and keeps the window's nine fp32 accumulators (the lane's 4
// channels of each) in registers; a plane routes its four weight x gradient products into the window
// by the footprint's position in it (selects, no memory), and the window is written out ONCE, at the
// end: 36 no-return buffer atomics per lane into the PIXEL-MAJOR gradient map (B, H, W, C) -- the 8
// lanes of a point cover a pixel's 128 bytes; a pixel outside the map is an offset beyond the buffer and
// the hardware drops it.  A footprint that leaves the window (never, for a lattice-aligned sweep) adds
// its products with atomics of its own.  A zero weight still multiplies (0 x Inf = NaN reaches the tap,
// as in ATen); an out-of-bounds tap is skipped.
// The gradient volume is read in the reference layout, 32 channels x 16 points per plane as 64-byte
// rows, and transposed through a wave-private LDS tile; plane d + 1 is requested before plane d is
// processed.
// (Two schemes for BOTH maps were built first and dropped, profiles/archive/r04_c40..c53: flushing an accumulator
// whenever its tap moves -- the forward's walking scheme mirrored -- 19.5 ms; a dense pixel-major scatter
// of every (point, plane) with full-wave atomics 21.7 ms and 26 GB of HBM writes for a 0.42 GB map, ~5 ms per
// dword atomic per lane and tap, proportional to their number: each of them re-visits every map line tens of
// times with the gradient volume streaming through the L2 in between, and a visit of a line that has left the
// L2 is a fill and a write-back.  This kernel visits each line once: 0.9 GB of writes.  The sc1 bit on the
// atomics changes neither time nor traffic.  The prev map, whose taps move, stays with the LDS-atomic tile
// kernel.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void walk_flush_if(float (&a)[4], unsigned long long need, unsigned off, const u32x4_t &rs)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "buffer_atomic_add_f32 %[a0], %[off], %[rs], 0 offen\n\t"
                 "buffer_atomic_add_f32 %[a1], %[off], %[rs], 0 offen offset:4\n\t"
                 "buffer_atomic_add_f32 %[a2], %[off], %[rs], 0 offen offset:8\n\t"
                 "buffer_atomic_add_f32 %[a3], %[off], %[rs], 0 offen offset:12\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save)
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [m] "s"(need), [off] "v"(off),
                   [rs] "s"(rs)
                 : "memory", "scc");
}

struct CurFoot {
    uint32_t f;  // bwd_footprint's packed corner + in-bounds bits
    float fw, fn;
    uint32_t pad;
};

__global__ __launch_bounds__(256, 3) void sweep_bwdc_kernel(
    SweepGeom g, SweepFast fast, int batch, int tiles, int passes, float *__restrict__ gcur, unsigned map_bytes_all,
    const float *__restrict__ depths, const float *__restrict__ P, const float *__restrict__ Pinv,
    const float *__restrict__ Tm, const float *__restrict__ gout)
{
    constexpr unsigned OOB = 0xf0000000u;
    constexpr int G = 2, PTS = 8 * G;
    constexpr int PITCH = 36;  // floats per POINT row of the tile (32 channels + 16 bytes)
    __shared__ __attribute__((aligned(16))) CurFoot foot_all[4][4][PTS];
    __shared__ int anchor_all[4][PTS][2];
    __shared__ __attribute__((aligned(16))) float tile_all[4][PTS * PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    CurFoot(*foot_s)[PTS] = foot_all[wave];
    int(*anchor_s)[2] = anchor_all[wave];
    float *tile_s = tile_all[wave];
    // block id = ((tile4 * passes + pass) * B + b): sample fastest (id % 8 == XCD)
    int th = blockIdx.x;
    const int b = th % batch;
    th /= batch;
    const int pass = th % passes;
    th /= passes;
    const int tile = th * 4 + wave;
    if (tile >= tiles) return;
    u32x4_t rs;
    {
        const unsigned long long base = (unsigned long long)gcur;
        rs.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        rs.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
        rs.z = (unsigned)__builtin_amdgcn_readfirstlane((int)map_bytes_all);
        rs.w = 0x00020000u;
    }
    const int sub = lane & 7, pin = lane >> 3;
    const int hw = g.h_out * g.w_out, H = g.h_in, W = g.w_in;
    const int p0 = tile * PTS;
    const unsigned boff = (unsigned)(pass * 8 + sub) * 16u;
    const unsigned pix_bytes = (unsigned)g.nblk * 16u;
    // footprint phase: lane = (plane of four, point)
    const int fp = p0 + (lane & 15);
    const int fhi = fp / g.w_out, fwi = fp - fhi * g.w_out;
    auto footprints = [&](int d) {
        const int k4 = lane >> 4, dd = min(d + k4, g.D - 1);
        float sx, sy, fw, fn;
        sweep_point_map<0>(g, fast, P + b * 16, Pinv + b * 16, Tm + b * 16, depths[dd], fhi, fwi, sx, sy);
        CurFoot cf;
        cf.f = bwd_footprint(sx, sy, H, W, fw, fn);
        cf.fw = fw;
        cf.fn = fn;
        cf.pad = 0u;
        foot_s[k4][lane & 15] = cf;
        if (d == 0 && k4 == 0) {
            // the window: the pixel nearest to the first plane's position and its eight neighbours
            const bool fin = (fabsf(sx) <= 1.0e9f) && (fabsf(sy) <= 1.0e9f);
            anchor_s[lane & 15][0] = fin ? (int)rintf(sx) - 1 : 0;
            anchor_s[lane & 15][1] = fin ? (int)rintf(sy) - 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    };
    float acc[G][3][3][4];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[gi][r][c][j] = 0.0f;
    // the two channel rows this lane reads: row i * 16 + (lane >> 2), points (lane & 3) * 4 ..
    const float *grow = gout + (((size_t)b * 2 * g.C + pass * 32 + (lane >> 2)) * g.D) * hw + p0 + (lane & 3) * 4;
    const size_t row16 = (size_t)16 * g.D * hw;
    u32x4_t gv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) gv[i] = __builtin_nontemporal_load((const u32x4_t *)(grow + i * row16));
    footprints(0);
    int ax[G], ay[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        ax[gi] = anchor_s[gi * 8 + pin][0];
        ay[gi] = anchor_s[gi * 8 + pin][1];
    }
    for (int d = 0; d < g.D; ++d) {
        if (d && (d & 3) == 0) footprints(d);
        // this plane's rows -> tile[point][channel]; the next plane's rows are requested
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = i * 16 + (lane >> 2), q = (lane & 3) * 4;
            tile_s[(q + 0) * PITCH + ch] = __uint_as_float(gv[i].x);
            tile_s[(q + 1) * PITCH + ch] = __uint_as_float(gv[i].y);
            tile_s[(q + 2) * PITCH + ch] = __uint_as_float(gv[i].z);
            tile_s[(q + 3) * PITCH + ch] = __uint_as_float(gv[i].w);
        }
        grow += hw;
        if (d + 1 < g.D) {
#pragma unroll
            for (int i = 0; i < 2; ++i) gv[i] = __builtin_nontemporal_load((const u32x4_t *)(grow + i * row16));
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int q = gi * 8 + pin;
            const u32x4_t fr = *(const u32x4_t *)&foot_s[d & 3][q];
            const f32x4w_t gq = *(const f32x4w_t *)(tile_s + q * PITCH + sub * 4);
            const uint32_t f = fr.x;
            const float fw = __uint_as_float(fr.y), fn = __uint_as_float(fr.z);
            const bool wok = f & (1u << 27), eok = f & (1u << 28), nok = f & (1u << 29), sok = f & (1u << 30);
            const int iyn = (int)(f & 0x1fffu) - 1, ixw = (int)((f >> 13) & 0x1fffu) - 1;
            const int dx = ixw - ax[gi], dy = iyn - ay[gi];
            const bool valid = f != 0u;
            const bool inwin = valid && (unsigned)dx <= 1u && (unsigned)dy <= 1u;
            const float cwt = 1.0f - fw, cet = fw, rnt = 1.0f - fn, rst = fn;
            const float wq[4] = {rnt * cwt, rnt * cet, rst * cwt, rst * cet};
            const bool okq[4] = {wok && nok, eok && nok, wok && sok, eok && sok};
            float pr[4][4];  // [tap][channel]: weight x gradient, 0 for a tap outside the map
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) pr[k][c] = okq[k] ? gq[c] * wq[k] : 0.0f;
            if (__builtin_amdgcn_ballot_w64(valid && !inwin) != 0ull) {
                // the footprint left the window: its products go to memory directly
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long need = __builtin_amdgcn_ballot_w64(valid && !inwin && okq[k]);
                    const unsigned off = ((unsigned)(b * H + iyn + (k >> 1)) * (unsigned)W + (unsigned)(ixw + (k & 1))) * pix_bytes + boff;
                    walk_flush_if(pr[k], need, off, rs);
                }
            }
            const bool x0 = inwin && dx == 0, x1 = inwin && dx == 1, y0 = dy == 0, y1 = dy == 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // columns of the window: west/east taps land on columns dx, dx + 1
                const float n0 = x0 ? pr[0][c] : 0.0f, n1 = x0 ? pr[1][c] : (x1 ? pr[0][c] : 0.0f), n2 = x1 ? pr[1][c] : 0.0f;
                const float s0 = x0 ? pr[2][c] : 0.0f, s1 = x0 ? pr[3][c] : (x1 ? pr[2][c] : 0.0f), s2 = x1 ? pr[3][c] : 0.0f;
                // rows: north/south taps land on rows dy, dy + 1
                acc[gi][0][0][c] += y0 ? n0 : 0.0f;
                acc[gi][0][1][c] += y0 ? n1 : 0.0f;
                acc[gi][0][2][c] += y0 ? n2 : 0.0f;
                acc[gi][1][0][c] += y0 ? s0 : (y1 ? n0 : 0.0f);
                acc[gi][1][1][c] += y0 ? s1 : (y1 ? n1 : 0.0f);
                acc[gi][1][2][c] += y0 ? s2 : (y1 ? n2 : 0.0f);
                acc[gi][2][0][c] += y1 ? s0 : 0.0f;
                acc[gi][2][1][c] += y1 ? s1 : 0.0f;
                acc[gi][2][2][c] += y1 ? s2 : 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the windows
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int y = ay[gi] + r, x = ax[gi] + c;
                const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                const unsigned off = in ? ((unsigned)(b * H + y) * (unsigned)W + (unsigned)x) * pix_bytes + boff : OOB;
                walk_flush_if(acc[gi][r][c], ~0ull, off, rs);
            }
}

size_t map_bytes(const dfm_sweep_desc *d)
{
    return ((size_t)d->batch * d->channels * d->h_in * d->w_in * (d->dtype == DFM_BF16 ? 2 : 4) + 255) &
           ~(size_t)255;
}

}  // namespace

namespace dfm {

// can the strided-sweep kernel above take this call? (16-byte alignment of every channel plane
// and of every 256-point tile, whole 16-byte channel blocks, 32-bit tap slots)
bool sweep_clt_supported(const dfm_sweep_desc *d, const void *out)
{
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    const long long hw = (long long)d->h_out * d->w_out;
    return d->channels % CB == 0 && hw % CB == 0 && ((uintptr_t)out & 15) == 0 &&
           map_bytes(d) / 16 < 0xffffffffull;
}

size_t sweep_clt_workspace_bytes(const dfm_sweep_desc *d)
{
    const size_t zero = ((size_t)d->channels * (d->dtype == DFM_BF16 ? 2 : 4) + 255) & ~(size_t)255;
    return zero + 2 * map_bytes(d);
}

// the depth-walking kernel: fp32, whole 32-channel passes, whole 32-point tiles
bool sweep_cltw_supported(const dfm_sweep_desc *d, const void *out)
{
    const long long hw = (long long)d->h_out * d->w_out;
#ifdef DFM_WALK_UNVERIFIED
    // build.py disassembles sweep_cltw_kernel after compiling it (tools/verify_walk_asm.py): its taps are
    // written by asynchronous buffer loads issued from inline asm, which is only sound if hipcc neither
    // copies, spills nor reads a tap register between a load and the counted s_waitcnt that covers it.
    // A build whose disassembly does not show that is compiled again with this macro: the per-plane
    // kernel (4) takes the calls, bit-identical results, 7-17 % slower.
    (void)hw;
    return false;
#endif
    // (tap offsets are 32-bit byte offsets into a map, 0xf0000000 marks an out-of-bounds corner)
    return d->dtype == DFM_F32 && d->channels % 32 == 0 && hw % 32 == 0 && sweep_clt_supported(d, out) &&
           map_bytes(d) < 0xe0000000ull;
}

int sweep_clt_launch(const dfm_sweep_desc *d, const void *cur, const void *prev, const float *depths,
                     const float *cam2img, const float *cam2img_inv, const float *cur2prev, void *out,
                     void *workspace, void *stream, bool nhwc, bool walk)
{
    const SweepGeom g = sweep_make_geom(d);
    const size_t esz = d->dtype == DFM_BF16 ? 2 : 4;
    const size_t zero = ((size_t)d->channels * esz + 255) & ~(size_t)255;
    const size_t mb = map_bytes(d);
    hipStream_t st = (hipStream_t)stream;
    char *w8 = (char *)workspace;
    hipError_t e = hipMemsetAsync(w8, 0, zero, st);
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    const long long HW = (long long)d->h_in * d->w_in;
    dim3 pg((unsigned)((HW + 63) / 64), (d->channels + 31) / 32, d->batch);
    ClGrid tg;
    tg.batch = d->batch;
    const long long hw = (long long)g.h_out * g.w_out;
    tg.tiles = (int)((hw + 255) / 256);
    tg.lpi_shift = 3;
    tg.zero_slot = 0;
    tg.cur_slot = (unsigned)(zero / 16);
    tg.prev_slot = (unsigned)((zero + mb) / 16);
    const long long nb = (long long)tg.tiles * g.D * d->batch;
    if (nb > 2147483647ll) return set_error(DFM_ERR_UNSUPPORTED, "too many lattice points");
    const uint4 *cur_maps = (const uint4 *)(nhwc ? (const char *)cur : w8 + zero);
    const uint4 *prev_maps = (const uint4 *)(nhwc ? (const char *)prev : w8 + zero + mb);
    if (nhwc) {
        // channels-last maps are sampled where they lie
    } else if (d->dtype == DFM_F32) {
        hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg, dim3(256), 0, st, (const float *)cur,
                           (float *)(w8 + zero), d->channels, d->channels, HW);
        hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg, dim3(256), 0, st, (const float *)prev,
                           (float *)(w8 + zero + mb), d->channels, d->channels, HW);
    } else {
        hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg, dim3(256), 0, st, (const bf16_t *)cur,
                           (bf16_t *)(w8 + zero), d->channels, d->channels, HW);
        hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg, dim3(256), 0, st, (const bf16_t *)prev,
                           (bf16_t *)(w8 + zero + mb), d->channels, d->channels, HW);
    }
    const bool timed = profile_mark(stream, false);
    const bool walking = walk && sweep_cltw_supported(d, out);
    if (walking) {
        WalkGrid wg;
        wg.batch = d->batch;
        wg.tiles = (int)(hw / 32);
        wg.chunk_planes = g.D >= 48 ? 24 : (g.D + 1) & ~1;
#ifdef DFM_WALK_CHUNK_ENV  // experiments
        if (const char *e = getenv("DFM_WALK_CHUNK")) wg.chunk_planes = atoi(e);
#endif
        wg.chunks = (g.D + wg.chunk_planes - 1) / wg.chunk_planes;
        wg.passes = g.nblk / 8;
        const long long nw = (long long)((wg.tiles + 3) / 4) * 2 * wg.passes * wg.chunks * d->batch;
        if (nw > 2147483647ll) return set_error(DFM_ERR_UNSUPPORTED, "too many lattice points");
        hipLaunchKernelGGL(sweep_cltw_kernel, dim3((unsigned)nw), dim3(256), 0, st, g, sweep_make_fast(d), wg,
                           cur_maps, prev_maps, (unsigned)mb, depths, cam2img, cam2img_inv, cur2prev, (float *)out);
    } else if (d->dtype == DFM_F32)
        hipLaunchKernelGGL((sweep_clt_kernel<float, 32>), dim3((unsigned)nb), dim3(256), 0, st, g, tg,
                           (const uint4 *)workspace, cur_maps, prev_maps, depths, cam2img, cam2img_inv, cur2prev,
                           (float *)out);
    else
        hipLaunchKernelGGL((sweep_clt_kernel<bf16_t, 64>), dim3((unsigned)nb), dim3(256), 0, st, g, tg,
                           (const uint4 *)workspace, cur_maps, prev_maps, depths, cam2img, cam2img_inv, cur2prev,
                           (bf16_t *)out);
    if (timed) profile_mark(stream, true);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    sweep_set_last_kernel(walking ? 5 : 4);  // which body ran: 5 = the depth-walking kernel, 4 = per plane
    return DFM_OK;
}

}  // namespace dfm

extern "C" {

DFM_API size_t dfm_plane_sweep_cl_workspace_bytes(const dfm_sweep_desc *d)
{
    if (sweep_check_desc(d) != DFM_OK) return 0;
    const size_t zero = ((size_t)d->channels * (d->dtype == DFM_BF16 ? 2 : 4) + 255) & ~(size_t)255;
    return zero + 2 * map_bytes(d);
}

static int sweep_cl_impl(const dfm_sweep_desc *d, const void *cur, const void *prev, const float *depths,
                         const float *cam2img, const float *cam2img_inv, const float *cur2prev, void *out,
                         void *workspace, size_t workspace_bytes, void *stream, bool nhwc)
{
    int rc = sweep_check_desc(d);
    if (rc != DFM_OK) return rc;
    if (!cur || !prev || !depths || !cam2img || !cam2img_inv || !cur2prev || !out)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const int CB = d->dtype == DFM_BF16 ? 8 : 4;
    if (d->channels % CB)
        return set_error(DFM_ERR_UNSUPPORTED,
                         "channels-last output needs channels to be a multiple of 16 bytes");
    const SweepGeom g = sweep_make_geom(d);
    const size_t esz = d->dtype == DFM_BF16 ? 2 : 4;
    const size_t zero = ((size_t)d->channels * esz + 255) & ~(size_t)255;
    const size_t mb = map_bytes(d);
    if (!workspace || workspace_bytes < (nhwc ? zero : dfm_plane_sweep_cl_workspace_bytes(d)))
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than dfm_plane_sweep_cl_workspace_bytes");
    if (nhwc && (((uintptr_t)cur | (uintptr_t)prev) & 15))
        return set_error(DFM_ERR_INVALID_ARG, "channels-last feature maps must be 16-byte aligned");
    if (mb / 16 >= 0xffffffffull)
        return set_error(DFM_ERR_UNSUPPORTED, "feature maps too large for 32-bit tap slots");
    hipStream_t st = (hipStream_t)stream;
    char *w8 = (char *)workspace;
    hipError_t e = hipMemsetAsync(w8, 0, zero, st);
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    const long long HW = (long long)d->h_in * d->w_in;
    dim3 pg((unsigned)((HW + 63) / 64), (d->channels + 31) / 32, d->batch);
    ClGrid tg;
    tg.batch = d->batch;
    const long long hw = (long long)g.h_out * g.w_out;
    tg.tiles = (int)((hw + 255) / 256);
    int sh = 0;
    while ((1 << sh) < g.nblk && sh < 5) ++sh;  // lanes per (point, map): 1..32
    tg.lpi_shift = sh;
    tg.zero_slot = 0;
    tg.cur_slot = (unsigned)(zero / 16);
    tg.prev_slot = (unsigned)((zero + mb) / 16);
    const long long nb = (long long)tg.tiles * g.D * d->batch;
    if (nb > 2147483647ll) return set_error(DFM_ERR_UNSUPPORTED, "too many lattice points");
    const uint4 *cur_maps = (const uint4 *)(nhwc ? (const char *)cur : w8 + zero);
    const uint4 *prev_maps = (const uint4 *)(nhwc ? (const char *)prev : w8 + zero + mb);
    if (nhwc) {
        // the caller's maps ARE pixel-major: nothing to re-lay
    } else if (d->dtype == DFM_F32) {
        hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg, dim3(256), 0, st, (const float *)cur,
                           (float *)(w8 + zero), d->channels, d->channels, HW);
        hipLaunchKernelGGL(pack_pixel_major_kernel<float>, pg, dim3(256), 0, st, (const float *)prev,
                           (float *)(w8 + zero + mb), d->channels, d->channels, HW);
    } else {
        hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg, dim3(256), 0, st, (const bf16_t *)cur,
                           (bf16_t *)(w8 + zero), d->channels, d->channels, HW);
        hipLaunchKernelGGL(pack_pixel_major_kernel<bf16_t>, pg, dim3(256), 0, st,
                           (const bf16_t *)prev, (bf16_t *)(w8 + zero + mb), d->channels,
                           d->channels, HW);
    }
    const bool timed = profile_mark(stream, false);
    if (d->dtype == DFM_F32)
        hipLaunchKernelGGL(sweep_cl_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, g, tg,
                           (const uint4 *)workspace, cur_maps, prev_maps, depths, cam2img, cam2img_inv,
                           cur2prev, (uint4 *)out);
    else
        hipLaunchKernelGGL(sweep_cl_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, g, tg,
                           (const uint4 *)workspace, cur_maps, prev_maps, depths, cam2img, cam2img_inv,
                           cur2prev, (uint4 *)out);
    if (timed) profile_mark(stream, true);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

DFM_API int dfm_plane_sweep_fwd_channels_last(const dfm_sweep_desc *d, const void *cur,
                                              const void *prev, const float *depths,
                                              const float *cam2img, const float *cam2img_inv,
                                              const float *cur2prev, void *out, void *workspace,
                                              size_t workspace_bytes, void *stream)
{
    return sweep_cl_impl(d, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace,
                         workspace_bytes, stream, false);
}

/* NHWC feature maps in, the REFERENCE layout (B, 2C, D, h_out, w_out) out: the pixel-major-tap +
 * LDS-transpose kernel (sweep_clt_kernel) on the caller's maps -- the strided sweeps of config K
 * without their pack passes (21 % of the step).  DFM_ERR_UNSUPPORTED when that kernel cannot take the
 * shape (see sweep_clt_supported): the caller falls back to NCHW maps and dfm_plane_sweep_fwd. */
DFM_API int dfm_plane_sweep_fwd_from_nhwc(const dfm_sweep_desc *d, const void *cur, const void *prev,
                                          const float *depths, const float *cam2img,
                                          const float *cam2img_inv, const float *cur2prev, void *out,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = sweep_check_desc(d);
    if (rc != DFM_OK) return rc;
    if (!cur || !prev || !depths || !cam2img || !cam2img_inv || !cur2prev || !out)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (!dfm::sweep_clt_supported(d, out) || (((uintptr_t)cur | (uintptr_t)prev) & 15))
        return set_error(DFM_ERR_UNSUPPORTED, "channels-last feature maps: shape not covered by the transpose kernel");
    const size_t zero = ((size_t)d->channels * (d->dtype == DFM_BF16 ? 2 : 4) + 255) & ~(size_t)255;
    if (!workspace || workspace_bytes < zero)
        return set_error(DFM_ERR_WORKSPACE, "workspace smaller than one zero pixel");
    return dfm::sweep_clt_launch(d, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace, stream, true);
}

DFM_API int dfm_plane_sweep_fwd_nhwc(const dfm_sweep_desc *d, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    return sweep_cl_impl(d, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace,
                         workspace_bytes, stream, true);
}

/* Backward of a strided fp32 sweep, CUR map only, into a PIXEL-MAJOR gradient map: grad_cur is (B, H, W, C)
 * fp32 in memory (torch: a channels_last tensor of shape (B, C, H, W)), zero-initialised by the caller and
 * accumulated into with atomics; grad_out is the reference layout (B, 2C, D, h_out, w_out), fp32, of which the
 * first C channels are read.  sweep_bwdc_kernel (csrc/plane_sweep_cl.hip): a wave keeps the 3x3 pixel window a
 * point's cur taps stay in over all depth planes in registers and writes it out once.  The prev map comes from
 * dfm_plane_sweep_bwd_opts with opts->kernel = 8 (the LDS-atomic tile kernel, prev map only).
 * DFM_ERR_UNSUPPORTED unless fp32, channels % 32 == 0 and h_out * w_out % 16 == 0 (the caller then uses
 * dfm_plane_sweep_bwd for both maps).  Replaces autograd of the first F.grid_sample call of build_dfm_cost
 * (reference dfm_backbone.py:296-303). */
DFM_API int dfm_plane_sweep_bwd_cur_nhwc(const dfm_sweep_desc *d, const void *grad_out, const float *depths,
                                         const float *cam2img, const float *cam2img_inv, const float *cur2prev,
                                         float *grad_cur, void *stream)
{
    int rc = sweep_check_desc(d);
    if (rc != DFM_OK) return rc;
    if (!grad_out || !depths || !cam2img || !cam2img_inv || !cur2prev || !grad_cur)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    const long long hw = (long long)d->h_out * d->w_out;
    if (d->dtype != DFM_F32 || d->channels % 32 || hw % 16 || (((uintptr_t)grad_out | (uintptr_t)grad_cur) & 15) ||
        map_bytes(d) >= 0xe0000000ull || d->h_in >= 4096 || d->w_in >= 8192)
        return set_error(DFM_ERR_UNSUPPORTED, "pixel-major cur backward: fp32, channels % 32 == 0, h_out * w_out % 16 == 0");
    const SweepGeom g = sweep_make_geom(d);
    const int tiles = (int)(hw / 16), passes = g.nblk / 8;
    const long long nw = (long long)((tiles + 3) / 4) * passes * d->batch;
    if (nw > 2147483647ll) return set_error(DFM_ERR_UNSUPPORTED, "too many lattice points");
    hipLaunchKernelGGL(sweep_bwdc_kernel, dim3((unsigned)nw), dim3(256), 0, (hipStream_t)stream, g, sweep_make_fast(d),
                       d->batch, tiles, passes, grad_cur, (unsigned)map_bytes(d), depths, cam2img, cam2img_inv, cur2prev,
                       (const float *)grad_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

}  // extern "C"

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
// spread over 512 channel planes (4.6-5.4 TB/s, profiles/r01_store_microbench*.txt).
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

using namespace dfm;

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

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

int sweep_clt_launch(const dfm_sweep_desc *d, const void *cur, const void *prev, const float *depths,
                     const float *cam2img, const float *cam2img_inv, const float *cur2prev, void *out,
                     void *workspace, void *stream, bool nhwc)
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
    if (d->dtype == DFM_F32)
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
    rc = dfm::sweep_clt_launch(d, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace, stream, true);
    if (rc == DFM_OK) dfm::sweep_set_last_kernel(4);
    return rc;
}

DFM_API int dfm_plane_sweep_fwd_nhwc(const dfm_sweep_desc *d, const void *cur, const void *prev,
                                     const float *depths, const float *cam2img,
                                     const float *cam2img_inv, const float *cur2prev, void *out,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    return sweep_cl_impl(d, cur, prev, depths, cam2img, cam2img_inv, cur2prev, out, workspace,
                         workspace_bytes, stream, true);
}

}  // extern "C"

// sweep_conv.hip -- the plane-sweep cost volume fused into the first aggregation convolutions
// (last clause): build_dfm_cost -> dres0 / dres0_mono without the
// (B, 2C, D, H, W) volume ever touching HBM.
//
// Reference (mmdet3d/models/backbones/dfm_backbone.py):
//   :161-172  cost_raw = build_dfm_cost(cur, prev, ...)            (B, 2C, D, H, W)
//   :175      stereo   = dres0(cost_raw)        Conv3d(2C -> 32, 3, 1, 1) + GN + ReLU
//   :189      mono     = dres0_mono(cost_raw[:, :C])   Conv3d(C -> 32, 3, 1, 1) + GN + ReLU
// Config K: C = 32, volume 64 x 72 x 80 x 320 = 236 MB bf16 written once and read three times
// (two halves of dres0 through an fp32 partial of another 236 MB, then dres0_mono).  Here the
// sampler produces the halo'd cur | prev block of an output tile straight into the convolution's
// LDS slab; the kernel emits the two pre-norm convolution outputs (bf16 NDHWC, 59 MB each) and their
// per-channel GroupNorm moment partials.  HBM traffic: the two feature maps (L2-resident), two
// 59 MB outputs.  The kernel is MFMA-bound (3 x 102 GFLOP at config K).
//
// Workgroup = 4 waves (one per SIMD, 512 registers each), output tile 8 rows x 32 columns, walking a
// chunk of depth planes with a ring of THREE depth slabs in LDS (10 x 34 pixels x (32 cur + 32 prev)
// bf16 channels, rows pitched to 36 pixels, 16-byte blocks XOR-swizzled for conflict-free
// ds_read_b128 of the 16x16x32 B operand).
//   waves 0, 1 (S): stereo output channels 0-15 / 16-31, K = 64 input channels x 27 taps: all 54
//                   weight fragments (216 registers) resident, v_mfma_f32_16x16x32_bf16 with
//                   D[cout 16][pixel 16] = W[cout][k 32] X[k][pixel]: 864 MFMAs per depth plane;
//   waves 2, 3 (M): mono output channels 0-15 / 16-31, K = 32 x 27 (27 fragments, 432 MFMAs per
//                   plane) -- and, in the MFMA time they do not need, the PRODUCER of the next depth
//                   slab: lane = lattice point of one map -> fp32 sampling position in the reference's op
//                   order (sweep_point_map) -> bilinear footprint -> its four corners' 64 bytes each from
//                   the pixel-major (NHWC) feature maps -> ATen-order blend -> bf16 -> LDS.
// No input channel split across waves (each wave owns 16 output channels with its full K), so there is
// no cross-wave reduction.  Two barriers per plane: X(d) "slab d-1 is free" after everyone's kd = 0
// taps, Y(d) "slab d+1 is complete" before anyone's kd = 2 taps; the M waves fill slab d+2 between X(d)
// and Y(d+1).
#include "dfm_common.h"

#include <stdlib.h>

using namespace dfm;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int SC_C = 32;                      // channels per feature map = output channels per branch
constexpr int SC_TR = 8, SC_TW = 32;          // output tile (rows x columns)
constexpr int SC_SH = SC_TR + 2, SC_SW = SC_TW + 2;
constexpr int SC_PITCH = 36;                  // slab row pitch in pixels (multiple of 4: row-independent swizzle)
constexpr int SC_ITEMS = SC_SH * SC_SW;       // 340 sampled pixels per map and slab
constexpr int SC_MAP_BYTES = SC_SH * SC_PITCH * 64;   // 23040
constexpr int SC_SLAB_BYTES = 2 * SC_MAP_BYTES;       // cur | prev
constexpr int SC_RING = 3;
constexpr int SC_DUMP_BYTES = 64 * 64;        // per-wave dump row (stores of the 11th chunk's empty tail)
constexpr int SC_LDS_BYTES = SC_RING * SC_SLAB_BYTES + 4 * SC_DUMP_BYTES;  // 154624
constexpr int SC_CHUNKS = (2 * SC_ITEMS + 63) / 64;   // 11 chunks of 64 (pixel, map) items per slab
constexpr int SC_FRAGS = 54;                  // weight fragments per wave role (M uses the first 27)

struct SCGeom {
    SweepGeom g;
    SweepFast f;
    int32_t tiles_w, tiles_h, dchunk, nchunks;
    int32_t per_xcd;       // work items per XCD (the grid is 8 * per_xcd workgroups)
    long long total_work;  // batch * tiles * nchunks
    int32_t ablate;  // debug builds only (DFM_SC_ABLATE): 1 no steady-state production, 2 no stereo MFMAs,
                     // 4 no mono MFMAs, 8 no epilogue
    unsigned long long *trace;  // debug builds only (dfm_debug_set_sc_trace): s_memtime stamps of one workgroup
};

#ifdef DFM_DEBUG_HOOKS
#define SC_AB(sc, bit) (((sc).ablate & (bit)) != 0)
// trace[wave][plane < 16][stamp < 16] of the workgroup (tile = tiles / 2, chunk 0, sample 0)
#define SC_STAMP(i)                                                                                  \
    do {                                                                                             \
        if (traced && lane == 0 && d - d0 < 16)                                                      \
            sc.trace[((size_t)wave * 16 + (d - d0)) * 16 + (i)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
unsigned long long *g_sc_trace = nullptr;
#else
#define SC_AB(sc, bit) false
#define SC_STAMP(i) do { } while (0)
#endif

// ---- weights -> MFMA A-operand fragments -------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 A operand: lane l holds A[m = l & 15][k = 8 (l >> 4) + j], j = 0..7.
// packed[role][frag][lane][8]:
//   role 0 / 1 (stereo couts 0-15 / 16-31): frag = tap * 2 + s, s = 0: input channels 0-31 (the cur
//       half of the volume), s = 1: channels 32-63 (prev half)   -- w_stereo (32, 64, 3, 3, 3)
//   role 2 / 3 (mono couts 0-15 / 16-31):   frag = tap                -- w_mono   (32, 32, 3, 3, 3)
template <typename TW>
__global__ void sweep_conv_pack_kernel(const TW *__restrict__ ws, const TW *__restrict__ wm, bf16_t *__restrict__ out)
{
    const int frag = blockIdx.x, role = blockIdx.y, l = threadIdx.x;
    const int cout = (role & 1) * 16 + (l & 15), k0 = 8 * (l >> 4);
    bf16_t *o = out + (((size_t)role * SC_FRAGS + frag) * 64 + l) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.0f;
        if (role < 2) {
            const int tap = frag >> 1, s = frag & 1;
            const size_t idx = ((size_t)cout * 64 + s * 32 + k0 + j) * 27 + tap;
            if constexpr (sizeof(TW) == 4) v = ((const float *)ws)[idx]; else v = bf16_to_f32(((const bf16_t *)ws)[idx]);
        } else if (frag < 27) {
            const size_t idx = ((size_t)cout * 32 + k0 + j) * 27 + frag;
            if constexpr (sizeof(TW) == 4) v = ((const float *)wm)[idx]; else v = bf16_to_f32(((const bf16_t *)wm)[idx]);
        }
        o[j] = f32_to_bf16(v);
    }
}

__device__ __forceinline__ void wg_barrier()
{
    // this wave's LDS reads / writes have completed; global stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- producer --------------------------------------------------------------------------------
// A depth slab is 11 chunks of 64 (pixel, map) items.  A chunk goes through three stages that the
// callers place around their MFMA phases: footprints (lane = item: sampling position, corner slots and
// weights -> a per-wave LDS scratch), loads (lane = (item, 16-byte channel block): the 16 tap loads of
// its four items are issued) and blend (ATen-order blend, bf16, swizzled store into the slab).
struct ProdCtx {
    SweepGeom g;
    SweepFast f;
    const uint4 *cur, *prev, *zero_page;
    const float *depths, *P, *Pinv, *Tm;
    int b, h0, w0, lane;
    int ablate;  // debug builds: 16 no tap loads, 32 no footprint arithmetic
};

struct Taps {
    uint4 tap[4][4];  // [corner nw / ne / sw / se][16-byte channel block]
    float w[4];
    int dst;          // byte offset of the item's pixel in the slab (block 0, unswizzled); < 0: dump row
    int swz;
};

// sweep_point_map (dfm_common.h) for a lane that samples either map: the same fp32 operations in the
// same order -- the prev map's extra cur2prev transform is computed by every lane and selected -- and no
// branch in the FAST specialisation (no flip, img_scale_factor 1, feat_sample_factor a power of two: the
// reference's un-augmented test-time geometry), so that two chunks' footprint arithmetic is ONE basic
// block the scheduler can interleave.
template <bool FAST>
__device__ __forceinline__ void sc_point(const SweepGeom &g, const SweepFast &f, const float *__restrict__ P,
                                         const float *__restrict__ Pinv, const float *__restrict__ Tm, float depth,
                                         int hi, int wi, bool is_prev, float &ox, float &oy)
{
    float x = ((float)wi * g.fsf) * g.csf;
    float y = ((float)hi * g.fsf) * g.csf;
    x = x + g.crop_x;
    y = y + g.crop_y;
    if constexpr (!FAST) {
        if (!f.scale_is_one) {
            x = x / g.scale;
            y = y / g.scale;
        }
        if (g.flip) x = g.org_w - x;
    }
    const float h0 = x * depth, h1 = y * depth, h2 = depth;
    float X0 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 0);
    float X1 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 4);
    float X2 = dot4_chain(h0, h1, h2, 1.0f, Pinv + 8);
    const float Y0 = dot4_chain(X0, X1, X2, 1.0f, Tm + 0);
    const float Y1 = dot4_chain(X0, X1, X2, 1.0f, Tm + 4);
    const float Y2 = dot4_chain(X0, X1, X2, 1.0f, Tm + 8);
    X0 = is_prev ? Y0 : X0;
    X1 = is_prev ? Y1 : X1;
    X2 = is_prev ? Y2 : X2;
    const float a = dot4_chain(X0, X1, X2, 1.0f, P + 0);
    const float b = dot4_chain(X0, X1, X2, 1.0f, P + 4);
    const float c = dot4_chain(X0, X1, X2, 1.0f, P + 8);
    float u = a / c, v = b / c;
    if constexpr (!FAST) {
        if (g.flip) u = g.org_w - u;
        if (!f.scale_is_one) {
            u = u * g.scale;
            v = v * g.scale;
        }
    }
    u = u - g.crop_x;
    v = v - g.crop_y;
    if (FAST || f.fsf_pow2) {
        u = u * f.inv_fsf;
        v = v * f.inv_fsf;
    } else {
        u = u / g.fsf;
        v = v / g.fsf;
    }
    const float wm1 = (float)(g.w_in - 1), hm1 = (float)(g.h_in - 1);
    const float nx = u / wm1 * 2.0f - 1.0f;
    const float ny = v / hm1 * 2.0f - 1.0f;
    ox = ((nx + 1.0f) * 0.5f) * wm1;
    oy = ((ny + 1.0f) * 0.5f) * hm1;
}

// lane = (pixel, map) item of chunk c: sampling position, footprint, and the 16 tap loads (4 corners x
// 4 channel blocks: 64 contiguous bytes per corner) -- no LDS round trip, nothing recomputed per block
template <bool FAST>
__device__ __forceinline__ void issue_stage(const ProdCtx &pc, int zi, int c, Taps &t)
{
    const SweepGeom &g = pc.g;
    const bool zok = zi >= 0 && zi < g.D;
    const float depth = pc.depths[zok ? zi : 0];
    const unsigned map_slot = (unsigned)pc.b * (unsigned)(g.h_in * g.w_in) * 4u;
    const int q = c * 64 + pc.lane;
    const int map = q >= SC_ITEMS ? 1 : 0;
    const int p = q - map * SC_ITEMS;
    const int j = p / SC_SW, i = p - j * SC_SW;
    const int hh = pc.h0 - 1 + j, ww = pc.w0 - 1 + i;
    const bool inside = zok && q < 2 * SC_ITEMS && hh >= 0 && hh < g.h_out && ww >= 0 && ww < g.w_out;
    // branch-free: the position of a (clamped) lattice point is always computed; an item outside the
    // volume (the convolution's zero padding) or past the slab's last item gets zero taps and weights
    float x = (float)ww, y = (float)hh;
    if (!SC_AB(pc, 32))
        sc_point<FAST>(g, pc.f, pc.P, pc.Pinv, pc.Tm, depth, min(max(hh, 0), g.h_out - 1), min(max(ww, 0), g.w_out - 1),
                       map != 0, x, y);
    const Tap tp = make_tap(x, y, g.h_in, g.w_in);
    const int i00 = tp.iy * g.w_in + tp.ix, i01 = i00 + tp.dx;
    const int i10 = i00 + tp.dy * g.w_in, i11 = i10 + tp.dx;
    const uint4 *mp = (map ? pc.prev : pc.cur) + map_slot;
    const uint4 *src[4];
    src[0] = (inside && (tp.ok & 1u)) ? mp + (unsigned)i00 * 4u : pc.zero_page;
    src[1] = (inside && (tp.ok & 2u)) ? mp + (unsigned)i01 * 4u : pc.zero_page;
    src[2] = (inside && (tp.ok & 4u)) ? mp + (unsigned)i10 * 4u : pc.zero_page;
    src[3] = (inside && (tp.ok & 8u)) ? mp + (unsigned)i11 * 4u : pc.zero_page;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (!SC_AB(pc, 16)) t.tap[k][blk] = src[k][blk];
            else t.tap[k][blk] = make_uint4((unsigned)(size_t)src[k], 0, 0, 0);
        }
    t.w[0] = inside ? tp.nw : 0.0f;
    t.w[1] = inside ? tp.ne : 0.0f;
    t.w[2] = inside ? tp.sw : 0.0f;
    t.w[3] = inside ? tp.se : 0.0f;
    t.swz = ((j & 1) ^ ((i >> 2) & 1)) << 1;
    // items past the slab's last one (the 11th chunk's tail) store into the wave's dump row
    t.dst = q < 2 * SC_ITEMS ? map * SC_MAP_BYTES + (j * SC_PITCH + i) * 64 : -1 - (pc.lane << 6);
}

__device__ __forceinline__ void blend_stage(const Taps &t, unsigned char *slab, unsigned char *dump)
{
    // ATen's accumulation order (plane_sweep_cl.hip: blend4), two channels per instruction:
    // v_pk_mul_f32 / v_pk_fma_f32 are the same IEEE operations per element
    const f32x2_t w0 = {t.w[0], t.w[0]}, w1 = {t.w[1], t.w[1]}, w2 = {t.w[2], t.w[2]}, w3 = {t.w[3], t.w[3]};
    unsigned char *px = t.dst >= 0 ? slab + t.dst : dump + (-1 - t.dst);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        float a[8], bb[8], cc[8], dd[8];
        unpack16(t.tap[0][blk], a);
        unpack16(t.tap[1][blk], bb);
        unpack16(t.tap[2][blk], cc);
        unpack16(t.tap[3][blk], dd);
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_t a2 = {a[e], a[e + 1]}, b2 = {bb[e], bb[e + 1]};
            const f32x2_t c2 = {cc[e], cc[e + 1]}, d2 = {dd[e], dd[e + 1]};
            f32x2_t acc = a2 * w0;
            acc = __builtin_elementwise_fma(b2, w1, acc);
            acc = __builtin_elementwise_fma(c2, w2, acc);
            acc = __builtin_elementwise_fma(d2, w3, acc);
            r[e] = acc[0];
            r[e + 1] = acc[1];
        }
        const uint4 v = make_uint4(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]), pack_bf16x2(r[4], r[5]),
                                   pack_bf16x2(r[6], r[7]));
        *(uint4 *)(px + ((blk ^ t.swz) << 4)) = v;
    }
}

// one chunk start to finish
template <bool FAST>
__device__ __forceinline__ void produce_one(const ProdCtx &pc, int zi, int c, unsigned char *slab, unsigned char *dump)
{
    Taps t;
    issue_stage<FAST>(pc, zi, c, t);
    blend_stage(t, slab, dump);
}

// two chunks with their independent instruction streams side by side
template <bool FAST>
__device__ __forceinline__ void produce_two(const ProdCtx &pc, int zi_a, int c_a, unsigned char *slab_a, int zi_b, int c_b,
                                            unsigned char *slab_b, unsigned char *dump)
{
    Taps ta, tb;
    issue_stage<FAST>(pc, zi_a, c_a, ta);
    issue_stage<FAST>(pc, zi_b, c_b, tb);
    blend_stage(ta, slab_a, dump);
    blend_stage(tb, slab_b, dump);
}

// ---- consumer: the 9 taps of one kernel depth slice on the whole tile --------------------------
// NS = 2 (stereo): fragments (tap, cur) and (tap, prev) into the same accumulator; NS = 1 (mono)
template <int NS, int KD>
__device__ __forceinline__ void conv_phase(f32x4_t (&acc)[2][SC_TR], const bf16x8_t *wf, const unsigned char *slab, int lane)
{
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            const int col = fr * 16 + kw + n;
            const int kgx = kg ^ (((col >> 2) & 1) << 1);
            const unsigned char *base = slab + col * 64;
            bf16x8_t xc[SC_SH], xp[SC_SH];
#pragma unroll
            for (int rr = 0; rr < SC_SH; ++rr) {
                const int off = rr * SC_PITCH * 64 + ((kgx ^ ((rr & 1) << 1)) << 4);
                xc[rr] = *(const bf16x8_t *)(base + off);
                if constexpr (NS == 2) xp[rr] = *(const bf16x8_t *)(base + SC_MAP_BYTES + off);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int tap = (KD * 3 + kh) * 3 + kw;
#pragma unroll
                for (int r = 0; r < SC_TR; ++r)
                    acc[fr][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap * NS], xc[r + kh], acc[fr][r], 0, 0, 0);
                if constexpr (NS == 2) {
#pragma unroll
                    for (int r = 0; r < SC_TR; ++r)
                        acc[fr][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap * 2 + 1], xp[r + kh], acc[fr][r], 0, 0, 0);
                }
            }
        }
    }
}

// bf16 stores of a wave's 16 output channels + the shifted per-channel moments of the stored values
struct Moments {
    float k[4], s1[4], s2[4], cnt;
    bool seeded;
};

// FULL: the tile lies entirely inside the volume (workgroup-uniform): no per-store predicate, no
// exec-mask regions between the MFMA phases
template <bool FULL>
__device__ __forceinline__ void epilogue_t(const SCGeom &sc, const f32x4_t (&acc)[2][SC_TR], bf16_t *__restrict__ y, int n,
                                           int d, int h0, int w0, int cbase, int lane, Moments &m)
{
    const SweepGeom &g = sc.g;
    const int px = lane & 15, cg = lane >> 4;
    static_assert(SC_TR % 2 == 0, "rows are stored in pairs");
#pragma unroll
    for (int fr = 0; fr < 2; ++fr) {
#pragma unroll
        for (int r2 = 0; r2 < SC_TR; r2 += 2) {
            u32x2_t pks[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = r2 + e;
                const int h = h0 + r, w = w0 + fr * 16 + px;
                const bool ok = FULL || (h < g.h_out && w < g.w_out);
                const u32x2_t pk = {pack_bf16x2(acc[fr][r][0], acc[fr][r][1]), pack_bf16x2(acc[fr][r][2], acc[fr][r][3])};
                pks[e] = pk;
                const float q[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u),
                                    __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)};
                if (!m.seeded) {
                    // the shift is common to the 16 pixel lanes of a channel group: the value of the group's
                    // first lane at the wave's first output position (in bounds by construction)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m.k[j] = __shfl(q[j], lane & 48);
                    m.seeded = true;
                }
                const float mk = ok ? 1.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dv = ok ? q[j] - m.k[j] : 0.0f;
                    m.s1[j] += dv;
                    m.s2[j] = __builtin_fmaf(dv, dv, m.s2[j]);
                }
                m.cnt += mk;
            }
            // 16-byte stores (round 6): a lane holds 4 channels of its pixel for each of the two rows; the lane 16
            // further holds the next 4 channels of the same pixels.  The even channel group keeps row r2 and receives
            // the partner's piece of it, the odd group gets both pieces of row r2 + 1 (v_permlane16_swap, one
            // instruction per dword): one 16-byte store per lane and row pair instead of two 8-byte ones.
            const auto sx = __builtin_amdgcn_permlane16_swap(pks[0].x, pks[1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(pks[0].y, pks[1].y, false, false);
            const dfm_u32x4 q16 = {sx[0], sy[0], sx[1], sy[1]};
            const int r = r2 + (cg & 1);
            const int h = h0 + r, w = w0 + fr * 16 + px;
            const bool ok = FULL || (h < g.h_out && w < g.w_out);
            const size_t vox = (((size_t)n * g.D + d) * g.h_out + h) * g.w_out + w;
            if (ok) *(dfm_u32x4 *)(y + vox * SC_C + cbase + 4 * (cg & ~1)) = q16;
        }
    }
}

__device__ __forceinline__ void epilogue(const SCGeom &sc, const f32x4_t (&acc)[2][SC_TR], bf16_t *__restrict__ y, int n,
                                         int d, int h0, int w0, int cbase, int lane, Moments &m)
{
    if (h0 + SC_TR <= sc.g.h_out && w0 + SC_TW <= sc.g.w_out)
        epilogue_t<true>(sc, acc, y, n, d, h0, w0, cbase, lane, m);
    else
        epilogue_t<false>(sc, acc, y, n, d, h0, w0, cbase, lane, m);
}

__device__ __forceinline__ void write_moments(const Moments &m, float *__restrict__ stats, int n, int cbase, int splits,
                                              int sidx, int lane)
{
    float cn = m.cnt;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) cn += __shfl_xor(cn, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a1 = m.s1[j], a2 = m.s2[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            a1 += __shfl_xor(a1, o);
            a2 += __shfl_xor(a2, o);
        }
        float mean = 0.0f, m2 = 0.0f;
        if (cn > 0.0f) {
            const float a = a1 / cn;
            mean = m.k[j] + a;
            m2 = fmaxf(a2 - a1 * a, 0.0f);
        }
        if ((lane & 15) == 0) {
            const int c = cbase + 4 * (lane >> 4) + j;
            float *o3 = stats + (((size_t)n * SC_C + c) * splits + sidx) * 3;
            o3[0] = cn; o3[1] = mean; o3[2] = m2;
        }
    }
}

template <bool FAST>
__global__ __launch_bounds__(256, 1) void sweep_conv_kernel(
    SCGeom sc, const uint4 *__restrict__ cur, const uint4 *__restrict__ prev, const float *__restrict__ depths,
    const float *__restrict__ P, const float *__restrict__ Pinv, const float *__restrict__ Tm,
    const uint4 *__restrict__ wpack, const uint4 *__restrict__ zero_page, bf16_t *__restrict__ y_stereo,
    bf16_t *__restrict__ y_mono, float *__restrict__ st_stereo, float *__restrict__ st_mono)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware work order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own
    // 4 MB L2), so the linear work list (sample, tile, depth chunk -- chunk fastest) is cut into 8
    // contiguous pieces, one per XCD: the depth chunks of a tile (same cur pixels, neighbouring prev
    // pixels) and adjacent tiles (shared halo) hit the same L2.  Measured on the traced build: with the
    // tiles dealt across all XCDs every L2 sees both whole feature maps and the tap loads miss to the
    // infinity cache (~5k cycles of load-issue stall per 64-item chunk).
    const int per_xcd = sc.per_xcd;
    const long long work = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || work >= sc.total_work) return;  // (whole workgroup: before any barrier)
    const int chunk_id = (int)(work % sc.nchunks);
    const long long tw_ = work / sc.nchunks;
    const int tile = (int)(tw_ % (sc.tiles_w * sc.tiles_h));
    const int tw = tile % sc.tiles_w, th = tile / sc.tiles_w;
    const int h0 = th * SC_TR, w0 = tw * SC_TW;
    const int d0 = chunk_id * sc.dchunk, d1 = min(d0 + sc.dchunk, sc.g.D);
    const int n = (int)(tw_ / (sc.tiles_w * sc.tiles_h));
    unsigned char *dump = lds + SC_RING * SC_SLAB_BYTES + wave * SC_DUMP_BYTES;
    auto slab_of = [&](int dz) -> unsigned char * { return lds + ((dz - d0 + 1) % SC_RING) * SC_SLAB_BYTES; };

    const bool stereo = wave < 2;
#ifdef DFM_DEBUG_HOOKS
    const bool traced = sc.trace && tile == (sc.tiles_w * sc.tiles_h) / 2 && chunk_id == 0 && n == 0;
#endif
    ProdCtx pc;
    pc.g = sc.g; pc.f = sc.f;
    pc.cur = cur; pc.prev = prev; pc.zero_page = zero_page;
    pc.depths = depths; pc.P = P + n * 16; pc.Pinv = Pinv + n * 16; pc.Tm = Tm + n * 16;
    pc.b = n; pc.h0 = h0; pc.w0 = w0; pc.lane = lane;
    pc.ablate = sc.ablate;
    // prologue: all four waves fill slabs d0-1, d0, d0+1: 33 chunks dealt round-robin, two at a time
    for (int ga = wave; ga < 3 * SC_CHUNKS; ga += 8) {
        const int gb = ga + 4;
        const int sa = ga / SC_CHUNKS, sb = gb / SC_CHUNKS;
        if (gb < 3 * SC_CHUNKS)
            produce_two<FAST>(pc, d0 - 1 + sa, ga - sa * SC_CHUNKS, slab_of(d0 - 1 + sa), d0 - 1 + sb, gb - sb * SC_CHUNKS,
                              slab_of(d0 - 1 + sb), dump);
        else
            produce_one<FAST>(pc, d0 - 1 + sa, ga - sa * SC_CHUNKS, slab_of(d0 - 1 + sa), dump);
    }
    wg_barrier();

    // weights: resident in registers for the whole launch
    bf16x8_t wf[SC_FRAGS];
    {
        const uint4 *wp = wpack + ((size_t)wave * SC_FRAGS) * 64 + lane;
#pragma unroll
        for (int f = 0; f < SC_FRAGS; ++f) {
            if (f < 27 || stereo) {
                const uint4 q = wp[(size_t)f * 64];
                __builtin_memcpy(&wf[f], &q, 16);
            }
        }
    }
    Moments mom;
#pragma unroll
    for (int j = 0; j < 4; ++j) { mom.k[j] = 0.0f; mom.s1[j] = 0.0f; mom.s2[j] = 0.0f; }
    mom.cnt = 0.0f;
    mom.seeded = false;
    const int cbase = (wave & 1) * 16;
    // Steady state, plane d: slab d+2 (11 chunks) is produced between X(d) and Y(d+1) by the two M waves
    // (432 MFMAs per plane against the S waves' 864): chunks {0,2 | 4,6 | 8,10} and {1,3 | 5,7 | 9}.  The
    // position arithmetic and tap loads of a pair are issued BEFORE an MFMA phase and blended after it:
    // the loads' latency passes under the phase's MFMAs (sched_barrier pins the stage boundaries -- left
    // to itself the scheduler hoists the blend arithmetic into the MFMA stream, which then stalls on vmcnt).
    // Measured alternatives (tools/sweep_conv_trace.py, profiles/archive/r03_*sweep_conv*): a chunk costs an M wave
    // ~3.4k cycles this way; produced inside an S wave (512-register pressure, latency exposed) 6.5-7k, so
    // the S waves only run MFMAs in the steady state.
    if (stereo) {
        for (int d = d0; d < d1; ++d) {
            f32x4_t acc[2][SC_TR];
#pragma unroll
            for (int fr = 0; fr < 2; ++fr)
#pragma unroll
                for (int r = 0; r < SC_TR; ++r) acc[fr][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            SC_STAMP(0);
            if (!SC_AB(sc, 2)) conv_phase<2, 0>(acc, wf, slab_of(d - 1), lane);
            SC_STAMP(1);
            wg_barrier();  // X(d): slab d-1 is free
            SC_STAMP(2);
            if (!SC_AB(sc, 2)) conv_phase<2, 1>(acc, wf, slab_of(d), lane);
            SC_STAMP(3);
            wg_barrier();  // Y(d): slab d+1 is complete
            SC_STAMP(4);
            if (!SC_AB(sc, 2)) conv_phase<2, 2>(acc, wf, slab_of(d + 1), lane);
            SC_STAMP(5);
            if (!SC_AB(sc, 8)) epilogue(sc, acc, y_stereo, n, d, h0, w0, cbase, lane, mom);
            SC_STAMP(6);
        }
        write_moments(mom, st_stereo, n, cbase, sc.tiles_w * sc.tiles_h * sc.nchunks, tile * sc.nchunks + chunk_id, lane);
    } else {
        const int share = wave - 2;
        for (int d = d0; d < d1; ++d) {
            f32x4_t acc[2][SC_TR];
#pragma unroll
            for (int fr = 0; fr < 2; ++fr)
#pragma unroll
                for (int r = 0; r < SC_TR; ++r) acc[fr][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const bool more = d + 2 <= d1 && !SC_AB(sc, 1);  // slab d+2 is plane d+1's far slab (chunk halo included)
            unsigned char *dst = slab_of(d + 2);
            Taps ta, tb;
            SC_STAMP(0);
            if (more) {  // pair 1
                issue_stage<FAST>(pc, d + 2, share, ta);
                issue_stage<FAST>(pc, d + 2, 2 + share, tb);
            }
            __builtin_amdgcn_sched_barrier(0);
            SC_STAMP(1);
            if (!SC_AB(sc, 4)) conv_phase<1, 0>(acc, wf, slab_of(d - 1), lane);
            SC_STAMP(2);
            wg_barrier();  // X(d)
            __builtin_amdgcn_sched_barrier(0);
            SC_STAMP(3);
            if (more) {
                blend_stage(ta, dst, dump);
                blend_stage(tb, dst, dump);
                issue_stage<FAST>(pc, d + 2, 4 + share, ta);   // pair 2
                issue_stage<FAST>(pc, d + 2, 6 + share, tb);
            }
            __builtin_amdgcn_sched_barrier(0);
            SC_STAMP(4);
            if (!SC_AB(sc, 4)) conv_phase<1, 1>(acc, wf, slab_of(d), lane);
            __builtin_amdgcn_sched_barrier(0);
            SC_STAMP(5);
            if (more) {
                blend_stage(ta, dst, dump);
                blend_stage(tb, dst, dump);
                issue_stage<FAST>(pc, d + 2, 8 + share, ta);   // pair 3 (the 11th chunk: M0 only)
                if (share == 0) issue_stage<FAST>(pc, d + 2, 10, tb);
            }
            __builtin_amdgcn_sched_barrier(0);
            SC_STAMP(6);
            wg_barrier();  // Y(d)
            SC_STAMP(7);
            if (!SC_AB(sc, 4)) conv_phase<1, 2>(acc, wf, slab_of(d + 1), lane);
            __builtin_amdgcn_sched_barrier(0);
            SC_STAMP(8);
            if (more) {
                blend_stage(ta, dst, dump);
                if (share == 0) blend_stage(tb, dst, dump);
            }
            SC_STAMP(9);
            if (!SC_AB(sc, 8)) epilogue(sc, acc, y_mono, n, d, h0, w0, cbase, lane, mom);
            SC_STAMP(10);
        }
        write_moments(mom, st_mono, n, cbase, sc.tiles_w * sc.tiles_h * sc.nchunks, tile * sc.nchunks + chunk_id, lane);
    }
}

// depth chunks: whole rounds over the 256 CUs (one workgroup per CU) with the least halo re-sampling
int sc_depth_chunk(long long cols, int d, int depth_chunk)
{
    if (depth_chunk > 0) return std::min(depth_chunk, d);
    int best = d;
    double best_cost = 1e30;
    for (int chunks = 1; chunks <= d; ++chunks) {
        const int dc = (d + chunks - 1) / chunks;
        const long long wgs = cols * ((d + dc - 1) / dc);
        const long long rounds = (wgs + 255) / 256;
        // a workgroup's time ~ prologue (3 slabs by 4 waves ~ 1.2 planes) + dc planes
        const double cost = (double)rounds * (dc + 1.2);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = dc; }
    }
    return best;
}

int sc_check(const dfm_sweep_desc *d)
{
    const int rc = sweep_check_desc(d);
    if (rc != DFM_OK) return rc;
    if (d->dtype != DFM_BF16) return set_error(DFM_ERR_UNSUPPORTED, "the fused sweep + dres0 kernel takes bf16 feature maps");
    if (d->channels != SC_C) return set_error(DFM_ERR_UNSUPPORTED, "the fused sweep + dres0 kernel takes 32-channel feature maps");
    if ((long long)d->batch * d->h_in * d->w_in * 4 >= (1ll << 32) - 1)
        return set_error(DFM_ERR_UNSUPPORTED, "feature maps too large for 32-bit tap slots");
    if (d->batch > 65535) return set_error(DFM_ERR_UNSUPPORTED, "batch > 65535");
    return DFM_OK;
}

}  // namespace

#ifdef DFM_DEBUG_HOOKS
// debug builds only (not in dfm_hip.h): device buffer of 4 waves x 16 planes x 16 stamps (u64)
extern "C" DFM_API void dfm_debug_set_sc_trace(void *buf) { g_sc_trace = (unsigned long long *)buf; }
#endif

extern "C" DFM_API size_t dfm_sweep_conv_weight_bytes(void) { return (size_t)4 * SC_FRAGS * 64 * 16 + 4096; }

extern "C" DFM_API int dfm_sweep_conv_pack_weights(const void *w_stereo, const void *w_mono, int32_t weight_dtype,
                                                   void *packed, void *stream)
{
    if (!w_stereo || !w_mono || !packed) return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (weight_dtype != DFM_F32 && weight_dtype != DFM_BF16)
        return set_error(DFM_ERR_UNSUPPORTED, "weight dtype must be DFM_F32 or DFM_BF16");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync((char *)packed + (size_t)4 * SC_FRAGS * 64 * 16, 0, 4096, st);  // the zero pixel
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    if (weight_dtype == DFM_F32)
        hipLaunchKernelGGL(sweep_conv_pack_kernel<float>, dim3(SC_FRAGS, 4), dim3(64), 0, st, (const float *)w_stereo,
                           (const float *)w_mono, (bf16_t *)packed);
    else
        hipLaunchKernelGGL(sweep_conv_pack_kernel<bf16_t>, dim3(SC_FRAGS, 4), dim3(64), 0, st, (const bf16_t *)w_stereo,
                           (const bf16_t *)w_mono, (bf16_t *)packed);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

extern "C" DFM_API int dfm_sweep_conv_stats_splits(const dfm_sweep_desc *desc, int32_t depth_chunk)
{
    if (sc_check(desc) != DFM_OK) return 0;
    const long long cols = (long long)((desc->w_out + SC_TW - 1) / SC_TW) * ((desc->h_out + SC_TR - 1) / SC_TR);
    const int dc = sc_depth_chunk(cols * desc->batch, desc->num_depths, depth_chunk);
    return (int)(cols * ((desc->num_depths + dc - 1) / dc));
}

extern "C" DFM_API int dfm_sweep_conv_fwd(const dfm_sweep_desc *desc, const void *cur_nhwc, const void *prev_nhwc,
                                          const float *depths, const float *cam2img, const float *cam2img_inv,
                                          const float *cur2prev, const void *packed_weights, void *y_stereo, void *y_mono,
                                          float *stats_stereo, float *stats_mono, int32_t depth_chunk, void *stream)
{
    int rc = sc_check(desc);
    if (rc != DFM_OK) return rc;
    if (!cur_nhwc || !prev_nhwc || !depths || !cam2img || !cam2img_inv || !cur2prev || !packed_weights || !y_stereo ||
        !y_mono || !stats_stereo || !stats_mono)
        return set_error(DFM_ERR_INVALID_ARG, "NULL device pointer");
    if (((uintptr_t)cur_nhwc | (uintptr_t)prev_nhwc | (uintptr_t)packed_weights | (uintptr_t)y_stereo | (uintptr_t)y_mono) & 15)
        return set_error(DFM_ERR_INVALID_ARG, "feature maps, weights and outputs must be 16-byte aligned");
    SCGeom sc;
    sc.g = sweep_make_geom(desc);
    sc.f.scale_is_one = desc->img_scale_factor == 1.0f;
    {
        int e = 0;
        const float m = frexpf(desc->feat_sample_factor, &e);
        sc.f.fsf_pow2 = (m == 0.5f) && e > -60 && e < 60;
        sc.f.inv_fsf = 1.0f / desc->feat_sample_factor;
    }
    sc.tiles_w = (desc->w_out + SC_TW - 1) / SC_TW;
    sc.tiles_h = (desc->h_out + SC_TR - 1) / SC_TR;
    const long long cols = (long long)sc.tiles_w * sc.tiles_h;
    sc.dchunk = sc_depth_chunk(cols * desc->batch, desc->num_depths, depth_chunk);
    sc.nchunks = (desc->num_depths + sc.dchunk - 1) / sc.dchunk;
    sc.ablate = 0;
    sc.trace = nullptr;
#ifdef DFM_DEBUG_HOOKS
    sc.trace = g_sc_trace;
    {
        const char *ab = getenv("DFM_SC_ABLATE");
        sc.ablate = ab ? atoi(ab) : 0;
    }
#endif
    sc.total_work = cols * sc.nchunks * desc->batch;
    if (sc.total_work >= (1ll << 31) - 8) return set_error(DFM_ERR_UNSUPPORTED, "grid too large");
    sc.per_xcd = (int)((sc.total_work + 7) / 8);
    // FAST: the un-augmented geometry (no flip, img_scale_factor 1, feat_sample_factor a power of two):
    // the footprint arithmetic has no branch and two chunks interleave
    const bool fast = !desc->flip && sc.f.scale_is_one && sc.f.fsf_pow2;
    const void *kern = fast ? (const void *)sweep_conv_kernel<true> : (const void *)sweep_conv_kernel<false>;
    rc = ensure_dynamic_lds(kern, SC_LDS_BYTES);
    if (rc != DFM_OK) return rc;
    const uint4 *wp = (const uint4 *)packed_weights;
    const uint4 *zero = wp + (size_t)4 * SC_FRAGS * 64;
    const bool timed = profile_mark(stream, false);
    const dim3 grid((unsigned)(8 * sc.per_xcd));
    if (fast)
        hipLaunchKernelGGL(sweep_conv_kernel<true>, grid, dim3(256), SC_LDS_BYTES, (hipStream_t)stream, sc,
                           (const uint4 *)cur_nhwc, (const uint4 *)prev_nhwc, depths, cam2img, cam2img_inv, cur2prev, wp,
                           zero, (bf16_t *)y_stereo, (bf16_t *)y_mono, stats_stereo, stats_mono);
    else
        hipLaunchKernelGGL(sweep_conv_kernel<false>, grid, dim3(256), SC_LDS_BYTES, (hipStream_t)stream, sc,
                           (const uint4 *)cur_nhwc, (const uint4 *)prev_nhwc, depths, cam2img, cam2img_inv, cur2prev, wp,
                           zero, (bf16_t *)y_stereo, (bf16_t *)y_mono, stats_stereo, stats_mono);
    if (timed) profile_mark(stream, true);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFM_ERR_HIP, hipGetErrorString(e));
    return DFM_OK;
}

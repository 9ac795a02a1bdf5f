// sweep_bench.cpp -- torch-free A/B harness for dfm_plane_sweep_fwd_opts (links libdfm_hip.so).
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/sweep_bench.cpp \
//         -L depth-from-motion_amd/lib -ldfm_hip -Wl,-rpath,'$ORIGIN/../depth-from-motion_amd/lib' \
//         -o tools/sweep_bench
//   tools/sweep_bench [--workload nstar|nstar_aug|kitti] [--mode fwd|nhwc|bwd] [--rounds R] [--launches L]
//                     [--batch B] cfg [cfg ...]
//   --mode nhwc: dfm_plane_sweep_fwd_nhwc (channels-last maps sampled in place; kitti); --mode bwd: the backward of the
//   workload -- dfm_plane_sweep_bwd (nstar) / dfm_plane_sweep_bwd_cur_nhwc + dfm_plane_sweep_bwd_prev_gather (kitti) --
//   with the volume as the gradient (round 6: the counter passes of bench.py's secondary rows)
//   cfg = comma list of kernel= lanes= lds= bpg= planes= chunk= ppl= pipe= align= pair=   ("default" = library defaults)
//
// Every round times each configuration once (L back-to-back launches between two HIP events on
// the launch stream), round-robin, so the variants see the same clock / thermal state
// (cdna_hip_programming.md 5.4 rule 24).  Reports per configuration the median and minimum time
// per launch and the algorithmic GB/s (SURVEY.md 8d bytes).  Also the harness the rocprofv3 --pmc
// passes run (no Python in the profiled process): --rounds 1 --launches 2 with ONE cfg.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "dfm_hip.h"

#define CK(e)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (e);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

// 64-bit checksum of the volume (sum of 32-bit words times an odd per-position weight): every
// configuration must reproduce the first one's volume bit for bit
__global__ void checksum_kernel(const uint32_t *__restrict__ p, size_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)p[i] * (2 * (i & 0xffff) + 1);
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
static inline float uni() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
static inline float gauss()
{
    const float u = std::max(uni(), 1e-7f), v = uni();
    return sqrtf(-2.0f * logf(u)) * cosf(6.2831853f * v);
}
static inline uint16_t to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static void inv4(const double *m, double *o)
{
    // Gauss-Jordan on [m | I]
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = j < 4 ? m[i * 4 + j] : (j - 4 == i ? 1.0 : 0.0);
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[p][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[i * 4 + j] = a[i][j + 4];
}

static dfm_sweep_opts parse_cfg(const std::string &s)
{
    dfm_sweep_opts o;
    memset(&o, 0, sizeof(o));
    if (s == "default") return o;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t e = s.find(',', pos);
        if (e == std::string::npos) e = s.size();
        const std::string kv = s.substr(pos, e - pos);
        const size_t q = kv.find('=');
        if (q == std::string::npos) { fprintf(stderr, "bad cfg item '%s'\n", kv.c_str()); exit(2); }
        const std::string k = kv.substr(0, q);
        const int v = atoi(kv.c_str() + q + 1);
        if (k == "kernel") o.kernel = v;
        else if (k == "lanes") o.lanes_per_workgroup = v;
        else if (k == "lds") o.lds_kib = v;
        else if (k == "bpg") o.blocks_per_group = v;
        else if (k == "planes") o.planes_per_workgroup = v;
        else if (k == "chunk") o.bands_per_chunk = v;
        else if (k == "ppl") o.points_per_lane = v;
        else if (k == "pipe") o.pipeline = v;
        else if (k == "align") o.store_align_points = v;
        else if (k == "pair") o.pair_stores = v;
        else if (k == "unpack") o.unpack = v;
        else { fprintf(stderr, "unknown cfg key '%s'\n", k.c_str()); exit(2); }
        pos = e + 1;
    }
    return o;
}

int main(int argc, char **argv)
{
    std::string workload = "nstar", mode = "fwd";
    int rounds = 7, launches = 3, batch = 8;
    std::vector<std::string> cfgs;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--workload" && i + 1 < argc) workload = argv[++i];
        else if (a == "--mode" && i + 1 < argc) mode = argv[++i];
        else if (a == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);
        else if (a == "--launches" && i + 1 < argc) launches = atoi(argv[++i]);
        else if (a == "--batch" && i + 1 < argc) batch = atoi(argv[++i]);
        else cfgs.push_back(a);
    }
    if (cfgs.empty()) cfgs.push_back("default");

    dfm_sweep_desc d;
    memset(&d, 0, sizeof(d));
    d.batch = batch;
    float dmin = 2.0f, dmax = 59.6f;
    if (workload == "kitti") {
        d.channels = 32; d.h_in = 320; d.w_in = 1280; d.num_depths = 72;
        d.feat_sample_factor = 1; d.cost_sample_factor = 4; d.h_out = 80; d.w_out = 320;
        d.img_scale_factor = 1; d.crop_x = 0; d.crop_y = 55; d.org_w = 1242; d.flip = 0; d.dtype = DFM_F32;
    } else {
        d.channels = 256; d.h_in = 94; d.w_in = 311; d.num_depths = 112;
        d.feat_sample_factor = 4; d.cost_sample_factor = 1; d.h_out = 94; d.w_out = 311;
        d.img_scale_factor = 1; d.crop_x = 0; d.crop_y = 0; d.org_w = 1242; d.flip = 0; d.dtype = DFM_BF16;
        if (workload == "nstar_aug") { d.img_scale_factor = 1.03f; d.crop_x = 11; d.crop_y = 55; d.flip = 1; }
        else if (workload != "nstar") { fprintf(stderr, "unknown workload\n"); return 2; }
    }
    const size_t esz = d.dtype == DFM_BF16 ? 2 : 4;
    const size_t feat_elems = (size_t)d.batch * d.channels * d.h_in * d.w_in;
    const size_t out_elems = (size_t)d.batch * 2 * d.channels * d.num_depths * d.h_out * d.w_out;
    const double alg_bytes = (double)esz * (2.0 * feat_elems + out_elems);

    // inputs
    std::vector<uint8_t> hc(feat_elems * esz), hp(feat_elems * esz);
    for (size_t i = 0; i < feat_elems; ++i) {
        const float a = gauss(), b = gauss();
        if (esz == 2) { ((uint16_t *)hc.data())[i] = to_bf16(a); ((uint16_t *)hp.data())[i] = to_bf16(b); }
        else { ((float *)hc.data())[i] = a; ((float *)hp.data())[i] = b; }
    }
    const double P2[16] = {721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791,
                           0, 0, 1, 0.002745884, 0, 0, 0, 1};
    double Pi[16];
    inv4(P2, Pi);
    std::vector<float> hP(16 * batch), hPi(16 * batch), hT(16 * batch), hD(d.num_depths);
    for (int b = 0; b < batch; ++b) {
        const double yaw = (uni() * 4 - 2) * M_PI / 180, tx = uni() * 0.2 - 0.1, tz = -1.5 + uni() * 1.2;
        const double T[16] = {cos(yaw), 0, sin(yaw), tx, 0, 1, 0, 0, -sin(yaw), 0, cos(yaw), tz, 0, 0, 0, 1};
        for (int k = 0; k < 16; ++k) { hP[b * 16 + k] = (float)P2[k]; hPi[b * 16 + k] = (float)Pi[k]; hT[b * 16 + k] = (float)T[k]; }
    }
    for (int i = 0; i < d.num_depths; ++i) hD[i] = dmin + (i + 0.5f) * ((dmax - dmin) / d.num_depths);

    void *cur, *prev, *out, *ws;
    float *P, *Pinv, *T, *depths;
    const size_t wsb = dfm_plane_sweep_workspace_bytes(&d);
    CK(hipMalloc(&cur, feat_elems * esz));
    CK(hipMalloc(&prev, feat_elems * esz));
    CK(hipMalloc(&out, out_elems * esz));
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc((void **)&P, 64 * batch));
    CK(hipMalloc((void **)&Pinv, 64 * batch));
    CK(hipMalloc((void **)&T, 64 * batch));
    CK(hipMalloc((void **)&depths, 4 * d.num_depths));
    CK(hipMemcpy(cur, hc.data(), feat_elems * esz, hipMemcpyHostToDevice));
    CK(hipMemcpy(prev, hp.data(), feat_elems * esz, hipMemcpyHostToDevice));
    CK(hipMemcpy(P, hP.data(), 64 * batch, hipMemcpyHostToDevice));
    CK(hipMemcpy(Pinv, hPi.data(), 64 * batch, hipMemcpyHostToDevice));
    CK(hipMemcpy(T, hT.data(), 64 * batch, hipMemcpyHostToDevice));
    CK(hipMemcpy(depths, hD.data(), 4 * d.num_depths, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    if (mode != "fwd") {
        // ---- the other rows: one configuration (the library's own dispatch), timed and left to the counters ----------
        float *g_cur = nullptr, *g_prev = nullptr;
        void *gws = nullptr;
        size_t gwsb = 0;
        double bytes = alg_bytes;
        if (mode == "bwd") {
            CK(hipMalloc((void **)&g_cur, feat_elems * 4));
            CK(hipMalloc((void **)&g_prev, feat_elems * 4));
            gwsb = dfm_plane_sweep_bwd_prev_gather_workspace_bytes(&d);
            CK(hipMalloc(&gws, gwsb ? gwsb : 256));
            // the gradient volume: the forward's result (finite, varied values)
            int rc = dfm_plane_sweep_fwd(&d, cur, prev, depths, P, Pinv, T, out, ws, wsb, st);
            if (rc != DFM_OK) { fprintf(stderr, "fwd: %s\n", dfm_last_error()); return 3; }
            bytes = (double)esz * out_elems + 2.0 * 4.0 * feat_elems;
        } else if (mode != "nhwc") { fprintf(stderr, "unknown mode\n"); return 2; }
        auto run_other = [&]() {
            int rc;
            if (mode == "nhwc") {
                rc = dfm_plane_sweep_fwd_nhwc(&d, cur, prev, depths, P, Pinv, T, out, ws, wsb, st);
            } else if (workload == "kitti") {
                CK(hipMemsetAsync(g_cur, 0, feat_elems * 4, st));
                rc = dfm_plane_sweep_bwd_cur_nhwc(&d, out, depths, P, Pinv, T, g_cur, st);
                if (rc == DFM_OK) rc = dfm_plane_sweep_bwd_prev_gather(&d, out, depths, P, Pinv, T, g_prev, gws, gwsb, st);
            } else {
                CK(hipMemsetAsync(g_cur, 0, feat_elems * 4, st));
                CK(hipMemsetAsync(g_prev, 0, feat_elems * 4, st));
                rc = dfm_plane_sweep_bwd(&d, out, depths, P, Pinv, T, g_cur, g_prev, st);
            }
            if (rc != DFM_OK) { fprintf(stderr, "%s %s: %s\n", workload.c_str(), mode.c_str(), dfm_last_error()); exit(3); }
        };
        run_other();
        CK(hipStreamSynchronize(st));
        std::vector<float> tms;
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, st));
            for (int l = 0; l < launches; ++l) run_other();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            tms.push_back(t / launches);
        }
        std::sort(tms.begin(), tms.end());
        printf("# workload %s mode %s  B=%d  algorithmic %.3f GB per step\n", workload.c_str(), mode.c_str(), batch, bytes / 1e9);
        printf("%-44s median %8.4f ms  min %8.4f ms   %7.1f GB/s (median)\n", "library dispatch", tms[tms.size() / 2], tms[0],
               bytes / (tms[tms.size() / 2] * 1e-3) / 1e9);
        return 0;
    }

    std::vector<dfm_sweep_opts> opts;
    for (auto &c : cfgs) opts.push_back(parse_cfg(c));
    std::vector<std::vector<float>> ms(cfgs.size());
    // reference checksum from the first configuration: every other one must reproduce the volume
    std::vector<unsigned long long> sums(cfgs.size(), 0);
    auto run = [&](size_t i) {
        int rc = dfm_plane_sweep_fwd_opts(&d, cur, prev, depths, P, Pinv, T, out, ws, wsb, st, &opts[i]);
        if (rc != DFM_OK) { fprintf(stderr, "cfg %s: %s\n", cfgs[i].c_str(), dfm_last_error()); exit(3); }
    };
    unsigned long long *dsum;
    CK(hipMalloc((void **)&dsum, 8));
    bool same = true;
    for (size_t i = 0; i < cfgs.size(); ++i) {  // warm up (and page in every code object)
        CK(hipMemsetAsync(out, 0xff, out_elems * esz, st));
        run(i);
        CK(hipMemsetAsync(dsum, 0, 8, st));
        hipLaunchKernelGGL(checksum_kernel, dim3(256 * 16), dim3(256), 0, st, (const uint32_t *)out,
                           out_elems * esz / 4, dsum);
        CK(hipMemcpyAsync(&sums[i], dsum, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (sums[i] != sums[0]) same = false;
    }
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < cfgs.size(); ++i) {
            CK(hipEventRecord(e0, st));
            for (int l = 0; l < launches; ++l) run(i);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms[i].push_back(t / launches);
        }
    printf("# workload %s  B=%d  algorithmic %.3f GB per launch  (%d rounds x %d launches, round-robin)\n",
           workload.c_str(), batch, alg_bytes / 1e9, rounds, launches);
    for (size_t i = 0; i < cfgs.size(); ++i) {
        std::sort(ms[i].begin(), ms[i].end());
        const float med = ms[i][ms[i].size() / 2], mn = ms[i][0];
        printf("%-44s median %8.4f ms  min %8.4f ms   %7.1f GB/s (median)  %7.1f vol/s  checksum %016llx%s\n",
               cfgs[i].c_str(), med, mn, alg_bytes / (med * 1e-3) / 1e9, batch / (med * 1e-3),
               (unsigned long long)sums[i], sums[i] == sums[0] ? "" : "  != first configuration");
    }
    if (!same) printf("# CHECKSUM MISMATCH (expected with a DFM_ABLATE debug build)\n");
    return 0;
}

// store_microbench3.hip -- which workgroup->address schedule writes the N* cost volume
// ([B][2C][D][H*W] bf16, 26.8 GB) fastest?  Stores only, one 16-B vector per lane per
// channel, same loop structure as sweep_tile_kernel (32 channel blocks x 8 channels).
// (profiles/archive/r01_store_microbench3.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct P { int B, C2, D, HWV, planes, order, nb, dg, nt, stagger, cbn, k; };

// HWV = vectors (8 points) per depth plane
__global__ void __launch_bounds__(256) pattern(uint4 *out, P p) {
    const int lpp = 256 / p.planes;
    long id = blockIdx.x;
    int b, half, dgrp, band;
    switch (p.order) {
    case 0:  // shipped: b fastest, depth group, map half, band
        b = id % p.B; id /= p.B; dgrp = id % p.dg; id /= p.dg; half = id % 2; band = id / 2; break;
    case 1:  // band fastest, half, b, depth group
        band = id % p.nb; id /= p.nb; half = id % 2; id /= 2; b = id % p.B; dgrp = id / p.B; break;
    case 2:  // b fastest, band, half, depth group
        b = id % p.B; id /= p.B; band = id % p.nb; id /= p.nb; half = id % 2; dgrp = id / 2; break;
    case 3:  // band fastest, depth group, half, b   (one sample at a time)
        band = id % p.nb; id /= p.nb; dgrp = id % p.dg; id /= p.dg; half = id % 2; b = id / 2; break;
    case 5: {  // b fastest, band within a chunk of k bands, depth group, half, band chunk
        b = id % p.B; id /= p.B;
        const int nchunk = (p.nb + p.k - 1) / p.k;
        const long per_chunk = (long)p.k * p.dg * 2;
        int chunk = id / per_chunk; long r = id % per_chunk;
        band = chunk * p.k + r % p.k; r /= p.k; dgrp = r % p.dg; half = r / p.dg;
        (void)nchunk; break; }
    case 6: {  // b fastest, half, band within chunk, depth group, band chunk
        b = id % p.B; id /= p.B; half = id % 2; id /= 2;
        const long per_chunk = (long)p.k * p.dg;
        int chunk = id / per_chunk; long r = id % per_chunk;
        band = chunk * p.k + r % p.k; dgrp = r / p.k; break; }
    default:  // 4: half fastest, b, band, depth group
        half = id % 2; id /= 2; b = id % p.B; id /= p.B; band = id % p.nb; dgrp = id / p.nb; break;
    }
    const int pl = threadIdx.x / lpp, ln = threadIdx.x % lpp;
    const int d = dgrp * p.planes + pl, v = band * lpp + ln;
    if (d >= p.D || v >= p.HWV || band >= p.nb) return;
    const size_t plane = (size_t)p.D * p.HWV;
    const int C = p.C2 / 2;
    uint4 *o = out + ((size_t)b * p.C2 + (size_t)half * C) * plane + (size_t)d * p.HWV + v;
    const int nblk = C / p.cbn;
    const int s0 = p.stagger ? (int)(blockIdx.x % nblk) : 0;
    for (int i = 0; i < nblk; ++i) {
        int blk = i + s0; if (blk >= nblk) blk -= nblk;
        uint4 *q = o + (size_t)blk * p.cbn * plane;
        for (int c = 0; c < p.cbn; ++c) {
            u32x4 val = {(unsigned)c, (unsigned)blk, 3u, 4u};
            if (p.nt) __builtin_nontemporal_store(val, (u32x4 *)(q + (size_t)c * plane));
            else *(u32x4 *)(q + (size_t)c * plane) = val;
        }
    }
}
int main() {
    P p{8, 512, 112, 3654, 2, 0, 0, 0, 1, 0, 8, 1};
    const size_t n16 = (size_t)p.B * p.C2 * p.D * p.HWV;
    uint4 *buf; CK(hipMalloc(&buf, n16 * 16));
    hipEvent_t a, z; CK(hipEventCreate(&a)); CK(hipEventCreate(&z));
    CK(hipFuncSetAttribute((const void *)pattern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    auto run = [&](int planes, int order, int k, int lds) {
        p.planes = planes; p.order = order; p.k = k;
        const int lpp = 256 / planes;
        p.nb = (p.HWV + lpp - 1) / lpp; p.dg = (p.D + planes - 1) / planes;
        const int nbk = (p.nb + k - 1) / k * k;
        const int nblocks = p.B * 2 * p.dg * (order >= 5 ? nbk : p.nb);
        pattern<<<nblocks, 256, lds>>>(buf, p); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) pattern<<<nblocks, 256, lds>>>(buf, p);
        CK(hipEventRecord(z)); CK(hipEventSynchronize(z));
        float ms; CK(hipEventElapsedTime(&ms, a, z)); ms /= 3;
        printf("planes %d order %d k %2d lds %2d KB  %7.3f ms  %7.1f GB/s\n", planes, order, k, lds / 1024, ms,
               n16 * 16 / ms / 1e6);
        fflush(stdout);
    };
    for (int lds : {0, 52 * 1024, 70 * 1024})
        for (int planes : {1, 2}) {
            for (int order : {0, 2, 4}) run(planes, order, 1, lds);
            for (int order : {5, 6})
                for (int k : {1, 2, 4, 8, 15, 29}) run(planes, order, k, lds);
        }
    return 0;
}

// Micro-benchmark: how fast can workgroups hand 16-byte-per-lane tiles to each other through memory when the
// total footprint is small enough for the 256 MiB Infinity Cache (or an XCD's 4 MiB L2), compared with a footprint
// that must stream through HBM?  Every workgroup rewrites its own region and reads a neighbour's region, `iters`
// times; the data values are irrelevant (no synchronisation: bandwidth only).
//   hipcc --offload-arch=gfx950 -O3 tools/mall_ring.hip -o tools/bin/mall_ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY, bool DO_W, bool DO_R>
__global__ __launch_bounds__(512) void rw_kernel(char *buf, size_t region, int iters, int nbr, unsigned *sink) {
    const int b = blockIdx.x, n = gridDim.x;
    char *mine = buf + (size_t)b * region;
    const char *theirs = buf + (size_t)((b + nbr) % n) * region;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4 v = {(unsigned)b, threadIdx.x, 3u, 4u};
    for (int it = 0; it < iters; ++it) {
        for (size_t o = (size_t)threadIdx.x * 16; o < region; o += 512 * 16) {
            if (DO_W) {
                u32x4 *p = reinterpret_cast<u32x4 *>(mine + o);
                if (POLICY == 0) *p = v;
                else if (POLICY == 1) __builtin_nontemporal_store(v, p);
                else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            }
            if (DO_R) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(theirs + o);
                u32x4 x;
                if (POLICY == 0) x = *p;
                else if (POLICY == 1) x = __builtin_nontemporal_load(p);
                else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(p) : "memory");
                if (POLICY == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc ^= x;
            }
        }
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

template <int POLICY, bool W, bool R>
static void run(const char *name, char *buf, size_t region, int nwg, int nbr, unsigned *sink) {
    const size_t per_iter = region * nwg * ((W ? 1 : 0) + (R ? 1 : 0));
    int iters = (int)(24e9 / (double)per_iter);
    if (iters < 2) iters = 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rw_kernel<POLICY, W, R><<<nwg, 512>>>(buf, region, 2, nbr, sink);
    hipEventRecord(e0);
    rw_kernel<POLICY, W, R><<<nwg, 512>>>(buf, region, iters, nbr, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s region %7zu KB  footprint %8.1f MB  nbr %3d  %7.2f ms  %7.2f TB/s\n", name, region >> 10,
           region * nwg / 1048576.0, nbr, ms, per_iter * (double)iters / (ms * 1e-3) / 1e12);
}

int main() {
    const int nwg = 256;
    const size_t maxreg = 16u << 20;
    char *buf; unsigned *sink;
    hipMalloc(&buf, maxreg * nwg); hipMalloc(&sink, 64);
    hipMemset(buf, 1, maxreg * nwg);
    const size_t regions[] = {64u << 10, 256u << 10, 512u << 10, 1u << 20, 4u << 20, 16u << 20};
    for (size_t r : regions) {
        for (int nbr : {8, 1}) {      // nbr 8: same XCD under round-robin placement; 1: the next XCD
            run<1, true, true>("nt store + nt load", buf, r, nwg, nbr, sink);
            run<0, true, true>("plain store + plain load", buf, r, nwg, nbr, sink);
            run<2, true, true>("sc1 store + sc1 load", buf, r, nwg, nbr, sink);
        }
        run<1, true, false>("nt store only", buf, r, nwg, 1, sink);
        run<1, false, true>("nt load only", buf, r, nwg, 1, sink);
    }
    return 0;
}

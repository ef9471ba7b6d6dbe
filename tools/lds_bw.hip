// Micro-benchmark (measurement only): LDS read bandwidth per CU for the fused kernels' A-fragment pattern
// (ds_read_b128, lane i reads 16 B at base + 16 i, every wave the same 18 KiB), NW waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NW, int SAME>
__global__ __launch_bounds__(NW * 64) void k(unsigned *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36 * 1024 / 4; i += NW * 64) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const char *base = smem + (SAME ? 0 : (wv & 1) * 18 * 1024) + lane * 16;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[18];
#pragma unroll
        for (int f = 0; f < 18; ++f) v[f] = *reinterpret_cast<const u32x4 *>(base + f * 1024);
#pragma unroll
        for (int f = 0; f < 18; ++f) acc ^= v[f];
        asm volatile("" ::: "memory");
    }
    if (acc[0] == 0x12345678u) out[threadIdx.x] = acc[1];
}

template <int NW, int SAME>
void run(const char *name) {
    unsigned *d;
    (void)hipMalloc(&d, 4096);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<NW, SAME>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NW, SAME><<<256, NW * 64, 40 * 1024>>>(d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NW, SAME><<<256, NW * 64, 40 * 1024>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * NW * iters * 18 * 1024;
    printf("%-28s %d waves/CU: %.1f TB/s chip, %.0f B/ns per CU (= B/clk at 1 GHz; /2.4 -> %.0f B/clk at 2.4 GHz)\n", name, NW, bytes / ms * 1e-9,
           bytes / 256 / (ms * 1e6), bytes / 256 / (ms * 1e6) / 2.4);
    (void)hipFree(d);
}

int main() {
    run<4, 1>("all waves same 18 KiB");
    run<8, 1>("all waves same 18 KiB");
    run<16, 1>("all waves same 18 KiB");
    run<8, 0>("two 18 KiB regions");
    return 0;
}

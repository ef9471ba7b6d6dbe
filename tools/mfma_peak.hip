// Micro-benchmark (measurement only, not part of the library): what one MI355X sustains on
// v_mfma_f32_32x32x16_bf16 for the fused kernels' operand pattern --
//   mode 0: registers only, one dependent accumulator chain per wave
//   mode 1: A fragment from LDS (ds_read_b128, 3 reads in flight), B in registers (= tile_matmul)
//   mode 2: as 1, plus a workgroup barrier every 16 MFMAs (= one ring step)
//   mode 4: registers only, two independent accumulator chains per wave
//   mode 5/6/7: mode 0 plus 2 / 4 / 8 independent VALU instructions (v_pk_mul_f32 on private registers) behind every MFMA:
//               does the vector ALU work of the two waves of a SIMD overlap the matrix pipe, or add to it?
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 18 * 1024 / 4; i += NW * 64) reinterpret_cast<float *>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.01f * ((lane + i + j) & 15));
    f32x16 acc = {};
    f32x16 keep = {};
    for (int j = 0; j < 16; ++j) keep[j] = 1.f + 0.001f * lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 1) & 15], b[ks], acc, 0, 0, 0);
        } else if (MODE >= 5) {
            constexpr int NV = MODE == 5 ? 2 : MODE == 6 ? 4 : 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 1) & 15], b[ks], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 t = {keep[2 * v], keep[2 * v + 1]};
                    t = t * (f32x2){1.0001f, 0.9999f};
                    keep[2 * v] = t[0]; keep[2 * v + 1] = t[1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 4) {        // two independent accumulator chains per wave
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 1) & 15], b[ks], acc, 0, 0, 0);
                keep = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 2) & 15], b[ks + 1], keep, 0, 0, 0);
            }
        } else {
            bf16x8 a[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + i * 1024 + lane * 16);
            __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + 3 < 16) a[(ks + 3) & 3] = *reinterpret_cast<const bf16x8 *>(smem + (ks + 3) * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 3], b[ks], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0x6);
            }
            if (MODE >= 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (MODE >= 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    b[it & 15][j] = (__bf16)fmaxf(acc[j], 0.f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) b[(it + 1) & 15][j] = (__bf16)fmaxf(acc[8 + j], 0.f);
            }
        }
    }
    for (int j = 0; j < 16; ++j) keep[j] += acc[j];
    float s = 0;
    for (int j = 0; j < 16; ++j) s += keep[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int NW>
void run(const char *name, int wg_per_cu) {
    float *d;
    hipMalloc(&d, 4096);
    const int iters = 20000, grid = 256 * wg_per_cu;
    hipFuncSetAttribute((const void *)k<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NW><<<grid, NW * 64, 32 * 1024>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NW><<<grid, NW * 64, 32 * 1024>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * NW * iters * 16 * 32768.0;
    const double cyc_per_mfma = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (NW * wg_per_cu / 4.0));
    printf("%-34s waves/SIMD %d  %.2f ms  %.0f TFLOP/s  (%.1f cyc@2.4GHz per MFMA per SIMD)\n", name, NW * wg_per_cu / 4, ms, flops / ms * 1e-9,
           cyc_per_mfma);
    hipFree(d);
}

int main() {
    run<0, 4>("regs only", 1);
    run<0, 8>("regs only", 1);
    run<4, 4>("regs only, 2 chains/wave", 1);
    run<4, 8>("regs only, 2 chains/wave", 1);
    run<1, 4>("A from LDS", 1);
    run<1, 8>("A from LDS", 1);
    run<2, 8>("A from LDS + barrier/16", 1);
    run<5, 8>("regs only + 2 VALU per MFMA", 1);
    run<6, 8>("regs only + 4 VALU per MFMA", 1);
    run<7, 8>("regs only + 8 VALU per MFMA", 1);
    run<7, 4>("regs only + 8 VALU per MFMA", 1);
    return 0;
}

// Energy-budget micro-benchmark (measurement only, round 4): isolated rows of what a training step is made of, each run
// back to back for S seconds on random data so that tools/smi_sample.py (started beside it) sees a steady board power:
//   mfma_regs     bf16 32x32x16 MFMA chains, operands in registers (2 waves per SIMD)
//   mfma_lds      the same with the A fragment of every MFMA read from LDS (1 KiB per MFMA and wave: the ring step's feed)
//   hbm_write     16-byte-per-lane non-temporal stores to fresh addresses (the tape writes), no MFMA
//   hbm_read_dma  1-KiB LDS-DMA pieces (buffer_load ... lds, nt) streamed from HBM (the dW kernel's tape reads), no MFMA
//   mfma_lds+write / mfma_regs+read : the two together (what co-running costs)
// Prints per row: unix start / end, launches, ms per launch, TFLOP/s, TB/s, in-kernel clock (s_memtime / s_memrealtime).
// build: hipcc -O3 --offload-arch=gfx950 tools/energy_bench.hip -o /tmp/energy_bench ; run: /tmp/energy_bench <seconds per row>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

static double now_unix() {
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

__device__ __forceinline__ unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

// MFMA: bit 0 = MFMAs on, bit 1 = A from LDS;  MEM: 0 none, 1 nt stores, 2 LDS-DMA reads
template <int MFMA, int MEM>
__global__ __launch_bounds__(512) void k(float *out, char *buf, long long bytes_per_wg, int iters, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned seed = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    for (int i = threadIdx.x; i < 18 * 1024 / 4; i += 512) {
        const float v = ((rnd(seed) >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f;
        reinterpret_cast<__bf16 *>(smem)[2 * i] = (__bf16)v;
        reinterpret_cast<__bf16 *>(smem)[2 * i + 1] = (__bf16)(0.7f * v + 0.1f);
    }
    __syncthreads();
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(((rnd(seed) >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f);
    f32x16 acc = {};
    char *mine = buf + (long long)blockIdx.x * bytes_per_wg;
    const long long pieces = bytes_per_wg / 1024;                 // 1 KiB per wave-instruction
    long long piece = wv;
    char *dma_dst = smem + 32 * 1024 + wv * 4096;                  // 4 x 1 KiB landing slots per wave
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MFMA & 1) {
            if (MFMA & 2) {
                bf16x8 a[4];
#pragma unroll
                for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + i * 1024 + lane * 16);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    if (ks + 3 < 16) a[(ks + 3) & 3] = *reinterpret_cast<const bf16x8 *>(smem + (ks + 3) * 1024 + lane * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 3], b[ks], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 1) & 15], b[ks], acc, 0, 0, 0);
            }
            // keep the accumulator bounded and data-dependent
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = acc[j] * 0.03125f + 0.25f;
        }
        if (MEM == 1) {             // two 16-byte stores per lane and "step": 2 KiB per wave, as the training forward's h tile
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                __builtin_nontemporal_store(b[(it + s) & 15], reinterpret_cast<bf16x8 *>(mine + piece * 1024 + lane * 16));
                piece += 8; if (piece >= pieces) piece = wv;
            }
        }
        if (MEM == 2) {             // four 1-KiB DMA pieces per wave and step, at most 8 in flight
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned long long u = reinterpret_cast<unsigned long long>(mine + piece * 1024);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
                const u32x4 rs = {lo, hi & 0xffffu, 1u << 20, 0x00020000u};
                const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(dma_dst + s * 1024));
                const unsigned voff = lane * 16;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(m), "v"(voff), "s"(rs) : "memory", "m0");
                piece += 8; if (piece >= pieces) piece = wv;
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[j];
    if (s == 12345.678f) out[threadIdx.x] = s + smem[32 * 1024 + 5];
}

template <int MFMA, int MEM>
void run(const char *name, double sec, int iters) {
    const int grid = 256;
    const long long per_wg = 1ll << 26;          // 64 MiB per workgroup, 16 GiB in all: far beyond the Infinity Cache
    float *d; char *buf = nullptr; unsigned long long *clk;
    hipMalloc(&d, 4096);
    hipMalloc(&clk, grid * 16);
    if (MEM) { hipMalloc(&buf, (size_t)grid * per_wg); hipMemset(buf, 0x11, (size_t)grid * per_wg); }
    hipFuncSetAttribute((const void *)k<MFMA, MEM>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k<MFMA, MEM><<<grid, 512, 64 * 1024>>>(d, buf, per_wg, 64, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> ghz;
    double ms_sum = 0; int n = 0;
    const double t_beg = now_unix();
    while (now_unix() - t_beg < sec) {
        hipEventRecord(e0);
        k<MFMA, MEM><<<grid, 512, 64 * 1024>>>(d, buf, per_wg, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms_sum += ms; ++n;
        std::vector<unsigned long long> h(2 * grid);
        hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
        std::vector<double> g;
        for (int i = 0; i < grid; ++i) g.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
        std::sort(g.begin(), g.end());
        ghz.push_back(g[grid / 2]);
    }
    const double t_end = now_unix();
    std::sort(ghz.begin(), ghz.end());
    const double flops = (MFMA & 1) ? (double)grid * 8 * (double)iters * 16 * 32768.0 * n : 0.0;
    const double bytes = MEM == 1 ? (double)grid * 8 * (double)iters * 2048.0 * n : MEM == 2 ? (double)grid * 8 * (double)iters * 4096.0 * n : 0.0;
    printf("%-16s unix %.1f .. %.1f  %4d launches %8.2f ms each  %6.0f TFLOP/s  %5.2f TB/s  in-kernel clock %.3f GHz\n", name, t_beg, t_end, n, ms_sum / n,
           flops / ms_sum * 1e-9, bytes / ms_sum * 1e-9, ghz[ghz.size() / 2]);
    fflush(stdout);
    hipFree(d); hipFree(clk); if (buf) hipFree(buf);
}

int main(int argc, char **argv) {
    const double sec = argc > 1 ? atof(argv[1]) : 4.0;
    run<1, 0>("mfma_regs", sec, 40000);
    run<3, 0>("mfma_lds", sec, 30000);
    run<0, 1>("hbm_write", sec, 20000);
    run<0, 2>("hbm_read_dma", sec, 12000);
    run<3, 1>("mfma_lds+write", sec, 20000);
    run<1, 2>("mfma_regs+read", sec, 12000);
    return 0;
}

// Device self-checks of the lane maps the fused kernels rely on (exact small-integer data):
//   [0] v_mfma_f32_32x32x16_bf16 A/B/C maps   [1] v_mfma_f32_32x32x2_f32 maps
//   [2] accumulator -> next B operand chaining (phi16 order), bf16     [3] same, f32
//   [4] ds_read_b64_tr_b16 as a [k][n] -> B-operand transposed read
//   [5]/[6] buffer_load_dwordx4 ... lds (LDS-DMA, dma_1k) lane placement, destinations below / above 64 KiB
//   [7] bf16 tape tile image (point on the lane) -> K = point fragments by ds_read_b64_tr_b16 (dW kernel)
#include "fused_common.h"

DEVI int ia(int i, int k) { return ((i * 3 + k * 5) % 7) - 3; }   // asymmetric integer operands
DEVI int ib(int k, int n) { return ((k * 2 + n * 7) % 9) - 4; }

__global__ void selftest_kernel(int *res, short *dump) {
    __shared__ __attribute__((aligned(16))) short img[64 * 32];   // [k][n], 64-byte rows
    const int lane = threadIdx.x & 63, r31 = lane & 31, h = lane >> 5;
    int bad;
    // ---- [0] bf16 32x32x16: A[row r31][k=8h+j], B[k=8h+j][col r31], C col=r31,row=(r&3)+8(r>>2)+4h
    {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)ia(r31, 8 * h + j); b[j] = (__bf16)(float)ib(8 * h + j, r31); }
        f32x16 c = {};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        bad = 0;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            int ref = 0;
            for (int k = 0; k < 16; ++k) ref += ia(row, k) * ib(k, r31);
            bad += (c[r] != (float)ref);
        }
        atomicAdd(res + 0, bad);
    }
    // ---- [1] f32 32x32x2: A[i=r31][k=h], B[k=h][j=r31]
    {
        f32x16 c = {};
        c = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ia(r31, h), (float)ib(h, r31), c, 0, 0, 0);
        bad = 0;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ref = ia(row, 0) * ib(0, r31) + ia(row, 1) * ib(1, r31);
            bad += (c[r] != (float)ref);
        }
        atomicAdd(res + 1, bad);
    }
    // ---- [2]/[3] chain: X = A1.B1 (32x32, K=16);  Y = A2.X with A2[i][feature], K=32 in phi16 order
    {
        // X[f][n] = sum_k ia(f,k) ib(k,n), |X| small integers (exact in bf16: |X| <= 16*3*4=192 < 256)
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)ia(r31, 8 * h + j); b[j] = (__bf16)(float)ib(8 * h + j, r31); }
        f32x16 x = {};
        x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
        f32x16 y = {}, y32 = {};
        for (int s = 0; s < 2; ++s) {
            bf16x8 xb, a2;
            for (int j = 0; j < 8; ++j) {
                xb[j] = (__bf16)x[8 * s + j];
                a2[j] = (__bf16)(float)(((r31 + 2 * (16 * s + phi16(h, j))) % 5) - 2);   // A2[i][f], f = 16s+phi
            }
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
            for (int j = 0; j < 8; ++j)
                y32 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)(((r31 + 2 * (16 * s + phi16(h, j))) % 5) - 2), x[8 * s + j], y32, 0, 0, 0);
        }
        bad = 0;
        int bad32 = 0;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            int ref = 0;
            for (int f = 0; f < 32; ++f) {
                int xf = 0;
                for (int k = 0; k < 16; ++k) xf += ia(f, k) * ib(k, r31);
                ref += (((row + 2 * f) % 5) - 2) * xf;
            }
            bad += (y[r] != (float)ref);
            bad32 += (y32[r] != (float)ref);
        }
        atomicAdd(res + 2, bad);
        atomicAdd(res + 3, bad32);
    }
    // ---- [4] transposed LDS read: image img[k][n] (k = 0..15 rows of 32 shorts), B operand wants
    //          lane (n = r31, h): elements j -> img[8h+j][n]
    {
        for (int t = threadIdx.x; t < 64 * 32; t += 64) img[t] = (short)(t + 1);
        __syncthreads();
        const int gi = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
        const int n0 = 16 * (gi & 1);
        bad = 0;
        for (int rd = 0; rd < 2; ++rd) {
            const int k0 = 8 * (gi >> 1) + 4 * rd;
            const short *addr = img + (k0 + q) * 32 + n0 + 4 * pp;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)addr);
            for (int e = 0; e < 4; ++e) {
                const short want = img[(8 * h + 4 * rd + e) * 32 + r31];
                bad += (v[e] != want);
                dump[(lane * 2 + rd) * 4 + e] = v[e];
            }
        }
        atomicAdd(res + 4, bad);
    }
    // ---- [7] bf16 tape tile: point-on-lane image (fused_bwd.hip TapeEmit::store_native offsets) read back with the
    //          transposed reads of the dW kernel (same address math as tr_frag): lane (n, kh), k-step s, element j
    //          must be X[feature n][point 16 s + phi16(kh, j)]
    {
        __syncthreads();
        char *tile = reinterpret_cast<char *>(img);                        // 2 KiB of the 4 KiB image
        auto X = [](int f, int p) { return (short)(1 + f * 37 + p); };
        const int pt = lane & 31;
        for (int s = 0; s < 2; ++s) {
            short v[8];
            for (int j = 0; j < 8; ++j) v[j] = X(16 * s + phi16(h, j), pt);
            const int off = 16 * (pt & 3) + 128 * s + 64 * h + 256 * (pt >> 2);
            for (int j = 0; j < 8; ++j) reinterpret_cast<short *>(tile + off)[j] = v[j];
        }
        __syncthreads();
        const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
        const int trl = 256 * (g >> 1) + 16 * q + 128 * (g & 1) + 64 * (pp & 1) + 8 * (pp >> 1);
        bad = 0;
        for (int s = 0; s < 2; ++s)
            for (int rd = 0; rd < 2; ++rd) {
                const char *addr = tile + 1024 * s + 512 * rd + trl;
                s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)addr);
                for (int e = 0; e < 4; ++e) bad += (v[e] != X(r31, 16 * s + phi16(h, 4 * rd + e)));
            }
        atomicAdd(res + 7, bad);
    }
}

// ---- [5],[6] LDS-DMA (dma_1k: buffer_load_dwordx4 ... lds): lane i's 16 bytes land at base + 16*i; destinations
//      below and above 64 KiB; counted vmcnt + barrier publish them to the other waves
__global__ void selftest_dma_kernel(const unsigned *src, int *res) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    // wave w copies KiB w of src to LDS offset 1024*w (low) and 96 KiB + 1024*w (high)
    for (int rep = 0; rep < 2; ++rep) {
        char *dst = dsm + (rep ? 96 * 1024 : 0) + wv * 1024;
        dma_1k(reinterpret_cast<const char *>(src) + wv * 1024, dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int bad_lo = 0, bad_hi = 0;
    for (int i = threadIdx.x; i < nw * 256; i += blockDim.x) {
        bad_lo += reinterpret_cast<const unsigned *>(dsm)[i] != src[i];
        bad_hi += reinterpret_cast<const unsigned *>(dsm + 96 * 1024)[i] != src[i];
    }
    atomicAdd(res + 5, bad_lo);
    atomicAdd(res + 6, bad_hi);
}

extern "C" int bhn_selftest(int32_t *results_host, void *scratch_dev, size_t scratch_bytes) {
    BHN_CHECK_ARG(results_host, "null results");
    BHN_CHECK_ARG(scratch_dev && scratch_bytes >= BHN_SELFTEST_SCRATCH_BYTES, "selftest needs %d bytes of device scratch", BHN_SELFTEST_SCRATCH_BYTES);
    // caller-owned scratch: [8 int results][512 short dump][8 KiB DMA source]
    int *d_res = reinterpret_cast<int *>(scratch_dev);
    short *d_dump = reinterpret_cast<short *>(reinterpret_cast<char *>(scratch_dev) + 64);
    unsigned *d_src = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(scratch_dev) + 2048);
    BHN_HIP(hipMemset(d_res, 0, 8 * sizeof(int)));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d_res, d_dump);
    BHN_HIP(hipGetLastError());
    {
        unsigned hsrc[2048];
        for (int i = 0; i < 2048; ++i) hsrc[i] = 0x9E3779B9u * (unsigned)(i + 1);
        BHN_HIP(hipMemcpy(d_src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice));
        BHN_HIP(hipFuncSetAttribute((const void *)selftest_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(selftest_dma_kernel, dim3(1), dim3(512), 104 * 1024, 0, d_src, d_res);
        BHN_HIP(hipGetLastError());
        BHN_HIP(hipDeviceSynchronize());
    }
    BHN_HIP(hipMemcpy(results_host, d_res, 8 * sizeof(int), hipMemcpyDeviceToHost));
    short dump[512];
    BHN_HIP(hipMemcpy(dump, d_dump, sizeof(dump), hipMemcpyDeviceToHost));
    if (results_host[4] != 0) {   // help diagnose the transposed-read map from one run
        char buf[400];
        int n = 0;
        for (int l = 0; l < 20 && n < 360; ++l) n += snprintf(buf + n, sizeof(buf) - n, "%d:%d,%d,%d,%d ", l, dump[l * 8], dump[l * 8 + 1], dump[l * 8 + 2], dump[l * 8 + 3]);
        bhn_set_error("tr16 read map mismatch; lane:first-read values = %s", buf);
    }
    return BHN_OK;
}

// ---------------------------------------------------------------------------------------------
// bhn_mfma_probe: the matrix-pipe ceiling of THIS board (bench.py: mfma_peak_this_box).  8 waves per workgroup (two per SIMD),
// each a chain of dependent v_mfma_f32_32x32x16_bf16 with every operand in registers, random operands (the data pattern sets the
// power a matrix instruction draws and with it the clock the board sustains: zeros run a third faster than real data).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mfma_probe_kernel(float *sink, int iters, long long *clk) {
    unsigned seed = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed; };
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(((rnd() >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f);
    f32x16 acc = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 1) & 15], b[ks], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = acc[j] * 0.03125f + 0.25f;      // bounded and data-dependent
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (clk && threadIdx.x == 0) { clk[2 * blockIdx.x] = (long long)(t1 - t0); clk[2 * blockIdx.x + 1] = (long long)(r1 - r0); }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[j];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

extern "C" int bhn_mfma_probe(int32_t grid, int32_t iters, int64_t *clk_dev, float *sink_dev, void *stream) {
    BHN_CHECK_ARG(grid > 0 && grid <= 65536 && iters > 0 && sink_dev, "bhn_mfma_probe: grid 1..65536, iters > 0, a sink buffer");
    hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, sink_dev, iters, reinterpret_cast<long long *>(clk_dev));
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

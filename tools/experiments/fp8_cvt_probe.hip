// Probe of the gfx950 scaled 8-bit conversions used by the 8-bit tape (hipcc --offload-arch=gfx950 fp8_cvt_probe.hip -o probe):
// what v_cvt_scalef32_pk_fp8_bf16 does with its scale operand, with values beyond +-448, with tiny values, and that
// v_cvt_scalef32_pk_bf16_fp8 with the same scale gives the value back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, float scale, unsigned *bytes, float *back, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const bf16x2 b = {(__bf16)in[i], (__bf16)(-in[i])};
    s16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, b, scale, false);
    bytes[i] = (unsigned)(unsigned short)r[0];
    const bf16x2 g = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((unsigned)(unsigned short)r[0], scale, false);
    back[2 * i] = (float)g[0]; back[2 * i + 1] = (float)g[1];
}
int main() {
    const float vals[] = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 3.f, 100.f, 448.f, 449.f, 480.f, 1000.f, 1e6f, 0.015625f, 0.013f, 0.002f, 0.001f, 0.0009f, 1e-5f};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *din, *dback; unsigned *dby;
    hipMalloc(&din, sizeof(vals)); hipMalloc(&dback, 2 * sizeof(vals)); hipMalloc(&dby, n * 4);
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (float scale : {1.f, 4.f, 0.25f, 3.f}) {
        k<<<1, 64>>>(din, scale, dby, dback, n);
        unsigned by[64]; float back[128];
        hipMemcpy(by, dby, n * 4, hipMemcpyDeviceToHost); hipMemcpy(back, dback, 2 * n * 4, hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int i = 0; i < n; ++i) printf("  x %-12g bytes %04x  back %-12g %-12g\n", vals[i], by[i], back[2 * i], back[2 * i + 1]);
    }
    return 0;
}

// Stand-alone pieces of the hot path: geometry fold, radiative-transfer ray sum (kgeo.py:595-622),
// chi^2 (network.py:476-484), Adam (network.py:173-174), plus the ABI bookkeeping.
#include <math.h>
#include <stdarg.h>

#include "common.h"

// ------------------------------------------------------------------------------------------
// error string / version / device cache
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void bhn_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bhn_version(void) { return BHN_ABI_VERSION; }
extern "C" const char *bhn_last_error(void) { return g_err; }

int bhn_num_cus(int device) {
    static DeviceOnce once;
    if (device < 0 || device >= BHN_MAX_DEVICES) return 0;        // callers: BHN_CHECK_DEVICE
    (void)once.run(device, [&](int &n) {
        n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
        return hipSuccess;
    });
    return once.value[device];
}

// ------------------------------------------------------------------------------------------
// MLP shape / flat parameter layout (flax tree order, network.py:56-62)
// ------------------------------------------------------------------------------------------
int bhn_mlp_shape(const bhn_model *m, MlpShape *s) {
    BHN_CHECK_ARG(m && s, "null model");
    BHN_CHECK_ARG(m->net_depth >= 2 && m->net_depth <= 8, "net_depth %d outside 2..8", m->net_depth);
    BHN_CHECK_ARG(m->net_width >= 1 && m->net_width <= BHN_GEN_WIDTH_MAX, "net_width %d outside 1..%d", m->net_width, BHN_GEN_WIDTH_MAX);
    BHN_CHECK_ARG(m->posenc_deg >= 0 && m->posenc_deg <= BHN_GEN_DEG_MAX, "posenc_deg %d outside 0..%d", m->posenc_deg, BHN_GEN_DEG_MAX);
    memset(s, 0, sizeof(*s));
    s->depth = m->net_depth;
    s->general = (m->posenc_deg > BHN_DEG_MAX || m->net_width > 256) ? 1 : 0;
    // the fused kernels exist for widths 32, 64, 128, 256: any other width runs on the next one with zero weights, biases
    // and gradients for the padding units (relu(0) = 0: they change nothing), the flat parameter layout keeps the true width
    s->width_true = m->net_width;
    s->width = m->net_width <= 32 ? 32 : m->net_width <= 64 ? 64 : m->net_width <= 128 ? 128 : 256;
    if (s->general) s->width = (m->net_width + 31) / 32 * 32;
    s->F = 3 + 6 * m->posenc_deg;
    s->deg = m->posenc_deg;
    const int skip_layer = m->net_depth / 2;
    int cur = s->F;
    int64_t off = 0;
    for (int i = 0; i <= s->depth; ++i) {
        s->in_dim[i] = cur;
        s->skip_in[i] = (cur == s->width_true + s->F) ? 1 : 0;
        const int out = (i == s->depth) ? 1 : s->width_true;
        s->kernel_off[i] = off;
        off += (int64_t)cur * out;
        s->bias_off[i] = off;
        off += out;
        cur = s->width_true;
        if (m->do_skip && i < s->depth && i % skip_layer == 0 && i > 0) cur = s->width_true + s->F;
    }
    s->nparams = off;      // (odd depths with do_skip feed the skip-concat into the OUTPUT layer, network.py:59-62: skip_in[depth])
    return BHN_OK;
}

extern "C" int64_t bhn_param_count(const bhn_model *m) {
    MlpShape s;
    if (bhn_mlp_shape(m, &s) != BHN_OK) return -1;
    return s.nparams;
}

extern "C" int bhn_param_layout(const bhn_model *m, int64_t *kernel_off, int64_t *bias_off, int32_t *in_dim) {
    MlpShape s;
    int rc = bhn_mlp_shape(m, &s);
    if (rc != BHN_OK) return rc;
    for (int i = 0; i <= s.depth; ++i) {
        if (kernel_off) kernel_off[i] = s.kernel_off[i];
        if (bias_off) bias_off[i] = s.bias_off[i];
        if (in_dim) in_dim[i] = s.in_dim[i];
    }
    return BHN_OK;
}

// ------------------------------------------------------------------------------------------
// geometry fold: w[s][p] = g^2 dtau Sigma J_s, dom[p] = inside the supervised region
// ------------------------------------------------------------------------------------------
__global__ void geom_prepare_kernel(const float *__restrict__ coords, const float *__restrict__ g,
                                    const float *__restrict__ dtau, const float *__restrict__ Sigma,
                                    const float *__restrict__ J, int S, int64_t P, float rmin2, float rmax2,
                                    float z_width, float *__restrict__ w, uint8_t *__restrict__ dom) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const float x = coords[p], y = coords[P + p], z = coords[2 * P + p];
        // emission.py:370-373: r^2 < rmin^2, r^2 > rmax^2, |z| > z_width are zeroed
        const float r2 = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
        dom[p] = (r2 < rmin2 || r2 > rmax2 || fabsf(z) > z_width) ? 0 : 1;
        const float gg = g[p];
        const float base = gg * gg * dtau[p] * Sigma[p];   // kgeo.py:621 factor order g^2 * e * dtau * Sigma
        if (S == 0) {
            w[p] = base;
        } else {
            for (int s = 0; s < S; ++s) w[(int64_t)s * P + p] = base * J[(int64_t)s * P + p];
        }
    }
}

extern "C" int bhn_geom_prepare(const float *coords, const float *g, const float *dtau, const float *Sigma,
                                const float *J, int32_t S, int64_t P, float rmin, float rmax, float z_width,
                                float *w_out, uint8_t *dom_out, void *stream) {
    BHN_CHECK_ARG(coords && g && dtau && Sigma && w_out && dom_out, "null pointer");
    BHN_CHECK_ARG(P > 0 && S >= 0 && S <= 4, "bad P=%lld or S=%d", (long long)P, S);
    BHN_CHECK_ARG(S == 0 || J, "S>0 needs J");
    const int threads = 256;
    const int blocks = (int)((P + threads - 1) / threads < 4096 ? (P + threads - 1) / threads : 4096);
    hipLaunchKernelGGL(geom_prepare_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, coords, g, dtau,
                       Sigma, J, S, P, rmin * rmin, rmax * rmax, z_width, w_out, dom_out);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// ------------------------------------------------------------------------------------------
// radiative transfer stand-alone (HBM-bound): a group of LPR lanes owns one ray, keeps the
// ray's weights g^2 dtau Sigma in registers and streams the N emission planes past them.
// Algorithmic bytes per launch: 4*N*R*G (e) + 12*R*G (g,dtau,Sigma) + 4*N*R (img).
// ------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));

template <int LPR, int KCH, bool VEC, bool BWD>
__global__ __launch_bounds__(256) void rt_kernel(const float *__restrict__ ein, const float *__restrict__ g,
                                                 const float *__restrict__ dtau, const float *__restrict__ Sigma,
                                                 float *__restrict__ out, int64_t N, int64_t R, int64_t G) {
    constexpr int RPB = 256 / LPR;   // rays per block
    const int sub = threadIdx.x % LPR;
    const int64_t ray = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool ray_ok = ray < R;
    const int64_t rbase = ray * G;
    float w[KCH][4];
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int64_t k = (int64_t)(c * LPR + sub) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[c][i] = 0.f;
        if (!ray_ok) continue;
        if (VEC) {
            if (k < G) {
                const float4 a = *reinterpret_cast<const float4 *>(g + rbase + k);
                const float4 b = *reinterpret_cast<const float4 *>(dtau + rbase + k);
                const float4 s = *reinterpret_cast<const float4 *>(Sigma + rbase + k);
                w[c][0] = a.x * a.x * b.x * s.x; w[c][1] = a.y * a.y * b.y * s.y;
                w[c][2] = a.z * a.z * b.z * s.z; w[c][3] = a.w * a.w * b.w * s.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (k + i < G) {
                    const float a = g[rbase + k + i];
                    w[c][i] = a * a * dtau[rbase + k + i] * Sigma[rbase + k + i];
                }
        }
    }
    const int64_t plane = R * G;
    for (int64_t n = 0; n < N; ++n) {
        if (!BWD) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int64_t k = (int64_t)(c * LPR + sub) * 4;
                if (!ray_ok) continue;
                if (VEC) {
                    if (k < G) {
                        const f4v e = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(ein + n * plane + rbase + k));
                        acc += w[c][0] * e.x + w[c][1] * e.y + w[c][2] * e.z + w[c][3] * e.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k + i < G) acc += w[c][i] * ein[n * plane + rbase + k + i];
                }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (sub == 0 && ray_ok) out[n * R + ray] = acc;
        } else {
            if (!ray_ok) continue;
            const float d = ein[n * R + ray];
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int64_t k = (int64_t)(c * LPR + sub) * 4;
                if (VEC) {
                    if (k < G) {
                        f4v v = {d * w[c][0], d * w[c][1], d * w[c][2], d * w[c][3]};
                        __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(out + n * plane + rbase + k));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k + i < G) out[n * plane + rbase + k + i] = d * w[c][i];
                }
            }
        }
    }
}

template <bool BWD>
static int rt_launch(const float *in, const float *g, const float *dtau, const float *Sigma, float *out, int64_t N,
                     int64_t R, int64_t G, hipStream_t st) {
    BHN_CHECK_ARG(in && g && dtau && Sigma && out, "null pointer");
    BHN_CHECK_ARG(N > 0 && R > 0 && G > 0, "bad sizes N=%lld R=%lld G=%lld", (long long)N, (long long)R, (long long)G);
    BHN_CHECK_ARG(G <= 1024, "G=%lld > 1024 samples per ray is not supported", (long long)G);
    const bool vec = (G % 4 == 0) && (((uintptr_t)in | (uintptr_t)g | (uintptr_t)dtau | (uintptr_t)Sigma | (uintptr_t)out) % 16 == 0);
    const int64_t quads = (G + 3) / 4;
    int lpr = 1;
    while (lpr < 64 && lpr < quads) lpr <<= 1;
    const int kch = (int)((quads + lpr - 1) / lpr);   // <= 4 because G <= 1024
#define RT_GO(L, K)                                                                                               \
    do {                                                                                                          \
        const int64_t blocks = (R + (256 / L) - 1) / (256 / L);                                                   \
        if (vec)                                                                                                  \
            hipLaunchKernelGGL((rt_kernel<L, K, true, BWD>), dim3((unsigned)blocks), dim3(256), 0, st, in, g, dtau, Sigma, out, N, R, G); \
        else                                                                                                      \
            hipLaunchKernelGGL((rt_kernel<L, K, false, BWD>), dim3((unsigned)blocks), dim3(256), 0, st, in, g, dtau, Sigma, out, N, R, G); \
    } while (0)
    if (lpr < 64) {
        switch (lpr) {
            case 1: RT_GO(1, 1); break;
            case 2: RT_GO(2, 1); break;
            case 4: RT_GO(4, 1); break;
            case 8: RT_GO(8, 1); break;
            case 16: RT_GO(16, 1); break;
            default: RT_GO(32, 1); break;
        }
    } else {
        switch (kch) {
            case 1: RT_GO(64, 1); break;
            case 2: RT_GO(64, 2); break;
            case 3: RT_GO(64, 3); break;
            default: RT_GO(64, 4); break;
        }
    }
#undef RT_GO
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

extern "C" int bhn_radiative_transfer_fwd(const float *e, const float *g, const float *dtau, const float *Sigma,
                                          float *img, int64_t N, int64_t R, int64_t G, void *stream) {
    return rt_launch<false>(e, g, dtau, Sigma, img, N, R, G, (hipStream_t)stream);
}

extern "C" int bhn_radiative_transfer_bwd(const float *dimg, const float *g, const float *dtau, const float *Sigma,
                                          float *de, int64_t N, int64_t R, int64_t G, void *stream) {
    return rt_launch<true>(dimg, g, dtau, Sigma, de, N, R, G, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// chi^2 on images (network.py:476-484).  One block per (frame, stokes) plane.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// 1024 threads, four pixels per thread and trip when the plane allows it (16-byte loads): one block per plane is all the
// parallelism a per-plane fixed-order sum has (8 blocks at config 2), so the kernel is bound by its dependent-load trips --
// 64 trips of 256 threads took 25 us, a fifth of the small-kernel time of a 4x128 step.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float block_sum_1024(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];       // every thread, the same order
    return t;
}

__global__ __launch_bounds__(1024) void chi2_image_kernel(const float *__restrict__ images, const float *__restrict__ target,
                                                          const float *__restrict__ sigma, const float *__restrict__ offset,
                                                          float scale, int dtype, int64_t R, float *__restrict__ loss,
                                                          float *__restrict__ dimages) {
    __shared__ float red[16];
    const int64_t plane = blockIdx.x;
    const float *img = images + plane * R;
    // 16-byte accesses when every plane of every array starts on a 16-byte boundary
    const bool vec = (R & 3) == 0 && ((reinterpret_cast<uintptr_t>(images) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(sigma) |
                                       reinterpret_cast<uintptr_t>(offset) | reinterpret_cast<uintptr_t>(dimages)) & 15) == 0;
    if (dtype == 0) {   // 'full': sum |(img - target - offset)/sigma|^2 (network.py:477)
        const float *tg = target + plane * R, *sg = sigma + plane * R, *of = offset + plane * R;
        float *dg = dimages ? dimages + plane * R : nullptr;
        float acc = 0.f;
        if (vec) {
            for (int64_t r = 4 * (int64_t)threadIdx.x; r < R; r += 4096) {
                const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sg + r), i4 = *reinterpret_cast<const f32x4 *>(img + r);
                const f32x4 t4 = *reinterpret_cast<const f32x4 *>(tg + r), o4 = *reinterpret_cast<const f32x4 *>(of + r);
                f32x4 g4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = (i4[e] - t4[e] - o4[e]) / s4[e];
                    acc += d * d;
                    g4[e] = 2.f * scale * d / s4[e];
                }
                if (dg) *reinterpret_cast<f32x4 *>(dg + r) = g4;
            }
        } else {
            for (int64_t r = threadIdx.x; r < R; r += 1024) {
                const float s = sg[r];
                const float d = (img[r] - tg[r] - of[r]) / s;
                acc += d * d;
                if (dg) dg[r] = 2.f * scale * d / s;
            }
        }
        const float tot = block_sum_1024(acc, red);
        if (threadIdx.x == 0) loss[1 + plane] = scale * tot;
    } else {            // 'lc': light curve = image summed over pixels (network.py:479-480)
        // The pixel sum and the residual are formed in DOUBLE (round 6).  A Stokes Q / U light curve is a sum of 65,536 pixels of
        // both signs that nearly cancel, and chi^2's gradient is proportional to (lc - target): in float the rounding of the sum
        // alone put 1e-5 .. 8e-4 of relative error on the gradient of the polarised configs (3, 5) -- the arithmetic of a float
        // light curve, which the float32 reference shares, but the f64 oracle this library is held to does not.  Fixed order
        // as before (bitwise reproducible); 64-bit adds run at the full vector rate on gfx950.
        __shared__ double red64[16];
        double acc = 0.0;
        if (vec) {
            for (int64_t r = 4 * (int64_t)threadIdx.x; r < R; r += 4096) {
                const f32x4 i4 = *reinterpret_cast<const f32x4 *>(img + r);
                acc += ((double)i4[0] + (double)i4[1]) + ((double)i4[2] + (double)i4[3]);
            }
        } else {
            for (int64_t r = threadIdx.x; r < R; r += 1024) acc += (double)img[r];
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((threadIdx.x & 63) == 0) red64[threadIdx.x >> 6] = acc;
        __syncthreads();
        double lc = 0.0;
        for (int i = 0; i < 16; ++i) lc += red64[i];       // every thread, the same order
        const double s = (double)sigma[plane];
        const double d = (lc - (double)target[plane] - (double)offset[plane]) / s;
        if (threadIdx.x == 0) loss[1 + plane] = (float)((double)scale * d * d);
        if (dimages) {
            const float gr = (float)(2.0 * (double)scale * d / s);
            for (int64_t r = threadIdx.x; r < R; r += 1024) dimages[plane * R + r] = gr;
        }
    }
}

// loss[0] = sum of the n partial terms loss[1..n] in a fixed order (one block): the logged loss is bitwise reproducible
__global__ __launch_bounds__(256) void loss_sum_kernel(float *__restrict__ loss, const float *__restrict__ part, int64_t n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += part[i];
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) loss[0] = tot;
}

extern "C" int bhn_chi2_image(const float *images, const float *target, const float *sigma, const float *offset,
                              float scale, int32_t dtype, int32_t B, int32_t Sx, int64_t R, float *loss,
                              float *dimages, void *stream) {
    BHN_CHECK_ARG(images && target && sigma && offset && loss, "null pointer");
    BHN_CHECK_ARG(dtype == 0 || dtype == 1, "image dtype (%d) not supported", dtype);
    BHN_CHECK_ARG(B > 0 && Sx > 0 && R > 0, "bad sizes");
    hipLaunchKernelGGL(chi2_image_kernel, dim3(B * Sx), dim3(1024), 0, (hipStream_t)stream, images, target, sigma,
                       offset, scale, dtype, R, loss, dimages);
    BHN_HIP(hipGetLastError());
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss, loss + 1, (int64_t)B * Sx);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// ------------------------------------------------------------------------------------------
// Adam with the reference's update form (optax.scale_by_adam + scale(-lr))
// ------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float c1,
                            float c2, float gs) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + eps);
    }
}

extern "C" int bhn_adam_step(float *params, const float *grads, float *m, float *v, int64_t n, int64_t t, float lr,
                             float b1, float b2, float eps, float grad_scale, void *stream) {
    BHN_CHECK_ARG(params && grads && m && v && n > 0 && t >= 1, "bad adam arguments");
    const float c1 = (float)(1.0 - pow((double)b1, (double)t));
    const float c2 = (float)(1.0 - pow((double)b2, (double)t));
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr, b1,
                       b2, eps, c1, c2, grad_scale);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// The same update with lr and the two bias corrections read from DEVICE memory (hyper = {lr, 1 - b1^t, 1 - b2^t}, written by
// the host through bhn_adam_hyper): the launch carries no per-step scalar, so a whole training step can be captured into a
// HIP graph and replayed (optimization.GraphedImageStep).  Same arithmetic as adam_kernel: bitwise-equal parameters.
__global__ void adam_dev_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                float *__restrict__ v, int64_t n, const float *__restrict__ hyper, float b1, float b2, float eps, float gs) {
    const float lr = hyper[0], c1 = hyper[1], c2 = hyper[2];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + eps);
    }
}

extern "C" int bhn_adam_hyper(int64_t t, float lr, float b1, float b2, float *hyper_host) {
    BHN_CHECK_ARG(hyper_host && t >= 1, "bad adam hyper arguments");
    hyper_host[0] = lr;
    hyper_host[1] = (float)(1.0 - pow((double)b1, (double)t));
    hyper_host[2] = (float)(1.0 - pow((double)b2, (double)t));
    return BHN_OK;
}

extern "C" int bhn_adam_step_dev(float *params, const float *grads, float *m, float *v, int64_t n, const float *hyper_dev,
                                 float b1, float b2, float eps, float grad_scale, void *stream) {
    BHN_CHECK_ARG(params && grads && m && v && hyper_dev && n > 0, "bad adam arguments");
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, hyper_dev, b1, b2, eps,
                       grad_scale);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// ------------------------------------------------------------------------------------------
// loss_fn_eht (network.py:541-564): visibilities = A . image, chi^2 on 'vis' | 'amp' | 'cphase'.
// A is complex64 (interleaved re,im), shape (N, C, nvis, R): C = 1 for vis/amp, 3 for closure
// phases (the product over the C axis is the bispectrum, network.py:558).  HBM-bound: A is read once
// in the forward GEMV and once in the backward; algorithmic bytes 2 * 8*N*C*nvis*R.
// ------------------------------------------------------------------------------------------
// Stage 1 of the visibility GEMV: block (row, split) sums its slice of the R axis; `part` is [row][RS].  With few rows
// (EHT2017: 8 frames x ~28 baselines) one block per row would leave most of the 256 CUs idle on 512 KB dot products, so
// the R axis is split until the grid has >= ~2048 blocks; stage 2 (eht_loss_kernel) adds the RS partial sums of a row in
// a fixed order -- no atomics, bitwise reproducible.  16-byte loads (two complex64 per lane).
__global__ __launch_bounds__(256) void eht_vis_kernel(const float *__restrict__ images, const float2 *__restrict__ A,
                                                      int C, int nvis, int64_t R, int RS, float2 *__restrict__ part, int wide) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;                         // (n, c, k)
    const int64_t n = row / ((int64_t)C * nvis);
    const int64_t span = ((R + RS - 1) / RS + 1) & ~(int64_t)1;          // even: slices start on 16-byte boundaries when R is even
    const int64_t r0 = blockIdx.y * span, r1 = r0 + span < R ? r0 + span : R;
    const float2 *a = A + row * R;
    const float *img = images + n * R;
    float re = 0.f, im = 0.f;
    if (wide) {                // R even AND A 16-byte / images 8-byte aligned (bhn_chi2_eht checks the caller's pointers)
        for (int64_t r = r0 + 2 * threadIdx.x; r < r1; r += 512) {      // r0, r1, R even: (r, r + 1) is a whole pair
            const float4 v = *reinterpret_cast<const float4 *>(a + r);       // (re0, im0, re1, im1)
            const float2 x = *reinterpret_cast<const float2 *>(img + r);
            re += v.x * x.x + v.z * x.y;
            im += v.y * x.x + v.w * x.y;
        }
    } else {
        for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
            const float2 v = a[r];
            const float x = img[r];
            re += v.x * x;
            im += v.y * x;
        }
    }
    const float sre = block_sum_256(re, red);
    const float sim = block_sum_256(im, red);
    if (threadIdx.x == 0) part[row * RS + blockIdx.y] = make_float2(sre, sim);
}

// per (n,k): chi^2 term and gv = dL/dRe(vis) + i dL/dIm(vis), written over vis
__global__ __launch_bounds__(256) void eht_loss_kernel(float2 *__restrict__ vis, const float2 *__restrict__ part, int RS,
                                                       const float *__restrict__ target, const float *__restrict__ sigma,
                                                       float scale, int dtype, int64_t N, int C, int nvis,
                                                       float *__restrict__ loss_part, int want_grad) {
    __shared__ float red[4];
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    float term = 0.f;
    if (t < N * nvis) {
        const int64_t n = t / nvis, k = t % nvis;
        const float s = sigma[t];
        for (int c = 0; c < C; ++c) {                 // stage 2 of the GEMV: the row's RS partial sums, fixed order
            const int64_t row = (n * C + c) * nvis + k;
            float sr = 0.f, si = 0.f;
            for (int j = 0; j < RS; ++j) { const float2 p = part[row * RS + j]; sr += p.x; si += p.y; }
            vis[row] = make_float2(sr, si);
        }
        if (dtype == 0) {            // 'vis': sum (|vis - target| / sigma)^2, target complex (network.py:548)
            const float2 v = vis[(n * C) * nvis + k];
            const float dr = v.x - target[2 * t], di = v.y - target[2 * t + 1];
            term = (dr * dr + di * di) / (s * s);
            if (want_grad) vis[(n * C) * nvis + k] = make_float2(2.f * scale * dr / (s * s), 2.f * scale * di / (s * s));
        } else if (dtype == 1) {     // 'amp': sum |(|vis| - target) / sigma|^2 (network.py:553)
            const float2 v = vis[(n * C) * nvis + k];
            const float amp = sqrtf(v.x * v.x + v.y * v.y);
            const float d = (amp - target[t]) / s;
            term = d * d;
            if (want_grad) {
                const float g = amp > 0.f ? 2.f * scale * d / (s * amp) : 0.f;
                vis[(n * C) * nvis + k] = make_float2(g * v.x, g * v.y);
            }
        } else {                     // 'cphase': sum (1 - cos(target - angle(prod_c vis_c))) / sigma^2 (network.py:558-559)
            float phi = 0.f;
            for (int c = 0; c < C; ++c) {
                const float2 v = vis[(n * C + c) * nvis + k];
                phi += atan2f(v.y, v.x);
            }
            const float d = target[t] - phi;
            term = (1.f - cosf(d)) / (s * s);
            if (want_grad) {
                const float dphi = -scale * sinf(d) / (s * s);          // dL/dphi
                for (int c = 0; c < C; ++c) {
                    const float2 v = vis[(n * C + c) * nvis + k];
                    const float m2 = v.x * v.x + v.y * v.y;
                    vis[(n * C + c) * nvis + k] = m2 > 0.f ? make_float2(-dphi * v.y / m2, dphi * v.x / m2) : make_float2(0.f, 0.f);
                }
            }
        }
    }
    const float tot = block_sum_256(term, red);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = scale * tot;
}

// dimg[n,r] = sum_{c,k} Re(gv) A_re + Im(gv) A_im
__global__ __launch_bounds__(256) void eht_bwd_kernel(const float2 *__restrict__ gv, const float2 *__restrict__ A, int C, int nvis,
                                                      int64_t R, float *__restrict__ dimages) {
    const int64_t n = blockIdx.y;
    const int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (r >= R) return;
    float acc = 0.f;
    const int rows = C * nvis;
    for (int j = 0; j < rows; ++j) {
        const float2 g = gv[n * rows + j];                  // wave-uniform: scalar loads
        const float2 a = A[(n * rows + j) * R + r];
        acc += g.x * a.x + g.y * a.y;
    }
    dimages[n * R + r] = acc;
}

// R-axis splits of the visibility GEMV: enough blocks to fill the chip, slices of >= 2048 elements
static int eht_splits(int64_t rows, int64_t R) {
    int64_t rs = (2048 + rows - 1) / rows;
    const int64_t most = R / 2048 > 1 ? R / 2048 : 1;
    if (rs > most) rs = most;
    if (rs > 64) rs = 64;
    return (int)(rs < 1 ? 1 : rs);
}

// workspace layout (floats): [vis: 2 rows][partial sums: 2 rows RS][loss terms per block: ceil(N nvis / 256)]
extern "C" size_t bhn_chi2_eht_ws_floats(int32_t N, int32_t C, int32_t nvis, int64_t R) {
    if (N <= 0 || C <= 0 || nvis <= 0 || R <= 0) return 0;
    const int64_t rows = (int64_t)N * C * nvis;
    return (size_t)(2 * rows * (1 + eht_splits(rows, R)) + ((int64_t)N * nvis + 255) / 256);
}

extern "C" int bhn_chi2_eht(const float *images, const float *A, const float *target, const float *sigma, float scale,
                            int32_t dtype, int32_t N, int32_t C, int32_t nvis, int64_t R, float *vis_ws, float *loss,
                            float *dimages, void *stream) {
    BHN_CHECK_ARG(images && A && target && sigma && vis_ws && loss, "null pointer");
    BHN_CHECK_ARG(dtype >= 0 && dtype <= 2, "eht dtype (%d) not supported", dtype);
    BHN_CHECK_ARG(N > 0 && nvis > 0 && R > 0, "bad sizes");
    BHN_CHECK_ARG((dtype == 2) ? (C >= 1 && C <= 8) : (C == 1), "A must have %s visibilities per closure", dtype == 2 ? "1..8" : "1");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)N * C * nvis, tot = (int64_t)N * nvis;
    const int RS = eht_splits(rows, R);
    float2 *vis = reinterpret_cast<float2 *>(vis_ws), *part = vis + rows;
    float *loss_part = vis_ws + 2 * rows * (1 + RS);
    const unsigned nblk = (unsigned)((tot + 255) / 256);
    // 16-byte loads of A and 8-byte loads of the images only when the caller's buffers allow them (a C caller may hand over
    // a 4- or 8-byte-aligned sub-view; torch allocations and row offsets are aligned): otherwise the scalar path
    const int wide = (R % 2 == 0) && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(images) & 7) == 0;
    hipLaunchKernelGGL(eht_vis_kernel, dim3((unsigned)rows, (unsigned)RS), dim3(256), 0, st, images,
                       reinterpret_cast<const float2 *>(A), C, nvis, R, RS, part, wide);
    BHN_HIP(hipGetLastError());
    hipLaunchKernelGGL(eht_loss_kernel, dim3(nblk), dim3(256), 0, st, vis, part, RS, target, sigma, scale, dtype, (int64_t)N,
                       C, nvis, loss_part, dimages ? 1 : 0);
    BHN_HIP(hipGetLastError());
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(256), 0, st, loss, loss_part, (int64_t)nblk);
    BHN_HIP(hipGetLastError());
    if (dimages) {
        hipLaunchKernelGGL(eht_bwd_kernel, dim3((unsigned)((R + 255) / 256), (unsigned)N), dim3(256), 0, st,
                           reinterpret_cast<const float2 *>(vis), reinterpret_cast<const float2 *>(A), C, nvis, R, dimages);
        BHN_HIP(hipGetLastError());
    }
    return BHN_OK;
}

// ------------------------------------------------------------------------------------------
// Voxel forward renderer: emission.image_plane_dynamics (emission.py:235-303) fused --
// velocity warp (emission.py:143-211) -> trilinear sampling of a 3-D emission grid
// (interpolate_coords, emission.py:213-233: scipy map_coordinates order=1, mode='constant', cval=0:
// NaN or any index outside [0, n-1] gives 0) -> x J g^2 dtau Sigma -> sum over the ray (kgeo.py:621).
// Gather-bound: the grid (<= a few MB) stays in L2, the geometry is streamed once per frame.
// ------------------------------------------------------------------------------------------
struct VoxelGrid {
    const float *data;          // (nx, ny, nz) C-order, or (B, nx, ny, nz) when frame_stride != 0
    long long frame_stride;
    int nx, ny, nz;
    float fx, fy, fz;           // extent of the grid along each axis (max - min of its coordinates)
};

__device__ __forceinline__ float trilinear(const VoxelGrid &v, const float *g, float x, float y, float z) {
    // utils.world_to_image_coords (utils.py:160-166): index = (c + fov/2) / fov * (n - 1)
    const float ix = (x + 0.5f * v.fx) / v.fx * (float)(v.nx - 1);
    const float iy = (y + 0.5f * v.fy) / v.fy * (float)(v.ny - 1);
    const float iz = (z + 0.5f * v.fz) / v.fz * (float)(v.nz - 1);
    if (!(ix >= 0.f && ix <= (float)(v.nx - 1) && iy >= 0.f && iy <= (float)(v.ny - 1) && iz >= 0.f && iz <= (float)(v.nz - 1)))
        return 0.f;                                              // also catches NaN (pre-injection)
    const int x0 = min((int)ix, max(v.nx - 2, 0)), y0 = min((int)iy, max(v.ny - 2, 0)), z0 = min((int)iz, max(v.nz - 2, 0));
    const int x1 = min(x0 + 1, v.nx - 1), y1 = min(y0 + 1, v.ny - 1), z1 = min(z0 + 1, v.nz - 1);
    const float tx = ix - (float)x0, ty = iy - (float)y0, tz = iz - (float)z0;
    const long long sx = (long long)v.ny * v.nz, sy = v.nz;
    const float c000 = g[x0 * sx + y0 * sy + z0], c001 = g[x0 * sx + y0 * sy + z1];
    const float c010 = g[x0 * sx + y1 * sy + z0], c011 = g[x0 * sx + y1 * sy + z1];
    const float c100 = g[x1 * sx + y0 * sy + z0], c101 = g[x1 * sx + y0 * sy + z1];
    const float c110 = g[x1 * sx + y1 * sy + z0], c111 = g[x1 * sx + y1 * sy + z1];
    const float c00 = c000 + (c001 - c000) * tz, c01 = c010 + (c011 - c010) * tz;
    const float c10 = c100 + (c101 - c100) * tz, c11 = c110 + (c111 - c110) * tz;
    const float c0 = c00 + (c01 - c00) * ty, c1 = c10 + (c11 - c10) * ty;
    return c0 + (c1 - c0) * tx;
}

template <int LPR>
__global__ __launch_bounds__(256) void voxel_render_kernel(bhn_geom geom, const double *__restrict__ tM0, int B, VoxelGrid v,
                                                           float *__restrict__ images) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long long ray_b = (long long)blockIdx.x * RPB + threadIdx.x / LPR;       // (frame, ray)
    const long long R = geom.R, G = geom.G, P = R * G;
    const int Sx = geom.S > 0 ? geom.S : 1;
    const bool ok = ray_b < (long long)B * R;
    const int b = ok ? (int)(ray_b / R) : 0;
    const long long ray = ok ? ray_b % R : 0;
    const float *grid = v.data + (long long)b * v.frame_stride;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const double t0 = tM0[b];
        for (long long k = sub; k < G; k += LPR) {
            const long long p = ray * G + k;
            const double tM = t0 + (double)geom.t_geo[p];
            float e = 0.f;
            if (!(tM < 0.0)) {                                      // emission.py:204-205: NaN before injection -> 0
                const double th = tM * (double)geom.Omega[p];
                double s, c;
                sincos(th, &s, &c);
                const float x = geom.x[p], y = geom.y[p];
                e = trilinear(v, grid, (float)(c * x + s * y), (float)(c * y - s * x), geom.z[p]);
            }
            if (e != 0.f)
                for (int s = 0; s < Sx; ++s) acc[s] += geom.w[(long long)s * P + p] * e;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) acc[s] += __shfl_xor(acc[s], o, 64);
    if (ok && sub == 0)
        for (int s = 0; s < Sx; ++s) images[((long long)b * Sx + s) * R + ray] = acc[s];
}

__global__ void trilinear_kernel(const float *__restrict__ pts, long long N, VoxelGrid v, float *__restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        out[i] = trilinear(v, v.data, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
}

static int voxel_grid_args(const float *grid, int32_t nx, int32_t ny, int32_t nz, int64_t frame_stride, const float *fov, VoxelGrid *v) {
    BHN_CHECK_ARG(grid && fov, "null pointer");
    BHN_CHECK_ARG(nx >= 1 && ny >= 1 && nz >= 1, "bad grid size %dx%dx%d", nx, ny, nz);
    BHN_CHECK_ARG(fov[0] > 0.f && fov[1] > 0.f && fov[2] > 0.f, "grid extents must be positive");
    v->data = grid; v->frame_stride = frame_stride; v->nx = nx; v->ny = ny; v->nz = nz;
    v->fx = fov[0]; v->fy = fov[1]; v->fz = fov[2];
    return BHN_OK;
}

extern "C" int bhn_voxel_render_fwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t nx, int32_t ny,
                                    int32_t nz, int64_t frame_stride, const float *fov_host, float *images, void *stream) {
    BHN_CHECK_ARG(geom && fr && images, "null pointer");
    BHN_CHECK_ARG(geom->x && geom->y && geom->z && geom->Omega && geom->t_geo && geom->w, "null geometry array");
    BHN_CHECK_ARG(geom->R > 0 && geom->G > 0 && geom->S >= 0 && geom->S <= 4 && fr->B > 0 && fr->tM0, "bad sizes");
    VoxelGrid v;
    int rc = voxel_grid_args(grid, nx, ny, nz, frame_stride, fov_host, &v);
    if (rc != BHN_OK) return rc;
    const long long rays = (long long)fr->B * geom->R;
    const int lpr = geom->G >= 48 ? 32 : (geom->G >= 12 ? 16 : 4);
    hipStream_t st = (hipStream_t)stream;
    if (lpr == 32) hipLaunchKernelGGL((voxel_render_kernel<32>), dim3((unsigned)((rays + 7) / 8)), dim3(256), 0, st, *geom, fr->tM0, fr->B, v, images);
    else if (lpr == 16) hipLaunchKernelGGL((voxel_render_kernel<16>), dim3((unsigned)((rays + 15) / 16)), dim3(256), 0, st, *geom, fr->tM0, fr->B, v, images);
    else hipLaunchKernelGGL((voxel_render_kernel<4>), dim3((unsigned)((rays + 63) / 64)), dim3(256), 0, st, *geom, fr->tM0, fr->B, v, images);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

extern "C" int bhn_trilinear(const float *points, int64_t N, const float *grid, int32_t nx, int32_t ny, int32_t nz,
                             const float *fov_host, float *out, void *stream) {
    BHN_CHECK_ARG(points && out && N > 0, "bad arguments");
    VoxelGrid v;
    int rc = voxel_grid_args(grid, nx, ny, nz, 0, fov_host, &v);
    if (rc != BHN_OK) return rc;
    const int blocks = (int)((N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048);
    hipLaunchKernelGGL(trilinear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, points, (long long)N, v, out);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// ------------------------------------------------------------------------------------------
// GRID_Predictor (network.py:254-353): the emission is a learnable res^3 voxel grid -- warp -> voxel index
// (u + scale)/(2 scale)(res - 1) -> trilinear sample (0 outside the grid: map_coordinates order 1, cval 0) ->
// sigmoid(. - 10) -> domain fill -> 0 before the injection -> x w -> ray sum.  MODE 0: emission (B,P);
// 1: images (B,Sx,R); 2: d loss / d grid from d loss / d images (float atomics into the res^3 grid, which is
// L2-resident: 1 MB at res 64; not bitwise reproducible, unlike the MLP gradient).
// ------------------------------------------------------------------------------------------
template <int LPR, int MODE>
__global__ __launch_bounds__(256) void grid_predictor_kernel(bhn_geom geom, const double *__restrict__ tM0, int B, VoxelGrid v,
                                                             float *__restrict__ out, const float *__restrict__ dimages,
                                                             float *__restrict__ dgrid) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long long ray_b = (long long)blockIdx.x * RPB + threadIdx.x / LPR;       // (frame, ray)
    const long long R = geom.R, G = geom.G, P = R * G;
    const int Sx = geom.S > 0 ? geom.S : 1;
    const bool ok = ray_b < (long long)B * R;
    const int b = ok ? (int)(ray_b / R) : 0;
    const long long ray = ok ? ray_b % R : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float dI[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 2 && ok)
        for (int s = 0; s < Sx; ++s) dI[s] = dimages[((long long)b * Sx + s) * R + ray];
    if (ok) {
        const double t0 = tM0[b];
        const float nm1 = (float)(v.nx - 1);
        for (long long k = sub; k < G; k += LPR) {
            const long long p = ray * G + k;
            const double tM = t0 + (double)geom.t_geo[p];
            float e = 0.f;
            float ix = 0.f, iy = 0.f, iz = 0.f;
            bool inside = false;
            if (!(tM < 0.0) && geom.dom[p]) {                       // emission.py:204-205, 370-373
                const double th = tM * (double)geom.Omega[p];
                double sn, cs;
                sincos(th, &sn, &cs);
                const float x = geom.x[p], y = geom.y[p];
                const float ux = (float)(cs * x + sn * y), uy = (float)(cs * y - sn * x), uz = geom.z[p];
                if (isfinite(ux) && isfinite(uy) && isfinite(uz)) {
                    ix = (ux + 0.5f * v.fx) / v.fx * nm1; iy = (uy + 0.5f * v.fy) / v.fy * nm1; iz = (uz + 0.5f * v.fz) / v.fz * nm1;
                    inside = ix >= 0.f && ix <= nm1 && iy >= 0.f && iy <= nm1 && iz >= 0.f && iz <= nm1;
                    const float val = inside ? trilinear(v, v.data, ux, uy, uz) : 0.f;
                    e = 1.f / (1.f + __expf(10.f - val));
                }
            }
            if (MODE == 0) out[(long long)b * P + p] = e;
            else if (MODE == 1) {
                if (e != 0.f)
                    for (int s = 0; s < Sx; ++s) acc[s] += geom.w[(long long)s * P + p] * e;
            } else if (inside && e != 0.f) {
                float dE = 0.f;
                for (int s = 0; s < Sx; ++s) dE += dI[s] * geom.w[(long long)s * P + p];
                const float d = dE * e * (1.f - e);
                const int n = v.nx;
                const int x0 = min((int)ix, max(n - 2, 0)), y0 = min((int)iy, max(n - 2, 0)), z0 = min((int)iz, max(n - 2, 0));
                const int x1 = min(x0 + 1, n - 1), y1 = min(y0 + 1, n - 1), z1 = min(z0 + 1, n - 1);
                const float tx = ix - (float)x0, ty = iy - (float)y0, tz = iz - (float)z0;
                const long long sx = (long long)n * n, sy = n;
                if (d != 0.f) {
                    atomicAdd(dgrid + x0 * sx + y0 * sy + z0, d * (1.f - tx) * (1.f - ty) * (1.f - tz));
                    atomicAdd(dgrid + x0 * sx + y0 * sy + z1, d * (1.f - tx) * (1.f - ty) * tz);
                    atomicAdd(dgrid + x0 * sx + y1 * sy + z0, d * (1.f - tx) * ty * (1.f - tz));
                    atomicAdd(dgrid + x0 * sx + y1 * sy + z1, d * (1.f - tx) * ty * tz);
                    atomicAdd(dgrid + x1 * sx + y0 * sy + z0, d * tx * (1.f - ty) * (1.f - tz));
                    atomicAdd(dgrid + x1 * sx + y0 * sy + z1, d * tx * (1.f - ty) * tz);
                    atomicAdd(dgrid + x1 * sx + y1 * sy + z0, d * tx * ty * (1.f - tz));
                    atomicAdd(dgrid + x1 * sx + y1 * sy + z1, d * tx * ty * tz);
                }
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc[s] += __shfl_xor(acc[s], o, 64);
        if (ok && sub == 0)
            for (int s = 0; s < Sx; ++s) out[((long long)b * Sx + s) * R + ray] = acc[s];
    }
}

template <int MODE>
static int grid_predictor_launch(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale, float *out,
                                 const float *dimages, float *dgrid, void *stream) {
    BHN_CHECK_ARG(geom && fr && grid, "null pointer");
    BHN_CHECK_ARG(geom->x && geom->y && geom->z && geom->Omega && geom->t_geo && geom->dom, "null geometry array");
    BHN_CHECK_ARG(MODE == 0 || geom->w, "render needs geom->w");
    BHN_CHECK_ARG(geom->R > 0 && geom->G > 0 && geom->S >= 0 && geom->S <= 4 && fr->B > 0 && fr->tM0, "bad sizes");
    BHN_CHECK_ARG(res >= 2 && res <= 1024 && scale > 0.f, "bad grid_res %d / scale %g", res, (double)scale);
    const float fov[3] = {2.f * scale, 2.f * scale, 2.f * scale};
    VoxelGrid v;
    int rc = voxel_grid_args(grid, res, res, res, 0, fov, &v);
    if (rc != BHN_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (MODE == 2) BHN_HIP(hipMemsetAsync(dgrid, 0, sizeof(float) * (size_t)res * res * res, st));
    const long long rays = (long long)fr->B * geom->R;
    const int lpr = geom->G >= 48 ? 32 : (geom->G >= 12 ? 16 : 4);
    if (lpr == 32) hipLaunchKernelGGL((grid_predictor_kernel<32, MODE>), dim3((unsigned)((rays + 7) / 8)), dim3(256), 0, st, *geom, fr->tM0, fr->B, v, out, dimages, dgrid);
    else if (lpr == 16) hipLaunchKernelGGL((grid_predictor_kernel<16, MODE>), dim3((unsigned)((rays + 15) / 16)), dim3(256), 0, st, *geom, fr->tM0, fr->B, v, out, dimages, dgrid);
    else hipLaunchKernelGGL((grid_predictor_kernel<4, MODE>), dim3((unsigned)((rays + 63) / 64)), dim3(256), 0, st, *geom, fr->tM0, fr->B, v, out, dimages, dgrid);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

extern "C" int bhn_grid_predict_fwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale,
                                    float *emission, void *stream) {
    BHN_CHECK_ARG(emission, "null emission");
    return grid_predictor_launch<0>(geom, fr, grid, res, scale, emission, nullptr, nullptr, stream);
}

extern "C" int bhn_grid_render_fwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale,
                                   float *images, void *stream) {
    BHN_CHECK_ARG(images, "null images");
    return grid_predictor_launch<1>(geom, fr, grid, res, scale, images, nullptr, nullptr, stream);
}

extern "C" int bhn_grid_render_bwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale,
                                   const float *dimages, float *dgrid, void *stream) {
    BHN_CHECK_ARG(dimages && dgrid, "null pointer");
    return grid_predictor_launch<2>(geom, fr, grid, res, scale, nullptr, dimages, dgrid, stream);
}

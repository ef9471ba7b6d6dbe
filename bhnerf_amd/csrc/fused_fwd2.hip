// Fused predictor / render forward, wide-tile variant (bf16): 4 waves per workgroup, one per SIMD, each wave
// owns NPT = 2 point tiles (64 points) and the whole 512-entry register file.  Compared with
// fused_fwd_kernel (8 waves x 32 points): every A fragment read from LDS feeds two MFMAs, half as many waves
// meet at each barrier, the weight chunks arrive by LDS-DMA (3-buffer ring, no staging registers), and the
// ReLU/pack VALU work of one point tile can issue in the MFMA shadow of the other (the two waves that shared
// a SIMD before ran in barrier lockstep and could not cover for each other -- DESIGN.md, round-1 ablations).
#include "fused_common.h"

struct PolBF16W : PolBF16 {
    static constexpr int NWAVES = 4;
    static constexpr int NTHREADS = NWAVES * 64;
};

template <int W, class Pol, int DEG, bool RENDER, int NPT>
__global__ __launch_bounds__(Pol::NTHREADS) void fused_fwd2_kernel(FusedArgs a) {
    using PK = Pack<W, Pol>;
    using frag = typename Pol::frag;
    using RG = DmaRing<PK::CHUNK_BYTES, Pol::NWAVES>;
    constexpr int CB = PK::CHUNK_BYTES, MT = PK::MT, KS = PK::KS;
    constexpr int DIST = 2, NB = DIST + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;                                              // NB x CB
    float *bias_lds = reinterpret_cast<float *>(smem + NB * CB);     // (depth+1) x W

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 31, h = lane >> 5;
    for (int i = tid; i < (a.depth + 1) * W; i += Pol::NTHREADS)
        bias_lds[i] = reinterpret_cast<const float *>(a.packed + a.bias_off)[i];

    const char *fwd = a.packed + a.fwd_off;
    const int NC = PK::fwd_chunks(a.depth);
    auto chunk_src = [&](int seq) { while (seq >= NC) seq -= NC; return fwd + (size_t)seq * CB; };
#pragma unroll
    for (int j = 0; j < DIST; ++j) RG::issue(chunk_src(j), ring + j * CB);
    RG::template wait_younger<RG::PPW * (DIST - 1)>();
    lds_barrier();
    int cur = 0;
#define STEP_BEGIN(seq) { const int nx = cur >= 1 ? cur - 1 : NB - 1; RG::issue(chunk_src((seq) + DIST), ring + nx * CB); } \
    const char *ch = ring + cur * CB;
#define STEP_END() RG::template wait_younger<RG::PPW * (DIST - 1)>(); lds_barrier(); cur = cur == NB - 1 ? 0 : cur + 1;

    for (long long tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        int b = 0;
        long long p[NPT];
        bool inb[NPT], live[NPT];
        frag enc[NPT][2];
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const PointIn in = load_point<Pol::NWAVES * NPT>(a, tile, wv * NPT + t, pl);
            b = in.b; p[t] = in.p; inb[t] = in.inb;
            point_prologue<Pol, DEG>(a, in, enc[t], live[t]);
        }
        frag act[NPT][KS], next[NPT][KS];
        int seq = 0;
        // ---- layer 0 (chunk 0: fragment m*2+ks) -------------------------------------------
        {
            STEP_BEGIN(seq)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const frag a0 = Pol::lds_frag(ch, 2 * m, lane), a1 = Pol::lds_frag(ch, 2 * m + 1, lane);
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
                    f32x16 acc = bias_acc(bias_lds, m, h);
                    acc = Pol::mma(a0, enc[t][0], acc);
                    acc = Pol::mma(a1, enc[t][1], acc);
                    relu_pack<W, Pol>(acc, m, act[t]);
                }
            }
            STEP_END()
            ++seq;
        }
        // ---- hidden layers 1..depth-1 -----------------------------------------------------
        for (int l = 1; l < a.depth; ++l) {
            const bool sk = (a.skip_mask >> l) & 1;
            const float *bl = bias_lds + l * W;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                STEP_BEGIN(seq)
                f32x16 acc[NPT];
#pragma unroll
                for (int t = 0; t < NPT; ++t) acc[t] = bias_acc(bl, m, h);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const frag af = Pol::lds_frag(ch, ks, lane);
#pragma unroll
                    for (int t = 0; t < NPT; ++t) acc[t] = Pol::mma(af, act[t][ks], acc[t]);
                }
                if (sk) {
                    const frag e0 = Pol::lds_frag(ch, KS, lane), e1 = Pol::lds_frag(ch, KS + 1, lane);
#pragma unroll
                    for (int t = 0; t < NPT; ++t) {
                        acc[t] = Pol::mma(e0, enc[t][0], acc[t]);
                        acc[t] = Pol::mma(e1, enc[t][1], acc[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < NPT; ++t) relu_pack<W, Pol>(acc[t], m, next[t]);
                STEP_END()
                ++seq;
            }
#pragma unroll
            for (int t = 0; t < NPT; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) act[t][ks] = next[t][ks];
        }
        // ---- output layer (row 0 of the tile is the pre-activation) -----------------------
        float outv[NPT];
        {
            STEP_BEGIN(seq)
            f32x16 acc[NPT];
#pragma unroll
            for (int t = 0; t < NPT; ++t) acc[t] = bias_acc(bias_lds + a.depth * W, 0, h);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag af = Pol::lds_frag(ch, ks, lane);
#pragma unroll
                for (int t = 0; t < NPT; ++t) acc[t] = Pol::mma(af, act[t][ks], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < NPT; ++t) outv[t] = acc[t][0];
            STEP_END()
            ++seq;
        }
        // ---- epilogue: sigmoid(out - 10), masks (network.py:230-232), render ---------------
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            float e = 0.f;
            if (h == 0 && live[t]) e = 1.f / (1.f + Pol::fexp(10.f - outv[t]));
            if (!RENDER) {
                if (h == 0 && inb[t]) a.emission[(long long)b * a.P + p[t]] = e;
            } else {
                const long long ray = inb[t] ? p[t] / a.G : -1;
                unsigned long long rem = __ballot(h == 0 && inb[t]);
                while (rem) {
                    const int first = __ffsll((long long)rem) - 1;
                    const long long r0 = __shfl(ray, first, 64);
                    const bool mine = (h == 0) && inb[t] && (ray == r0);
                    for (int s = 0; s < a.Sx; ++s) {
                        float v = (mine && e != 0.f) ? a.w[(long long)s * a.P + p[t]] * e : 0.f;
                        v = half_wave_sum(v);
                        if (lane == first) atomicAdd(a.images + ((long long)b * a.Sx + s) * a.R + r0, v);
                    }
                    rem &= ~__ballot(mine);
                }
            }
        }
    }
#undef STEP_BEGIN
#undef STEP_END
}

int fused_fill_args(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                    const bhn_frames *fr, bool need_w, FusedArgs *a, MlpShape *s, int nwaves);

template <int W, bool RENDER>
static int launch_fwd2_w(FusedArgs &a, hipStream_t st) {
    using Pol = PolBF16W;
    using PK = Pack<W, Pol>;
    constexpr int NPT = 2;
    const size_t lds = 3 * PK::CHUNK_BYTES + (size_t)(a.depth + 1) * W * 4;
    auto kern = fused_fwd2_kernel<W, Pol, 3, RENDER, NPT>;
    static bool attr_done = false;
    if (!attr_done) {
        BHN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    long long grid = bhn_num_cus(dev);
    if (grid > a.total_tiles) grid = a.total_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Pol::NTHREADS), lds, st, a);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// bf16 forward entry used by bhn_predict_fwd / bhn_render_fwd (fused_fwd.hip); groups per tile = 4 waves x 2
int launch_fwd2_bf16(const bhn_model *m, const void *packed, const bhn_geom *geom, const bhn_frames *fr, bool render,
                     float *out, hipStream_t st) {
    FusedArgs a;
    MlpShape s;
    int rc = fused_fill_args(m, BHN_BF16, packed, geom, fr, render, &a, &s, PolBF16W::NWAVES * 2);
    if (rc != BHN_OK) return rc;
    if (render) {
        a.images = out;
        BHN_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)a.B * a.Sx * a.R, st));
    } else {
        a.emission = out;
        if (geom->groups) BHN_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)a.B * a.P, st));
    }
    switch (s.width) {
        case 64: return render ? launch_fwd2_w<64, true>(a, st) : launch_fwd2_w<64, false>(a, st);
        case 128: return render ? launch_fwd2_w<128, true>(a, st) : launch_fwd2_w<128, false>(a, st);
        case 256: return render ? launch_fwd2_w<256, true>(a, st) : launch_fwd2_w<256, false>(a, st);
        default: return -1;      // caller falls back to the 8-wave kernel (width 32)
    }
}

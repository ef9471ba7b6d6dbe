// Fused NeRF_Predictor forward (network.py:191-237) and fused image-plane prediction
// (network.py:373-420): warp -> posenc -> skip-MLP -> sigmoid/masks [-> x w -> sum over the ray].
#include <type_traits>
#include "fused_common.h"

// ---------------------------------------------------------------------------------------------
// packed weight image
// ---------------------------------------------------------------------------------------------
struct PackedLayout {
    int CH, chunk_bytes, n_fwd, n_bwd;
    size_t fwd_off, bwd_off, bias_off, wout_off, total;
};

static void packed_layout(const MlpShape &s, int mode, PackedLayout *L) {
    const int frag_bytes = (mode == BHN_BF16) ? 1024 : 2048;
    const int MT = s.width / 32, KS = s.width / 16;
    L->CH = KS + 2;
    L->chunk_bytes = L->CH * frag_bytes;
    L->n_fwd = 1 + (s.depth - 1) * MT + 1;
    L->n_bwd = (s.depth - 1) * MT;
    L->fwd_off = 0;
    L->bwd_off = (size_t)L->n_fwd * L->chunk_bytes;
    L->bias_off = L->bwd_off + (size_t)L->n_bwd * L->chunk_bytes;
    L->wout_off = L->bias_off + (size_t)(s.depth + 1) * s.width * 4;
    L->total = L->wout_off + (size_t)s.width * 4;
}

extern "C" size_t bhn_packed_bytes(const bhn_model *m, int32_t mode) {
    mode = bhn_norm_mode(mode);
    MlpShape s;
    if (bhn_mlp_shape(m, &s) != BHN_OK) return 0;
    if (s.general) return gen_packed_bytes(s, bhn_norm_mode(mode));
    PackedLayout L;
    packed_layout(s, mode, &L);
    return L.total;
}

struct PackArgs {
    MlpShape s;
    PackedLayout L;
    int mode;
    const float *params;
    char *packed;
};

__global__ void pack_weights_kernel(PackArgs a) {
    const MlpShape &s = a.s;
    // W: kernel width (fragment geometry); WT: the model's width (strides of the flat parameters).  Hidden units >= WT
    // are padding: zero weights in and out, zero bias.
    const int W = s.width, WT = s.width_true, MT = W / 32, KS = W / 16, CH = a.L.CH, D = s.depth;
    const long long n_frag_elems = (long long)(a.L.n_fwd + a.L.n_bwd) * CH * 512;
    const long long n_bias = (long long)(D + 1) * W;
    const long long total = n_frag_elems + n_bias + W;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        if (t >= n_frag_elems) {
            const long long u = t - n_frag_elems;
            if (u < n_bias) {
                const int l = (int)(u / W), o = (int)(u % W);
                float v = 0.f;
                if (l < D) { if (o < WT) v = a.params[s.bias_off[l] + o]; }
                else if (o == 0) v = a.params[s.bias_off[D]];
                reinterpret_cast<float *>(a.packed + a.L.bias_off)[u] = v;
            } else {
                const int k = (int)(u - n_bias);
                reinterpret_cast<float *>(a.packed + a.L.wout_off)[k] = k < WT ? a.params[s.kernel_off[D] + k] : 0.f;
            }
            continue;
        }
        const int j = (int)(t & 7);
        const int lane = (int)((t >> 3) & 63);
        const int f = (int)((t >> 9) % CH);
        const int c = (int)((t >> 9) / CH);       // chunk index over fwd then bwd images
        const int i = lane & 31, h = lane >> 5;
        const int ph = 8 * (j >> 2) + 4 * h + (j & 3);
        float v = 0.f;
        bool fwd = c < a.L.n_fwd;
        if (fwd) {
            if (c == 0) {                                   // layer 0: frag = m*2 + ks
                if (f < 2 * MT) {
                    const int m = f >> 1, q = 16 * (f & 1) + ph;
                    const int fe = bhn_enc_slot_feature(q, s.deg);
                    if (fe >= 0 && 32 * m + i < WT) v = a.params[s.kernel_off[0] + (long long)fe * WT + 32 * m + i];
                }
            } else if (c == a.L.n_fwd - 1) {                // output layer, only row 0 is real
                if (f < KS) {
                    if (i == 0 && 16 * f + ph < WT) v = a.params[s.kernel_off[D] + 16 * f + ph];
                } else if (s.skip_in[D] && i == 0) {        // odd depths: the output layer takes concat[h, enc]
                    const int fe = bhn_enc_slot_feature(16 * (f - KS) + ph, s.deg);
                    if (fe >= 0) v = a.params[s.kernel_off[D] + WT + fe];
                }
            } else {
                const int l = 1 + (c - 1) / MT, m = (c - 1) % MT;
                const int o = 32 * m + i;
                if (f < KS) {
                    const int k = 16 * f + ph;
                    if (k < WT && o < WT) v = a.params[s.kernel_off[l] + (long long)k * WT + o];
                } else if (s.skip_in[l]) {
                    const int q = 16 * (f - KS) + ph;
                    const int fe = bhn_enc_slot_feature(q, s.deg);
                    if (fe >= 0 && o < WT) v = a.params[s.kernel_off[l] + (long long)(WT + fe) * WT + o];
                }
            }
        } else {
            const int cb = c - a.L.n_fwd;
            const int l = 1 + cb / MT, m = cb % MT;
            const int k = 32 * m + i, o = 16 * f + ph;     // transposed image: row = input unit k, column = output unit o
            if (f < KS && k < WT && o < WT) {
                v = a.params[s.kernel_off[l] + (long long)k * WT + o];
                // layer depth-1: column o carries W_out[o] (the delta chain's B operand is relu' (.) dout only; common.h)
                if (l == D - 1 && bhn_folds_wout(a.mode, D)) v *= a.params[s.kernel_off[D] + o];
            }
        }
        char *img = a.packed + (fwd ? a.L.fwd_off + (size_t)c * a.L.chunk_bytes
                                    : a.L.bwd_off + (size_t)(c - a.L.n_fwd) * a.L.chunk_bytes);
        if (a.mode == BHN_BF16) {
            reinterpret_cast<__bf16 *>(img + f * 1024 + lane * 16)[j] = (__bf16)v;
        } else {
            reinterpret_cast<float *>(img + f * 2048 + (j >> 2) * 1024 + lane * 16)[j & 3] = v;
        }
    }
}

extern "C" int bhn_pack_weights(const bhn_model *m, int32_t mode, const float *params, void *packed, void *stream) {
    mode = bhn_norm_mode(mode);
    BHN_CHECK_ARG(params && packed, "null pointer");
    BHN_CHECK_ARG(mode == BHN_F32 || mode == BHN_BF16, "bad mode %d", mode);
    PackArgs a;
    int rc = bhn_mlp_shape(m, &a.s);
    if (rc != BHN_OK) return rc;
    if (a.s.general) return gen_pack_weights(a.s, mode, params, packed, (hipStream_t)stream);
    packed_layout(a.s, mode, &a.L);
    a.mode = mode;
    a.params = params;
    a.packed = (char *)packed;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, a);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// ---------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------
// DBG: measurement build (bit 128: per-wave time stamps [compute done, barrier passed] of the ring steps of one tile); (tools/dbg_fwd_ablate.py): a.debug bits knock out one cost at a time -- 1 hidden/output MFMAs,
// 2 relu+pack, 4 weight DMA + its waits, 8 barriers, 16 posenc trig, 32 epilogue.  Results are then meaningless.
// RES: the whole weight image resident in LDS (ResidentRing: no DMA, no per-chunk barrier), when it fits
template <int W, class Pol, int DEG, bool RENDER, bool DBG = false, bool RES = false>
__global__ __launch_bounds__(Pol::NTHREADS) __attribute__((amdgpu_waves_per_eu(Pol::WPE, Pol::WPE))) void fused_fwd_kernel(FusedArgs a) {
    const int dbg = DBG ? a.debug : 0;
    clock_stamp(a.clk, BHN_CLK_FWD, 0);
    using PK = Pack<W, Pol>;
    using frag = typename Pol::frag;
    constexpr int CB = PK::CHUNK_BYTES, MT = PK::MT, KS = PK::KS;
    using EB = EncBlock<W, Pol>;
    constexpr bool ENCR = EB::ON && !RES;                           // the ring copies the hidden fragments only; encoded-input block resident (EncBlock)
    constexpr int NFR = ENCR ? KS : KS + 2;
    using RG = DmaRing<ENCR ? KS * Pol::FRAG_BYTES : CB, Pol::NWAVES>;
    constexpr int DIST = Pol::FWD_DIST;                                                // LDS-DMA weight ring: chunks in flight
    using RS = std::conditional_t<RES, ResidentRing<RG, CB, MT>, RingState<RG, CB, DIST, Pol::PHASE_LAG, MT, DBG>>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;                                              // NB x (bytes copied per chunk) (resident: all chunks)
    float *bias_lds = reinterpret_cast<float *>(smem + RS::lds_bytes(PK::fwd_chunks(a.depth)));     // (depth+1) x W
    char *seg_lds = reinterpret_cast<char *>(bias_lds + (a.depth + 1) * W);      // RaySum scratch
    char *encblk = seg_lds + RaySum<Pol::NWAVES>::bytes(a.Sx);                   // EncBlock (ENCR)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 31, h = lane >> 5;
    for (int i = tid; i < (a.depth + 1) * W; i += Pol::NTHREADS)
        bias_lds[i] = reinterpret_cast<const float *>(a.packed + a.bias_off)[i];
    if constexpr (ENCR) EB::fill(encblk, a.packed + a.fwd_off, a.depth, a.skip_mask);

    // weight ring (LDS-DMA), software-pipelined steps: fused_common.h "Software-pipelined ring steps"
    RS rs;
    rs.start(ring, a.packed + a.fwd_off, PK::fwd_chunks(a.depth), nullptr, 0, dbg, (wv >= Pol::NWAVES / 2 && !(a.debug & 64)) ? 1 : 0);
    if (rs.lag) rs.idle_step();
    APipe<Pol> ap;
    ap.prime(rs.ch(), bias_lds);

    PointIn nxt = load_point<Pol::NWAVES>(a, blockIdx.x, wv, pl);
    for (long long tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const PointIn in = nxt;
        const int b = in.b;
        const long long p = in.p;
        const bool inb = in.inb;
        frag enc[2];
        bool live;
        if (!(dbg & 16)) point_prologue<Pol, DEG>(a, in, enc, live);
        else { live = in.dom; for (int j = 0; j < 8; ++j) { Pol::set(enc[0], j, in.x); Pol::set(enc[1], j, in.tg); } }
        nxt = load_point<Pol::NWAVES>(a, tile + gridDim.x, wv, pl);      // next tile's inputs fly during this tile
        float w0 = 0.f;                                                  // quadrature weight of s = 0, needed at the very end
        if (RENDER && h == 0 && inb) w0 = a.w[p];

        frag act[KS], next[KS];
        if (DBG) for (int ks = 0; ks < KS; ++ks) next[ks] = act[ks] = Pol::zero();
        if (DBG) {      // bit 128: stamp the ring steps of this workgroup's 4th tile, waves 0 and NWAVES/2 -> a.emission
            const bool on = (dbg & 128) && blockIdx.x == 0 && tile == blockIdx.x + 3 * (long long)gridDim.x;
            rs.ts = on ? reinterpret_cast<long long *>(a.emission) + wv * 64 : nullptr;
        }
        f32x16 pend;
        PackTile0<Pol> l0;
        layer0_step<W, Pol, RG, 0, RS, PackTile0<Pol>, NFR>(rs, ap, enc, act, bias_lds, h, pend, l0);
        // ---- hidden layers 1..depth-1, ping-pong act <-> next (no register copies) -------------
        float outv;
        {
            // one copy of the layer code (the fully unrolled 3-layer body did not fit the instruction cache:
            // every step then paid ~1000 cycles of instruction fetch); the price is 56 v_mov per layer
#pragma nounroll
            for (int l = 1; l < a.depth; ++l) {
                hidden_layer<W, Pol, RG, RS, NFR>(rs, ap, act, next, enc, (a.skip_mask >> l) & 1, bias_lds + l * W, pend, encblk);
#pragma unroll
                for (int ks = 0; ks < KS - 2; ++ks) act[ks] = next[ks];      // the pending tile lands in act[KS-2], act[KS-1]
            }
            // ---- output layer (row 0 of the tile is the pre-activation) ------------------------
            const char *ch = rs.ch(), *chn = rs.chn();
            const DmaJob dj = rs.job();
            PackPost<Pol> post(pend, act[KS - 2], act[KS - 1]);
            const f32x16 acc = ring_step<W, Pol, RG, PackPost<Pol>, NFR>(ch, chn, ap, act, enc, (a.skip_mask >> a.depth) & 1, bias_lds /* next tile, layer 0 */, post, dj, dbg,
                                                                         encblk + 2 * MT * Pol::FRAG_BYTES);
            outv = acc[0];
            rs.step_end();
        }
        // ---- epilogue: sigmoid(out - 10), masks (network.py:230-232) ----------------------
        float e = 0.f;
        if (h == 0 && live) e = 1.f / (1.f + Pol::fexp(10.f - outv));
        if (dbg & 32) {
            if (e == 12345.f) a.images[0] = e;
        } else if (!RENDER) {
            if (h == 0 && inb) a.emission[(long long)b * a.P + p] = e;
        } else {
            // x J g^2 dtau Sigma and the sum over the ray: segment sums per wave, combined per workgroup tile (RaySum)
            RaySum<Pol::NWAVES>::run(a, seg_lds, b, p, inb, e, w0, true);
            // resident weights: no ring barriers separate this tile's combine from the next tile's segment sums
            if constexpr (RES) { if (!a.ray_direct) lds_barrier(); }
        }
    }
    if (!rs.lag) rs.idle_step();                        // every wave runs the same number of ring steps (barriers)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
    clock_stamp(a.clk, BHN_CLK_FWD, 1);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int fused_fill_args(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                    const bhn_frames *fr, bool need_w, FusedArgs *a, MlpShape *s, int nwaves) {
    BHN_CHECK_ARG(m && packed && geom && fr, "null pointer");
    BHN_CHECK_ARG(mode == BHN_F32 || mode == BHN_BF16, "bad mode %d", mode);
    int rc = bhn_mlp_shape(m, s);
    if (rc != BHN_OK) return rc;
    BHN_CHECK_ARG(geom->R > 0 && geom->G > 0 && geom->S >= 0 && geom->S <= 4, "bad geometry sizes");
    BHN_CHECK_ARG(geom->x && geom->y && geom->z && geom->Omega && geom->t_geo && geom->dom, "null geometry array");
    BHN_CHECK_ARG(!need_w || geom->w, "render needs geom->w");
    BHN_CHECK_ARG(fr->B > 0 && fr->tM0, "bad frames");
    BHN_CHECK_ARG(m->scale > 0.f, "scale must be positive");
    PackedLayout L;
    packed_layout(*s, mode, &L);
    memset(a, 0, sizeof(*a));
    a->depth = s->depth;
    a->deg = s->deg;
    for (int l = 0; l <= s->depth; ++l) a->skip_mask |= s->skip_in[l] << l;
    a->scale = m->scale;
    a->inv_scale = (float)(1.0 / (double)m->scale);
    a->x = geom->x; a->y = geom->y; a->z = geom->z; a->Omega = geom->Omega; a->t_geo = geom->t_geo;
    a->w = geom->w; a->dom = geom->dom;
    a->R = geom->R; a->G = geom->G; a->P = geom->R * geom->G;
    if (geom->ray_idx) {                                   // point-level compaction of the domain mask
        BHN_CHECK_ARG(geom->n_points > 0 && geom->n_points % 32 == 0 && !geom->groups,
                      "compacted geometry: n_points %lld must be a positive multiple of 32 and groups NULL", (long long)geom->n_points);
        a->P = geom->n_points;
        a->ray_idx = reinterpret_cast<const int *>(geom->ray_idx);
    }
    a->Sx = geom->S > 0 ? geom->S : 1;
    a->tM0 = fr->tM0; a->B = fr->B;
    a->clk = reinterpret_cast<long long *>(fr->clock_probe);
    a->packed = (const char *)packed;
    a->fwd_off = (unsigned)L.fwd_off; a->bwd_off = (unsigned)L.bwd_off;
    a->bias_off = (unsigned)L.bias_off; a->wout_off = (unsigned)L.wout_off;
    a->groups = reinterpret_cast<const int *>(geom->groups);
    a->n_groups = geom->groups ? geom->n_groups : (a->P + 31) / 32;
    BHN_CHECK_ARG(a->n_groups > 0 && a->n_groups <= (a->P + 31) / 32, "bad n_groups %lld", (long long)a->n_groups);
    a->tiles_per_frame = (int)((a->n_groups + nwaves - 1) / nwaves);
    a->total_tiles = (long long)a->tiles_per_frame * a->B;
    BHN_CHECK_ARG(a->total_tiles < (1ll << 31) && a->P < (1ll << 31) && a->G < (1ll << 31),
                  "problem too large for one call: %lld tiles, %lld points per frame (limit 2^31)", a->total_tiles, (long long)a->P);
    a->fd_tpf = FastDiv::make((unsigned)a->tiles_per_frame);
    a->fd_G = FastDiv::make((unsigned)a->G);
    a->ray_direct = a->ray_idx ? (geom->ray_span == 1 || geom->ray_span == 2) : (a->G <= 33 || (a->G <= 64 && a->G % 32 == 0));
#ifdef BHN_FORCE_RAY_DIRECT          // measurement build: per-wave atomics whatever the ray layout (pixel sums then depend on the arrival order)
    a->ray_direct = 1;
#endif
    return BHN_OK;
}


#ifndef BHN_RESIDENT
#define BHN_RESIDENT 1           // 0: never keep the weight image resident in LDS (A/B builds)
#endif
template <int W, class Pol, bool RENDER, bool DBG = false, bool RES = false>
static int launch_fwd_w(FusedArgs &a, hipStream_t st) {
    using PK = Pack<W, Pol>;
    const size_t lds_fixed = (size_t)(a.depth + 1) * W * 4 + RaySum<Pol::NWAVES>::bytes(a.Sx);
    if constexpr (!RES && !DBG && W <= 128 && BHN_RESIDENT != 0 && Pol::ELEM_BYTES == 2) {     // (f32, one wave per SIMD: measured 11 % slower resident)
        // small networks: all chunks of the forward image resident in LDS, waves run without the per-chunk barrier
        if ((size_t)PK::fwd_chunks(a.depth) * PK::CHUNK_BYTES + lds_fixed <= 160 * 1024) return launch_fwd_w<W, Pol, RENDER, DBG, true>(a, st);
    }
    constexpr bool ENCR = EncBlock<W, Pol>::ON && !RES;             // (the kernel's own condition)
    const size_t lds = RES ? (size_t)PK::fwd_chunks(a.depth) * PK::CHUNK_BYTES + lds_fixed
                           : (size_t)(Pol::FWD_DIST + (Pol::PHASE_LAG ? 2 : 1)) * (ENCR ? (size_t)PK::KS * Pol::FRAG_BYTES : (size_t)PK::CHUNK_BYTES) + lds_fixed +
                             (ENCR ? (size_t)EncBlock<W, Pol>::BYTES : 0);
    auto kern = fused_fwd_kernel<W, Pol, 3, RENDER, DBG, RES>;
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    BHN_CHECK_DEVICE(dev);
    static DeviceOnce once;                 // per template instantiation and device: LDS attribute + occupancy
    BHN_HIP(once.run(dev, [&](int &occ) {
        occ = 1;
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kern, Pol::NTHREADS, lds) == hipSuccess && o > 0) occ = o;
        return hipSuccess;
    }));
    const long long grid = bhn_balanced_grid(a.total_tiles, (long long)bhn_num_cus(dev) * once.value[dev]);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Pol::NTHREADS), lds, st, a);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// 32-point groups per workgroup tile of the forward kernels a model and ray set take (fused_fill_args: nwaves)
static int fwd_tile_groups(const bhn_model *m, int mode, const bhn_geom *geom) {
    MlpShape s;
    if (m && bhn_mlp_shape(m, &s) == BHN_OK && !s.general && bhn_fwd_w12(mode, s.width, s.depth, bhn_groups_per_frame(geom))) return PolBF16X::NWAVES;
    return (mode == BHN_BF16) ? PolBF16::NWAVES : PolF32::NWAVES;
}

template <class Pol, bool RENDER>
static int launch_fwd(FusedArgs &a, int width, hipStream_t st) {
    switch (width) {
        case 32: return launch_fwd_w<32, Pol, RENDER>(a, st);
        case 64: return launch_fwd_w<64, Pol, RENDER>(a, st);
        case 128:
            if constexpr (Pol::ELEM_BYTES == 2) {
                if (bhn_fwd_w12(BHN_BF16, 128, a.depth, a.n_groups)) return launch_fwd_w<128, PolBF16X, RENDER>(a, st);
            }
            return launch_fwd_w<128, Pol, RENDER>(a, st);
        case 256:
            return launch_fwd_w<256, Pol, RENDER>(a, st);
        default:
            bhn_set_error("net_width %d: fused kernels are built for 32, 64, 128, 256", width);
            return BHN_EUNSUPPORTED;
    }
}

#ifdef BHN_DEBUG
// Measurement build only (make debug -> libbhnerf_hip_dbg.so, include/bhnerf_hip_debug.h): low 4 bits 1 = production
// kernel (default), 3 = ablation build of the 4x256 render kernel with the flags of fused_fwd_kernel<DBG> in bits 4.. .
static thread_local int g_fwd_variant = 1;
extern "C" int bhn_debug_set_fwd_variant(int32_t v) {
    g_fwd_variant = v;
    return BHN_OK;
}
static void *g_dbg_buf = nullptr;
void *bhn_debug_buffer() {        // 4 KiB device scratch of the measurement builds (time stamps)
    if (!g_dbg_buf) {
        if (hipMalloc(&g_dbg_buf, 4096) != hipSuccess) return nullptr;
        (void)hipMemset(g_dbg_buf, 0, 4096);
    }
    return g_dbg_buf;
}
extern "C" int bhn_debug_read(void *dst_host, size_t bytes) {
    BHN_CHECK_ARG(dst_host && bytes <= 4096, "bad debug read");
    BHN_CHECK_ARG(g_dbg_buf, "no ablation launch has run");
    BHN_HIP(hipMemcpy(dst_host, g_dbg_buf, bytes, hipMemcpyDeviceToHost));
    return BHN_OK;
}
#endif

extern "C" int bhn_predict_fwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                               const bhn_frames *fr, float *emission, void *stream) {
    FusedArgs a;
    MlpShape s;
    BHN_CHECK_ARG(emission, "null emission");
    mode = bhn_norm_mode(mode);
    if (m && bhn_mlp_shape(m, &s) == BHN_OK && s.general) return gen_forward(false, m, mode, packed, geom, fr, emission, (hipStream_t)stream);
    const int nw = fwd_tile_groups(m, mode, geom);
    int rc = fused_fill_args(m, mode, packed, geom, fr, false, &a, &s, nw);
    if (rc != BHN_OK) return rc;
    a.emission = emission;
    if (geom->groups)   // points of skipped groups are outside the domain: emission 0
        BHN_HIP(hipMemsetAsync(emission, 0, sizeof(float) * (size_t)a.B * a.P, (hipStream_t)stream));
    return mode == BHN_BF16 ? launch_fwd<PolBF16, false>(a, s.width, (hipStream_t)stream)
                            : launch_fwd<PolF32, false>(a, s.width, (hipStream_t)stream);
}

extern "C" int bhn_render_fwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                              const bhn_frames *fr, float *images, void *stream) {
    FusedArgs a;
    MlpShape s;
    BHN_CHECK_ARG(images, "null images");
    mode = bhn_norm_mode(mode);
    if (m && bhn_mlp_shape(m, &s) == BHN_OK && s.general) return gen_forward(true, m, mode, packed, geom, fr, images, (hipStream_t)stream);
    const int nw = fwd_tile_groups(m, mode, geom);
    int rc = fused_fill_args(m, mode, packed, geom, fr, true, &a, &s, nw);
    if (rc != BHN_OK) return rc;
    a.images = images;
    BHN_HIP(hipMemsetAsync(images, 0, sizeof(float) * (size_t)a.B * a.Sx * a.R, (hipStream_t)stream));
#ifdef BHN_DEBUG
    a.debug = (g_fwd_variant >> 4) & 64;                                       // bit 64: no phase lag (A/B measurements)
    if (mode == BHN_BF16 && s.width == 256 && (g_fwd_variant & 15) == 3) {     // ablation build, see fused_fwd_kernel
        a.debug = g_fwd_variant >> 4;                                          // includes bit 64
        a.emission = reinterpret_cast<float *>(bhn_debug_buffer());
        BHN_CHECK_ARG(a.emission, "no debug buffer");
        return launch_fwd_w<256, PolBF16, true, true>(a, (hipStream_t)stream);
    }
    if (mode == BHN_BF16 && s.width == 128 && (g_fwd_variant & 15) == 3) {     // round 6: the same ablation flags on the resident-weights kernel of width 128
        a.debug = g_fwd_variant >> 4;
        a.emission = reinterpret_cast<float *>(bhn_debug_buffer());
        BHN_CHECK_ARG(a.emission, "no debug buffer");
        return nw == PolBF16X::NWAVES ? launch_fwd_w<128, PolBF16X, true, true, true>(a, (hipStream_t)stream)
                                      : launch_fwd_w<128, PolBF16, true, true, true>(a, (hipStream_t)stream);
    }
#endif
    return mode == BHN_BF16 ? launch_fwd<PolBF16, true>(a, s.width, (hipStream_t)stream)
                            : launch_fwd<PolF32, true>(a, s.width, (hipStream_t)stream);
}

// NeRF_Predictor shapes the fused kernels are not built for: posenc_deg 5..10 (network.py:98-122; 3 + 6 deg <= 63 encoded
// features) and net_width 257..512 (network.py:154).  No reference driver sets either, so this path is written for
// completeness, not for the roofline: f32 arithmetic in BOTH modes (v_mfma_f32_32x32x2_f32; BHN_BF16 is accepted and
// computed in f32), activations of a 32-point group in LDS, weights streamed from L2 by every workgroup, and a backward
// that goes through an HBM tape:
//
//   gen_mlp_kernel<PREDICT | RENDER>   one workgroup = a TILE of eight consecutive 32-point groups, one group at a time: warp +
//                                      posenc (emission.py:200-210, network.py:118-122) -> LDS, layers as [feature][point] images
//                                      in two LDS buffers (the waves share the 32-row output tiles of a layer), sigmoid / masks,
//                                      emission or ray sums -- the eight groups' ray segments are combined in LDS (RaySum<8>), so a
//                                      pixel receives one float atomic per tile its ray touches: at most two for rays of <= 257
//                                      samples, bitwise reproducible like the fused kernels (round 5: one atomic per group);
//   gen_mlp_kernel<RECORD>             bhn_render_fwd_train: RENDER + every layer input, the encoded inputs and e on the tape (round 6:
//                                      the training step no longer runs its forward twice);
//   gen_mlp_kernel<CHAIN_TAPE>         bhn_render_bwd_tape: dout and the delta chain gA_{l-1} = relu' (.) K_l gA_l (transposed weight
//                                      copies) from that tape, writing every gA_l;
//   gen_mlp_kernel<CHAIN>              bhn_render_bwd (any workspace from 16 groups of tape on): both, chunk by chunk;
//   gen_dw_kernel                      dK_l = in_l^T gA_l over the tape, K = points: workgroup (job, split) = one 32-row x
//                                      128-column block of one layer over one share of the chunk's groups, into its own slab
//                                      (the bias is the job whose input tile is a row of ones);
//   gen_reduce_kernel                  slabs summed in split order -> flat gradient (deterministic).
//
// Packed image (bhn_pack_weights, all f32): per layer K_l zero-padded to [hrows | erows][outp] (hrows = width padded to 32
// for l > 0, erows = encoded inputs padded to 32 for layer 0 and the skip layers, network.py:59-61; outp = padded width, 32
// for the output layer whose column 0 is real), its transposed hidden part [outp][hrows] for the delta chain, its bias.
#include <algorithm>
#include "fused_common.h"

namespace {

enum { GEN_PREDICT = 0, GEN_RENDER = 1, GEN_CHAIN = 2, GEN_RECORD = 3, GEN_CHAIN_TAPE = 4 };
constexpr int GEN_TG = 8;             // 32-point groups per workgroup tile (the unit of the ray-sum combine and of the tape chunks)

struct GenLayer {
    unsigned k_off, kt_off, b_off;     // float offsets in the packed image
    unsigned slab_off;                 // float offset of this layer's [(hrows + erows + 32) x outp] block in a gradient slab
    int hrows, erows, outp;
    int h_true, e_true, out_true;      // the flat parameters: kernel ((h_true + e_true) x out_true) at pk_off, bias at pb_off
    long long pk_off, pb_off;
};

// bf16 mode (round 6): byte offsets of the bf16 fragment images behind the f32 image, and of the parts of one group's tape
struct Gen16 {
    unsigned f_off[BHN_MAX_LAYERS];    // forward A fragments of layer l: [output tile m][k-step over (hidden | encoded) inputs][1 KiB]
    unsigned t_off[BHN_MAX_LAYERS];    // transposed A fragments (delta chain through layer l, 1 <= l < D): [input tile][k-step over outputs][1 KiB]
    int ksf[BHN_MAX_LAYERS];           // k-steps of the forward image of layer l: (hrows + erows) / 16
    unsigned image_off, image_bytes;   // the bf16 images inside the packed buffer
    // tape of one 32-point group: encT | hT_1 .. hT_D | gaT_0 .. gaT_{D-1} | relu bits of h_1 .. h_D | e | dout; "T" tiles are 2 KiB with
    // the FEATURE on the lane and 16 points in the two fragments: the operand layout of the weight-gradient GEMMs (K = points)
    long long encT, hT, gaT, bits, e, dout, bytes;
};

struct GenArgs {
    FusedArgs f;
    Gen16 g16;
    int bf16;                          // 1: the bf16 kernels (mode BHN_BF16)
    char *tape16;
    GenLayer L[BHN_MAX_LAYERS];
    int D, Wp, Ep, F;
    const float *pk;                   // packed image
    float *tape;                       // [group of this chunk][tape_tile floats]: enc | h_1 .. h_D | gA_0 .. gA_{D-1} | gA_D (32 rows) | e (32)
    long long tape_tile;
    long long tile0, ntiles;           // the chunk, in tiles of GEN_TG groups
    const float *dimages;
    float *slabs;
    long long slab_floats;
    int nsplit, njobs, accumulate;
    float *dparams;
    long long nparams;
    const float *params;               // pack
    float *packed_out;
    long long packed_floats;
};

DEVI long long tape_h(const GenArgs &A, int l) { return (long long)A.Ep * 32 + (long long)(l - 1) * A.Wp * 32; }        // l = 1..D
DEVI long long tape_ga(const GenArgs &A, int l) { return (long long)A.Ep * 32 + (long long)(A.D + l) * A.Wp * 32; }     // l = 0..D
DEVI long long tape_e(const GenArgs &A) { return A.tape_tile - 32; }

// acc (features 32 m .. 32 m + 31 x the 32 points) += sum_k K[k][32 m + i] in[k][point]: K rows `rows` (a multiple of 32)
// of `stride` floats, `in` an LDS image [row][32].  Lane (i, h) feeds k = 2 t + h of every pair.
DEVI f32x16 gen_rows(f32x16 acc, const float *__restrict__ K, int stride, int m, const float *in, int rows, int lane) {
    // 16 rows (8 MFMAs) per iteration, the next iteration's 8 weight loads issued in front of them (32 rows / 16 loads in flight:
    // measured no faster -- the kernel is bound by the f32 matrix pipe, 0.5-0.6 of its peak, not by the L2 latency)
    constexpr int NJ = 8;
    const int i = lane & 31, h = lane >> 5;
    const float *Kc = K + 32 * m + i + (long long)h * stride;
    const float *ic = in + 32 * h + i;
    float an[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) an[j] = Kc[(long long)(2 * j) * stride];
    for (int k0 = 0; k0 < rows; k0 += 2 * NJ) {
        float ac[NJ], bc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { ac[j] = an[j]; bc[j] = ic[(k0 + 2 * j) * 32]; }
        if (k0 + 2 * NJ < rows) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) an[j] = Kc[(long long)(k0 + 2 * NJ + 2 * j) * stride];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[j], bc[j], acc, 0, 0, 0);
    }
    return acc;
}

// accumulator element r of lane (i, h) is feature 32 m + gen_row(r, h) of point i
DEVI int gen_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int MODE>
__global__ __launch_bounds__(512) void gen_mlp_kernel(GenArgs A) {
    constexpr bool FWD = MODE != GEN_CHAIN_TAPE;                             // runs the forward
    constexpr bool REC = MODE == GEN_RECORD || MODE == GEN_CHAIN;            // writes the layer inputs to the tape
    constexpr bool IMG = MODE == GEN_RENDER || MODE == GEN_RECORD;           // adds the ray sums to the images
    constexpr bool BWD = MODE == GEN_CHAIN || MODE == GEN_CHAIN_TAPE;        // dout and the delta chain
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedArgs &a = A.f;
    const int Wp = A.Wp, Ep = A.Ep, D = A.D, MTp = Wp >> 5;
    float *buf0 = reinterpret_cast<float *>(smem), *buf1 = buf0 + Wp * 32, *encS = buf1 + Wp * 32;
    float *red = encS + Ep * 32;                       // [16][32] partial sums of the output layer
    float *doutv = red + 16 * 32;                      // [32]
    int *livev = reinterpret_cast<int *>(doutv + 32);  // [32]
    char *seg = reinterpret_cast<char *>(livev + 32);  // RaySum<GEN_TG> scratch
    const int tid = threadIdx.x, lane = tid & 63, pl = tid & 31, part = tid >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = (int)(blockDim.x >> 6), NP = (int)(blockDim.x >> 5), NT = (int)blockDim.x;
    const int li = lane & 31, lh = lane >> 5;

    for (long long tile = A.tile0 + blockIdx.x; tile < A.tile0 + A.ntiles; tile += gridDim.x) {
      int tile_b = 0;
      for (int g = 0; g < GEN_TG; ++g) {
        float *tp = (REC || BWD) ? A.tape + ((tile - A.tile0) * GEN_TG + g) * A.tape_tile : nullptr;
        // ---- velocity warp + positional encoding of the group's 32 points (every 32-thread part holds all of them) ----
        const PointIn q = load_point<GEN_TG>(a, tile, g, pl);
        tile_b = q.b;
        float *in = buf0, *out = buf1;
        float e = 0.f;
        if constexpr (FWD) {
        bool live;
        {
            const double tM = q.tM0d + (double)q.tg;                       // emission.py:200-201 (t_M in double)
            const bool pre = tM < 0.0;                                     // emission.py:204-205
            const double rev_d = tM * (double)q.om * 0.15915494309189535;
            const double fr = rev_d - floor(rev_d);
            float s, c;
            sincosf((float)(fr * 6.283185307179586), &s, &c);
            float u[3] = {c * q.x + s * q.y, c * q.y - s * q.x, q.z};      // rot_z(-theta), utils.py:126-132
            const bool finite_theta = !pre && (fr == fr);
            bool valid0 = false;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool v = finite_theta && isfinite(u[k]);              // network.py:226
                if (k == 0) valid0 = v;
                u[k] = v ? u[k] / a.scale : 0.f;                            // network.py:227, 229
            }
            live = q.inb && q.dom && valid0;
            // features in the reference's order [u | sin(2^i u_k) at 3 + 3 i + k | cos(...) at 3 + 3 deg + 3 i + k]
            for (int i = part; i < a.deg; i += NP)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float sv, cv;
                    sincosf(u[k] * (float)(1 << i), &sv, &cv);
                    encS[(3 + 3 * i + k) * 32 + pl] = sv;
                    encS[(3 + 3 * a.deg + 3 * i + k) * 32 + pl] = cv;
                }
            if (part == NP - 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k) encS[k * 32 + pl] = u[k];
                livev[pl] = live ? 1 : 0;
            }
            for (int r = A.F + part; r < Ep; r += NP) encS[r * 32 + pl] = 0.f;
        }
        __syncthreads();
        if (REC)
            for (int idx = tid; idx < Ep * 32; idx += NT) tp[idx] = encS[idx];
        // ---- hidden layers ----
        for (int l = 0; l < D; ++l) {
            const GenLayer Lr = A.L[l];
            for (int m = wv; m < MTp; m += nwv) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = A.pk[Lr.b_off + 32 * m + gen_row(r, lh)];
                if (Lr.hrows) acc = gen_rows(acc, A.pk + Lr.k_off, Lr.outp, m, in, Lr.hrows, lane);
                if (Lr.erows) acc = gen_rows(acc, A.pk + Lr.k_off + (long long)Lr.hrows * Lr.outp, Lr.outp, m, encS, Lr.erows, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[r] > 0.f ? acc[r] : 0.f;
                    const int o = (32 * m + gen_row(r, lh)) * 32 + li;
                    out[o] = v;
                    if (REC) tp[tape_h(A, l + 1) + o] = v;
                }
            }
            __syncthreads();
            float *t = in; in = out; out = t;
        }
        // ---- output layer (column 0 of its [rows][32] image), sigmoid(. - 10), masks ----
        {
            const GenLayer Lo = A.L[D];
            const float *ko = A.pk + Lo.k_off;
            float s = 0.f;
            for (int k = part; k < Lo.hrows; k += NP) s += ko[(long long)k * 32] * in[k * 32 + pl];
            for (int k = part; k < Lo.erows; k += NP) s += ko[(long long)(Lo.hrows + k) * 32] * encS[k * 32 + pl];
            red[part * 32 + pl] = s;
            __syncthreads();
            if (tid < 32) {
                float o = A.pk[Lo.b_off];
                for (int p2 = 0; p2 < NP; ++p2) o += red[p2 * 32 + pl];
                if (live) e = 1.f / (1.f + expf(10.f - o));                      // network.py:231-233
                if (REC) tp[tape_e(A) + pl] = e;
            }
        }
        } else {
            // ---- bhn_render_bwd_tape: e and h_D as the forward recorded them ----
            if (tid < 32) e = tp[tape_e(A) + pl];
            for (int idx = tid; idx < Wp * 32; idx += NT) in[idx] = tp[tape_h(A, D) + idx];
            __syncthreads();
        }
        if (MODE == GEN_PREDICT) {
            if (tid < 32 && q.inb) a.emission[(long long)q.b * a.P + q.p] = e;
        }
        if constexpr (IMG) {
            // the group's ray segments: straight to the pixels where that is order-independent already (ray_direct), else to the
            // tile's scratch (combined behind the eighth group)
            if (wv == 0) RaySum<GEN_TG>::put(a, seg, g, q.p, q.inb, e, 0.f, false, q.b);
        }
        if constexpr (BWD) {
            const GenLayer Lo = A.L[D];
            // ---- dout = d loss / d (pre-sigmoid output): dE e (1 - e), dE = sum_s dimages[b, s, ray] w[s, p] ----
            if (tid < 32) {
                float d = 0.f;
                if (q.inb && e != 0.f) {
                    const long long ray = a.ray_idx ? (long long)a.ray_idx[q.p] : (long long)a.fd_G.div((unsigned)q.p);
                    float dE = 0.f;
                    for (int s = 0; s < a.Sx; ++s) dE += A.dimages[((long long)q.b * a.Sx + s) * a.R + ray] * a.w[(long long)s * a.P + q.p];
                    d = dE * e * (1.f - e);
                }
                doutv[pl] = d;
            }
            __syncthreads();
            // gA_D (32 rows, row 0 real) and gA_{D-1} = relu'(h_D) (.) K_out dout
            for (int idx = tid; idx < 1024; idx += NT) tp[tape_ga(A, D) + idx] = idx < 32 ? doutv[idx] : 0.f;
            {
                const float *ko = A.pk + Lo.k_off;
                for (int idx = tid; idx < Wp * 32; idx += NT) {
                    const float gv = in[idx] > 0.f ? ko[(long long)(idx >> 5) * 32] * doutv[idx & 31] : 0.f;
                    out[idx] = gv;
                    tp[tape_ga(A, D - 1) + idx] = gv;
                }
            }
            __syncthreads();
            { float *t = in; in = out; out = t; }                            // `in` = gA_l from here on
            // ---- delta chain: gA_{l-1} = relu'(h_l) (.) K_l[hidden rows] gA_l, l = D-1 .. 1 (h_l read back from the tape) ----
            for (int l = D - 1; l >= 1; --l) {
                const GenLayer Lr = A.L[l];
                for (int m = wv; m < MTp; m += nwv) {
                    f32x16 acc = {};
                    acc = gen_rows(acc, A.pk + Lr.kt_off, Lr.hrows, m, in, Lr.outp, lane);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = (32 * m + gen_row(r, lh)) * 32 + li;
                        const float gv = tp[tape_h(A, l) + o] > 0.f ? acc[r] : 0.f;
                        out[o] = gv;
                        tp[tape_ga(A, l - 1) + o] = gv;
                    }
                }
                __syncthreads();
                float *t = in; in = out; out = t;
            }
        }
        __syncthreads();
      }
      if constexpr (IMG) {
          // the tile's ray segments -> pixels: every ray of the tile gets ONE atomic, its segments added in group order
          if (!a.ray_direct) {
              for (int vw = wv; vw < GEN_TG; vw += nwv) RaySum<GEN_TG>::combine(a, seg, vw, tile_b);
              __syncthreads();
          }
      }
    }
}

// dK_l[in rows 32 mi ..][out columns 128 strip ..] += sum over the split's groups of in_l^T gA_l (the bias: in = a row of ones)
__global__ __launch_bounds__(64) void gen_dw_kernel(GenArgs A) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    // job -> (layer, input tile, output strip)
    int job = blockIdx.x, l = 0, mi = 0, strip = 0;
    for (l = 0; l <= A.D; ++l) {
        const int nin = ((A.L[l].hrows + A.L[l].erows) >> 5) + 1, nst = ((A.L[l].outp >> 5) + 3) >> 2;
        if (job < nin * nst) { mi = job / nst; strip = job - mi * nst; break; }
        job -= nin * nst;
    }
    if (l > A.D) return;
    const GenLayer Lr = A.L[l];
    const int nt = min(4, (Lr.outp >> 5) - 4 * strip);                     // 32-column tiles of this strip
    const int htiles = Lr.hrows >> 5, etiles = Lr.erows >> 5;
    const bool bias_job = mi == htiles + etiles;
    // this lane's 16 points of a group: 16 h .. 16 h + 15 (any pairing of points with k-steps serves a sum over points)
    long long a_off = 0;
    if (mi < htiles) a_off = tape_h(A, l) + (long long)(32 * mi + i) * 32 + 16 * h;
    else if (!bias_job) a_off = (long long)(32 * (mi - htiles) + i) * 32 + 16 * h;
    const long long b_off = tape_ga(A, l) + (long long)(128 * strip + i) * 32 + 16 * h;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const f32x16 z = {}; acc[j] = z; }
    const long long ngr = A.ntiles * GEN_TG;                              // groups of the chunk
    const long long t0 = ngr * blockIdx.y / A.nsplit, t1 = ngr * (blockIdx.y + 1) / A.nsplit;
    for (long long t = t0; t < t1; ++t) {
        const float *tp = A.tape + t * A.tape_tile;
        f32x4 av[4], bv[4][4];
        if (bias_job) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float one = i == 0 ? 1.f : 0.f; av[k] = f32x4{one, one, one, one}; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) av[k] = *reinterpret_cast<const f32x4 *>(tp + a_off + 4 * k);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nt) {
#pragma unroll
                for (int k = 0; k < 4; ++k) bv[j][k] = *reinterpret_cast<const f32x4 *>(tp + b_off + (long long)j * 1024 + 4 * k);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nt) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k >> 2][k & 3], bv[j][k >> 2][k & 3], acc[j], 0, 0, 0);
            }
    }
    float *slab = A.slabs + (long long)blockIdx.y * A.slab_floats + Lr.slab_off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *d = slab + (long long)(32 * mi + gen_row(r, h)) * Lr.outp + 128 * strip + 32 * j + i;
                *d = A.accumulate ? *d + acc[j][r] : acc[j][r];
            }
        }
}

__global__ void gen_reduce_kernel(GenArgs A) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < A.nparams; idx += (long long)gridDim.x * blockDim.x) {
        int l = 0;
        while (l < A.D && idx >= A.L[l + 1].pk_off) ++l;
        const GenLayer Lr = A.L[l];
        long long row, o;
        if (idx >= Lr.pb_off) { row = Lr.hrows + Lr.erows; o = idx - Lr.pb_off; }
        else {
            const long long r = (idx - Lr.pk_off) / Lr.out_true;
            o = (idx - Lr.pk_off) - r * Lr.out_true;
            row = r < Lr.h_true ? r : Lr.hrows + (r - Lr.h_true);
        }
        const float *s = A.slabs + Lr.slab_off + row * Lr.outp + o;
        float sum = 0.f;
        for (int k = 0; k < A.nsplit; ++k) sum += s[(long long)k * A.slab_floats];
        A.dparams[idx] = sum;
    }
}

__global__ void gen_pack_fill_kernel(GenArgs A) {
    const long long gtid = blockIdx.x * (long long)blockDim.x + threadIdx.x, gn = (long long)gridDim.x * blockDim.x;
    for (int l = 0; l <= A.D; ++l) {
        const GenLayer Lr = A.L[l];
        const int rows = Lr.hrows + Lr.erows;
        const long long nk = (long long)rows * Lr.outp;
        for (long long idx = gtid; idx < nk; idx += gn) {
            const int r = (int)(idx / Lr.outp), o = (int)(idx - (long long)r * Lr.outp);
            long long fr = -1;
            if (r < Lr.hrows) { if (r < Lr.h_true) fr = r; }
            else if (r - Lr.hrows < Lr.e_true) fr = Lr.h_true + (r - Lr.hrows);
            A.packed_out[Lr.k_off + idx] = (fr >= 0 && o < Lr.out_true) ? A.params[Lr.pk_off + fr * Lr.out_true + o] : 0.f;
        }
        if (l >= 1 && l < A.D) {
            const long long nt = (long long)Lr.outp * Lr.hrows;
            for (long long idx = gtid; idx < nt; idx += gn) {
                const int o = (int)(idx / Lr.hrows), k = (int)(idx - (long long)o * Lr.hrows);
                A.packed_out[Lr.kt_off + idx] = (k < Lr.h_true && o < Lr.out_true) ? A.params[Lr.pk_off + (long long)k * Lr.out_true + o] : 0.f;
            }
        }
        for (long long o = gtid; o < Lr.outp; o += gn) A.packed_out[Lr.b_off + o] = o < Lr.out_true ? A.params[Lr.pb_off + o] : 0.f;
    }
}

// =============================================================================================================================
// bf16 mode (BHN_BF16) of the general path, round 6: v_mfma_f32_32x32x16_bf16 with every operand in MFMA fragment order.
//   * weights: bf16 fragment images (gen_pack16_kernel), the A operand straight from L2 / L1 -- one 16-byte load per lane and
//     MFMA, shared by the NG groups a workgroup runs side by side (NG accumulator tiles per wave);
//   * activations: per group two LDS images of B fragments ([k-step][lane][8 bf16]); as in the fused kernels the accumulator of
//     an output tile IS the pair of B fragments of the next layer's k-steps 2 m, 2 m + 1 (fused_common.h), so a finished tile is
//     relu'd, rounded and written back with two ds_write_b128;
//   * tape: what the weight-gradient GEMMs (K = points) read -- layer inputs and pre-activation gradients as 2-KiB "T" tiles
//     (feature on the lane, 16 points in the two fragments), transposed on the way out by two MFMAs against identity fragments
//     (exact: one product by 1.0 per element) -- plus the relu bits, e and dout.  gen_dw16_kernel loads its operands from it with
//     16-byte global loads: no LDS, no conversions.  2.2 KB per point at 4x128 / degree 5 (f32 tape: 4.4 KB).
// Same tiles of eight groups, the same modes, slabs and reduce as the f32 kernels above.
// =============================================================================================================================
typedef __bf16 g16frag __attribute__((ext_vector_type(8)));

DEVI g16frag g16_ident(int s, int lane) {          // identity k-step s: element j of lane (n, h) is 1 where 16 s + phi16(h, j) == n
    const int n = lane & 31, h = lane >> 5;
    g16frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (phi16(h, j) + 16 * s == n) ? (__bf16)1.f : (__bf16)0.f;
    return f;
}
// a 32-feature block held as two B fragments (point on the lane) -> its T tile in global memory (feature on the lane; register r of
// the transposed accumulator is point (r & 3) + 8 (r >> 2) + 4 (lane >> 5): fragment s element j = point (j & 3) + 8 (j >> 2) + 16 s + 4 h)
DEVI void g16_store_T(char *tile, const g16frag &f0, const g16frag &f1, const g16frag (&id)[2], int lane) {
    f32x16 t = {};
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, id[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, id[1], t, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        g16frag o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)t[8 * s + j];
        *reinterpret_cast<g16frag *>(tile + s * 1024 + lane * 16) = o;
    }
}
// the point a T-tile fragment element stands for
DEVI int g16_point(int s, int j, int h) { return (j & 3) + 8 * (j >> 2) + 16 * s + 4 * h; }

__global__ void gen_pack16_kernel(GenArgs A) {
    // from the f32 image gen_pack_fill_kernel has just written (zero-padded [rows][outp] per layer)
    const long long gtid = blockIdx.x * (long long)blockDim.x + threadIdx.x, gn = (long long)gridDim.x * blockDim.x;
    __bf16 *img = reinterpret_cast<__bf16 *>(reinterpret_cast<char *>(A.packed_out) + A.g16.image_off);
    for (int l = 0; l <= A.D; ++l) {
        const GenLayer Lr = A.L[l];
        const int ksf = A.g16.ksf[l];
        const long long nf = (long long)(Lr.outp / 32) * ksf * 512;
        for (long long idx = gtid; idx < nf; idx += gn) {
            const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
            const long long frag = idx >> 9;
            const int m = (int)(frag / ksf), ks = (int)(frag - (long long)m * ksf);
            const int k = 16 * ks + phi16(lane >> 5, j), o = 32 * m + (lane & 31);
            img[A.g16.f_off[l] / 2 + idx] = (__bf16)A.packed_out[Lr.k_off + (long long)k * Lr.outp + o];
        }
        if (l >= 1 && l < A.D) {
            const int kst = Lr.outp / 16;
            const long long nt = (long long)(Lr.hrows / 32) * kst * 512;
            for (long long idx = gtid; idx < nt; idx += gn) {
                const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
                const long long frag = idx >> 9;
                const int m = (int)(frag / kst), ks = (int)(frag - (long long)m * kst);
                const int kin = 32 * m + (lane & 31), o = 16 * ks + phi16(lane >> 5, j);
                img[A.g16.t_off[l] / 2 + idx] = (__bf16)A.packed_out[Lr.k_off + (long long)kin * Lr.outp + o];
            }
        }
    }
}

// acc[g] += sum over `nks` k-steps of A fragment (global image, 1 KiB apart) x B fragment of group g (LDS image `bbase + g * slot`)
template <int NG>
DEVI void g16_ksteps(f32x16 (&acc)[NG], const char *afrag, const char *bbase, int slot, int nks, int lane) {
    const char *ap = afrag + lane * 16;
    const char *bp = bbase + lane * 16;
    g16frag an = *reinterpret_cast<const g16frag *>(ap);
    for (int ks = 0; ks < nks; ++ks) {
        const g16frag a = an;
        if (ks + 1 < nks) an = *reinterpret_cast<const g16frag *>(ap + (long long)(ks + 1) * 1024);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const g16frag b = *reinterpret_cast<const g16frag *>(bp + g * slot + ks * 1024);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g], 0, 0, 0);
        }
    }
}

template <int MODE, int NG>
__global__ __launch_bounds__(512) void gen_mlp16_kernel(GenArgs A) {
    constexpr bool FWD = MODE != GEN_CHAIN_TAPE, REC = MODE == GEN_RECORD || MODE == GEN_CHAIN;
    constexpr bool IMG = MODE == GEN_RENDER || MODE == GEN_RECORD, BWD = MODE == GEN_CHAIN || MODE == GEN_CHAIN_TAPE;
    static_assert(GEN_TG % NG == 0, "a tile is a whole number of rounds");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedArgs &a = A.f;
    const Gen16 &G = A.g16;
    const int Wp = A.Wp, Ep = A.Ep, D = A.D, MTp = Wp >> 5;
    const int BUF = Wp * 64, SLOT = 2 * BUF + Ep * 64;                    // per group: two activation images + the encoded inputs (B fragments)
    float *ev = reinterpret_cast<float *>(smem + NG * SLOT);              // [NG][32]
    float *doutv = ev + NG * 32;
    int *livev = reinterpret_cast<int *>(doutv + NG * 32);
    char *seg = reinterpret_cast<char *>(livev + NG * 32);                // RaySum<GEN_TG> scratch
    const int tid = threadIdx.x, lane = tid & 63, pl = tid & 31, part = tid >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = (int)(blockDim.x >> 6), NP = (int)(blockDim.x >> 5), NT = (int)blockDim.x;
    const int li = lane & 31, lh = lane >> 5;
    const char *pk16 = reinterpret_cast<const char *>(A.pk) + G.image_off;
    g16frag id[2] = {g16_ident(0, lane), g16_ident(1, lane)};
    // the unused slots of the encoded-input images stay zero for the whole launch
    for (int idx = tid; idx < NG * Ep * 16; idx += NT) {
        const int g = idx / (Ep * 16), w = idx - g * (Ep * 16);
        reinterpret_cast<unsigned *>(smem + g * SLOT + 2 * BUF)[w] = 0u;
    }
    __syncthreads();
    auto enc_put = [&](char *encb, int q, int pt, float v) {             // feature q of point pt -> its place in the B fragments
        const int s = q >> 4, c = q & 15, h = (c >> 2) & 1, j = 4 * (c >> 3) + (c & 3);
        *reinterpret_cast<__bf16 *>(encb + s * 1024 + (32 * h + pt) * 16 + 2 * j) = (__bf16)v;
    };

    for (long long tile = A.tile0 + blockIdx.x; tile < A.tile0 + A.ntiles; tile += gridDim.x) {
      for (int g0 = 0; g0 < GEN_TG; g0 += NG) {
        char *tp0 = (REC || BWD) ? A.tape16 + ((tile - A.tile0) * GEN_TG + g0) * G.bytes : nullptr;      // group g0 + g: + g * G.bytes
        int cur = 0;                                                       // which activation image holds the running layer's input
        if constexpr (FWD) {
            // ---- velocity warp + positional encoding: 32-thread part (gsel, sub) does group gsel's octaves sub, sub + nsub, ... ----
            {
                const int gsel = part % NG, sub = part / NG, nsub = NP / NG;
                const PointIn q = load_point<GEN_TG>(a, tile, g0 + gsel, pl);
                const double tM = q.tM0d + (double)q.tg;                   // emission.py:200-201 (t_M in double)
                const bool pre = tM < 0.0;                                 // emission.py:204-205
                const double rev_d = tM * (double)q.om * 0.15915494309189535;
                const double fr = rev_d - floor(rev_d);
                const float s0 = __builtin_amdgcn_sinf((float)fr), c0 = __builtin_amdgcn_cosf((float)fr);
                float u[3] = {c0 * q.x + s0 * q.y, c0 * q.y - s0 * q.x, q.z};    // rot_z(-theta), utils.py:126-132
                const bool finite_theta = !pre && (fr == fr);
                bool valid0 = false;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool v = finite_theta && isfinite(u[k]);          // network.py:226
                    if (k == 0) valid0 = v;
                    u[k] = v ? u[k] * a.inv_scale : 0.f;                    // network.py:227, 229
                }
                char *encb = smem + gsel * SLOT + 2 * BUF;
                if (sub < nsub) {
                    for (int i = sub; i < a.deg; i += nsub)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float rev = __builtin_amdgcn_fractf(u[k] * (float)(1 << i) * 0.15915494309189535f);
                            enc_put(encb, 3 + 3 * i + k, pl, __builtin_amdgcn_sinf(rev));
                            enc_put(encb, 3 + 3 * a.deg + 3 * i + k, pl, __builtin_amdgcn_cosf(rev));
                        }
                    if (sub == 0) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) enc_put(encb, k, pl, u[k]);
                        livev[gsel * 32 + pl] = (q.inb && q.dom && valid0) ? 1 : 0;
                    }
                }
            }
            __syncthreads();
            if constexpr (REC) {                                            // the encoded inputs as T tiles (dW_0 and the skip layer)
                for (int it = wv; it < NG * (Ep >> 5); it += nwv) {
                    const int g = it % NG, eb = it / NG;
                    const char *encb = smem + g * SLOT + 2 * BUF;
                    const g16frag f0 = *reinterpret_cast<const g16frag *>(encb + (2 * eb) * 1024 + lane * 16);
                    const g16frag f1 = *reinterpret_cast<const g16frag *>(encb + (2 * eb + 1) * 1024 + lane * 16);
                    g16_store_T(tp0 + g * G.bytes + G.encT + eb * 2048, f0, f1, id, lane);
                }
            }
            // ---- hidden layers ----
            for (int l = 0; l < D; ++l) {
                const GenLayer Lr = A.L[l];
                const int hks = Lr.hrows >> 4, eks = Lr.erows >> 4;
                for (int m = wv; m < MTp; m += nwv) {
                    f32x16 acc[NG];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bv = A.pk[Lr.b_off + 32 * m + gen_row(r, lh)];
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[g][r] = bv;
                    }
                    const char *af = pk16 + G.f_off[l] + (long long)m * G.ksf[l] * 1024;
                    if (hks) g16_ksteps<NG>(acc, af, smem + cur * BUF, SLOT, hks, lane);
                    if (eks) g16_ksteps<NG>(acc, af + (long long)hks * 1024, smem + 2 * BUF, SLOT, eks, lane);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        g16frag f[2];
                        unsigned mask = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool on = acc[g][r] > 0.f;
                            f[r >> 3][r & 7] = (__bf16)(on ? acc[g][r] : 0.f);
                            mask |= (on ? 1u : 0u) << r;
                        }
                        char *ob = smem + g * SLOT + (cur ^ 1) * BUF;
                        *reinterpret_cast<g16frag *>(ob + (2 * m) * 1024 + lane * 16) = f[0];
                        *reinterpret_cast<g16frag *>(ob + (2 * m + 1) * 1024 + lane * 16) = f[1];
                        if constexpr (REC) {
                            char *tg = tp0 + g * G.bytes;
                            g16_store_T(tg + G.hT + ((long long)l * MTp + m) * 2048, f[0], f[1], id, lane);
                            *reinterpret_cast<unsigned *>(tg + G.bits + ((long long)l * MTp + m) * 256 + lane * 4) = mask;
                        }
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
            // ---- output layer: one tile whose row 0 is real; sigmoid(. - 10), masks (network.py:231-233) ----
            {
                const GenLayer Lo = A.L[D];
                const int hks = Lo.hrows >> 4, eks = Lo.erows >> 4;
                for (int g = wv; g < NG; g += nwv) {
                    f32x16 acc1[1];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
                    const char *af = pk16 + G.f_off[D];
                    g16_ksteps<1>(acc1, af, smem + g * SLOT + cur * BUF, 0, hks, lane);
                    if (eks) g16_ksteps<1>(acc1, af + (long long)hks * 1024, smem + g * SLOT + 2 * BUF, 0, eks, lane);
                    if (lh == 0) {
                        float e = 0.f;
                        if (livev[g * 32 + li]) e = 1.f / (1.f + __expf(10.f - (acc1[0][0] + A.pk[Lo.b_off])));
                        ev[g * 32 + li] = e;
                        if constexpr (REC) *reinterpret_cast<float *>(tp0 + g * G.bytes + G.e + li * 4) = e;
                    }
                }
            }
            __syncthreads();
        } else {
            if (tid < 32 * NG) ev[tid] = *reinterpret_cast<const float *>(tp0 + (tid >> 5) * G.bytes + G.e + pl * 4);
            __syncthreads();
        }
        // ---- emission / ray sums / dout: one 32-thread part per group ----
        if (tid < 32 * NG) {
            const int g = tid >> 5;
            const PointIn q = load_point<GEN_TG>(a, tile, g0 + g, pl);
            const float e = ev[g * 32 + pl];
            if (MODE == GEN_PREDICT) { if (q.inb) a.emission[(long long)q.b * a.P + q.p] = e; }
            if constexpr (BWD) {
                float d = 0.f;
                if (q.inb && e != 0.f) {
                    const long long ray = a.ray_idx ? (long long)a.ray_idx[q.p] : (long long)a.fd_G.div((unsigned)q.p);
                    float dE = 0.f;
                    for (int s = 0; s < a.Sx; ++s) dE += A.dimages[((long long)q.b * a.Sx + s) * a.R + ray] * a.w[(long long)s * a.P + q.p];
                    d = dE * e * (1.f - e);
                }
                doutv[g * 32 + pl] = d;
                *reinterpret_cast<float *>(tp0 + g * G.bytes + G.dout + pl * 4) = d;
            }
        }
        if constexpr (IMG) {
            if (wv == 0) {
                for (int g = 0; g < NG; ++g) {
                    const PointIn q = load_point<GEN_TG>(a, tile, g0 + g, li);
                    RaySum<GEN_TG>::put(a, seg, g0 + g, q.p, q.inb, lh == 0 ? ev[g * 32 + li] : 0.f, 0.f, false, q.b);
                }
            }
        }
        if constexpr (BWD) {
            __syncthreads();
            // ---- gA_{D-1} = relu'(h_D) (.) K_out dout as B fragments, from the recorded relu bits ----
            const GenLayer Lo = A.L[D];
            for (int m = wv; m < MTp; m += nwv) {
                float kf[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) kf[r] = A.pk[Lo.k_off + (long long)(32 * m + gen_row(r, lh)) * 32];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    char *tg = tp0 + g * G.bytes;
                    const unsigned mask = *reinterpret_cast<const unsigned *>(tg + G.bits + ((long long)(D - 1) * MTp + m) * 256 + lane * 4);
                    const float dv = doutv[g * 32 + li];
                    g16frag f[2];
#pragma unroll
                    for (int r = 0; r < 16; ++r) f[r >> 3][r & 7] = (__bf16)(((mask >> r) & 1u) ? kf[r] * dv : 0.f);
                    char *ob = smem + g * SLOT + (cur ^ 1) * BUF;
                    *reinterpret_cast<g16frag *>(ob + (2 * m) * 1024 + lane * 16) = f[0];
                    *reinterpret_cast<g16frag *>(ob + (2 * m + 1) * 1024 + lane * 16) = f[1];
                    g16_store_T(tg + G.gaT + ((long long)(D - 1) * MTp + m) * 2048, f[0], f[1], id, lane);
                }
            }
            __syncthreads();
            cur ^= 1;
            // ---- delta chain: gA_{l-1} = relu'(h_l) (.) K_l[hidden rows] gA_l, l = D-1 .. 1 ----
            for (int l = D - 1; l >= 1; --l) {
                const GenLayer Lr = A.L[l];
                const int kst = Lr.outp >> 4;
                for (int m = wv; m < MTp; m += nwv) {
                    f32x16 acc[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) { const f32x16 z = {}; acc[g] = z; }
                    g16_ksteps<NG>(acc, pk16 + G.t_off[l] + (long long)m * kst * 1024, smem + cur * BUF, SLOT, kst, lane);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        char *tg = tp0 + g * G.bytes;
                        const unsigned mask = *reinterpret_cast<const unsigned *>(tg + G.bits + ((long long)(l - 1) * MTp + m) * 256 + lane * 4);
                        g16frag f[2];
#pragma unroll
                        for (int r = 0; r < 16; ++r) f[r >> 3][r & 7] = (__bf16)(((mask >> r) & 1u) ? acc[g][r] : 0.f);
                        char *ob = smem + g * SLOT + (cur ^ 1) * BUF;
                        *reinterpret_cast<g16frag *>(ob + (2 * m) * 1024 + lane * 16) = f[0];
                        *reinterpret_cast<g16frag *>(ob + (2 * m + 1) * 1024 + lane * 16) = f[1];
                        g16_store_T(tg + G.gaT + ((long long)(l - 1) * MTp + m) * 2048, f[0], f[1], id, lane);
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
        }
        __syncthreads();
      }
      if constexpr (IMG) {
          if (!a.ray_direct) {
              for (int vw = wv; vw < GEN_TG; vw += nwv) RaySum<GEN_TG>::combine(a, seg, vw, (int)a.fd_tpf.div((unsigned)tile));
              __syncthreads();
          }
      }
    }
}

// dK_l[in rows 32 mi ..][out columns 128 strip ..] += sum over the split's groups of in_l^T gA_l, both operands T tiles of the tape
__global__ __launch_bounds__(64) void gen_dw16_kernel(GenArgs A) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const Gen16 &G = A.g16;
    const int MTp = A.Wp >> 5;
    int job = blockIdx.x, l = 0, mi = 0, strip = 0;
    for (l = 0; l <= A.D; ++l) {
        const int nin = ((A.L[l].hrows + A.L[l].erows) >> 5) + 1, nst = ((A.L[l].outp >> 5) + 3) >> 2;
        if (job < nin * nst) { mi = job / nst; strip = job - mi * nst; break; }
        job -= nin * nst;
    }
    if (l > A.D) return;
    const GenLayer Lr = A.L[l];
    const int nt = min(4, (Lr.outp >> 5) - 4 * strip);                     // 32-column tiles of this strip
    const int htiles = Lr.hrows >> 5, etiles = Lr.erows >> 5;
    const bool bias_job = mi == htiles + etiles, out_job = l == A.D;
    long long a_off = 0;                                                   // the input tile: h_l tile mi, or encoded-input tile mi - htiles
    if (mi < htiles) a_off = G.hT + ((long long)(l - 1) * MTp + mi) * 2048;
    else if (!bias_job) a_off = G.encT + (long long)(mi - htiles) * 2048;
    const long long b_off = out_job ? G.dout : G.gaT + ((long long)l * MTp + 4 * strip) * 2048;
    g16frag ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (i == 0) ? (__bf16)1.f : (__bf16)0.f;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const f32x16 z = {}; acc[j] = z; }
    const long long ngr = A.ntiles * GEN_TG;
    const long long t0 = ngr * blockIdx.y / A.nsplit, t1 = ngr * (blockIdx.y + 1) / A.nsplit;
    struct Ops { g16frag a[2], b[4][2]; };
    auto load = [&](long long t) {
        Ops o;
        const char *tp = A.tape16 + t * G.bytes;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            o.a[s] = bias_job ? ones : *reinterpret_cast<const g16frag *>(tp + a_off + s * 1024 + lane * 16);
            if (out_job) {                                                 // gA_D: column 0 = dout of the fragment's eight points
                const float *dv = reinterpret_cast<const float *>(tp + b_off);
#pragma unroll
                for (int j = 0; j < 8; ++j) o.b[0][s][j] = (i == 0) ? (__bf16)dv[g16_point(s, j, h)] : (__bf16)0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nt) o.b[j][s] = *reinterpret_cast<const g16frag *>(tp + b_off + (long long)j * 2048 + s * 1024 + lane * 16);
            }
        }
        return o;
    };
    if (t0 < t1) {
        Ops nx = load(t0);
        for (long long t = t0; t < t1; ++t) {
            const Ops c = nx;
            if (t + 1 < t1) nx = load(t + 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nt) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.a[s], c.b[j][s], acc[j], 0, 0, 0);
        }
    }
    float *slab = A.slabs + (long long)blockIdx.y * A.slab_floats + Lr.slab_off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *d = slab + (long long)(32 * mi + gen_row(r, h)) * Lr.outp + 128 * strip + 32 * j + i;
                *d = A.accumulate ? *d + acc[j][r] : acc[j][r];
            }
        }
}

// layer tables + sizes of the packed image / a gradient slab / a tape tile
void gen_layout(const MlpShape &s, GenArgs *A) {
    const int Wp = (s.width_true + 31) / 32 * 32, Ep = (s.F + 31) / 32 * 32;
    A->D = s.depth; A->Wp = Wp; A->Ep = Ep; A->F = s.F;
    unsigned off = 0, soff = 0;
    for (int l = 0; l <= s.depth; ++l) {
        GenLayer &L = A->L[l];
        L.hrows = l > 0 ? Wp : 0;
        L.erows = (l == 0 || s.skip_in[l]) ? Ep : 0;
        L.outp = l == s.depth ? 32 : Wp;
        L.h_true = l > 0 ? s.width_true : 0;
        L.e_true = (l == 0 || s.skip_in[l]) ? s.F : 0;
        L.out_true = l == s.depth ? 1 : s.width_true;
        L.pk_off = s.kernel_off[l]; L.pb_off = s.bias_off[l];
        L.k_off = off; off += (unsigned)(L.hrows + L.erows) * L.outp;
        L.kt_off = off; if (l >= 1 && l < s.depth) off += (unsigned)L.outp * L.hrows;
        L.b_off = off; off += L.outp;
        L.slab_off = soff; soff += (unsigned)(L.hrows + L.erows + 32) * L.outp;
    }
    A->packed_floats = off;
    A->slab_floats = soff;
    // gen_dw_kernel: one wave per (block of one layer's gradient, share of the chunk's groups).  Narrow networks have few blocks
    // (25 at 4x128): enough shares that ~6000 waves fill the 1024 SIMDs (8 shares left 4x128 at 200 waves: 68 % of its step)
    int njobs = 0;
    for (int l = 0; l <= s.depth; ++l) njobs += (((A->L[l].hrows + A->L[l].erows) >> 5) + 1) * (((A->L[l].outp >> 5) + 3) >> 2);
    A->njobs = njobs;
    A->nsplit = std::min(256, std::max(8, (6144 + njobs - 1) / njobs));
    A->tape_tile = (long long)Ep * 32 + 2ll * s.depth * Wp * 32 + 1024 + 32;
    A->nparams = s.nparams;
    // bf16 mode: fragment images (1 KiB per 32 x 16 fragment) and the per-group tape
    Gen16 &g = A->g16;
    unsigned o16 = 0;
    for (int l = 0; l <= s.depth; ++l) {
        const GenLayer &L = A->L[l];
        g.ksf[l] = (L.hrows + L.erows) / 16;
        g.f_off[l] = o16; o16 += (unsigned)(L.outp / 32) * g.ksf[l] * 1024u;
        g.t_off[l] = o16; if (l >= 1 && l < s.depth) o16 += (unsigned)(L.hrows / 32) * (L.outp / 16) * 1024u;
    }
    g.image_off = (unsigned)((A->packed_floats * 4 + 255) / 256 * 256);
    g.image_bytes = o16;
    const long long mt = Wp / 32;
    g.encT = 0; g.hT = (long long)(Ep / 32) * 2048; g.gaT = g.hT + s.depth * mt * 2048; g.bits = g.gaT + s.depth * mt * 2048;
    g.e = g.bits + s.depth * mt * 256; g.dout = g.e + 128; g.bytes = g.dout + 128;
}

size_t gen_lds_bytes(const GenArgs &A, int Sx) { return (size_t)(2 * A.Wp + A.Ep) * 32 * 4 + (16 * 32 + 64) * 4 + RaySum<GEN_TG>::bytes(Sx); }
int gen_block(const GenArgs &A) { const int mt = A.Wp / 32; return mt >= 8 ? 512 : mt >= 4 ? 256 : 128; }

// bf16 kernels: groups a workgroup runs side by side (4 when their LDS images fit, else 2) and the LDS bytes of that choice
size_t gen_lds16_bytes(const GenArgs &A, int Sx, int ng) { return (size_t)ng * (2 * A.Wp * 64 + A.Ep * 64) + (size_t)ng * 32 * 4 * 3 + RaySum<GEN_TG>::bytes(Sx); }
int gen_ng16(const GenArgs &A, int Sx) { return gen_lds16_bytes(A, Sx, 4) <= 160 * 1024 ? 4 : 2; }

template <int MODE, int NG>
int gen_launch_mlp16(const GenArgs &A, int grid, hipStream_t st) {
    const size_t lds = gen_lds16_bytes(A, A.f.Sx, NG);
    BHN_CHECK_ARG(lds <= 160 * 1024, "internal: general bf16 path needs %zu bytes of LDS", lds);
    static DeviceOnce once;
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    BHN_HIP(once.run(dev, [&](int &) { return hipFuncSetAttribute((const void *)gen_mlp16_kernel<MODE, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
    hipLaunchKernelGGL((gen_mlp16_kernel<MODE, NG>), dim3(grid), dim3(gen_block(A)), lds, st, A);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

template <int MODE>
int gen_launch_mlp(const GenArgs &A, int grid, hipStream_t st) {
    if (A.bf16) return gen_ng16(A, A.f.Sx) == 4 ? gen_launch_mlp16<MODE, 4>(A, grid, st) : gen_launch_mlp16<MODE, 2>(A, grid, st);
    const size_t lds = gen_lds_bytes(A, A.f.Sx);
    static DeviceOnce once;
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    BHN_HIP(once.run(dev, [&](int &) { return hipFuncSetAttribute((const void *)gen_mlp_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
    hipLaunchKernelGGL(gen_mlp_kernel<MODE>, dim3(grid), dim3(gen_block(A)), lds, st, A);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

}  // namespace

// f32 image; bf16 mode: + the bf16 fragment images behind it
size_t gen_packed_bytes(const MlpShape &s, int32_t mode) {
    GenArgs A;
    gen_layout(s, &A);
    return mode == BHN_BF16 ? (size_t)A.g16.image_off + A.g16.image_bytes : (size_t)A.packed_floats * 4;
}

int gen_pack_weights(const MlpShape &s, int32_t mode, const float *params, void *packed, hipStream_t st) {
    GenArgs A;
    memset(&A, 0, sizeof(A));
    gen_layout(s, &A);
    A.params = params;
    A.packed_out = reinterpret_cast<float *>(packed);
    hipLaunchKernelGGL(gen_pack_fill_kernel, dim3(512), dim3(256), 0, st, A);
    if (mode == BHN_BF16) hipLaunchKernelGGL(gen_pack16_kernel, dim3(512), dim3(256), 0, st, A);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

int fused_fill_args(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom, const bhn_frames *fr, bool need_w,
                    FusedArgs *a, MlpShape *s, int nwaves);     // fused_fwd.hip

// FusedArgs for the general kernels: tiles of GEN_TG consecutive 32-point groups of one frame
static int gen_fill(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom, const bhn_frames *fr, bool need_w,
                    GenArgs *A, MlpShape *s) {
    memset(A, 0, sizeof(*A));
    const int rc = fused_fill_args(m, mode, packed, geom, fr, need_w, &A->f, s, GEN_TG);
    if (rc != BHN_OK) return rc;
    gen_layout(*s, A);
    A->bf16 = mode == BHN_BF16 ? 1 : 0;
    A->pk = reinterpret_cast<const float *>(packed);
    A->tile0 = 0;
    A->ntiles = A->f.total_tiles;
    return BHN_OK;
}

static size_t gen_align(size_t v) { return (v + 255) & ~(size_t)255; }
static int gen_grid(const GenArgs &A, long long ntiles, int dev) {
    const size_t lds = A.bf16 ? gen_lds16_bytes(A, A.f.Sx, gen_ng16(A, A.f.Sx)) : gen_lds_bytes(A, A.f.Sx);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(160 * 1024 / lds, 2048 / gen_block(A)));
    return (int)bhn_balanced_grid(ntiles, (long long)bhn_num_cus(dev) * per_cu);
}
// bytes of the gradient slabs / of the tape of ALL tiles of a call
static size_t gen_slab_bytes(const GenArgs &A) { return gen_align((size_t)A.nsplit * A.slab_floats * 4); }
static size_t gen_tape_bytes(const GenArgs &A, long long tiles) { return (size_t)tiles * GEN_TG * (A.bf16 ? (size_t)A.g16.bytes : (size_t)A.tape_tile * 4); }

// bhn_predict_fwd / bhn_render_fwd; with a workspace that holds the whole tape (bhn_render_fwd_train): the render that records it
int gen_forward(bool render, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom, const bhn_frames *fr,
                float *out, hipStream_t st, void *workspace, size_t workspace_bytes) {
    GenArgs A;
    MlpShape s;
    int rc = gen_fill(m, mode, packed, geom, fr, render, &A, &s);
    if (rc != BHN_OK) return rc;
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    BHN_CHECK_DEVICE(dev);
    const int grid = gen_grid(A, A.ntiles, dev);
    if (render) {
        A.f.images = out;
        BHN_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)A.f.B * A.f.Sx * A.f.R, st));
        // the training forward records the tape when the workspace holds ALL of it (what bhn_render_bwd_tape then asks for);
        // with a smaller workspace it is the plain render and the gradient comes from bhn_render_bwd, chunk by chunk
        if (workspace && workspace_bytes >= gen_slab_bytes(A) + gen_tape_bytes(A, A.ntiles)) {
            A.tape = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + gen_slab_bytes(A));
            A.tape16 = reinterpret_cast<char *>(A.tape);
            return gen_launch_mlp<GEN_RECORD>(A, grid, st);
        }
        return gen_launch_mlp<GEN_RENDER>(A, grid, st);
    }
    A.f.emission = out;
    if (geom->groups) BHN_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)A.f.B * A.f.P, st));
    return gen_launch_mlp<GEN_PREDICT>(A, grid, st);
}

static constexpr long long GEN_MIN_CHUNK_TILES = 2;       // smallest tape chunk (tiles of GEN_TG groups) gen_backward accepts

// slabs + the tape of all B frames: what the training pair (bhn_render_fwd_train / bhn_render_bwd_tape) needs.  bhn_render_bwd
// takes ANY workspace from slabs + 16 groups of tape on and walks the tiles in chunks (the tile -> (frame, group) map goes through
// fd_tpf: a chunk may start and end anywhere) -- callers that cannot afford the whole tape (one frame of an 8x512 network on a
// 256 x 256 x 128 ray set is 277 GB) allocate what they have and call that (engine.workspace).
size_t gen_bwd_workspace_bytes(const MlpShape &s, int32_t mode, int32_t B, int64_t P) {
    GenArgs A;
    gen_layout(s, &A);
    A.bf16 = mode == BHN_BF16 ? 1 : 0;
    const long long tiles = ((P + 31) / 32 + GEN_TG - 1) / GEN_TG * B;
    return gen_slab_bytes(A) + gen_tape_bytes(A, tiles);
}

// tape_only: bhn_render_bwd_tape (the tape of the whole call was recorded by gen_forward); else bhn_render_bwd (recompute, chunks)
int gen_backward(bool tape_only, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom, const bhn_frames *fr,
                 const float *dimages, float *dparams, void *workspace, size_t workspace_bytes, hipStream_t st) {
    GenArgs A;
    MlpShape s;
    int rc = gen_fill(m, mode, packed, geom, fr, true, &A, &s);
    if (rc != BHN_OK) return rc;
    BHN_CHECK_ARG(workspace && dimages && dparams, "null pointer");
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    BHN_CHECK_DEVICE(dev);
    const size_t slab_bytes = gen_slab_bytes(A);
    const long long min_tiles = tape_only ? A.f.total_tiles : std::min<long long>(A.f.total_tiles, GEN_MIN_CHUNK_TILES);
    if (workspace_bytes < slab_bytes + gen_tape_bytes(A, min_tiles)) {
        bhn_set_error(tape_only ? "the recorded-tape path needs a workspace for the whole tape (%zu bytes given, need >= %zu: slabs %zu + %lld groups of tape); "
                                  "use bhn_render_fwd + bhn_render_bwd, which walk the groups in chunks"
                                : "render_bwd workspace too small: %zu bytes, need >= %zu (slabs %zu + %lld groups of tape)",
                      workspace_bytes, slab_bytes + gen_tape_bytes(A, min_tiles), slab_bytes, min_tiles * GEN_TG);
        return BHN_EWORKSPACE;
    }
    A.slabs = reinterpret_cast<float *>(workspace);
    A.tape = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + slab_bytes);
    A.tape16 = reinterpret_cast<char *>(A.tape);
    A.dimages = dimages;
    A.dparams = dparams;
    const long long chunk = std::min<long long>(A.f.total_tiles, (long long)((workspace_bytes - slab_bytes) / gen_tape_bytes(A, 1)));
    for (long long c0 = 0; c0 < A.f.total_tiles; c0 += chunk) {
        A.tile0 = c0;
        A.ntiles = std::min<long long>(chunk, A.f.total_tiles - c0);
        A.accumulate = c0 > 0;
        rc = tape_only ? gen_launch_mlp<GEN_CHAIN_TAPE>(A, gen_grid(A, A.ntiles, dev), st) : gen_launch_mlp<GEN_CHAIN>(A, gen_grid(A, A.ntiles, dev), st);
        if (rc != BHN_OK) return rc;
        if (A.bf16) hipLaunchKernelGGL(gen_dw16_kernel, dim3(A.njobs, A.nsplit), dim3(64), 0, st, A);
        else hipLaunchKernelGGL(gen_dw_kernel, dim3(A.njobs, A.nsplit), dim3(64), 0, st, A);
        BHN_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gen_reduce_kernel, dim3(1024), dim3(256), 0, st, A);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

/*
 * bhnerf_hip.h -- C ABI of libbhnerf_hip.so: the MI355X (gfx950) implementation of the one
 * hot path of aviadlevis/bhnerf (velocity warp -> positional encoding -> skip-MLP ->
 * sigmoid/masks -> radiative-transfer ray sum -> chi^2 -> Adam).
 *
 * The reference has no FFI layer: its seams are Python callables.  Each entry
 * point below names the reference function(s) (file:line under the reference tree) whose
 * arithmetic it replaces; the bhnerf_amd Python modules bind them with ctypes behind the reference's own
 * Python signatures (INTEGRATION.md shows the stub a bhnerf maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, a BHN_E* code otherwise; bhn_last_error() gives a
 *     thread-local message.  Nothing aborts, nothing prints.
 *   - the CALLER owns and allocates every buffer (device memory unless marked host); workspace
 *     sizes are queried with bhn_*_bytes().  No hidden allocation, no hidden synchronisation,
 *     no global mutable state except the thread-local error string and per-device one-time caches
 *     (CU count, kernel attributes; std::call_once): re-entrant across threads, streams and devices.
 *     The CURRENT device of the calling thread must be the device that owns `stream` and the buffers.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call returns
 *     immediately (graph-capturable: no malloc/free/sync inside).
 *   - arrays are C-contiguous float32 unless stated.  P = R*G points per frame, flat index
 *     p = ray*G + sample (the reference's (H,W,G) layout flattened).  S = number of Stokes
 *     planes, 0 meaning "J is the scalar 1" (unpolarised); Sx = max(S,1).
 *   - `mode` selects the arithmetic of the MLP GEMMs: BHN_F32 = f32 MFMA (exact fmaf chains,
 *     parity mode), BHN_BF16 = bf16 operands, f32 accumulate (throughput mode).
 */
#ifndef BHNERF_HIP_H
#define BHNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points declared here (BHN_API) are its only dynamic symbols. */
#ifndef BHN_API
#if defined(__GNUC__)
#define BHN_API __attribute__((visibility("default")))
#else
#define BHN_API
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define BHN_ABI_VERSION 5      /* 3: BHN_BF16_T8 / BHN_T8_CALIBRATE, bhn_adam_hyper, bhn_adam_step_dev, bhn_render_bwd_tape_kernel_name_for;
                                * 4: bhn_geom.ray_span; posenc_deg <= 10, net_width <= 512 (general path);
                                * 5: bhn_frames.clock_probe, bhn_tape_info, bhn_mfma_probe */

enum { BHN_OK = 0, BHN_EINVAL = 1, BHN_EUNSUPPORTED = 2, BHN_EHIP = 3, BHN_EWORKSPACE = 4 };
enum { BHN_F32 = 0, BHN_BF16 = 1, BHN_BF16_T8 = 2 };
/* BHN_BF16_T8 ("8-bit tape", width 256, depth >= 3 only; an opt-in training mode, NOT a parity mode): the arithmetic of
 * BHN_BF16, but the backward's tape keeps the two operands of the weight-gradient GEMMs (layer inputs h_l, pre-activation
 * gradients gA_l) as OCP e4m3 bytes -- half the tape traffic; the gradient differs from BHN_BF16's by ~2e-3 of its norm.
 * gA_l is stored with one power-of-two scale per layer: (largest |gA_l| / largest |dimages| of the PREVIOUS backward call on
 * the same workspace, kept in the workspace) x (largest |dimages| of this call) x 16 head room.  OR BHN_T8_CALIBRATE into
 * `mode` on the first backward call on a workspace (or after the weights were replaced wholesale): that call runs the delta
 * chain twice to take the ratios from itself.  Every
 * entry point that does not touch the tape treats BHN_BF16_T8 as BHN_BF16. */
#define BHN_T8_CALIBRATE 0x100

/* NeRF_Predictor hyper-parameters: network.py:147-157 (scale,rmin,rmax,z_width,posenc_deg,
 * net_depth,net_width,do_skip; activation=relu and out_channel=1 are fixed as in every driver). */
typedef struct {
    int32_t net_depth;    /* hidden Dense+ReLU layers, 2..8 (network.py:153) */
    int32_t net_width;    /* 1..512 (network.py:154); widths other than 32/64/128/256 run zero-padded on the next one;
                           * 257..512: the general layer-by-layer path (below)                                            */
    int32_t posenc_deg;   /* 0..10 (network.py:151); the fused kernels take 0..4 (3+6*deg <= 32 features incl. a bias
                           * slot), 5..10 run on the general path                                                          */
    int32_t do_skip;      /* skip-concat after layer depth/2 (network.py:59-61) */
    float scale, rmin, rmax, z_width;
} bhn_model;

/* The general path (csrc/general_mlp.hip; posenc_deg > 4 or net_width > 256 -- shapes no reference driver uses): every entry
 * point below accepts these models with the same arguments and the same results, computed layer by layer on tiles of eight
 * 32-point groups: BHN_F32 with f32 MFMAs, BHN_BF16 with bf16 MFMAs on fragment-ordered operands (round 6; BHN_BF16_T8 is
 * BHN_EUNSUPPORTED).  bhn_packed_bytes differs by mode (BHN_BF16: the f32 image + bf16 fragment images).
 * bhn_render_bwd_workspace_bytes(B, P) = gradient slabs + the tape of all B frames: with a workspace of that size
 * bhn_render_fwd_train records the tape and bhn_render_bwd_tape runs the delta chain and the weight gradients from it (with less:
 * BHN_EWORKSPACE, as for the fused paths).  bhn_render_bwd recomputes chunk by chunk and accepts ANY workspace from the slabs + 16
 * groups of tape on (not a whole frame's: one frame of an 8x512 network on 256 x 256 x 128 rays is 277 GB of f32 tape).
 * Images, losses and gradients of these models are bitwise reproducible like the fused kernels' (the ray segments of a tile are
 * combined in LDS: one float atomic per tile and ray, at most two per pixel for rays of <= 257 samples; slabs summed in a fixed order). */

/* Geodesic-side inputs, prepared once per ray set by bhn_geom_prepare (arrays of P floats). */
typedef struct {
    int64_t R, G;         /* rays (H*W) and samples per ray (ngeo)            */
    int32_t S;            /* Stokes planes (0 = unpolarised)                  */
    const float *x, *y, *z;   /* coords (3,H,W,G) planes, network.py:874      */
    const float *Omega;   /* angular velocity per point (emission.py:204)     */
    const float *t_geo;   /* slow-light time per point (emission.py:200)      */
    const float *w;       /* (Sx,P) folded g^2*dtau*Sigma*J_s (kgeo.py:621, network.py:417) */
    const uint8_t *dom;   /* (P) 1 inside rmin<=r<=rmax,|z|<=z_width (emission.py:370-373)   */
    /* Optional compaction of the static domain mask: ascending indices of the 32-point groups
     * (points 32*i .. 32*i+31) that contain at least one in-domain point.  The fused kernels and the
     * backward tape visit only these groups; every other point has emission 0 by emission.py:370-373.
     * NULL = all ceil(P/32) groups. */
    const int32_t *groups;
    int64_t n_groups;
    /* Optional point-level compaction (fused predictor / render kernels only): the arrays above then hold only the
     * `n_points` in-domain points of the ray set (ray-major order kept, padded with dom = 0 points to a multiple of 32),
     * w is (Sx, n_points), and ray_idx[i] is the ray (0..R-1) point i belongs to -- the reference's `p / G`
     * (kgeo.py:621 sums over the last axis).  Out-of-domain samples have emission 0 (emission.py:370-373), so leaving
     * them out changes no image and no gradient.  NULL / 0 = dense layout, P = R*G. */
    const int32_t *ray_idx;
    int64_t n_points;
    /* Point-compacted layouts only, optional (0 = not known): the largest number of consecutive 32-point groups the points of
     * ONE ray lie in.  1 or 2: every pixel then receives at most two partial sums, so the render kernels add each wave's ray
     * segments straight to the pixels (bitwise reproducible all the same) and skip the per-tile combine through LDS and its
     * workgroup barriers (19 % of the width-128 forward of BASELINE config 5).  A value that is too small makes pixel sums
     * depend on the arrival order of their addends, nothing else.  Dense layouts: decided from G by the library. */
    int32_t ray_span;
} bhn_geom;

/* Frames of one step.  tM0[b] = (t_frames[b]-t_start_obs)/GM_c3 - t_injection, float64, device
 * (emission.py:200-201 evaluated per frame on the host in double). */
typedef struct {
    int32_t B;
    const double *tM0;
    /* Optional measurement aid (NULL = off; bench.py's `sustained_clock_mhz`): device buffer of 4 * BHN_CLK_SLOTS int64.  Workgroup 0
     * of each fused MLP kernel writes {s_memtime, s_memrealtime} at its start and at its end to entries 4 * slot .. 4 * slot + 3 of
     * its slot: (t1 - t0) / (r1 - r0) x 100 MHz is the core clock the kernel sustained.  Nothing else changes. */
    int64_t *clock_probe;
} bhn_frames;
enum { BHN_CLK_FWD = 0,        /* bhn_predict_fwd / bhn_render_fwd                                              */
       BHN_CLK_FWD_TRAIN = 1,  /* bhn_render_fwd_train                                                          */
       BHN_CLK_CHAIN = 2,      /* the delta chain (width-128 depth-4 bf16 networks: the fused chain + dW kernel) */
       BHN_CLK_DW = 3,         /* the weight-gradient kernel                                                    */
       BHN_CLK_SLOTS = 4 };

BHN_API int bhn_version(void);
BHN_API const char *bhn_last_error(void);

/* Number of float32 parameters / offset table of the flat parameter buffer.  Layout = flax
 * param tree order: Dense_0.kernel (in,out) row-major, Dense_0.bias, Dense_1.kernel, ...
 * (network.py:56-62). */
BHN_API int64_t bhn_param_count(const bhn_model *m);
/* kernel_off/bias_off: host arrays of net_depth+1 entries; in_dim likewise (may be NULL). */
BHN_API int bhn_param_layout(const bhn_model *m, int64_t *kernel_off, int64_t *bias_off, int32_t *in_dim);

/* One-off fold of the static per-point factors.  Replaces the per-iteration broadcasts of
 * kgeo.py:618-621 (g^2*dtau*Sigma), network.py:416-417 (J) and emission.py:370-373 (domain
 * mask, evaluated on the UN-warped coords).  J may be NULL (S=0).  coords is (3,P). */
BHN_API int bhn_geom_prepare(const float *coords, const float *g, const float *dtau, const float *Sigma,
                     const float *J, int32_t S, int64_t P, float rmin, float rmax, float z_width,
                     float *w_out, uint8_t *dom_out, void *stream);

/* kgeo.radiative_trasfer (kgeo.py:595-622) stand-alone: img[n,r] = sum_k g^2 e[n,r,k] dtau Sigma.
 * e is (N,R,G) with N = product of leading axes; g,dtau,Sigma are (R,G).  bwd: de = dimg * w. */
BHN_API int bhn_radiative_transfer_fwd(const float *e, const float *g, const float *dtau, const float *Sigma,
                               float *img, int64_t N, int64_t R, int64_t G, void *stream);
BHN_API int bhn_radiative_transfer_bwd(const float *dimg, const float *g, const float *dtau, const float *Sigma,
                               float *de, int64_t N, int64_t R, int64_t G, void *stream);

/* Re-layout of the flat f32 parameters into MFMA-fragment order (forward and transposed images,
 * biases).  Must be re-run after every parameter update; `packed` is bhn_packed_bytes() big. */
BHN_API size_t bhn_packed_bytes(const bhn_model *m, int32_t mode);
BHN_API int bhn_pack_weights(const bhn_model *m, int32_t mode, const float *params, void *packed, void *stream);

/* NeRF_Predictor.__call__ (network.py:191-237) fused: warp (emission.py:143-211) -> posenc
 * (network.py:98-122) -> MLP (network.py:18-64) -> sigmoid(out-10) -> domain fill -> injection
 * mask.  emission is (B,P).  geom->w may be NULL here. */
BHN_API int bhn_predict_fwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                    const bhn_frames *fr, float *emission, void *stream);

/* image_plane_prediction (network.py:373-420) fused with the predictor: images (B,Sx,R), no
 * emission is materialised.  The kernel zero-fills `images` itself. */
BHN_API int bhn_render_fwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                   const bhn_frames *fr, float *images, void *stream);

/* Reverse of bhn_render_fwd w.r.t. the parameters (jax.value_and_grad in network.py:617):
 * dimages (B,Sx,R) -> dparams (flat f32, overwritten).  The training forward is run again (tape only)
 * and then the delta chain, both register-chained per 32-point tile; layer inputs and pre-activation
 * gradients are streamed to a fragment-ordered tape inside `workspace`, from which the weight-gradient GEMMs (K = points)
 * accumulate one layer per workgroup in registers; per-workgroup slabs are reduced at the end
 * (deterministic, no float atomics).  bhn_render_bwd_workspace_bytes(B,P) is the size that holds
 * all B frames at once; any workspace that holds the slabs plus ONE frame of tape is accepted and
 * makes the call iterate over groups of frames. */
BHN_API size_t bhn_render_bwd_workspace_bytes(const bhn_model *m, int32_t mode, int32_t B, int64_t P, int32_t device);
BHN_API int bhn_render_bwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                   const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Training fast path (same result as bhn_render_fwd + bhn_render_bwd, one forward less):
 * bhn_render_fwd_train renders `images` AND records layer inputs, ReLU bits and emission on the tape in
 * `workspace`; bhn_render_bwd_tape then runs only the delta chain + weight-gradient GEMMs from that
 * tape.  Both calls must see the same model/geometry/frames and a workspace of at least
 * bhn_render_bwd_workspace_bytes(B,P) bytes (BHN_EWORKSPACE otherwise: use the pair above).
 * bhn_render_bwd_tape only READS what the forward recorded (its own intermediates go to regions of their own), so it may
 * be called any number of times on one recorded tape, with the same or with other `dimages`. */
BHN_API int bhn_render_fwd_train(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                         const bhn_frames *fr, float *images, void *workspace, size_t workspace_bytes,
                         void *stream);
BHN_API int bhn_render_bwd_tape(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                        const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                        size_t workspace_bytes, void *stream);

/* loss_fn_image (network.py:476-484): dtype 0 = 'full', 1 = 'lc'.  target/sigma/offset are
 * (B,Sx,R) for 'full', (B,Sx) for 'lc'.  `loss` holds 1 + B*Sx floats: loss[1 + plane] = scale*chi^2 of that
 * (frame, Stokes) plane, loss[0] = their sum, added in a fixed order (bitwise reproducible; no atomics).
 * dimages = dloss/dimages (pass NULL to skip the gradient). */
BHN_API int bhn_chi2_image(const float *images, const float *target, const float *sigma, const float *offset,
                   float scale, int32_t dtype, int32_t B, int32_t Sx, int64_t R, float *loss,
                   float *dimages, void *stream);

/* loss_fn_eht (network.py:486-564): visibilities = A . image, then chi^2 of dtype 0 = 'vis' (complex
 * target), 1 = 'amp', 2 = 'cphase' (closure phase of the product over the C axis).  images (N,R) with
 * N = B*Sx planes; A complex64 interleaved (N,C,nvis,R), C = 1 for vis/amp; target (N,nvis) (complex
 * interleaved for 'vis'); sigma (N,nvis); vis_ws: caller scratch of bhn_chi2_eht_ws_floats(N,C,nvis,R) floats.
 * Writes loss[0] = scale*chi^2 and, unless NULL, dimages (N,R).  The visibility GEMV is split over the R axis into
 * enough blocks to fill the chip and reduced in two fixed-order stages (no atomics: loss and gradient are bitwise
 * reproducible). */
BHN_API size_t bhn_chi2_eht_ws_floats(int32_t N, int32_t C, int32_t nvis, int64_t R);
BHN_API int bhn_chi2_eht(const float *images, const float *A, const float *target, const float *sigma, float scale,
                 int32_t dtype, int32_t N, int32_t C, int32_t nvis, int64_t R, float *vis_ws, float *loss,
                 float *dimages, void *stream);

/* Voxel forward renderer: emission.image_plane_dynamics (emission.py:235-303) fused: warp ->
 * trilinear sampling of a 3-D emission grid (interpolate_coords, emission.py:213-233; scipy map_coordinates
 * order=1, cval=0) -> x J g^2 dtau Sigma -> ray sum.  grid is (nx,ny,nz) C-order with frame_stride 0, or one
 * grid per frame with frame_stride = nx*ny*nz; fov_host = 3 host floats, the coordinate extent of each grid
 * axis (grid centred on 0 as utils.world_to_image_coords assumes).  geom->dom is not used (no domain fill in
 * this function); images is (B,Sx,R).  bhn_trilinear: interpolate_coords alone, points (N,3) -> out (N). */
BHN_API int bhn_voxel_render_fwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t nx, int32_t ny,
                         int32_t nz, int64_t frame_stride, const float *fov_host, float *images, void *stream);
BHN_API int bhn_trilinear(const float *points, int64_t N, const float *grid, int32_t nx, int32_t ny, int32_t nz,
                  const float *fov_host, float *out, void *stream);

/* GRID_Predictor (network.py:254-353): the emission as a learnable (res,res,res) voxel grid sampled trilinearly at the
 * velocity-warped points, index = (u + scale)/(2 scale)(res - 1), 0 outside the grid (map_coordinates order 1,
 * cval 0), sigmoid(. - 10), domain fill (geom->dom), 0 before the injection.  predict: emission (B,P);
 * render: images (B,Sx,R) as bhn_render_fwd; render_bwd: dgrid (res^3, overwritten) = d loss / d grid for the given
 * d loss / d images (float atomics: not bitwise reproducible). */
BHN_API int bhn_grid_predict_fwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale,
                         float *emission, void *stream);
BHN_API int bhn_grid_render_fwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale,
                        float *images, void *stream);
BHN_API int bhn_grid_render_bwd(const bhn_geom *geom, const bhn_frames *fr, const float *grid, int32_t res, float scale,
                        const float *dimages, float *dgrid, void *stream);

/* optax.adam + polynomial_schedule(power=1) as used by init_state (network.py:173-174, 621):
 * g' = g*grad_scale (the 1/ndev of pmean, network.py:620); t = 1-based update count. */
BHN_API int bhn_adam_step(float *params, const float *grads, float *m, float *v, int64_t n, int64_t t, float lr,
                  float b1, float b2, float eps, float grad_scale, void *stream);

/* The same update for a step that is captured into a HIP graph: lr and the two bias corrections come from DEVICE memory
 * (hyper_dev[3] = {lr, 1 - b1^t, 1 - b2^t}), so the launch carries no per-step scalar.  bhn_adam_hyper fills the three floats on
 * the HOST exactly as bhn_adam_step computes them (the caller copies them to hyper_dev before each replay): parameters
 * bitwise equal to bhn_adam_step's. */
BHN_API int bhn_adam_hyper(int64_t t, float lr, float b1, float b2, float *hyper_host);
BHN_API int bhn_adam_step_dev(float *params, const float *grads, float *m, float *v, int64_t n, const float *hyper_dev,
                      float b1, float b2, float eps, float grad_scale, void *stream);

/* bhn_render_bwd_tape with the caller's HIP events recorded on `stream` at its kernel boundaries, so that each kernel
 * of the backward can be timed live (bench.py's roofline): events[0] before the first kernel, events[i] behind kernel
 * i - 1 (i = 1..BHN_BWD_TAPE_KERNELS); n_events <= BHN_BWD_TAPE_KERNELS + 1, NULL entries are skipped.  The events are
 * hipEvent_t handles created by the caller; nothing is synchronised or allocated here.
 * bhn_render_bwd_tape_kernel_name(i): the name of kernel i as it appears in a rocprofv3 kernel trace (prefix). */
#define BHN_BWD_TAPE_KERNELS 3
BHN_API int bhn_render_bwd_tape_timed(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                              const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                              size_t workspace_bytes, void *stream, void *const *events, int32_t n_events);
BHN_API const char *bhn_render_bwd_tape_kernel_name(int32_t i);
/* The same for the path a given network takes: width-128 bf16 networks of depth 4 (the reference's default MLP, network.py:19-20)
 * run the delta chain and the weight-gradient GEMMs as ONE kernel (slot 0: "bwd128_kernel", behind a small per-point
 * "dout128_kernel"; slot 1 is empty: "-"; slot 2: "reduce128_kernel"). */
BHN_API const char *bhn_render_bwd_tape_kernel_name_for(const bhn_model *m, int32_t mode, int32_t i);

/* What the backward's tape of a network costs, from the library's own layout (bench.py's `tape_stream` figures): info[0] bytes the
 * training forward writes per 32-point group, [1] / [2] bytes the delta chain writes / reads, [3] bytes the weight-gradient kernel
 * (or the fused width-128 backward) reads, [4] flags (BHN_TAPE_*), [5] 32-point groups per workgroup tile of the training forward
 * for a ray set of `groups_per_frame` groups.  Host-only, no device call. */
#define BHN_TAPE_INFO_N 8
enum { BHN_TAPE_DROP_H1 = 1,     /* h_1 is recomputed from the encoded inputs                                    */
       BHN_TAPE_DROP_GA = 2,     /* gA_{depth-1} is rebuilt by its dW job; the output row rides on that job       */
       BHN_TAPE_GA0_CHAIN = 4,   /* dW_0 is accumulated inside the delta chain (no layer-0 dW job, no gA_0)      */
       BHN_TAPE_FUSED128 = 8,    /* fused delta chain + weight gradients (width 128, depth 4, bf16)              */
       BHN_TAPE_DROP_HD = 16,    /* h_depth is not recorded (its relu bits are)                                  */
       BHN_TAPE_LBITS = 32,      /* the dW job of layer depth-1 works from relu bits                             */
       BHN_TAPE_GENERAL = 64 };  /* general path (posenc_deg > 4 or net_width > 256): f32 tape in chunks         */
BHN_API int bhn_tape_info(const bhn_model *m, int32_t mode, int64_t groups_per_frame, int64_t *info, int32_t n_info);

/* Measurement aid (bench.py's `mfma_peak_this_box`): `iters` x 16 dependent v_mfma_f32_32x32x16_bf16 per wave with every operand
 * in registers, random data, 8 waves on each of `grid` workgroups: the matrix-pipe ceiling of THIS board under its power cap,
 * timed by the caller (HIP events around the call).  flop = grid x 8 x iters x 16 x 32768.  clk_dev: optional, 2 int64 per
 * workgroup {s_memtime ticks, s_memrealtime ticks}; sink_dev: 4 KiB the kernel may write to (it does not). */
BHN_API int bhn_mfma_probe(int32_t grid, int32_t iters, int64_t *clk_dev, float *sink_dev, void *stream);

/* Device self-checks of the MFMA / LDS-transpose / LDS-DMA lane maps the kernels rely on (exact integer
 * data).  results: 8 int32 mismatch counts on the host, all 0 when the maps hold.  scratch_dev: caller-owned device
 * memory of >= BHN_SELFTEST_SCRATCH_BYTES.  The one synchronous call of the library (runs on the null stream and
 * waits for the device). */
#define BHN_SELFTEST_SCRATCH_BYTES 16384
BHN_API int bhn_selftest(int32_t *results_host, void *scratch_dev, size_t scratch_bytes);

#ifdef __cplusplus
}
#endif
#endif /* BHNERF_HIP_H */

// Shared host/device definitions for libbhnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <stdio.h>
#include <string.h>

#include "../../include/bhnerf_hip.h"
#ifdef BHN_DEBUG
#include <stdlib.h>
#include "../../include/bhnerf_hip_debug.h"
#endif

#define BHN_MAX_LAYERS 9   // net_depth <= 8 hidden layers + the output layer
#define BHN_ENC_PAD 32     // encoded input (3 + 6*deg <= 27) padded to one 32-feature block
#define BHN_DEG_MAX 4      // posenc degrees 0..4 share ONE kernel-side slot layout of the 32-feature block
// Slot layout (round 6).  An MFMA B fragment gives lane half h, element j of k-step s the slot 16 s + 8 (j >> 2) + 4 h + (j & 3): the
// two lane halves of one register hold slots q and q + 4.  The layout makes every such register pair a (sin, cos) pair of ONE
// argument -- register n = 8 s + j (0..15) of a lane:
//     n = 0..2            u_n on half 0 (slot n), nothing on half 1 (slot n + 4: zero weight rows)
//     n = 3 + 3 i + k     sin(2^i u_k) on half 0, cos(2^i u_k) on half 1, octave i < deg, coordinate k < 3
//     n = 15              nothing on half 0 (slot 27); slot 31 (half 1) carries the constant 1 on the tapes whose dW GEMMs take the
//                         bias gradient from it (TapeLayout::fused128 / ga0_chain)
// so that a lane evaluates ONE transcendental per register, sin(2 pi (rev + h / 4)), instead of the sine AND the cosine of every
// argument on both halves followed by a select (rounds 1-5: slots [u | sin block | cos block], 24 + 2 quarter-rate instructions
// and 32 selects per lane; now 12 + 2 and none).  bhn_pack_weights and the reduce kernels reach the reference's feature order
// [u | sin block (3 deg) | cos block (3 deg)] (network.py:118-122) only through the two maps below.
#ifndef BHN_ENC_PAIRS
#define BHN_ENC_PAIRS 1    // 0: the old [u | sin | cos] slot layout (A/B builds)
#endif
// slot of register n (0..15) on lane half 0
static inline __host__ __device__ int bhn_enc_reg_slot(int n) { return 16 * (n >> 3) + 8 * ((n >> 2) & 1) + (n & 3); }
// reference feature index (network.py:118-122: [u | sin block (3 deg) | cos block (3 deg)]) of kernel slot q, or -1
static inline __host__ __device__ int bhn_enc_slot_feature(int q, int deg) {
#if BHN_ENC_PAIRS
    const int h = (q >> 2) & 1, n = 8 * (q >> 4) + 4 * ((q >> 3) & 1) + (q & 3);
    if (n < 3) return h == 0 ? n : -1;
    const int t = n - 3, i = t / 3;
    if (t >= 3 * BHN_DEG_MAX || i >= deg) return -1;
    return (h == 0 ? 3 : 3 + 3 * deg) + t;
#else
    if (q < 3 + 3 * deg) return q;
    if (q >= 3 + 3 * BHN_DEG_MAX && q < 3 + 3 * BHN_DEG_MAX + 3 * deg) return 3 + 3 * deg + (q - 3 - 3 * BHN_DEG_MAX);
    return -1;
#endif
}
// kernel slot of reference feature index f (0 <= f < 3 + 6 deg)
static inline __host__ __device__ int bhn_enc_feature_slot(int f, int deg) {
#if BHN_ENC_PAIRS
    if (f < 3) return f;
    return f < 3 + 3 * deg ? bhn_enc_reg_slot(f) : bhn_enc_reg_slot(f - 3 * deg) + 4;
#else
    return f < 3 + 3 * deg ? f : 3 + 3 * BHN_DEG_MAX + (f - 3 - 3 * deg);
#endif
}

void bhn_set_error(const char *fmt, ...);

#define BHN_CHECK_ARG(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            bhn_set_error(__VA_ARGS__);                           \
            return BHN_EINVAL;                                    \
        }                                                         \
    } while (0)

#define BHN_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            bhn_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                       \
            return BHN_EHIP;                                                               \
        }                                                                                  \
    } while (0)

// Shape of the MLP of network.py:49-62 as the kernels see it.
struct MlpShape {
    int depth;                  // hidden layers
    int width;                  // KERNEL width: the model's width padded to 32, 64, 128 or 256 (zero weights for the padding units)
    int width_true;             // the model's hidden width (network.py:154): what the flat parameter layout uses
    int F;                      // encoded input features 3 + 6*deg (network.py:118-122)
    int deg;                    // posenc degree (0..BHN_DEG_MAX; general path: .. BHN_GEN_DEG_MAX)
    int general;                // 1: a shape outside the fused kernels (posenc_deg > 4 or net_width > 256) -> general_mlp.hip
    int skip_in[BHN_MAX_LAYERS];   // 1 if layer l takes concat[h, enc] as input (network.py:59-61)
    int in_dim[BHN_MAX_LAYERS];    // true fan-in of layer l (l = depth is the output layer)
    int64_t kernel_off[BHN_MAX_LAYERS], bias_off[BHN_MAX_LAYERS];
    int64_t nparams;
};

int bhn_mlp_shape(const bhn_model *m, MlpShape *s);   // validates, returns BHN_* code

// general_mlp.hip: the f32 layer-by-layer path of the shapes the fused kernels are not built for (MlpShape::general)
#define BHN_GEN_DEG_MAX 10     // 3 + 6 deg <= 63 encoded features
#define BHN_GEN_WIDTH_MAX 512
size_t gen_packed_bytes(const MlpShape &s, int32_t mode);
int gen_pack_weights(const MlpShape &s, int32_t mode, const float *params, void *packed, hipStream_t st);
int gen_forward(bool render, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom, const bhn_frames *fr,
                float *out, hipStream_t st, void *workspace = nullptr, size_t workspace_bytes = 0);     // (workspace: bhn_render_fwd_train records the tape)
size_t gen_bwd_workspace_bytes(const MlpShape &s, int32_t mode, int32_t B, int64_t P);
int gen_backward(bool tape_only, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom, const bhn_frames *fr,
                 const float *dimages, float *dparams, void *workspace, size_t workspace_bytes, hipStream_t st);

// bf16 networks with >= 3 hidden layers: the backward never materialises gA_{depth-1} = W_out (.) dout (.) relu' --
// the dW job of layer depth-1 rebuilds it from h_depth and dout (TapeLayout::drop_ga), and the delta chain feeds
// relu' (.) bf16(dout) into a transposed weight image of layer depth-1 whose columns are pre-scaled by W_out
// (bhn_pack_weights): the same product with the factor W_out[k] moved from the B operand to the A operand.
__host__ __device__ static inline bool bhn_folds_wout(int mode, int depth) { return mode == BHN_BF16 && depth >= 3; }
// BHN_BF16_T8 (+ BHN_T8_CALIBRATE) is BHN_BF16 for everything but the tape of the backward
static inline int bhn_norm_mode(int mode) { return ((mode & 0xff) == BHN_BF16_T8 && !(mode & ~(0xff | BHN_T8_CALIBRATE))) ? BHN_BF16 : mode; }

// Number of compute units of a device (cached); 0 for a device id outside [0, BHN_MAX_DEVICES) -- callers turn that into
// BHN_EINVAL (BHN_CHECK_DEVICE), the same answer DeviceOnce::run gives for such an id.
int bhn_num_cus(int device);

// Grid of a persistent kernel over `units` equal work items on at most `max_grid` workgroups: the smallest grid with the same
// number of rounds as max_grid (365 items on 256 CUs take two rounds either way: 183 workgroups then do what 256 would, with
// fewer weight preludes and fewer per-workgroup gradient slabs to flush and reduce).
static inline long long bhn_balanced_grid(long long units, long long max_grid) {
    if (units <= 0 || max_grid <= 0) return 1;
    if (units <= max_grid) return units;
    const long long rounds = (units + max_grid - 1) / max_grid;
    return (units + rounds - 1) / rounds;
}

// Per-device one-time setup of a kernel (hipFuncSetAttribute applies to the device that is current when it is
// called; a process may drive several devices, from several threads): run `f` once per device, remember its result.
// Only SUCCESS is remembered: a transient failure (e.g. an earlier sticky asynchronous error on that device) is returned
// to the caller and the set-up is tried again by the next call, under the mutex.
#define BHN_MAX_DEVICES 64
#define BHN_CHECK_DEVICE(dev) BHN_CHECK_ARG((dev) >= 0 && (dev) < BHN_MAX_DEVICES, "device %d outside [0, %d)", (int)(dev), BHN_MAX_DEVICES)
struct DeviceOnce {
    std::mutex mu;
    std::atomic<bool> done[BHN_MAX_DEVICES];
    int value[BHN_MAX_DEVICES];
    DeviceOnce() { for (auto &d : done) d.store(false, std::memory_order_relaxed); }
    template <class F>
    hipError_t run(int dev, F &&f) {
        if (dev < 0 || dev >= BHN_MAX_DEVICES) return hipErrorInvalidDevice;
        if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
        std::lock_guard<std::mutex> lock(mu);
        if (done[dev].load(std::memory_order_relaxed)) return hipSuccess;
        const hipError_t rc = f(value[dev]);
        if (rc == hipSuccess) done[dev].store(true, std::memory_order_release);
        return rc;
    }
};

// Shared host/device definitions for libbhnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/bhnerf_hip.h"

#define BHN_MAX_LAYERS 9   // net_depth <= 8 hidden layers + the output layer
#define BHN_ENC_PAD 32     // encoded input (3 + 6*deg <= 27) padded to one 32-feature block

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
    int width;                  // hidden width (multiple of 32)
    int F;                      // encoded input features 3 + 6*deg (network.py:118-122)
    int skip_in[BHN_MAX_LAYERS];   // 1 if layer l takes concat[h, enc] as input (network.py:59-61)
    int in_dim[BHN_MAX_LAYERS];    // true fan-in of layer l (l = depth is the output layer)
    int64_t kernel_off[BHN_MAX_LAYERS], bias_off[BHN_MAX_LAYERS];
    int64_t nparams;
};

int bhn_mlp_shape(const bhn_model *m, MlpShape *s);   // validates, returns BHN_* code

// Number of compute units of a device (cached).
int bhn_num_cus(int device);

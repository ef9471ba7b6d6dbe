#include "fused_common.h"
extern "C" size_t bhn_render_bwd_workspace_bytes(const bhn_model *m, int32_t mode, int32_t device) { return 0; }
extern "C" int bhn_render_bwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                              const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                              size_t workspace_bytes, void *stream) {
    bhn_set_error("render_bwd not built yet");
    return BHN_EUNSUPPORTED;
}

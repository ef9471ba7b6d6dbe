/*
 * bhnerf_hip_debug.h -- measurement switches of the DEBUG build of the library (make -C bhnerf_amd/csrc debug ->
 * libbhnerf_hip_dbg.so, compiled with -DBHN_DEBUG).  Not part of the product ABI: the release library
 * (libbhnerf_hip.so, include/bhnerf_hip.h) exports none of these, allocates nothing and reads no environment
 * variables.  Used by the scripts under tools/ (kernel ablations, ring-step time stamps).
 */
#ifndef BHNERF_HIP_DEBUG_H
#define BHNERF_HIP_DEBUG_H
#include "bhnerf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Run only some kernels of the backward on this thread: bit 0 = chain kernel, bit 1 = dW GEMM kernel, bit 2 = slab
 * reduction (default 7); bits 3.. = ablation flags of the kernels (fused_bwd.hip). */
BHN_API int bhn_debug_set_bwd_stages(int32_t mask);
/* bf16 forward kernel variant.  Low 4 bits: 1 = production kernel (default), 3 = ablation build of the 4x256 render
 * kernel; bits 4.. = its ablation flags (fused_fwd.hip). */
BHN_API int bhn_debug_set_fwd_variant(int32_t variant);
/* Copy the first `bytes` (<= 4096) of the ablation build's stamp buffer (allocated by the library) to the host. */
BHN_API int bhn_debug_read(void *dst_host, size_t bytes);
/* Environment variables read by the debug build: BHN_DEBUG_DW_GRID, BHN_DEBUG_JOB1_W, BHN_DEBUG_JOBL_W. */
#ifdef __cplusplus
}
#endif
#endif

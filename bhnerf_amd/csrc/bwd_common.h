// Declarations shared by the backward translation units (fused_bwd.hip: generic tape backward; fused_bwd128.hip: the
// fused delta-chain + weight-gradient kernel of width-128 networks).
#pragma once
#include "fused_common.h"

struct TapeLayout {
    long long NQ;                              // 32-point groups on the tape
    long long h_off[BHN_MAX_LAYERS + 1];       // h_l, l = 1..depth  (inputs of layer l)
    long long ga_off[BHN_MAX_LAYERS];          // gA_l, l = 0..depth-1
    long long enc_off, dout_off, mask_off, e_off, total;   // mask: relu bits [group][layer][word][lane]; e: [group][32] f32
    // the recorded h_l / gA_l tensors are equally spaced: h_off[l] = h_lin + l * lin_stride, ga_off[l] = ga_lin + l * lin_stride.
    // The producers address them this way: indexing the offset ARRAYS with the run-time layer made the compiler fetch
    // the entry from the kernel-argument segment in every ring step (s_load + s_waitcnt lgkmcnt(0), which also drains
    // the LDS prefetch queue).
    long long h_lin, ga_lin, lin_stride;
    // bf16: h_1 = relu(W_0^T enc + b_0) is NOT on the tape; the dW job of layer 1 recomputes it from the encoded
    // inputs kept a second time in their forward (point-on-lane) fragment form -- 64 B instead of 512 B per point
    long long encp_off;
    int drop_h1;
    // bf16, depth >= 3: gA_{depth-1} = relu'(a_{depth-1}) * W_out * dout is NOT on the tape either; the dW job of layer
    // depth-1 rebuilds it from the h_depth tiles (relu bits = "!= 0"), W_out and dout kept in f32 behind the dout
    // tile, and also makes dW_out from the same h_depth tiles (no separate output-layer job): -1 KB per point of
    // tape traffic (chain write + dW read of gA_{depth-1}, second dW read of h_depth)
    int drop_ga;
    // bf16, width 256, depth >= 3 (round 5): gA_0 is not on the tape and NOBODY streams it: the delta chain itself accumulates
    // dW_0 = gA_0^T [enc | 1] -- every wave stages its finished gA_0 tile in LDS, wave m adds tile m of all eight waves to the
    // ONE accumulator tile it owns (fused_bwd.hip "dW_0 inside the delta chain") -- and flushes it to BwdArgs::slab0; the dW
    // kernel has no layer-0 job.  The encoded-input tile on the tape carries 1 in slot 31 (the bias column), as for fused128.
    int ga0_chain;
    long long dout_stride;                     // bytes per group of the dout region: tile (+ 32 f32 when drop_ga)
    // width-128 bf16 networks of depth <= 4 (fused_bwd128.hip): the tape holds only what the FORWARD knows -- h_1 .. h_depth,
    // the encoded inputs (slot 31 set to 1: the bias column of the fused dW GEMMs) and e; no relu bits (the fused
    // backward reads them off the h tiles), no gA, no dout
    int fused128;
    // fused128, round 5: h_depth is NOT on the tape either.  The backward needs it for two things: relu'(a_{depth-1}) -- the
    // forward records those relu bits instead (maskd_off: [group][tile pair][lane] words, TapePost: 16 B per point for 256) --
    // and the output layer's row dW_out = sum_p dout_p h_depth[p], which the reduce makes from the layer's own gradient:
    // h = relu(a) = relu'(a) a and a = K^T h_{depth-1} + b give  dW_out[f] = sum_k K[k][f] G[k][f] + b[f] g[f]  with G, g the
    // (un-folded) weight and bias gradients of layer depth-1 (fused_bwd128.hip: reduce128_kernel).
    int drop_hd;
    // generic bf16 path (round 5): the dW job of layer depth-1 reads the relu-bit words instead of the h_depth tiles and makes the
    // output layer's row at its flush (dw_body2 LBITS); with drop_hd the forward does not store those tiles either
    int lbits;
    long long maskd_off, scratch_off;          // scratch: [5][128] f32 partial sums of that product (per input tile)
};

struct BwdArgs {
    FusedArgs f;
    char *tape;
    TapeLayout t;
    // dW jobs: job j = layer j (0..depth), workgroups [wg_begin[j], wg_begin[j+1])
    int wg_begin[BHN_MAX_LAYERS + 2];
    int accumulate;                            // 1: add to what the slabs already hold
    int debug;                                 // measurement aid: 1 skip MFMA work, 2 skip tape loads
    long long *ts_buf;                         // measurement aid: ring-step time stamps (debug bit 9)
    long long wrap;                            // measurement aid (debug build): tape tile addresses wrap after this many groups
    int policy;                                // measurement aid (debug build): 0 nt, 1 plain, 2 sc1 tape stores / loads
    float *dparams;
    long long kernel_off[BHN_MAX_LAYERS + 1], bias_off[BHN_MAX_LAYERS + 1];
    int in_dim[BHN_MAX_LAYERS + 1];
    int F;
    int width_true;                            // the model's hidden width (flat parameter layout); W is the kernel width
    long long nparams;
    // 8-bit tape (PolBF16T8): the state block at the head of the tape region of the workspace (fused_bwd.hip, t8_prepare_kernel):
    // [0 .. 7] the power-of-two scale of gA_l in this call, [8 .. 15] the largest |gA_l| it saw, [16 .. 23] their ratios to
    // |dimages|max carried to the next call, [24] |dimages|max
    float *t8;
    // TapeLayout::ga0_chain: dW_0 slabs of the delta chain's workgroups, [workgroup][tile m][1024] in the dW slab tile layout,
    // and how many workgroups wrote them (the reduce kernel sums them in workgroup order)
    float *slab0;
    int n_chain_wg;
    int chain_step;                            // stride of the dW_0 slabs that hold the stage-1 sums (chain_slab_stage1)
    int fwd_nw;                                // 32-point groups per tile of the training forward that recorded the tape (8; fused 4x128: 12, PolBF16X)
};


// fused_bwd128.hip: the fused delta-chain + weight-gradient backward of width-128 bf16 networks (depth 4)
bool bwd128_supported(int mode, int kernel_width, int depth);
size_t bwd128_slab_bytes(int grid);
void bwd128_tape_layout(int depth, long long NQ, TapeLayout *t);
int bwd128_launch(const BwdArgs &A, int depth, int grid, hipStream_t st);
int reduce128_launch(const BwdArgs &A, int depth, int nslabs, hipStream_t st);

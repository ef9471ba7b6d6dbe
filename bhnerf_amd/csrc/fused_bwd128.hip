// Fused backward of width-128 bf16 networks (the reference's default MLP, network.py:19-20): delta chain AND weight-gradient
// GEMMs in ONE kernel, with the gradient of every layer accumulated in the registers of the workgroup.
//
// Why a kernel of its own.  At width 256 the weight gradient of all layers (833 KB of f32) does not fit a compute unit, so
// the generic backward (fused_bwd.hip) sends every layer input h_l AND every pre-activation gradient gA_l through an HBM
// tape to a second kernel whose workgroups own one layer each: 3.3 KB per point and step.  At width 128 the whole
// gradient is 57,344 + 128 floats = 225 KB: it fits the 512 KB register file of one CU (4 waves x 256 accumulator
// registers), so gA_l never leaves the chip and the tape shrinks to what the forward knows: h_2 .. h_{depth-1}, the relu BITS of
// the last hidden layer, the encoded inputs, e -- 600 B per point, written once and read once.  (h_1 is recomputed from the
// encoded inputs; h_depth is not needed at all, round 5: the backward wants relu'(a_{depth-1}) -- 16 B of bits per point instead
// of 256 B of values -- and the output layer's row sum_p dout_p h_depth[p], which equals sum_k K[k][f] G[k][f] + b[f] g[f] for
// the layer's own gradient G, g because h = relu(a) = relu'(a) a: reduce128_kernel, TapeLayout::drop_hd.)
//
// Structure (one workgroup = 4 waves, one per SIMD, 512 registers each; one workgroup per CU, persistent).  A workgroup
// iteration takes 128 points (four 32-point tape groups).  Activations live in LDS as [point][feature] images whose 256-byte
// rows are stored in 16-byte chunks with the XOR swizzle that makes BOTH access patterns conflict-free
// (cdna_hip_programming.md T10, "one image for row reads and transposed reads", form (b)):
//     off(row, ch) = 256 row + 16 (ch ^ swz(row)),   swz(row) = ((row & 3) << 2) | (((row >> 2) & 3) ^ (row & 2))
// (round 5: the `^ (row & 2)` -- without it the ds_write_b128 of the chain's output tiles, 8 consecutive rows per lane group, hit
//  every bank twice: SQ_LDS_BANK_CONFLICT 21 % of the LDS-active cycles, profiles/r4_w128_sq_counters.txt; the three conditions --
//  writes: swz & 7 distinct over 8 aligned rows; row reads: swz a bijection on 16 rows; transposed reads: swz >> 2 distinct over
//  4 aligned rows -- are checked by tests/test_host_logic_cpu.py)
//   * row reads  (ds_read_b128): lane (point, half) takes chunk 2 ks + half = the B fragment of k-step ks of a product that
//     sums over FEATURES (the delta chain  gA_{l-1} = relu' (.) W_l gA_l);
//   * transposed reads (ds_read_b64_tr_b16): lane = feature, 8 points in the registers = the A / B fragments of a product
//     that sums over POINTS (dW_l^T = gA_l^T [h_l | enc]).
// The 8 features of a chunk are in the forward's canonical fragment order (fused_common.h), so a chunk IS the fragment the
// producer holds: the forward's tape tiles are copied in by LDS-DMA with the lanes' global offsets permuted (free), the
// chain's output tiles are written as they leave the accumulators (two ds_write_b128 per tile).  Feature positions inside a
// 32-feature tile are therefore permuted ("virtual" position v = 8 (2 s + h) + e holds feature 16 s + 4 h + (e & 3) +
// 8 (e >> 2)) identically for every operand; reduce128_kernel undoes the permutation when it writes the flat gradient.
//
// Per layer l = depth-1 .. 1, wave (i, j):
//   chain   tiles m in {2i, 2i+1} x point blocks {2j, 2j+1} of gA_{l-1}: A = its 16 fragments of the transposed weight image
//           (registers, loaded from L2 one phase ahead), B = row reads of gA_l: every B fragment feeds two MFMAs;
//   dW_l    accumulator tiles m in {2i, 2i+1} x n in {2j, 2j+1} of gA_l^T h_l plus (m = 2i + j) x the encoded-input tile:
//           the encoded inputs carry a 1 in slot 31, so that tile's column 31 is the bias gradient of EVERY layer and its
//           other columns the skip layer's encoded-input rows (discarded for the other layers); K = the 128 points.
// Layer 0: dW_0 = gA_0^T enc, one tile per wave.  gA_{depth-1} = relu' (.) bf16(dout) from the recorded relu bits, with W_out
// folded into the weight image and applied to dW_{depth-1} at the reduce (bhn_folds_wout: the same arithmetic as the generic
// path); the output layer's row comes out of the reduce (above), its bias is the sum of dout.
// 16 accumulator tiles (256 registers) per wave at depth 4; deeper networks use the generic path.
#include <type_traits>
#include "bwd_common.h"

namespace {

constexpr int MT = 4, KS = 8, TB = 2048;                           // width 128
constexpr int IMG = 32768;                                         // 128 points x 128 features, bf16
constexpr int OFF_GA = 0, OFF_H = 2 * IMG, OFF_E = 4 * IMG;         // two gA images, two h images, two encoded-input images (8 KiB)
constexpr int ENC_IMG = 8192;
constexpr int OFF_DOUT = OFF_E + 2 * ENC_IMG;                       // f32 dout of the 128 points
constexpr int OFF_DPK = OFF_DOUT + 512;                             // the same as bf16 (dW_out operand)
constexpr int OFF_W0 = OFF_DPK + 256;                              // layer 0's forward weight fragments (8 KiB) and bias (h_1 recompute)
constexpr int OFF_B0 = OFF_W0 + 8192;
constexpr int LDS_BYTES = OFF_B0 + 512;
constexpr int CB = (KS + 2) * 1024;                                 // chunk bytes of the packed weight images (fused_common.h Pack<128>)
constexpr int SLAB_TILES = 65;                                      // 60 hidden tiles + 4 layer-0 tiles + the output layer's row / biases
constexpr int SLAB_FLOATS128 = SLAB_TILES * 1024;

typedef PolBF16 Pol;
typedef Pol::frag frag;

DEVI frag lds_row(const char *smem, unsigned off) { return *reinterpret_cast<const frag *>(smem + off); }
DEVI void lds_put(char *smem, unsigned off, const frag &f) { *reinterpret_cast<frag *>(smem + off) = f; }

// two transposed reads = one K = points fragment (8 points of one feature per lane)
DEVI frag tr2(const char *smem, unsigned off_a, unsigned off_b) {
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smem + off_a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(smem + off_b));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(frag, v);
}

DEVI u32x4 make_rsrc(const char *p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return u32x4{lo, hi & 0xffffu, 0x7fffffffu, 0x00020000u};
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (only: "inline asm clobber list contains reserved registers: M0")
// one wave copies 1 KiB global -> LDS: lane i's 16 bytes come from rs.base + soff + voff(i) and land at lds + 16 i
DEVI void dma_piece(const u32x4 &rs, unsigned soff, unsigned lds, unsigned voff) {
#ifdef BHN_B128_ABL
    if (BHN_B128_ABL & 8) return;
#endif
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

// g with every 16-bit half zeroed whose half of h is zero (h: non-negative bf16 pairs, i.e. relu outputs).  Three instructions
// per pair of elements: h + 0x7fff7fff sets a half's sign bit iff it is nonzero (no carry between halves), a packed arithmetic
// >> 15 spreads it, and.  (min(h, 1) * g as packed 16-bit integers would be two, but hipcc 7.2 expands the packed unsigned min
// into compares and selects: measured slower.)
DEVI unsigned keep_where_nz(unsigned g, unsigned h) {
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    const unsigned sgn = h + 0x7fff7fffu;
    return g & __builtin_bit_cast(unsigned, __builtin_bit_cast(i16x2, sgn) >> (i16x2){15, 15});
}


// SKIPL: the hidden layer whose input is concat[h, enc] (network.py:59-61; 3 at depth 4 with do_skip), or 0 for none
template <int DEPTH, int SKIPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bwd128_kernel(BwdArgs A) {
    static_assert(DEPTH >= 2 && DEPTH <= 4, "14 accumulator tiles per wave at depth 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedArgs &a = A.f;
    clock_stamp(a.clk, BHN_CLK_CHAIN, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const int pl = lane & 31, hh = lane >> 5;

    // ---- per-lane address parts (bytes) ------------------------------------------------------------------------------
    // row access of point block 2j + pi: 256 (32 pb + n) + 16 ((half ^ swz) & 15); chunk pair C (even) is reached by ^ 16 C
    unsigned rowb[2];
    {
        const int swz = ((pl & 3) << 2) | (((pl >> 2) & 3) ^ (pl & 2));
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) rowb[pi] = 256u * (32 * (2 * wj + pi) + pl) + 16u * ((hh ^ swz) & 15);
    }
    // transposed read of feature tile T: first / second 4-point block of the lane's 8 points of a 16-point k-step
    const int tg = lane >> 4, tcg = tg & 1, tkh = tg >> 1, tq = (lane & 15) >> 2, tp = lane & 3;
    auto tr_first = [&](int T) -> unsigned {
        const int lc = 2 * tcg + (tp >> 1), swa = (tq << 2) | ((2 * tkh) ^ (tq & 2));          // swz(row 8 tkh + tq)
        return 256u * (8 * tkh + tq) + 16u * (((4 * T) ^ lc ^ swa) & 15) + 8u * (tp & 1);
    };
    auto tr_second = [&](int T) -> unsigned {
        const int lc = 2 * tcg + (tp >> 1), swb = (tq << 2) | ((2 * tkh + 1) ^ (tq & 2));      // swz(row 8 tkh + tq + 4)
        return 256u * (8 * tkh + tq + 4) + 16u * (((4 * T) ^ lc ^ swb) & 15) + 8u * (tp & 1);
    };
    // this wave's tiles: chain / dW rows m0 = 2i, m1 = 2i + 1; dW columns n0 = 2j, n1 = 2j + 1; enc tile with m_e = 2i + j;
    // layer 0 and the output layer: tile wv
    const unsigned trA_m0 = tr_first(2 * wi), trB_m0 = tr_second(2 * wi), trA_m1 = tr_first(2 * wi + 1), trB_m1 = tr_second(2 * wi + 1);
    const unsigned trA_n0 = tr_first(2 * wj), trB_n0 = tr_second(2 * wj), trA_n1 = tr_first(2 * wj + 1), trB_n1 = tr_second(2 * wj + 1);
    const unsigned trA_w = tr_first(wv), trB_w = tr_second(wv);
    const unsigned trA_me = tr_first(2 * wi + wj), trB_me = tr_second(2 * wi + wj);
    const unsigned trE = 64u * (8 * tkh + tq) + 32u * tcg + 8u * tp;      // encoded-input image: 64-byte rows, no swizzle

    // LDS-DMA lane offsets on the GLOBAL side (tape tiles are in the producer's slot layout, TapeEmit::native_off)
    // (LDS lane i of piece k lands on row 4 k + (i >> 4), chunk position i & 15, i.e. feature chunk (i & 15) ^ swz(row) with
    //  row & 3 = i >> 4, (row >> 2) & 3 = wv: tile = chunk >> 2, fragment (s, h) = chunk & 3)
    const unsigned voffH = 2048u * ((((lane & 15) >> 2) ^ (lane >> 4)) & 3) + 64u * (((lane & 3) ^ wv ^ ((lane >> 4) & 2)) & 3) + 16u * (lane >> 4);
    const unsigned voffE = 16u * ((lane >> 2) & 3) + 64u * (lane & 3) + 256u * (lane >> 4);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(smem));

    const long long nquads = A.t.NQ >> 2;
    const long long h_stride = A.t.lin_stride;
    // image of h_l (l = 1..DEPTH) of quad Q -> H buffer hb (0 / 1): 32 pieces of 4 rows, wave w issues pieces w + 4 t
    auto dma_h = [&](int l, long long Q, int hb) {
        const u32x4 rs = make_rsrc(A.tape + A.t.h_lin + (long long)l * h_stride + Q * (4ll * MT * TB));
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int pb = t >> 1, kk = wv + 4 * (t & 1), k = wv + 4 * t;
            dma_piece(rs, (unsigned)(pb * (MT * TB) + 256 * kk), lds0 + OFF_H + hb * IMG + 1024 * k, voffH);
        }
    };
    // the same, one piece at a time (issued from the MFMA shadows of the phases): rs from h_rsrc(l, Q)
    auto h_rsrc = [&](int l, long long Q) { return make_rsrc(A.tape + A.t.h_lin + (long long)l * h_stride + Q * (4ll * MT * TB)); };
    auto dma_h_piece = [&](const u32x4 &rs, int hb, int t) {
        const int pb = t >> 1, kk = wv + 4 * (t & 1), k = wv + 4 * t;
        dma_piece(rs, (unsigned)(pb * (MT * TB) + 256 * kk), lds0 + OFF_H + hb * IMG + 1024 * k, voffH);
    };
    auto dma_enc = [&](long long Q, int eb) {
        const u32x4 rs = make_rsrc(A.tape + A.t.enc_off + Q * (4ll * TB));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = wv + 4 * t, pb = e >> 1, half = e & 1;
            dma_piece(rs, (unsigned)(pb * TB + 1024 * half), lds0 + OFF_E + eb * ENC_IMG + 1024 * e, voffE);
        }
    };
    // dout of point pl of group 4 Q + wv: dout128_kernel has made dout = dE e (1 - e) from the tape's e (a region of its own: the
    // recorded tape stays as the forward left it, a second backward call on it gives the same gradient)
    const float *dout_g = reinterpret_cast<const float *>(A.tape + A.t.dout_off);
    auto point_dout = [&](long long Q) -> float { return dout_g[(4 * Q + wv) * 32 + pl]; };
    auto put_dout = [&](float d) {
        if (lane < 32) reinterpret_cast<float *>(smem + OFF_DOUT)[32 * wv + pl] = d;
    };
    // relu bits of the last hidden layer (TapeLayout::drop_hd: recorded by the forward in place of the h_depth tiles), point blocks
    // 2j, 2j+1 of quad Q: the word of THIS lane (point pl, half hh -- the lane that held the point in the forward) for row tiles
    // 2i (low 16 bits) and 2i+1 (high): bit k = accumulator element 2k, bit 8+k = element 2k+1 (Pol::mask_code)
    const unsigned *maskd_g = reinterpret_cast<const unsigned *>(A.tape + A.t.maskd_off);
    auto load_maskd = [&](long long Qx, unsigned (&mw)[2]) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) mw[pi] = maskd_g[((4 * Qx + 2 * wj + pi) * 2 + wi) * 64 + lane];
    };
    // this wave's 16 fragments of the transposed weight image of hidden layer l (rows m0, m1), plain loads from L2
    // (buffer loads: one resource over the image, the fragment as a scalar offset, 16 lane -- no 64-bit address per fragment:
    //  with plain pointers hipcc kept fourteen of them across the loop, in scratch)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(a.packed + a.bwd_off + (size_t)(2 * wi) * CB), 0, 0x7fffffff, 0x00020000);
    auto load_w1 = [&](int l, int mi, int ks) -> frag {
        return __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, ((l - 1) * MT + mi) * CB + ks * 1024, 0));
    };
    auto load_w = [&](int l, frag (&wf)[2][KS]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[mi][ks] = load_w1(l, mi, ks);
    };
    auto use_w = [&](frag (&wf)[2][KS]) {          // the loads have landed (the compiler waits here, at a drain point of the schedule)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ks = 0; ks < KS; ks += 4) asm volatile("" : "+v"(wf[mi][ks]), "+v"(wf[mi][ks + 1]), "+v"(wf[mi][ks + 2]), "+v"(wf[mi][ks + 3]));
    };
#ifndef BHN_B128_ABL
#define BHN_B128_ABL 0              // measurement builds (results wrong): 1 no dW MFMAs, 2 dW fragments read once per phase, 4 no chain MFMAs, 8 no tape DMA, 16 no chain epilogue, 32 no front, 64 no barriers
#endif
#ifndef BHN_B128_STAMPS
#define BHN_B128_STAMPS 0           // 1 (measurement build): s_memtime stamps of one iteration of workgroup 0 -> slab tile 64 (tools/dbg_bwd128_stamps.py)
#endif
    long long *ts = nullptr;
    int ts_i = 0;
    auto stamp = [&]() {
        if constexpr (BHN_B128_STAMPS != 0) {
            if (ts) { const long long t = __builtin_readcyclecounter(); if (lane == 0) ts[ts_i] = t; ++ts_i; }
        }
    };
    auto drain_and_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp();
        if (!(BHN_B128_ABL & 64)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp();
    };

    // ---- accumulators: hidden layer l = 1..DEPTH-1: (m0,n0) (m0,n1) (m1,n0) (m1,n1); the skip layer's (m_e, enc) tile;
    //      layer 0: (wv, enc).  The enc tile's column 31 is the bias gradient (slot 31 of the recorded inputs is 1); the
    //      other hidden layers sum their bias from the A fragments (v_dot2 against (1, 1): bsum) ----
    f32x16 acc[DEPTH - 1][4], acc_e, acc0;
#pragma unroll
    for (int l = 0; l < DEPTH - 1; ++l)
#pragma unroll
        for (int t = 0; t < 4; ++t) { const f32x16 z = {}; acc[l][t] = z; }
    { const f32x16 z = {}; acc_e = z; acc0 = z; }
    float bsum[DEPTH - 1];
#pragma unroll
    for (int l = 0; l < DEPTH - 1; ++l) bsum[l] = 0.f;
    float bout = 0.f;

    // ---- phases ------------------------------------------------------------------------------------------------------
    // delta chain through hidden layer l, point block 2j + pi: tiles m0, m1 of W_l gA_l (ga_in: image offset of gA_l) ...
    // `side(ks)`: work of other phases issued in the shadow of this step's MFMAs (one wave per SIMD: nothing else hides it)
    auto chain_mma = [&](const frag (&wf)[2][KS], unsigned ga_in, int pi, f32x16 (&c)[2], auto &&side) {
        { const f32x16 z = {}; c[0] = z; c[1] = z; }
#ifndef BHN_B128_CPF
#define BHN_B128_CPF 3              // chain B fragments in flight
#endif
        constexpr int CPF = BHN_B128_CPF;
        frag bq[CPF + 1];
#pragma unroll
        for (int i = 0; i < CPF; ++i) bq[i] = lds_row(smem, ga_in + (rowb[pi] ^ (32u * i)));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + CPF < KS) bq[(ks + CPF) % (CPF + 1)] = lds_row(smem, ga_in + (rowb[pi] ^ (32u * (ks + CPF))));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BHN_B128_ABL & 4) { asm volatile("" :: "v"(bq[ks % (CPF + 1)])); }
            else {
            c[0] = Pol::mma(wf[0][ks], bq[ks % (CPF + 1)], c[0]);
            c[1] = Pol::mma(wf[1][ks], bq[ks % (CPF + 1)], c[1]);
            }
            side(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (hipcc places these two tiles in the 32 AGPRs the dW accumulators leave free and reads them out for the epilogue, 32
        //  v_accvgpr_read per point block; forcing them into VGPRs inside the loop made it copy back and forth every k-step)
        asm volatile("" : "+v"(c[0]), "+v"(c[1]));
    };
    // ... and their epilogue: gA_{l-1} = relu'(a_{l-1}) (.) c, relu' read off the h_l image (h_l = relu(a_{l-1}) as bf16)
    // the same epilogue in eight slices (slice j: row tile j >> 2, k-step (j >> 1) & 1, half j & 1 of its eight elements),
    // issued from the MFMA shadows of the following phase; the h chunk of a slice pair is read one slice ahead
    u32x4 ps_hv, ps_hvn, ps_o;
    auto post_off = [&](int pi, int jp) -> unsigned { return rowb[pi] ^ (16u * (4 * (2 * wi + (jp >> 1)) + 2 * (jp & 1))); };
    auto post_begin = [&](int pi, unsigned h_img) { ps_hvn = __builtin_bit_cast(u32x4, lds_row(smem, h_img + post_off(pi, 0))); };
    auto post_slice = [&](const f32x16 (&c)[2], int pi, unsigned h_img, unsigned ga_out, int j) {
        const int mi = j >> 2, s2 = (j >> 1) & 1, half = j & 1;
        if (half == 0) ps_hv = ps_hvn;
        else if (j + 1 < 8) ps_hvn = __builtin_bit_cast(u32x4, lds_row(smem, h_img + post_off(pi, (j + 1) >> 1)));
#pragma unroll
        for (int d = 2 * half; d < 2 * half + 2; ++d) ps_o[d] = keep_where_nz(Pol::pack_a(c[mi][8 * s2 + 2 * d], c[mi][8 * s2 + 2 * d + 1]), ps_hv[d]);
        if (half == 1) lds_put(smem, ga_out + post_off(pi, j >> 1), __builtin_bit_cast(frag, ps_o));
    };
    // dW of a hidden layer: t[0..3] += gA^T h over the wave's 2 x 2 tiles; ENC: te += gA[m_e]^T enc, else bs += the bias of
    // row tile m_e (sum over the points of the A fragments); K = the 128 points
    auto dw_phase = [&](auto enc_tag, unsigned ga_img, unsigned h_img, unsigned e_img, f32x16 (&t)[4], f32x16 &te, float &bs, auto &&side) {
        constexpr bool ENC = decltype(enc_tag)::value;
        // ae: row tile m_e = 2i + j once more, read through its own address (a select between a0 and a1 costs four v_cndmask
        // per k-step; a wave-uniform branch around the MFMA made hipcc wait for the MFMA and move the accumulator: 3.7k cycles
        // per phase against 1.7k)
        struct KF { frag a0, a1, b0, b1, ae, be; };
        auto fetch = [&](int k) {
            KF f;
            const unsigned ko = 4096u * k;
            f.a0 = tr2(smem, ga_img + trA_m0 + ko, ga_img + trB_m0 + ko);
            f.a1 = tr2(smem, ga_img + trA_m1 + ko, ga_img + trB_m1 + ko);
            f.b0 = tr2(smem, h_img + trA_n0 + ko, h_img + trB_n0 + ko);
            f.b1 = tr2(smem, h_img + trA_n1 + ko, h_img + trB_n1 + ko);
            f.ae = tr2(smem, ga_img + trA_me + ko, ga_img + trB_me + ko);
            if constexpr (ENC) f.be = tr2(smem, e_img + trE + 1024u * k, e_img + trE + 1024u * k + 256u);
            else f.be = f.b0;
            return f;
        };
#ifndef BHN_B128_DWDB
#define BHN_B128_DWDB 0             // 1: the next k-step's fragments are fetched BEFORE this step's MFMAs (a second fragment set: 24 registers)
#endif
        KF cur = fetch(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            KF nx = cur;
            if constexpr (BHN_B128_DWDB != 0) { if (k + 1 < 8) nx = fetch(k + 1); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BHN_B128_ABL & 1) { asm volatile("" :: "v"(cur.a0), "v"(cur.a1), "v"(cur.b0), "v"(cur.b1), "v"(cur.be), "v"(cur.ae)); }
            else {
            t[0] = Pol::mma(cur.a0, cur.b0, t[0]);
            t[1] = Pol::mma(cur.a0, cur.b1, t[1]);
            t[2] = Pol::mma(cur.a1, cur.b0, t[2]);
            t[3] = Pol::mma(cur.a1, cur.b1, t[3]);
            }
            if constexpr (BHN_B128_ABL & 1) {}
            else if constexpr (ENC) te = Pol::mma(cur.ae, cur.be, te);
            else bs = Pol::sum8(cur.ae, bs);
            // the next k-step's fragments are fetched BEHIND this step's MFMAs (160 cycles of matrix work cover the LDS
            // latency; a second fragment set in flight costs 40 registers this kernel does not have)
            if constexpr (BHN_B128_DWDB != 0) cur = nx;
            else if (k + 1 < 8 && !(BHN_B128_ABL & 2)) cur = fetch(k + 1);
            side(k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // dW of layer 0: tile m = wv of gA_0^T enc
    auto dw0_phase = [&](unsigned ga_img, unsigned e_img, auto &&side) {
        frag af = tr2(smem, ga_img + trA_w, ga_img + trB_w);
        frag be = tr2(smem, e_img + trE, e_img + trE + 256u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            frag afn = af, ben = be;
            if (k + 1 < 8) {
                const unsigned ko = 4096u * (k + 1);
                afn = tr2(smem, ga_img + trA_w + ko, ga_img + trB_w + ko);
                ben = tr2(smem, e_img + trE + 1024u * (k + 1), e_img + trE + 1024u * (k + 1) + 256u);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = Pol::mma(af, be, acc0);
            side(k);
            __builtin_amdgcn_sched_barrier(0);
            af = afn; be = ben;
        }
    };
    // front: gA_{depth-1} (without W_out, common.h bhn_folds_wout) = relu'(a_{depth-1}) (.) bf16(dout) from the recorded relu bits
    // (round 5: no h_depth image any more; the output layer's row is made by the reduce from layer depth-1's own gradient,
    //  TapeLayout::drop_hd).  Chunk (tile T, k-step s2, half hh) of point pl holds accumulator elements 8 s2 .. 8 s2 + 7 of lane
    // (pl, hh): dword d = elements 8 s2 + 2 d, + 1 = bits 4 s2 + d and 8 + 4 s2 + d of the tile's 16-bit code.
    auto front = [&](const unsigned (&mw)[2], unsigned ga_out) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const float dvp = reinterpret_cast<const float *>(smem + OFF_DOUT)[32 * (2 * wj + pi) + pl];
            const Pol::bf16x2 d2 = {(__bf16)dvp, (__bf16)dvp};
            const unsigned dd = __builtin_bit_cast(unsigned, d2);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const unsigned code = mi ? mw[pi] >> 16 : mw[pi];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int lo = (int)(code << (31 - (4 * s2 + d))) >> 31, hi = (int)(code << (31 - (8 + 4 * s2 + d))) >> 31;
                        o[d] = dd & (((unsigned)lo & 0xffffu) | ((unsigned)hi & 0xffff0000u));
                    }
                    lds_put(smem, ga_out + (rowb[pi] ^ (16u * (4 * (2 * wi + mi) + 2 * s2))), __builtin_bit_cast(frag, o));
                }
            }
        }
    };

    // ---- h_1 = relu(W_0^T enc + b_0) recomputed into an h image (same operands, same order, same rounding as the forward's layer
    //      0: bit-identical to the tile the forward held): wave (i, j) makes row tiles 2i, 2i+1 of point blocks 2j, 2j+1 ----
    for (int i = tid; i < 8192 / 16; i += 256)
        reinterpret_cast<u32x4 *>(smem + OFF_W0)[i] = reinterpret_cast<const u32x4 *>(a.packed + a.fwd_off)[i];
    if (tid < 128) reinterpret_cast<float *>(smem + OFF_B0)[tid] = reinterpret_cast<const float *>(a.packed + a.bias_off)[tid];
    auto make_h1 = [&](unsigned e_img, unsigned h_out) {
        frag eb[2][2], wa[2][2];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) eb[pi][ks] = lds_row(smem, e_img + 64u * (32 * (2 * wj + pi) + pl) + 16u * (2 * ks + hh));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wa[mi][ks] = lds_row(smem, OFF_W0 + 1024u * (2 * (2 * wi + mi) + ks) + 16u * lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const f32x16 b0 = bias_acc(reinterpret_cast<const float *>(smem + OFF_B0), 2 * wi + mi, hh);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                f32x16 t = b0;
                t = Pol::mma(wa[mi][0], eb[pi][0], t);
                t = Pol::mma(wa[mi][1], eb[pi][1], t);
                asm volatile("" : "+v"(t));
                frag o[2];
                unsigned unused = 0;
#pragma unroll
                for (int r = 0; r < 16; r += 2) Pol::relu_pair(o[r >> 3], (r & 7) >> 1, r >> 1, t[r], t[r + 1], unused);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) lds_put(smem, h_out + (rowb[pi] ^ (16u * (4 * (2 * wi + mi) + 2 * s2))), o[s2]);
            }
        }
    };

    // ---- prelude: the first quad's h_depth, h_{depth-1}, enc; its dout; the top layer's weights --------------------------
    long long Q = blockIdx.x;
    frag wf[2][KS];
    float dnext = 0.f;
    int eb = 0;
    unsigned mnext[2] = {0u, 0u};                              // relu words of the quad whose top phase runs next
    if (Q < nquads) {
        dma_h(DEPTH - 1, Q, 1);
        dma_enc(Q, 0);
        dnext = point_dout(Q);
        load_maskd(Q, mnext);
        load_w(DEPTH - 1, wf);
        use_w(wf);
        if (lane < 32) bout += dnext;
        put_dout(dnext);
        drain_and_barrier();                                   // the first quad's images landed, its dout visible
        front(mnext, OFF_GA + 0 * IMG);                        // top phase of the first quad: gA_{D-1} -> GA0
        drain_and_barrier();
    }
    for (; Q < nquads; Q += gridDim.x) {
        // the accumulators stay in the AGPRs, in place, across the back edge (hipcc otherwise shuffles them through VGPRs)
#pragma unroll
        for (int l = 0; l < DEPTH - 1; ++l) asm volatile("" : "+a"(acc[l][0]), "+a"(acc[l][1]), "+a"(acc[l][2]), "+a"(acc[l][3]));
        asm volatile("" : "+a"(acc_e), "+a"(acc0));
        const bool has_next = Q + gridDim.x < nquads;
        const long long Qn = has_next ? Q + gridDim.x : Q;     // the last iteration re-loads its own quad (with dout = 0: no contribution)
        if constexpr (BHN_B128_STAMPS != 0) {
            ts = (blockIdx.x == 0 && Q == blockIdx.x + 100ll * gridDim.x) ? reinterpret_cast<long long *>(a.slabs + 64 * 1024 + 256) + 64 * wv : nullptr;
            ts_i = 0;
            stamp();
        }
        const unsigned e_img = OFF_E + eb * ENC_IMG;
        // (the top phase of THIS quad -- gA_{D-1} -> GA0, dW_out -- ran at the end of the previous iteration, beside layer 0's dW)
        // Layers D-1 .. 1.  Buffers alternate: gA_l in GA[(D-1-l) & 1], h_l in H[(D-l) & 1]; the h image that the mask of
        // layer l+1 has just released takes h_{l-1} (or the next quad's h_D).
#pragma unroll
        for (int l = DEPTH - 1; l >= 1; --l) {
            const int gi = (DEPTH - 1 - l) & 1, hi = (DEPTH - l) & 1;
            const unsigned ga_in = OFF_GA + gi * IMG, ga_out = OFF_GA + (gi ^ 1) * IMG, h_img = OFF_H + hi * IMG;
            // the h image that layer l+1 has released takes h_{l-1} (l == 1: the next quad's h_D), piece by piece under the
            // first chain step's MFMAs; the epilogue of point block 0 runs under the MFMAs of point block 1, that of point block
            // 1 and the loads of the next layer's weights under the dW MFMAs
            const u32x4 rs_h = h_rsrc(l - 1 >= 2 ? l - 1 : 2, Q);
            if (l == 2) make_h1(e_img, OFF_H + (hi ^ 1) * IMG);          // (the image h_3 has left; published by this layer's barrier)
            if (l == 1) { dnext = point_dout(Qn); load_maskd(Qn, mnext); }   // loads issued here, consumed behind this layer's dW phase / in the next top phase
            stamp();
            f32x16 c0[2], c1[2];
            chain_mma(wf, ga_in, 0, c0, [&](int ks) { if (l - 1 >= 2) dma_h_piece(rs_h, hi ^ 1, ks); });
            stamp();
            if constexpr (BHN_B128_ABL & 16) {
                chain_mma(wf, ga_in, 1, c1, [&](int) {});
            } else {
                post_begin(0, h_img);
                chain_mma(wf, ga_in, 1, c1, [&](int ks) { post_slice(c0, 0, h_img, ga_out, ks); });
                post_begin(1, h_img);
            }
            stamp();
            const int ln = (l - 1 >= 1) ? l - 1 : DEPTH - 1;  // next layer's weights (this layer's are dead behind the chain MFMAs)
            auto side_dw = [&](int k) {
                if constexpr (!(BHN_B128_ABL & 16)) post_slice(c1, 1, h_img, ga_out, k);
                if (k < 4) {                                  // (all sixteen issued in the first half: the last ones have four k-steps to land)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) { wf[mi][2 * k] = load_w1(ln, mi, 2 * k); wf[mi][2 * k + 1] = load_w1(ln, mi, 2 * k + 1); }
                }
            };
            if (l == SKIPL) dw_phase(std::true_type{}, ga_in, h_img, e_img, acc[l - 1], acc_e, bsum[l - 1], side_dw);
            else dw_phase(std::false_type{}, ga_in, h_img, e_img, acc[l - 1], acc_e, bsum[l - 1], side_dw);
            stamp();
            use_w(wf);
            if (l == 1) {
                asm volatile("" : "+v"(dnext));                // (the compiler's wait for this load belongs here, at a drain point)
                const float d = has_next ? dnext : 0.f;        // the next quad's dout: DOUT was last read by this quad's top phase
                if (lane < 32) bout += d;
                put_dout(d);
            }
            stamp();
            drain_and_barrier();                               // gA_{l-1} complete; the h image issued above has landed
        }
        // ---- layer 0: dW_0 = gA_0^T enc; the images of the next quad fly under it ----
        {
            const int gi0 = (DEPTH - 1) & 1, hfree = (DEPTH - 1) & 1;          // gA_0 image; H image that held h_1
            const u32x4 rs_h = h_rsrc(DEPTH - 1, Qn);
            dma_enc(Qn, eb ^ 1);
            stamp();
            dw0_phase(OFF_GA + gi0 * IMG, e_img, [&](int k) { dma_h_piece(rs_h, hfree, k); });
            stamp();
            // ... and the NEXT quad's top phase beside it (its h_D landed behind layer 1's barrier; GA0 was last read by layer 1):
            // one barrier and one drain fewer per iteration than a top phase of its own
            if (!(BHN_B128_ABL & 32)) front(mnext, OFF_GA + 0 * IMG);
            stamp();
            drain_and_barrier();                               // GA0 complete; h_{D-1} (H1) and enc of the next quad landed
        }
        eb ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- flush: slab[tile][r >> 2][lane][r & 3] (the layout reduce128_kernel reads) -------------------------------------
    float *slab = a.slabs + (long long)blockIdx.x * SLAB_FLOATS128;
    auto flush_tile = [&](int tile, const f32x16 &t) {
        float *tp = slab + (long long)tile * 1024;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t[4 * g4 + e];
            f32x4 *dst = reinterpret_cast<f32x4 *>(tp + g4 * 256 + lane * 4);
            if (A.accumulate) {
                const f32x4 old = *dst;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += old[e];
            }
            *dst = v;
        }
    };
#pragma unroll
    for (int l = 1; l < DEPTH; ++l) {
        const int base = (l - 1) * 20;                          // tile (l, m, n) = (l-1) 20 + 5 m + n
        flush_tile(base + 5 * (2 * wi) + 2 * wj, acc[l - 1][0]);
        flush_tile(base + 5 * (2 * wi) + 2 * wj + 1, acc[l - 1][1]);
        flush_tile(base + 5 * (2 * wi + 1) + 2 * wj, acc[l - 1][2]);
        flush_tile(base + 5 * (2 * wi + 1) + 2 * wj + 1, acc[l - 1][3]);
        if (l == SKIPL) flush_tile(base + 5 * (2 * wi + wj) + 4, acc_e);
        else {
            // bias of row tile m_e: virtual row (lane & 31), halves added; stored as column 31 of row (lane & 31) of tile (l, m_e, 4):
            // the place the enc tile's bias column has (register r of lane 31 + 32 h holds row (r & 3) + 8 (r >> 2) + 4 h)
            float v = bsum[l - 1] + __shfl_xor(bsum[l - 1], 32, 64);
            float *tp = slab + (long long)(base + 5 * (2 * wi + wj) + 4) * 1024;
            if (lane < 32) {
                const int row = lane, h2 = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
                float *dst = tp + (r >> 2) * 256 + (31 + 32 * h2) * 4 + (r & 3);
                if (A.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
    flush_tile(60 + wv, acc0);
    {   // tile 64: [128 + wv] this wave's share of the output layer's bias (its row: reduce128_kernel, from layer depth-1's gradient)
        float *tp = slab + 64ll * 1024;
        float bs = bout;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) bs += __shfl_xor(bs, o, 64);
        if (lane == 0) {
            float *dst = tp + 128 + wv;
            if (A.accumulate) bs += *dst;
            *dst = bs;
        }
    }
    clock_stamp(a.clk, BHN_CLK_CHAIN, 1);
}

// dout = dE e (1 - e) per point from the e the forward recorded (sigmoid'(out - 10) = e (1 - e); dE = sum_s dimg w) -> the tape's
// dout region
__global__ __launch_bounds__(256) void dout128_kernel(BwdArgs A) {
    const FusedArgs &a = A.f;
    const long long n = A.t.NQ * 32;
    const float *eg = reinterpret_cast<const float *>(A.tape + A.t.e_off);
    float *dg = reinterpret_cast<float *>(A.tape + A.t.dout_off);
    if (blockIdx.x == 0 && threadIdx.x == 0)            // the arrival counter of reduce128_kernel's output-row blocks (this kernel runs first in every backward call)
        *reinterpret_cast<unsigned *>(reinterpret_cast<float *>(A.tape + A.t.scratch_off) + 5 * 128) = 0u;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long q = i >> 5;
        int b; long long p; bool inb;
        // (groups per forward tile: 12 (PolBF16X) or 8 -- constants, so that the division is a multiply: a 64-bit run-time
        //  division here cost this kernel 0.08 ms at config 2)
        const unsigned qu = (unsigned)q;
        if (A.fwd_nw == 12) tile_point_nw(a, qu / 12u, (int)(qu % 12u), (int)(i & 31), 12, b, p, inb);
        else tile_point_nw(a, qu >> 3, (int)(qu & 7u), (int)(i & 31), 8, b, p, inb);
        const float e = eg[i];
        float d = 0.f;
        if (inb && e != 0.f) {
            const long long ray = a.ray_idx ? (long long)a.ray_idx[p] : (long long)a.fd_G.div((unsigned)p);
            float dE = 0.f;
            for (int s = 0; s < a.Sx; ++s) dE += a.dimages[((long long)b * a.Sx + s) * a.R + ray] * a.w[(long long)s * a.P + p];
            d = dE * e * (1.f - e);
        }
        dg[i] = d;
    }
}

// feature of virtual position v (0..31) of a 32-feature tile: chunk (s, h) = v >> 3, element e = v & 7 of the canonical fragment
__host__ __device__ inline int virt_feature(int v) { return 16 * ((v >> 4) & 1) + 4 * ((v >> 3) & 1) + (v & 3) + 8 * ((v >> 2) & 1); }

// Stage 1 of the slab reduction: block (tile, part) adds the slabs [part * per, (part + 1) * per) of its tile in order and leaves
// the partial sum in the first slab of its range (its own tile of its own range: no block reads what another one writes).
// 65 x 8 blocks instead of 65: the sum over 256 slabs of 260 KB each is latency-bound per thread.
__global__ __launch_bounds__(256) void reduce128_stage1(BwdArgs A, int nslabs, int per) {
    const int tile = blockIdx.x, part = blockIdx.y, s0 = part * per, s1 = (s0 + per < nslabs) ? s0 + per : nslabs;
    if (s0 >= s1) return;
    float *base = A.f.slabs + (long long)tile * 1024 + threadIdx.x * 4;
    f32x4 sum = *reinterpret_cast<const f32x4 *>(base + (long long)s0 * SLAB_FLOATS128);
    for (int wg = s0 + 1; wg < s1; ++wg) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long long)wg * SLAB_FLOATS128);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
    *reinterpret_cast<f32x4 *>(base + (long long)s0 * SLAB_FLOATS128) = sum;
}

// Sum the slabs of all workgroups (fixed order: bitwise reproducible) and write the flat gradient (flax tree order).
// One block per slab tile; thread (g4, lane) owns the float4 at [g4][lane] of the tile: accumulator registers 4 g4 .. 4 g4 + 3 of
// lane `lane`, i.e. rows (virtual output position) e + 8 g4 + 4 (lane >> 5), column (virtual input position) lane & 31.
template <int DEPTH>
__global__ __launch_bounds__(256) void reduce128_kernel(BwdArgs A, int nslabs, int step) {      // step: stride of the slabs that hold stage-1 sums
    const int tile = blockIdx.x, tid = threadIdx.x, g4 = tid >> 6, lane = tid & 63, hh = lane >> 5, col = lane & 31;
    const float *src = A.f.slabs + (long long)tile * 1024 + g4 * 256 + lane * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int wg = 0; wg < nslabs; wg += step) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (long long)wg * SLAB_FLOATS128);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
    const int WT = A.width_true;
    const float *wout = reinterpret_cast<const float *>(A.f.packed + A.f.wout_off);
    if (tile == 64) {                                            // output layer: floats 128..131 = the four waves' shares of its bias
        if (tid == 32) A.dparams[A.bias_off[DEPTH]] = (sum[0] + sum[1]) + (sum[2] + sum[3]);
        return;
    }
    int l, m, n;
    if (tile >= 60) { l = 0; m = tile - 60; n = 4; }
    else { l = 1 + tile / 20; m = (tile % 20) / 5; n = tile % 5; }
    if (l >= DEPTH) return;
    const bool skip = (A.f.skip_mask >> l) & 1;
    const bool fold = l == DEPTH - 1 && bhn_folds_wout(BHN_BF16, DEPTH);
    if (l == DEPTH - 1) {
        // The output layer's row from THIS layer's gradient (TapeLayout::drop_hd): h_D = relu(a) = relu'(a) a, a = K^T [h | enc] + b, so
        //   dW_out[o] = sum_p dout_p h_D[p][o] = sum_k K[k][o] G[k][o] + b[o] g[o],   G, g = the un-folded sums of this tile's
        // layer (gA without its W_out factor).  K = the bf16 weights the forward multiplied with (packed image), b in f32.
        // This block adds its 32 input columns; the last block of the layer adds the five input tiles of every row, in order.
        const char *wimg = A.f.packed + A.f.fwd_off + (size_t)(1 + (l - 1) * MT + m) * CB;          // chunk of output tile m
        const float *bias = reinterpret_cast<const float *>(A.f.packed + A.f.bias_off) + l * 128;
        float *scr = reinterpret_cast<float *>(A.tape + A.t.scratch_off) + n * 128;
        const int q = n < 4 ? 32 * n + virt_feature(col) : virt_feature(col);      // hidden input unit / encoded-input slot of this column
        const int fr = n < 4 ? q >> 4 : KS + (q >> 4), ph = q & 15, h2 = (ph >> 2) & 1, jj = (ph & 3) + 4 * (ph >> 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = e + 8 * g4 + 4 * hh, i = virt_feature(v);
            float wq;
            if (n == 4 && q == 31) wq = bias[32 * m + i];
            else if (n == 4 && !skip) wq = 0.f;
            else wq = (float)reinterpret_cast<const __bf16 *>(wimg + fr * 1024 + (i + 32 * h2) * 16)[jj];
            float part = wq * sum[e];
#pragma unroll
            for (int o2 = 16; o2 > 0; o2 >>= 1) part += __shfl_xor(part, o2, 64);
            if (col == 0) scr[32 * m + i] = part;
        }
        // the LAST of the layer's twenty blocks to get here adds the five partial sums of every row, in order (a fixed order: the
        // result does not depend on which block is last) -- a third kernel for 128 floats cost 3 us per step, 3 % of a config-5 share
        __shared__ unsigned ticket;
        __threadfence();
        __syncthreads();
        unsigned *count = reinterpret_cast<unsigned *>(reinterpret_cast<float *>(A.tape + A.t.scratch_off) + 5 * 128);   // zeroed by dout128_kernel
        if (tid == 0) ticket = atomicAdd(count, 1u);
        __syncthreads();
        if (ticket == 4 * 5 - 1) {
            __threadfence();
            const volatile float *all = reinterpret_cast<const volatile float *>(A.tape + A.t.scratch_off);
            if (tid < 128) {
                float s = all[tid];
#pragma unroll
                for (int k = 1; k < 5; ++k) s += all[k * 128 + tid];
                if (tid < WT) A.dparams[A.kernel_off[DEPTH] + tid] = s;
            }
        }
    }
    // input of this column
    long long kin = -1;
    bool is_bias = false;
    if (n < 4) {
        const int k = 32 * n + virt_feature(col);
        if (k < WT) kin = k;
    } else {
        const int slot = virt_feature(col);
        if (slot == 31) is_bias = true;
        else {
            const int fe = bhn_enc_slot_feature(slot, A.f.deg);
            if (fe >= 0 && (l == 0 || skip)) kin = (l == 0 ? 0 : WT) + fe;
        }
    }
    if (kin < 0 && !is_bias) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = e + 8 * g4 + 4 * hh;                       // row of the accumulator = virtual output position
        const int o = 32 * m + virt_feature(v);
        if (o >= WT) continue;
        float val = sum[e];
        if (fold) val *= wout[o];
        A.dparams[is_bias ? A.bias_off[l] + o : A.kernel_off[l] + kin * WT + o] = val;
    }
}

}   // namespace

// ---- host side (called from bwd_run of fused_bwd.hip) --------------------------------------------------------------------
bool bwd128_supported(int mode, int kernel_width, int depth) {
#ifdef BHN_NO_FUSED128
    return false;
#else
    return mode == BHN_BF16 && kernel_width == 128 && depth == 4;
#endif
}

size_t bwd128_slab_bytes(int grid) { return (size_t)grid * SLAB_FLOATS128 * 4; }

void bwd128_tape_layout(int depth, long long NQ, TapeLayout *t) {
    memset(t, 0, sizeof(*t));
    t->NQ = NQ;
    t->fused128 = 1;
#ifndef BHN_F128_PAD
#define BHN_F128_PAD 0             // (measurement builds: bytes between the h tensors of the tape beyond their size)
#endif
    const long long per_tensor = NQ * (long long)MT * TB + BHN_F128_PAD;
    long long off = 0;
    t->drop_h1 = 1;                                              // h_1 = relu(W_0^T enc + b_0) is recomputed by the backward (2 MFMAs per tile)
    t->h_off[1] = -1;
    t->drop_hd = 1;                                              // h_depth: its relu bits instead (bwd_common.h)
    for (int l = 2; l < depth; ++l) { t->h_off[l] = off; off += per_tensor; }
    t->h_off[depth] = -1;
    t->maskd_off = off; off += NQ * 2ll * 256;                  // [group][tile pair][lane] words
    for (int l = 0; l < depth; ++l) t->ga_off[l] = -1;
    t->lin_stride = per_tensor;
    t->h_lin = -2 * per_tensor;                                  // h_off[l] = h_lin + l * lin_stride, l >= 2
    t->enc_off = off; off += NQ * (long long)TB;
    t->e_off = off; off += NQ * 128;
    t->dout_off = off; off += NQ * 128;                         // f32 dout per point (dout128_kernel), beside the recorded e
    t->dout_stride = 128;
    t->mask_off = -1; t->encp_off = -1;
    off = (off + 255) / 256 * 256;
    t->scratch_off = off; off += 5 * 128 * 4 + 256;             // + the arrival counter
    t->total = (long long)(((size_t)off + 1024 + 255) / 256 * 256);
}

int bwd128_launch(const BwdArgs &A, int depth, int grid, hipStream_t st) {
    static DeviceOnce once;
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    BHN_CHECK_DEVICE(dev);
    BHN_HIP(once.run(dev, [&](int &) {
        return hipFuncSetAttribute((const void *)&bwd128_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }));
    BHN_CHECK_ARG(depth == 4 && (A.t.NQ & 3) == 0, "fused width-128 backward: depth %d, %lld groups", depth, A.t.NQ);
    {
        long long nb = (A.t.NQ * 32 + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(dout128_kernel, dim3((unsigned)nb), dim3(256), 0, st, A);
        BHN_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((bwd128_kernel<4, 3>), dim3((unsigned)grid), dim3(256), LDS_BYTES, st, A);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

int reduce128_launch(const BwdArgs &A, int depth, int nslabs, hipStream_t st) {
    BHN_CHECK_ARG(depth == 4, "fused width-128 backward: depth %d", depth);
    const int parts = 8, per = (nslabs + parts - 1) / parts;
    hipLaunchKernelGGL(reduce128_stage1, dim3(SLAB_TILES, parts), dim3(256), 0, st, A, nslabs, per);
    BHN_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce128_kernel<4>, dim3(SLAB_TILES), dim3(256), 0, st, A, nslabs, per);
    BHN_HIP(hipGetLastError());
    return BHN_OK;
}

// Reverse of the fused render w.r.t. the MLP parameters (jax.value_and_grad of loss_fn_image,
// network.py:617).  Three kernels (DESIGN.md "backward"):
//
//   chain_kernel  per 32-point wave tile: (re)compute the forward, dE = sum_s dimg*w, dout = dE*e*(1-e),
//                 delta chain gA_{l-1} = (W_l gA_l) * relu'(a_{l-1}) with the same register-chained MFMA
//                 structure and software-pipelined ring steps as the forward (fused_common.h).  Every 32x32
//                 tile of h_l (layer inputs) and gA_l (gradients w.r.t. pre-activations) is transposed by two
//                 MFMAs against identity fragments (TapeEmit) and streamed to an HBM "tape" already in MFMA
//                 fragment order [lane = feature][8 points].
//   dw_kernel     weight-gradient GEMMs dW_l^T[out x in] = sum_points gA_l^T . in_l with K = points:
//                 a workgroup owns ONE layer's whole dW^T in registers (<= 288 KB) and streams its
//                 share of the tape through LDS; bias gradients come from a constant ones B-fragment.
//                 One slab flush per workgroup at the end: no float atomics, deterministic.
//   reduce_kernel sums the slabs of each layer's workgroups into the flat dparams (flax tree order).
#include <utility>
#include <type_traits>
#include "fused_common.h"
#include "bwd_common.h"

#ifndef BHN_CHAIN_STAMPS
#define BHN_CHAIN_STAMPS 0      // 1: ring-step time stamps in the chain kernels (tools/dbg_chain_steps.py needs this build)
#endif
#ifndef BHN_JOBLB_W
#define BHN_JOBLB_W 6           // LBITS: weight of that job's compute (mask expansion, the output row at the flush) in B tiles at width 256
#endif
#ifndef BHN_DROP_HD
#define BHN_DROP_HD 0           // with LBITS, generic tape path: the training forward does not store the h_depth tiles (TapeLayout::drop_hd).
#endif                          // OFF, as is LBITS: 9 GB less tape traffic per step at 4x256 (48.4 -> 39.5) and NOT faster -- the dW kernel does not get
                                // faster with 18 % fewer bytes (it is issue-bound per group, not HBM-bound) and every variant of the forward's
                                // change costs its ring steps 3 % (profiles/r5_ab_drop_hd_width256.txt).  The fused 4x128 path has it on (bwd128).
#ifndef BHN_LBITS
#define BHN_LBITS 0             // 1: the dW job of layer depth-1 works from the relu bits (dw_body2 LBITS) instead of the h_depth tiles (see BHN_DROP_HD)
#endif
#ifndef BHN_JOB1_W
#define BHN_JOB1_W 13          // weight of the layer-1 dW job in B tiles at width 256 (measured optimum; scaled with the width)
#endif
#ifndef BHN_JOBL_W
#define BHN_JOBL_W 7            // extra weight (in tiles at width 256, scaled with the width) of the layer depth-1 dW job when it rebuilds gA_{depth-1} and carries the output row
                                // (round 5, re-swept without the layer-0 job, profiles/r5_dw_job_weights.txt: 6-7 beat 8 by 1.2 % of the dW kernel)
#endif
// Run-time measurement switches exist only in the debug build (make debug); the release kernels see the constant 0
#ifdef BHN_DEBUG
#define BHN_DBG(x) (x)
#else
#define BHN_DBG(x) 0
#endif
#ifndef BHN_GA0_CHAIN
#define BHN_GA0_CHAIN 1          // bf16, width 256, depth >= 3: the delta chain accumulates dW_0 itself (TapeLayout::ga0_chain); 0: A/B builds
#endif
#ifndef BHN_GA0C_ABL
#define BHN_GA0C_ABL 0           // measurement builds (dW_0 wrong): 1 no consumer (no extra MFMAs / staged-tile reads), 2 no staging writes, 4 no encoded-input DMA
#endif
#ifndef BHN_GA0C_DIST
#define BHN_GA0C_DIST 4          // weight chunks in flight in that delta chain (its LDS also holds 64 KB of staging images)
#endif
#ifndef BHN_GA0C_SWZ
#define BHN_GA0C_SWZ 1           // staging images of the gA_0 tiles with swizzled rows (chain_kernel: stage_off); 0: A/B builds
#endif
#ifndef BHN_TAPED_DIST
#define BHN_TAPED_DIST 7         // weight chunks in flight in the training-forward / delta-chain kernels (bf16; 4 measured 2 % slower)
#endif

template <int W, class Pol>
struct BwdGeom {
    static constexpr int MT = W / 32;
    static constexpr int TILE_BYTES = 2 * Pol::FRAG_BYTES;          // 32 features x 32 points
    // bytes of one h / gA tile ON THE TAPE: the 8-bit tape (Pol::TAPE8) halves these; the encoded-input tile stays bf16
    static constexpr int TAPE_TILE = Pol::TAPE8 ? 1024 : TILE_BYTES;
    static constexpr int NTMAX = MT + 2;                            // h tiles + enc tile + ones tile
    static constexpr int SLAB_FLOATS = (MT + 1) * NTMAX * 1024;     // row MT: the output layer's row when it rides on job depth-1
    // wave grid of the dW kernel
    static constexpr int WR = (MT >= 4) ? 4 : MT;
    static constexpr int WC = Pol::NWAVES / WR >= 1 ? Pol::NWAVES / WR : 1;
    static constexpr int WRR = (Pol::NWAVES < WR) ? Pol::NWAVES : WR;   // f32 policy has 4 waves
    static constexpr int WCC = Pol::NWAVES / WRR;
    static constexpr int MPW = (MT + WRR - 1) / WRR;
    static constexpr int NPW_ALL = (NTMAX + WCC - 1) / WCC;          // B tiles owned per wave
    // B tiles accumulated per sweep of the tape: 5 with 8 waves (2 per SIMD, 256 registers each); all of them with the
    // f32 policy's 4 waves (1 per SIMD, 512 registers: 16 accumulator tiles in the AGPRs) -- ONE sweep instead of two
    static constexpr int SWEEP = (Pol::NWAVES <= 4) ? NTMAX : 5;
    static constexpr int NPASS = (NPW_ALL + SWEEP - 1) / SWEEP;
    static constexpr int NPW = (NPW_ALL + NPASS - 1) / NPASS;
    // A tiles + h tiles + enc tile (8-bit tape: the largest group image is the layer-1 job's -- 8-bit A tiles, the recomputed
    // bf16 h_1 tiles, the encoded inputs)
    static constexpr int GROUP_BYTES = Pol::TAPE8 ? MT * TAPE_TILE + (MT + 1) * TILE_BYTES : (2 * MT + 1) * TILE_BYTES;
    static constexpr int GROUP_BYTES_LAST2 = GROUP_BYTES + 1024;               // + the KiB that starts with the f32 dout (dw_body2 LAST)
    // chain kernels: prefetch distance of the LDS-DMA weight ring (RING_DIST_TAPED+1 buffers of one chunk)
    static constexpr int RING_DIST_TAPED = (Pol::ELEM_BYTES == 2) ? BHN_TAPED_DIST : 3;
    // dW kernel: LDS-DMA ring of NBUF groups (counted vmcnt, raw s_barrier); f32: 2 buffers
    static constexpr int NBUF = (Pol::ELEM_BYTES == 2) ? ((160 * 1024) / GROUP_BYTES >= 4 ? 4 : 3) : 2;
    static constexpr int NPIECE = GROUP_BYTES / 1024;               // 1 KiB = one wave-wide 16-B DMA
    static constexpr int PPW = (NPIECE + Pol::NWAVES - 1) / Pol::NWAVES;
};

// ---------------------------------------------------------------------------------------------
// tile emission: 32x32 (feature x point) register tile -> fragment-ordered tape tile
// ---------------------------------------------------------------------------------------------
// Tile emission through the matrix core instead of LDS: the two B fragments of a 32-feature block are fed as the
// A operand (rows = points) against identity fragments, D = H^T . I, and the accumulator then holds the tile with
// its FEATURE on the lane and 16 points in the registers -- the operand layout of the dW GEMM (K = points).  Exact
// (one product by 1.0 per output), no LDS traffic, no waits; the points of a tile end up in the fixed order
// p = (r&3) + 8(r>>2) + 4(lane>>5), the same for every tile, which a sum over points does not care about.
template <class Pol>
struct TapeEmit {
    // identity k-steps, one fragment pair in LDS (read when needed: holding them costs 8 registers the 4x256 bf16
    // kernels do not have): element j of lane (n, h) is 1 where feature phi16(h,j)+16s == n
    static constexpr int LDS_BYTES = 2 * Pol::FRAG_BYTES;
    const char *table;
    DEVI void init(char *lds) {         // call from every thread of the workgroup before the first barrier
        table = lds;
        if (Pol::ELEM_BYTES != 2 && threadIdx.x < 64) {      // (bf16 tiles are not transposed by the matrix core any more)
            const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
            typename Pol::frag id[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) Pol::set(id[s], j, phi16(h, j) + 16 * s == n ? 1.f : 0.f);
                if constexpr (Pol::ELEM_BYTES == 2) {
                    *reinterpret_cast<typename Pol::frag *>(lds + s * Pol::FRAG_BYTES + lane * 16) = id[s];
                } else {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = Pol::get(id[s], 4 * hf + e);
                        *reinterpret_cast<f32x4 *>(lds + s * Pol::FRAG_BYTES + hf * 1024 + lane * 16) = v;
                    }
                }
            }
        }
    }
    DEVI void load_id(typename Pol::frag (&id)[2]) const {
        const int lane = threadIdx.x & 63;
        id[0] = Pol::lds_frag(table, 0, lane);
        id[1] = Pol::lds_frag(table, 1, lane);
    }
    static DEVI f32x16 transpose(const typename Pol::frag &f0, const typename Pol::frag &f1, const typename Pol::frag (&id)[2]) {
        f32x16 t = {};
        t = Pol::mma(f0, id[0], t);
        t = Pol::mma(f1, id[1], t);
        return t;
    }
    static DEVI void store(char *dst, const f32x16 &t, int dbg) {
        const int lane = threadIdx.x & 63;
        typename Pol::frag o[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) Pol::set(o[s], j, t[8 * s + j]);
        if constexpr (Pol::ELEM_BYTES == 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!(dbg & 1)) __builtin_nontemporal_store(o[s], reinterpret_cast<typename Pol::frag *>(dst + s * 1024 + lane * 16));
                else asm volatile("" ::"v"(o[s]));
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = Pol::get(o[s], 4 * hf + e);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst + s * 2048 + hf * 1024 + lane * 16));
                }
        }
    }
    // bf16: the tile goes to the tape AS THE PRODUCER HOLDS IT (point on the lane: the two B fragments of the 32-feature
    // block, no transposing MFMAs, no converts) and the dW kernel transposes while it reads the LDS image, with
    // ds_read_b64_tr_b16 (tr_frag below).  The 16 bytes of lane (pt, h) of fragment s go to slot
    //     (pt & 3) + 4 (2 s + h) + 16 (pt >> 2)
    // of the 2 KiB tile, so that the 32 8-byte chunks one half-wave gathers for a transposed read (4 points x
    // {s, h, 8-byte half}) cover one 256-byte LDS row exactly -- conflict-free -- and one store instruction still
    // writes eight whole 128-byte lines.
    static DEVI int native_off(int s) {
        const int lane = threadIdx.x & 63, pt = lane & 31, h = lane >> 5;
        return 16 * (pt & 3) + 128 * s + 64 * h + 256 * (pt >> 2);
    }
    static DEVI void store_native(char *dst, const typename Pol::frag &f0, const typename Pol::frag &f1, int dbg) {
        if (BHN_DBG(dbg & 12)) {          // measurement: cache policy of the tape stores (4: plain, 8: sc1)
            typename Pol::frag *p0 = reinterpret_cast<typename Pol::frag *>(dst + native_off(0));
            typename Pol::frag *p1 = reinterpret_cast<typename Pol::frag *>(dst + native_off(1));
            if (dbg & 4) { *p0 = f0; *p1 = f1; }
            else {
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p0), "v"(f0) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p1), "v"(f1) : "memory");
            }
        } else if (!(dbg & 1)) {
            __builtin_nontemporal_store(f0, reinterpret_cast<typename Pol::frag *>(dst + native_off(0)));
            __builtin_nontemporal_store(f1, reinterpret_cast<typename Pol::frag *>(dst + native_off(1)));
        } else {
            asm volatile("" ::"v"(f0), "v"(f1));
        }
    }
    DEVI void emit(char *dst, const typename Pol::frag &f0, const typename Pol::frag &f1, int dbg) const {
        if (dbg & 2) return;
        if constexpr (Pol::ELEM_BYTES == 2) {
            store_native(dst, f0, f1, dbg);
        } else {
            typename Pol::frag id[2];
            load_id(id);
            store(dst, transpose(f0, f1, id), dbg);
        }
    }
};

// A wave-uniform 64-bit value moved into SGPRs.  The compiler keeps the result of a 64-bit division (no scalar divide) in
// VGPRs and then does every address computation that depends on it on the vector ALU, followed by v_readfirstlane for
// the scalar operands of the DMA: ~60 VALU instructions per group in the dW stream (round-2 ISA census).
DEVI long long uniform64(long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Transposed read of a bf16 tape tile (TapeEmit::store_native image in LDS): the A / B fragment of k-step s of a dW
// GEMM (K = points) for lane (n = lane & 31, kh = lane >> 5): feature n, points 16 s + phi16(kh, j), j = 0..7 -- the
// same point order as the f32 tape tiles.  Two ds_read_b64_tr_b16; per 16-lane group g the instruction gathers 4 points
// (rows) x 16 features (columns, fragment g & 1 of the tile) and hands lane i of the group column i.  `trl` is the
// lane's part of the address (tr_lane_off), the rest is an immediate.  EXEC must be all ones.
DEVI int tr_lane_off() {
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    return 256 * (g >> 1) + 16 * q + 128 * (g & 1) + 64 * (pp & 1) + 8 * (pp >> 1);
}
DEVI bf16x8 tr_frag(const char *tile, int s, int trl) {
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const char *a = tile + 1024 * s + trl;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a + 512));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// ---- 8-bit tape (PolBF16T8): transposed read of a PAIR of e4m3 tape tiles (TapePost::store_tile8 images, tiles 2M and 2M+1
// of a layer, 1 KiB each, consecutive in LDS) --------------------------------------------------------------------------------
// ds_read_b64_tr_b16 moves 16-bit units; a unit of the 8-bit image is the byte pair (element 2u, element 2u+1) of one
// point, so lane i of a 16-lane group receives, for 4 points per read, the two features j = 2 (i & 3) + {0, 1} of fragment
// s = (i >> 2) & 1, lane half h = i >> 3 of ITS tile (groups 0 / 2: tile 2M, groups 1 / 3: tile 2M+1; groups 2, 3 = k half 1).
// Two reads = 8 points, in the point order of the bf16 fragments (tr_frag: element j of k half kh = point
// 16 s2 + 8 (j >> 2) + 4 kh + (j & 3)) -- the other operand may be a bf16 tile; the bytes are sorted per feature (2 v_perm per read pair and feature) and widened to bf16 by
// v_cvt_scalef32_pk_bf16_fp8, which also multiplies by the layer's power-of-two scale: out come the fragments (feature on
// the lane, 8 points of k-step s2 in the registers) of two VIRTUAL 32-row tiles -- `fa`: the even elements, `fb`: the odd
// ones -- i.e. row n of virtual tile 2M + ab is feature t8_feature(2M + ab, n) of the layer.  The dW slabs are indexed by
// virtual rows / columns; reduce_kernel undoes the permutation.
struct Raw8 { u32x2 lo, hi; };
DEVI int tr8_lane_off() {
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15, tsel = g & 1, kh = g >> 1;
    return 1024 * tsel + 128 * (kh ^ tsel) + 32 * (li >> 2) + 8 * (li & 3);          // (odd tiles swap their 4-point blocks); the second read: + 256
}
DEVI Raw8 tr8_read(const char *pair, int s2, int tr8) {
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    Raw8 r;
    r.lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pair + 512 * s2 + tr8)));
    r.hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pair + 512 * s2 + 256 + tr8)));
    return r;
}
// the two e4m3 fragments (8 points of one feature each: the operands of v_mfma_f32_32x32x16_fp8_fp8) of a raw pair read
struct Pair8 { u32x2 a, b; };
DEVI Pair8 tr8_sort(const Raw8 &r) {
    Pair8 p;
    p.a = (u32x2){__builtin_amdgcn_perm(r.lo[1], r.lo[0], 0x06040200u), __builtin_amdgcn_perm(r.hi[1], r.hi[0], 0x06040200u)};
    p.b = (u32x2){__builtin_amdgcn_perm(r.lo[1], r.lo[0], 0x07050301u), __builtin_amdgcn_perm(r.hi[1], r.hi[0], 0x07050301u)};
    return p;
}
// e4m3 fragment -> bf16 fragment, times a power-of-two scale (exact)
DEVI bf16x8 widen8(const u32x2 &f, float scale) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 a0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(f[0], scale, false), a1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(f[0], scale, true);
    const b2 a2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(f[1], scale, false), a3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(f[1], scale, true);
    return (bf16x8){a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
}
DEVI void tr8_widen(const Raw8 &r, float scale, bf16x8 &fa, bf16x8 &fb) {
    const Pair8 p = tr8_sort(r);
    fa = widen8(p.a, scale);
    fb = widen8(p.b, scale);
}
DEVI f32x16 mma8(const u32x2 &a, const u32x2 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}
// feature of row (or column) n of virtual tile t (see above)
__host__ __device__ static inline int t8_feature(int t, int n) {
    const int i = n & 15, hh = i >> 3, s = (i >> 2) & 1, j = 2 * (i & 3) + (t & 1);
    return 32 * ((t & ~1) + (n >> 4)) + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
}

// Post object of the training kernels (fused_common.h "Software-pipelined ring steps"): finishes the PENDING
// output tile in the MFMA shadows of the running step --
//   k-steps 0..7   RELU: relu + relu bits (forward) / !RELU: apply the recorded relu bits `mask` (delta chain),
//                  repack into the B fragments d0, d1 of the next layer
//   k-step  6, 8   identity fragments from LDS; relu bits -> tape word, two
//                  transposing MFMAs
//   k-step 12      bf16 convert + two 1 KiB non-temporal stores to the tape (after the step's DMA issue at k-step 9)
// Delta chain with TapeLayout::ga0_chain: the wave that OWNS dW_0's accumulator tile m adds the staged gA_0 tile m of every wave of
// the workgroup to it -- sixteen MFMAs (source wave p >> 1, k half p & 1), operands by transposed reads of the LDS staging
// images (point on the K index, exactly as the dW kernel reads its tape tiles), two pairs of fragments in flight.  A block of
// its own between the wave's ring step and the step's barrier: interleaved with the ring step's MFMAs (a second copy of every
// step body) it cost the consumer the same ~1000 cycles and every OTHER step ~150 more (profiles/r5_chain_stamps.txt).
struct Ga0Consumer {
    const char *ga, *enc;        // staged gA_0 tiles / encoded-input tiles of the eight source waves (2 KiB apart)
    int trl;                     // tr_lane_off(): the encoded-input tiles (DMA'd from the tape: its slot layout)
    int trl_ga;                  // the same for the staged gA_0 tiles, whose odd 256-byte rows swap their 64-byte lane halves (ga0_stage_swz)
#ifndef BHN_GA0C_PF
#define BHN_GA0C_PF 2            // fragment pairs in flight in the consumer block (8 registers each); round 6 re-sweep: 2 beats 3 and 4 by 1 % of the kernel (profiles/r6_tunable_sweep.txt)
#endif
    DEVI void run(f32x16 &acc, int npairs) const {
        constexpr int PF = BHN_GA0C_PF;
        bf16x8 a[PF], b[PF];
#pragma unroll
        for (int p = 0; p < PF - 1; ++p) {
            a[p] = tr_frag(ga + (p >> 1) * 2048, p & 1, trl_ga);
            b[p] = tr_frag(enc + (p >> 1) * 2048, p & 1, trl);
        }
#pragma unroll
        for (int p = 0; p < npairs; ++p) {
            const int q = p + PF - 1;
            if (q < npairs) {
                a[q % PF] = tr_frag(ga + (q >> 1) * 2048, q & 1, trl_ga);
                b[q % PF] = tr_frag(enc + (q >> 1) * 2048, q & 1, trl);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p % PF], b[p % PF], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};
template <class Pol, bool RELU, bool STAGE = false>
struct TapePost {
    char *stage = nullptr;       // STAGE: the finished tile also goes to this LDS image (staging image of a gA_0 tile, or the sink)
    int stage_off = 0;           // TapeEmit::native_off(0) of the lane
    const f32x16 &pend;
    typename Pol::frag &d0, &d1;
    unsigned mask;
    char *dst;
    // RELU: the relu bits of two consecutive tiles share a tape word [group][layer][tile/2][lane]: `macc` carries the
    // even tile's half to the odd tile's step, which stores the word; `stash` (optional) is a second, LDS destination
    unsigned *mword, *stash;
    unsigned &macc;
    bool hi, last;          // odd tile of the word (8-bit tape: of the tile pair) / last tile of the layer (stores a half-filled word when MT is odd)
    int edbg;
    const TapeEmit<Pol> &em;
    typename Pol::frag id[2];
    f32x16 tr;
    // 8-bit tape, delta chain: the power-of-two scale of the pending tile's layer (gA / t8_sc is what the tape holds) and the
    // running largest |gA| of the layer (for the NEXT call's scale); the forward's h tiles have scale 1
    float t8_sc;
    unsigned &t8_amax;       // (two bf16 magnitudes: the packed maximum over the dwords of the recorded tiles; an f32 maximum over
                             //  the accumulators in elems() made hipcc spill 70 registers in the delta chain)
    DEVI TapePost(const f32x16 &p, typename Pol::frag &a, typename Pol::frag &b, unsigned mask_in, const TapeEmit<Pol> &em_, char *dst_,
                  unsigned *mword_, unsigned *stash_, unsigned &macc_, bool hi_, bool last_, int edbg_, float t8_sc_, unsigned &t8_amax_)
        : pend(p), d0(a), d1(b), mask(RELU ? 0u : Pol::mask_spread(mask_in)), dst(dst_), mword(mword_), stash(stash_),
          macc(macc_), hi(hi_), last(last_), edbg(edbg_), em(em_), t8_sc(t8_sc_), t8_amax(t8_amax_) {}
    template <int R0, int N>
    DEVI void elems() {
        if constexpr (RELU) pack_elems<Pol, R0, N>(pend, d0, d1, mask);
        else {
#pragma unroll
            for (int r = R0; r < R0 + N; r += 2) {
                if constexpr (Pol::TAPE8) {
                    // the tape holds gA / t8_sc as e4m3 (|.| <= 448; the conversion makes NaN of anything larger): the pair is
                    // limited to +-448 t8_sc BEFORE it is rounded to bf16 -- a no-op unless the gradient grew by more than the
                    // 16x head room since the call the scale was taken from (DESIGN.md, 8-bit tape)
                    const float lim = 448.f * t8_sc;
                    const float x = pend[r], y = pend[r + 1];
                    Pol::mask_pair(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, __builtin_amdgcn_fmed3f(x, -lim, lim), __builtin_amdgcn_fmed3f(y, -lim, lim), mask);
                } else Pol::mask_pair(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, pend[r], pend[r + 1], mask);
            }
            if (R0 < 8) asm volatile("" : "+v"(d0));
            if (R0 + N > 8) asm volatile("" : "+v"(d1));
        }
    }
    // 8-bit tape: the tile as 16 e4m3 bytes per lane (its two bf16 B fragments through v_cvt_scalef32_pk_fp8_bf16: byte
    // 2 i + e of fragment s = element 2 i + e, i.e. feature 16 s + phi16(h, 2 i + e)), ONE 16-byte store.  Tile image (1 KiB):
    // point pt at byte 32 (pt ^ 4 par), par = tile & 1 -- lane half h at + 16 h, fragment s at + 8 s -- so that the
    // transposed LDS reads of the dW kernel (tr8_pair: a 16-lane group gathers 4 points x 32 bytes = 128 contiguous bytes,
    // the groups of an even and an odd tile in the same instruction) fall on disjoint bank halves.
    DEVI void store_tile8() {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        const u32x4 w0 = __builtin_bit_cast(u32x4, d0), w1 = __builtin_bit_cast(u32x4, d1);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned lo = i < 2 ? w0[2 * i] : w1[2 * i - 4], hi2 = i < 2 ? w0[2 * i + 1] : w1[2 * i - 3];
            if constexpr (RELU) {      // h >= 0: one signed packed min limits it to 448 (0x43e0)
                lo = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(i16x2, lo), (i16x2){0x43e0, 0x43e0}));
                hi2 = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(i16x2, hi2), (i16x2){0x43e0, 0x43e0}));
            } else {                   // the largest magnitudes recorded so far (non-negative halves: the signed packed max)
                const i16x2 m = __builtin_elementwise_max(__builtin_bit_cast(i16x2, lo & 0x7fff7fffu), __builtin_bit_cast(i16x2, hi2 & 0x7fff7fffu));
                t8_amax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, t8_amax), m));
            }
            s16x2 r = {0, 0};
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, __builtin_bit_cast(typename Pol::bf16x2, lo), t8_sc, false);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, __builtin_bit_cast(typename Pol::bf16x2, hi2), t8_sc, true);
            o[i] = __builtin_bit_cast(unsigned, r);
        }
        const int lane = threadIdx.x & 63, pt = lane & 31, hh = lane >> 5;
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst + ((32 * pt + 16 * hh) ^ (hi ? 128 : 0))));
    }
    DEVI void bits_and_transpose() {
        if constexpr (RELU) {
            const unsigned code = Pol::mask_code(mask);
            const unsigned word = hi ? (macc | (code << 16)) : code;
            macc = word;
            if (hi || last) {
                if (mword) __builtin_nontemporal_store(word, mword);
                if (stash) *stash = word;
            }
        }
        if constexpr (Pol::ELEM_BYTES != 2) {
            if (!(edbg & 2)) tr = TapeEmit<Pol>::transpose(d0, d1, id);
        }
    }
    DEVI void store_tile() {
        if constexpr (STAGE) {
            // GA0C kernels: a finished gA_0 tile goes to its LDS staging image instead of the tape: two ds_write_b128 in the tape's slot
            // layout (TapeEmit::native_off).  (Writing EVERY tile to LDS -- the others to a sink -- to save this wave-uniform branch
            // measured 4 % slower on the kernel: profiles/r5_ab_ga0_chain.txt.)
            if (stage) {
                *reinterpret_cast<typename Pol::frag *>(stage + stage_off) = d0;
                *reinterpret_cast<typename Pol::frag *>(stage + stage_off + 128) = d1;
                return;
            }
        }
        if (edbg & 2) return;
        if (BHN_DROP_HD != 0 && (edbg & 16)) {                // (TapeLayout::drop_hd on a ring kernel: the emission's two stores, four bytes each, `dst` = a line of the tape's scratch area)
            __builtin_nontemporal_store(0u, reinterpret_cast<unsigned *>(dst));
            __builtin_nontemporal_store(0u, reinterpret_cast<unsigned *>(dst) + 1);
            return;
        }
        if constexpr (Pol::TAPE8) store_tile8();
        else if constexpr (Pol::ELEM_BYTES == 2) TapeEmit<Pol>::store_native(dst, d0, d1, edbg);
        else TapeEmit<Pol>::store(dst, tr, edbg);
    }
    unsigned w = 0;                             // the pair rounded in the previous k-step (pack_pipe)
    DEVI void at(int t) {
        if constexpr (RELU && Pol::ELEM_BYTES == 2) pack_pipe<Pol, true>(t, pend, d0, d1, w, mask);     // fragments complete after k-step 8, bits after 9
        else {
            if (t == 0) elems<0, 2>();
            if (t == 1) elems<2, 2>();
            if (t == 2) elems<4, 2>();
            if (t == 3) elems<6, 2>();
            if (t == 4) elems<8, 2>();
            if (t == 5) elems<10, 2>();
            if (t == 6) elems<12, 2>();
            if (t == 7) elems<14, 2>();
        }
        if (Pol::ELEM_BYTES != 2 && t == 6 && !(edbg & 2)) em.load_id(id);
        if (t == 10) bits_and_transpose();
        if (t == 12) store_tile();
    }
    DEVI void all() {
        elems<0, 16>();
        if (Pol::ELEM_BYTES != 2 && !(edbg & 2)) em.load_id(id);
        bits_and_transpose();
        store_tile();
    }
    DEVI void finish() {}
};

#ifdef BHN_DEBUG
void *bhn_debug_buffer();
#endif
// ---------------------------------------------------------------------------------------------
// chain kernel
// ---------------------------------------------------------------------------------------------
enum { MODE_FWD_TRAIN = 1, MODE_CHAIN = 2 };

// MODE_FWD_TRAIN: the training forward: render (images, unless a.images is null) AND record h tiles, relu bits and
//                 e on the tape.
// MODE_CHAIN:     delta chain from the relu bits / e recorded by MODE_FWD_TRAIN.
// bhn_render_bwd (gradient for arbitrary dimages, any workspace size) = both, frame group by frame group; a single
// kernel doing both ("recompute" mode) needed the working sets of both halves in one register allocation and
// spilled in its hot loop (12.8 ms vs 4.6 + 4.7 ms at config 2).
// Weight-chunk stream of one tile: forward chunks 0..NCF-1 (MODE_FWD_TRAIN) or the transposed chunks of hidden
// layers depth-1 .. 1 (MODE_CHAIN); the ring wraps to the next tile's first chunk.
// RES: the whole chunk sequence resident in LDS (fused_common.h ResidentRing: no DMA, no per-chunk barrier), when it fits
template <int W, class Pol, int DEG, int MODE, bool RES = false, bool GA0C = false>
__global__ __launch_bounds__(Pol::NTHREADS) void chain_kernel(BwdArgs A) {
    static_assert(!GA0C || (MODE == MODE_CHAIN && !RES && Pol::ELEM_BYTES == 2 && W / 32 == Pol::NWAVES),
                  "dW_0 inside the delta chain: bf16, one gA_0 tile per wave, the barrier-stepped ring");
    using PK = Pack<W, Pol>;
    using BG = BwdGeom<W, Pol>;
    using frag = typename Pol::frag;
    constexpr int CB = PK::CHUNK_BYTES, MT = PK::MT, KS = PK::KS;
    constexpr int MW = (MT + 1) / 2;                                   // mask words per layer per lane
    constexpr int TB = BG::TILE_BYTES;
    constexpr int TT = BG::TAPE_TILE;                                  // an h / gA tile on the tape (8-bit tape: 1 KiB)
    constexpr bool T8 = Pol::TAPE8;
    const FusedArgs &a = A.f;
    clock_stamp(a.clk, MODE == MODE_CHAIN ? BHN_CLK_CHAIN : BHN_CLK_FWD_TRAIN, 0);
    const int edbg = BHN_DBG(((A.debug >> 6) & 3) | (A.policy << 2));  // measurement aid for the tape emission (bits 2,3: store policy)
    constexpr int sdbg = 0;                        // (a run-time MFMA-skip flag put every MFMA in its own basic block)
    // the delta chain's transposed image never uses the two encoded-input fragments of a chunk: its ring copies (and its steps
    // stream) only the KS hidden fragments; the training forward (bf16) keeps them in a resident block of their own (EncBlock)
    // (KS >= 8: the A-fragment prefetch of a step runs LDS_PREFETCH - 1 fragments into the NEXT chunk, which must have that many)
    using EB = EncBlock<W, Pol>;
    constexpr bool ENCR = MODE != MODE_CHAIN && EB::ON && !RES;
    constexpr int NFR = ((MODE == MODE_CHAIN && KS >= 8) || ENCR) ? KS : KS + 2;
    using RG = DmaRing<RES ? CB : NFR * Pol::FRAG_BYTES, Pol::NWAVES, GA0C>;     // (GA0C: transposed LDS reads in the kernel -> asm DMA)
    constexpr int DIST = GA0C ? BHN_GA0C_DIST : BG::RING_DIST_TAPED;
    using RS = std::conditional_t<RES, ResidentRing<RG, CB, MT>, RingState<RG, CB, DIST, false, MT, BHN_CHAIN_STAMPS != 0>>;
    // stores guaranteed younger than chunk c+2 at the end of step c (RingState::step_end): every interval between
    // two DMA issues holds the >= ES stores of one pending-tile emission; with >= 16 k-steps the running step's
    // own emission (k-step 12) also follows its DMA issue (k-step 9).  The DMA pieces of the DIST-2 younger chunks
    // cancel out of the balance, so only this store count has to be a lower bound (relu-bit words, epilogue stores and
    // prefetch loads only add slack); the exceptions are the steps right after a layer-0 step without h_1 emission.
    constexpr int ES = T8 ? 1 : (Pol::ELEM_BYTES == 2) ? 2 : 4;        // global stores of one tile emission
#ifndef BHN_YS_EXTRA
#define BHN_YS_EXTRA 0          // EXPERIMENT ONLY (-DBHN_YS_EXTRA=n): lets n more stores stay in flight than the ring proof allows
#endif                          // (racy: wrong results possible) -- measures what the in-order vmcnt coupling of stores and weight DMA costs
    constexpr int YS = ES * (DIST - 2) + ((KS >= 16 && Pol::ELEM_BYTES == 2) ? ES : 0) + BHN_YS_EXTRA;   // (f32: kept conservative)
    constexpr int YS0 = ES * (DIST - 2) + BHN_YS_EXTRA;                // steps whose own stores precede their DMA issue
    constexpr int YS_L1r = (KS >= 16) ? YS - ES : 0;                   // first steps of layer 1 when h_1 is not emitted
    constexpr int YS_L1 = YS_L1r > 0 ? YS_L1r : 0;
    const int NCF = (MODE == MODE_CHAIN) ? 0 : PK::fwd_chunks(A.f.depth);
    // delta chain: hidden layers depth-1 .. LEND = 1 produce gA_{l-1}
    constexpr int LEND = 1;
    const int NLB = (MODE == MODE_FWD_TRAIN) ? 0 : A.f.depth - LEND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;
    float *bias_lds = reinterpret_cast<float *>(smem + RS::lds_bytes(NCF + NLB * MT));  // depth x W + the 32 rows of the output tile, then one zero row
    const int nbias = a.depth * W + 32;                               // (the output layer's tile is read as one 32-row tile:
    float *zero_lds = bias_lds + nbias;                               //  a full W row for it put the f32 8x256 kernel over 160 KB)
    float *wout_lds = zero_lds + 32;
    char *id_lds = reinterpret_cast<char *>(wout_lds + W);
    char *seg_lds = id_lds + (Pol::ELEM_BYTES == 2 ? 0 : 2 * Pol::FRAG_BYTES);   // RaySum scratch (bf16 has no identity table)
    // 8-bit tape, delta chain (no RaySum there): 8 layer scales, then one |gA|max slot per layer and wave
    float *t8_lds = reinterpret_cast<float *>(seg_lds);
    // GA0C: staging images behind the fixed part (host: lds_fixed) -- the encoded-input tiles of the workgroup's eight groups
    // (two tile parities) and the finished gA_0 tiles of the running / the previous step, all in the tape's slot layout
    constexpr int STG = Pol::NWAVES * TB;                              // one image: a 2-KiB tile per wave
    char *encS = seg_lds + RaySum<Pol::NWAVES>::bytes(A.f.Sx);
    char *gaS = encS + 2 * STG;
    char *encblk = encS;                                               // training forward (ENCR): the resident encoded-input weight block
    // Staging images of the gA_0 tiles: the tape's slot layout with the two 64-byte lane halves of every ODD 256-byte row (points
    // 4 r .. 4 r + 3) swapped.  A ds_write_b128 retires 8 consecutive lanes (128 bytes) per LDS cycle -- points 8 i .. 8 i + 7 of one lane
    // half, i.e. the first 64 bytes of two consecutive rows: the same 16 banks twice in the plain layout (SQ_LDS_BANK_CONFLICT 6.7 % of
    // this kernel's LDS cycles in round 5).  The consumer's transposed reads gather whole rows and only swap the halves back.
    const int stage_off = TapeEmit<Pol>::native_off(0) ^ ((GA0C && BHN_GA0C_SWZ) ? ((int)(threadIdx.x >> 2) & 1) << 6 : 0);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 31, h = lane >> 5;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);           // the wave index as a scalar: tape addresses stay in SGPRs
    auto t8_flush = [&](int l, unsigned packed) {          // packed: two bf16 magnitudes (TapePost::t8_amax)
        float v = __builtin_fmaxf(__uint_as_float(packed << 16), __uint_as_float(packed & 0xffff0000u));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
        if (lane == 0) {
            float *slot = t8_lds + 8 + l * Pol::NWAVES + wv;
            *slot = __builtin_fmaxf(*slot, v);
        }
    };
    // scalars of the tape layout, pinned in registers (the compiler otherwise re-fetches kernel arguments inside the steps)
    long long h_lin = A.t.h_lin, ga_lin = A.t.ga_lin, lin_stride = A.t.lin_stride;
    asm volatile("" : "+s"(h_lin), "+s"(ga_lin), "+s"(lin_stride));
    TapeEmit<Pol> em;
    em.init(id_lds);
    for (int i = tid; i < nbias; i += Pol::NTHREADS)
        bias_lds[i] = reinterpret_cast<const float *>(a.packed + a.bias_off)[i];
    if (tid < 32) zero_lds[tid] = 0.f;
    for (int i = tid; i < W; i += Pol::NTHREADS) wout_lds[i] = reinterpret_cast<const float *>(a.packed + a.wout_off)[i];
    if constexpr (ENCR) EB::fill(encblk, a.packed + a.fwd_off, a.depth, a.skip_mask);
    if constexpr (T8 && MODE == MODE_CHAIN) {
        if (tid < 8) {
            const float v = A.t8[tid];
            t8_lds[tid] = (v > 0.f && v < __builtin_inff()) ? v : 1.f;       // (a workspace that was never calibrated)
        }
        if (tid < a.depth * Pol::NWAVES) t8_lds[8 + tid] = 0.f;
    }

    const bool have_ring = NCF + NLB * MT > 0;
    // first tile of the sequence starts with bias (forward) or zero (delta chain) accumulators
    const float *first_bias = (MODE == MODE_CHAIN) ? zero_lds : bias_lds;
    RS rs;
    APipe<Pol> ap;
    if (have_ring) {
        // (the transposed images are stored layer-major from layer 1 on: the ring starts at layer LEND's)
        rs.start(ring, a.packed + a.fwd_off, NCF, a.packed + a.bwd_off + (size_t)(LEND - 1) * MT * CB, NLB, sdbg ? 1 : 0, 0);
        ap.prime(rs.ch(), first_bias);
    } else {
        __syncthreads();
    }

    // inputs of the next tile are fetched one tile ahead (geometry for the forward modes; for MODE_CHAIN the
    // recorded emission and dE = sum_s dimg * w, which only need the point index)
    // ... and the relu-bit words of layer depth-1 (all of it, for gA_{depth-1}).  The words of the chain sequence (layers depth-2 .. 0,
    // MW words each) come through a FIFO of whole LAYERS, filled at least MWL - 1 layers (>= 4 ring steps; 8 at width 256) ahead of
    // their use -- see `layer_words` below.
    struct ChainIn { int b; long long p; bool inb; float e, dE; unsigned mtop[MW]; };
    // Relu-bit words of the layer at position `pos` of the chain sequence that starts at layer 0 of tile `tile0` and runs on into the
    // workgroup's next tiles (NL = depth - 1 layers per tile).  WHY whole layers far ahead (round 5): vmcnt retires in issue order, so
    // the wait in front of the first use of a loaded word also waits for every tape store issued BEFORE the load -- and under 3-4 TB/s
    // of tape writes a store takes ~5 ring steps to retire.  With the words fetched two words (four steps) ahead every second step of
    // the delta chain stalled for 500-1500 cycles (ring-step stamps: 50.4 k ticks per tile against 39.1 k with the stores switched
    // off, the training forward -- no loads in its loop -- 49.3 k against 45.8 k; profiles/r5_chain_stamps.txt).
    constexpr int MWL = MT >= 4 ? 2 : (MT == 2 ? 3 : 5);              // layers in the FIFO
    const int NL = a.depth - LEND;                                     // chain layers per tile
    auto layer_words = [&](long long tile0, int pos, unsigned (&w)[MW]) {
        long long t = tile0;
        int j = pos;
        while (j >= NL && NL > 0) { j -= NL; t += gridDim.x; }
        const bool on = NL > 0 && t < a.total_tiles;
        const unsigned *mg = reinterpret_cast<const unsigned *>(A.tape + A.t.mask_off) + (t * Pol::NWAVES + wvu) * (long long)(a.depth * MW * 64);
#pragma unroll
        for (int i = 0; i < MW; ++i) w[i] = on ? __builtin_nontemporal_load(mg + ((a.depth - 2 - j) * MW + i) * 64 + lane) : 0u;
    };
    auto load_chain = [&](long long tile) {
        ChainIn c;
        c.b = 0; c.p = 0; c.inb = false; c.e = 0.f; c.dE = 0.f;
#pragma unroll
        for (int i = 0; i < MW; ++i) c.mtop[i] = 0u;
        if (tile < a.total_tiles) {
            const unsigned *mg = reinterpret_cast<const unsigned *>(A.tape + A.t.mask_off) +
                                 (tile * Pol::NWAVES + wvu) * (long long)(a.depth * MW * 64);
#pragma unroll
            for (int i = 0; i < MW; ++i) c.mtop[i] = __builtin_nontemporal_load(mg + ((a.depth - 1) * MW + i) * 64 + lane);
            tile_point<Pol::NWAVES>(a, tile, wv, pl, c.b, c.p, c.inb);
            if (h == 0) c.e = (reinterpret_cast<const float *>(A.tape + A.t.e_off) + (tile * Pol::NWAVES + wvu) * 32)[pl];
            if (h == 0 && c.inb) {
                const long long ray = a.ray_idx ? (long long)a.ray_idx[c.p] : (long long)a.fd_G.div((unsigned)c.p);
                for (int s = 0; s < a.Sx; ++s)
                    c.dE += a.dimages[((long long)c.b * a.Sx + s) * a.R + ray] * a.w[(long long)s * a.P + c.p];
            }
        }
        return c;
    };
    // (the forward modes load their geometry at the tile start: a one-tile-ahead prefetch measured no faster and
    // costs 12 registers the 4x256 kernel does not have)
    ChainIn cnxt;
    unsigned mwf[MWL][MW];                           // relu-bit words: mwf[0] = the running chain layer's, mwf[k] = k layers ahead
    if constexpr (MODE == MODE_CHAIN) {
        cnxt = load_chain(blockIdx.x);
#pragma unroll
        for (int k = 0; k < MWL; ++k) layer_words(blockIdx.x, k, mwf[k]);
    }
    // GA0C: dW_0 accumulator tiles (rows = an output tile of layer 0, columns = encoded-input slots, column 31 = bias).  The OLDER wave
    // of every SIMD (waves 0 .. NWAVES/2 - 1) owns two -- tiles w and w + NWAVES/2 -- and the younger none: the older wave wins
    // the issue arbitration and reaches every barrier ~500 cycles before its partner (ring-step stamps), so the
    // consumer block runs in time the wave would otherwise spend waiting
    constexpr int NCW = Pol::NWAVES / 2;             // consumer waves
    f32x16 cacc0 = {}, cacc1 = {};
    Ga0Consumer cons;
    cons.trl = tr_lane_off(); cons.ga = cons.enc = nullptr;
    cons.trl_ga = cons.trl ^ ((GA0C && BHN_GA0C_SWZ) ? ((int)(threadIdx.x >> 5) & 1) << 6 : 0);      // (a read's two row pairs: rows g >> 1 and + 2 -- the parity is lane >> 5)
    int tpar = 0;                                    // parity of the tile: which encS image it uses
    bool have_prev = false;                          // the previous tile's gA_0 tiles MT-2, MT-1 are staged and not yet consumed
    bool has_next = false;                           // this workgroup has another tile after the running one (its inputs are being prefetched)
    for (long long tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        PointIn in;
        if constexpr (MODE != MODE_CHAIN) in = load_point<Pol::NWAVES>(a, tile, wv, pl);
        const ChainIn cin = cnxt;
        const int b = (MODE != MODE_CHAIN) ? in.b : cin.b;
        const long long p = (MODE != MODE_CHAIN) ? in.p : cin.p;
        const bool inb = (MODE != MODE_CHAIN) ? in.inb : cin.inb;
        const long long q = tile * Pol::NWAVES + wvu;                    // 32-point group on the tape
        const long long qs = BHN_DBG(A.wrap) ? q % A.wrap : q;           // (debug: h / gA tiles wrap into a cache-resident window)
        unsigned *mask_g = A.t.fused128 ? nullptr : reinterpret_cast<unsigned *>(A.tape + A.t.mask_off) + q * (long long)(a.depth * MW * 64);
        // (TapeLayout::drop_hd: the relu bits of the last hidden layer, recorded in place of its output tiles)
        unsigned *maskd_g = (MODE == MODE_FWD_TRAIN && A.t.drop_hd && A.t.fused128) ? reinterpret_cast<unsigned *>(A.tape + A.t.maskd_off) + q * (long long)(MW * 64) + lane : nullptr;
        float *e_g = reinterpret_cast<float *>(A.tape + A.t.e_off) + q * 32;
        frag enc[2], act[KS], next[KS];
        bool live = false;
        float e = 0.f;
        if (BHN_CHAIN_STAMPS && have_ring) rs.ts = (A.ts_buf && blockIdx.x == 0 && tile == 3 * (long long)gridDim.x) ? A.ts_buf + __builtin_amdgcn_readfirstlane(wv) * 64 : nullptr;
        if constexpr (MODE != MODE_CHAIN) {
            point_prologue<Pol, DEG>(a, in, enc, live);
            // the encoded inputs are the B operand of dW_0 and of the skip layer
            if (A.t.fused128 || A.t.ga0_chain) {
                // slot 31 (k-step 1, lane half 1, element 7; an unused slot: zero weight rows) carries 1 on the tape: against
                // it the fused dW GEMMs of fused_bwd128.hip (and the delta chain's own dW_0, ga0_chain) produce the bias gradients
                frag e1 = enc[1];
                if (h) Pol::set(e1, 7, 1.f);
                em.emit(A.tape + A.t.enc_off + q * TB, enc[0], e1, edbg);
            } else em.emit(A.tape + A.t.enc_off + q * TB, enc[0], enc[1], edbg);
            const bool drop_h1 = A.t.drop_h1;
            if (drop_h1 && !A.t.fused128 && !(edbg & 2)) {        // ... and, as they are, the A operand from which dW_1 recomputes h_1
                __builtin_nontemporal_store(enc[0], reinterpret_cast<frag *>(A.tape + A.t.encp_off + q * TB + lane * 16));
                __builtin_nontemporal_store(enc[1], reinterpret_cast<frag *>(A.tape + A.t.encp_off + q * TB + Pol::FRAG_BYTES + lane * 16));
            }
            // ---- forward, layer 0: tile m-1 is packed, recorded and emitted behind the MFMAs of tile m ------
            struct Tile0 {
                const TapeEmit<Pol> &em;
                char *dst;
                unsigned *mword, *stash;
                unsigned macc;
                int edbg;
                DEVI void tile(int m, const f32x16 &acc, frag &d0, frag &d1) {
                    unsigned mk = 0;
                    pack_elems<Pol, 0, 16>(acc, d0, d1, mk);
                    mk = Pol::mask_code(mk);
                    macc = (m & 1) ? (macc | (mk << 16)) : mk;
                    if (m & 1) {
                        if (mword) __builtin_nontemporal_store(macc, mword + (m >> 1) * 64);
                        if (stash) stash[(m >> 1) * 64] = macc;
                    }
                    if (dst) em.emit(dst + (long long)m * TB, d0, d1, edbg);
                }
            } l0{em, drop_h1 ? nullptr : A.tape + h_lin + lin_stride + qs * MT * TT, mask_g ? mask_g + lane : nullptr,
                 nullptr, 0u, edbg};
            f32x16 pend;
            layer0_step<W, Pol, RG, YS0, RS, Tile0, NFR>(rs, ap, enc, act, bias_lds, h, pend, l0);
            // ---- hidden layers 1..depth-1 and the output layer: the pending tile is (l-1, MT-1) at m = 0 ------
            int pl_layer = 0;                        // layer of the pending tile
            unsigned t8_none = 0u;                   // (the forward's h tiles: scale 1, no maxima)
            unsigned macc = l0.macc;                 // relu bits of the even tile of the running mask word
#pragma nounroll
            for (int l = 1; l <= a.depth; ++l) {
                const bool out = l == a.depth;
                const bool sk = (a.skip_mask >> l) & 1;              // (odd depths: also the output layer)
                const float *bl = bias_lds + l * W;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (out && m > 0) break;
                    const char *ch = rs.ch(), *chn = rs.chn();
                    const DmaJob dj = rs.job();
                    const int pm = m == 0 ? MT - 1 : m - 1;           // pending tile (layer pl_layer)
                    frag &d0 = m == 0 ? act[KS - 2] : next[2 * (m > 0 ? m - 1 : 0)];
                    frag &d1 = m == 0 ? act[KS - 1] : next[2 * (m > 0 ? m - 1 : 0) + 1];
                    const int widx = (pl_layer * MW + (pm >> 1)) * 64 + lane;
                    // h_depth: relu bits only (drop_hd).  fused128 (resident weights, no ring): no store at all; ring kernels: the
                    // emission's two stores stay, four bytes each to a 256-KiB scratch area of the tape (bit 4; one line per tile: all on one line was a hot spot) -- the waits of the weight ring
                    // count the stores of every step (YS above), and a step without them would need counts of its own
                    constexpr bool CAN_HD = BHN_DROP_HD != 0 || (W == 128 && Pol::ELEM_BYTES == 2 && !T8);      // (fused128: width 128, bf16)
                    const bool no_hd = CAN_HD && A.t.drop_hd && pl_layer == a.depth - 1;
                    const bool no_h = (drop_h1 && pl_layer == 0) || (no_hd && RES);       // layer 0's last tile: relu bits only
                    const int edbg_t = no_h ? (edbg | 2) : (no_hd ? (edbg | 16) : edbg);
                    TapePost<Pol, true> post(pend, d0, d1, 0u, em, (no_hd && !RES) ? A.tape + A.t.scratch_off + (((q * MT + pm) & 4095) << 6) : A.tape + h_lin + (pl_layer + 1) * lin_stride + (qs * MT + pm) * TT,
                                                 mask_g ? mask_g + widx : (no_hd && maskd_g ? maskd_g + (pm >> 1) * 64 : nullptr), nullptr,
                                                 macc, pm & 1, pm == MT - 1, edbg_t, 1.f, t8_none);
                    // bias rows of the next tile: (l, m+1), or the first tile of the next sequence part
                    const float *bn = (out || (m == MT - 1 && l + 1 > a.depth)) ? nullptr : bl + 32 * (m + 1);
                    if (out) bn = bias_lds;                           // next tile, layer 0
                    const f32x16 acc = ring_step<W, Pol, RG, TapePost<Pol, true>, NFR>(ch, chn, ap, act, enc, sk, bn, post, dj, sdbg,
                                                                                       encblk + 2 * (out ? MT : m) * Pol::FRAG_BYTES);
                    // without the h_1 emission the interval after this layer's first DMA issue holds no store: the
                    // three step ends that count it allow one emission less in flight (small widths: none)
                    if (drop_h1 && l == 1 && m <= 2) rs.template step_end<YS_L1>();
                    else if (drop_h1 && l == 1) rs.template step_end<(KS >= 16 ? YS : 0)>();
                    else rs.template step_end<YS>();
                    pend = acc;
                    pl_layer = l;
                    if (m == 0) pl_layer = l;                         // from here on the pending tiles are layer l's
                }
                if (!out) {
#pragma unroll
                    for (int ks = 0; ks < KS - 2; ++ks) act[ks] = next[ks];      // the pending tile lands in act[KS-2], act[KS-1]
                }
            }
            if (h == 0 && live) e = 1.f / (1.f + Pol::fexp(10.f - pend[0]));
        } else {
            cnxt = load_chain(tile + gridDim.x);
            e = cin.e;
            has_next = tile + gridDim.x < a.total_tiles;
            if constexpr (GA0C) {                    // this group's encoded-input tile (tape, slot 31 = 1) -> encS[tpar][wave]
                const char *esrc = A.tape + A.t.enc_off + q * TB;
                char *edst = encS + tpar * STG + wvu * TB;
                if (!(BHN_GA0C_ABL & 4)) {
                    dma_1k_asm<1>(esrc, edst);
                    dma_1k_asm<1>(esrc + 1024, edst + 1024);
                }
            }
        }
        // ---- e ; dE, dout -----------------------------------------------------------------------
        float dout = 0.f;
        if constexpr (MODE == MODE_FWD_TRAIN) {
            // record what the delta chain needs, then the render epilogue of fused_fwd_kernel
            if (h == 0) e_g[pl] = e;
            if (a.images) RaySum<Pol::NWAVES>::run(a, seg_lds, b, p, inb, e, 0.f, false);      // (bhn_render_bwd: tape only)
            if constexpr (RES) { if (a.images && !a.ray_direct) lds_barrier(); }                // (no ring barriers behind the combine)
        } else {
            float d = 0.f;
            if (h == 0 && inb && e != 0.f) {
                d = cin.dE * e * (1.f - e);                        // sigmoid'(out-10) = e(1-e)
            }
            dout = __shfl(d, pl, 64);                               // both lane halves need it
            char *dgrp = A.tape + A.t.dout_off + q * A.t.dout_stride;
            if (A.t.drop_ga) {
                // 32 f32 per group: the job of layer depth-1 rebuilds gA_{depth-1} and makes the output layer's row from them
                // (8-bit tape: that job turns them into e4m3(dout / scale) -- which makes NaN of anything beyond 448 scale)
                float dt = d;
                if constexpr (T8) { const float lim = 448.f * t8_lds[a.depth - 1]; dt = __builtin_amdgcn_fmed3f(d, -lim, lim); }
                if (h == 0 && !(edbg & 2)) __builtin_nontemporal_store(dt, reinterpret_cast<float *>(dgrp) + pl);
            } else {
                // dout as a 32x32 (feature x point) tile whose feature 0 is dout: the A operand of dW_out
                frag d0 = Pol::zero(), d1 = Pol::zero();
                Pol::set(d0, 0, h == 0 ? d : 0.f);
                em.emit(dgrp, d0, d1, edbg);
            }
        }
        if constexpr (MODE == MODE_CHAIN) {
            // ---- gA_{depth-1} = wout * dout * relu'(a_{depth-1}); its last tile stays pending ---------------
            frag dl[KS];
            f32x16 pend = {};
            unsigned last_mask = 0xffffu;            // relu bits still to be applied to the pending tile of gA_{depth-1}
            {
                const bool keep_ga = !A.t.drop_ga;                    // else the dW kernel rebuilds gA_{depth-1}
                char *gdst = A.tape + ga_lin + (a.depth - 1) * lin_stride + qs * MT * TT;
                if constexpr (Pol::ELEM_BYTES == 2) {
                    if (A.t.drop_ga) {
                        // W_out sits in the columns of this layer's transposed weight image (bhn_folds_wout): the B operand
                        // is relu' (.) bf16(dout) -- one packed dword ANDed with the spread relu bits, 3 VALU per pair
                        // (building W_out . dout . relu' element by element was ~550 of the 970 VALU instructions of this
                        // kernel's tile prologue, which all eight waves execute with the matrix pipe idle)
                        typedef short i16x2 __attribute__((ext_vector_type(2)));
                        const typename Pol::bf16x2 d2 = {(__bf16)dout, (__bf16)dout};
                        const unsigned dd = __builtin_bit_cast(unsigned, d2);
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const unsigned mw = (cin.mtop[m >> 1] >> ((m & 1) * 16)) & 0xffffu;
                            if (m == MT - 1 && a.depth > 1) {
                                // the last tile stays pending as f32; the first ring step masks and packs it (TapePost)
#pragma unroll
                                for (int r = 0; r < 16; ++r) pend[r] = dout;
                                last_mask = mw;
                                continue;
                            }
                            const unsigned spread = Pol::mask_spread(mw);
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const i16x2 on = __builtin_bit_cast(i16x2, spread << (15 - k)) >> (i16x2){15, 15};
                                Pol::put_dword(dl[2 * m + (k >> 2)], k & 3, dd & __builtin_bit_cast(unsigned, on));
                            }
                        }
                    }
                }
                if (!(Pol::ELEM_BYTES == 2 && A.t.drop_ga)) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const unsigned mw = cin.mtop[m >> 1] >> ((m & 1) * 16);
                    f32x16 g;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 wv4 = *reinterpret_cast<const f32x4 *>(wout_lds + 32 * m + 8 * g4 + 4 * h);
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const int r = 4 * g4 + e4;                 // ReLU code: bit r/2 (even r) or 8 + r/2 (odd r)
                            g[r] = ((mw >> ((r >> 1) + 8 * (r & 1))) & 1) ? wv4[e4] * dout : 0.f;
                        }
                    }
                    if (m < MT - 1 || a.depth == 1) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int j = 0; j < 8; ++j) Pol::set(dl[2 * m + s2], j, g[8 * s2 + j]);
                        if (keep_ga) em.emit(gdst + m * TB, dl[2 * m], dl[2 * m + 1], edbg);
                    } else pend = g;
                }
                }
            }
            // ---- delta chain through hidden layers depth-1 .. 1: step (l, m) makes tile m of gA_{l-1} -------
            int pnd_layer = a.depth - 1;             // layer of the pending gA tile
            unsigned pnd_mask = last_mask;           // its relu bits (0xffff: gA_{depth-1} is masked already)
            unsigned no_acc = 0u;
            float t8_sc = 1.f;
            // (seeded with |dout|: the first flush below -- layer depth-1, whose gA is not recorded -- leaves the largest |dout|, the
            //  scale of the e4m3(dout) the dW job of that layer forms)
            unsigned t8_amax = T8 ? (unsigned)__builtin_bit_cast(unsigned short, (__bf16)__builtin_fabsf(dout)) : 0u;     // 8-bit tape: scale of the pending tile's layer, |gA|max of that layer so far
            if constexpr (T8) t8_sc = 0x1p60f;        // (gA_{depth-1} is not recorded -- TapeLayout::drop_ga -- and must not be limited either)
#pragma nounroll
            for (int l = a.depth - 1; l >= LEND; --l) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const char *ch = rs.ch(), *chn = rs.chn();
                    const DmaJob dj = rs.job();
                    const int pm = m == 0 ? MT - 1 : m - 1;
                    frag &d0 = m == 0 ? dl[KS - 2] : next[2 * (m > 0 ? m - 1 : 0)];
                    frag &d1 = m == 0 ? dl[KS - 1] : next[2 * (m > 0 ? m - 1 : 0) + 1];
                    const bool no_ga = (A.t.drop_ga && pnd_layer == a.depth - 1) || (GA0C && pnd_layer == 0);     // gA_{depth-1}'s last tile: not recorded; GA0C: gA_0 is staged
                    // GA0C: a finished gA_0 tile goes to the LDS staging image gaS[tile & 1] instead of the tape
                    const bool staged = GA0C && pnd_layer == 0 && !(BHN_GA0C_ABL & 2);
                    char *stage = staged ? gaS + (pm & 1) * STG + wvu * TB : nullptr;
                    // GA0C: the consumer of this step is wave (m - 2) mod MT: in layer 1 it adds the gA_0 tile m - 2 (staged in step
                    // m - 1, published by that step's barrier) of all waves to its accumulator; in the first two steps of the NEXT
                    // tile waves MT-2, MT-1 do the same for the previous tile's last two gA_0 tiles
                    bool cons_on = false;
                    const int ctile = (m + MT - 2) % MT;             // the gA_0 tile consumed in step m (of layer 1; m <= 1: of the next tile's first layer)
                    if constexpr (GA0C) {
                        if (!(BHN_GA0C_ABL & 1) && wvu == ctile % NCW) {
                            if (l == 1 && m >= 2) { cons_on = true; cons.ga = gaS + (m & 1) * STG; cons.enc = encS + tpar * STG; }
                            else if (l == a.depth - 1 && m <= 1 && have_prev) { cons_on = true; cons.ga = gaS + m * STG; cons.enc = encS + (tpar ^ 1) * STG; }
                        }
                    }
                    const float *bias_nx = (l == LEND && m == MT - 1) ? first_bias : zero_lds;
                    char *tdst = A.tape + ga_lin + pnd_layer * lin_stride + (qs * MT + pm) * TT;
                    TapePost<Pol, false, GA0C> post(pend, d0, d1, pnd_mask, em, tdst, nullptr, nullptr, no_acc, T8 && (pm & 1), false,
                                                    no_ga ? (edbg | 2) : edbg, t8_sc, t8_amax);
                    post.stage = stage; post.stage_off = stage_off;
                    const f32x16 acc = ring_step<W, Pol, RG, TapePost<Pol, false, GA0C>, NFR, GA0C>(ch, chn, ap, dl, enc, false, bias_nx, post, dj, sdbg);
                    if constexpr (GA0C) {
                        if (cons_on) { if (ctile / NCW) cons.run(cacc1, 2 * MT); else cons.run(cacc0, 2 * MT); }      // (m is unrolled: folded)
                    }
                    // without the gA_{depth-1} emissions the intervals around this layer's first DMA issue hold one
                    // emission less: the step ends whose window reaches back to it count one less (small widths: none)
                    if constexpr (GA0C) {
                        // tile emissions (ES stores each) among this step and the DIST - 2 before it: none in the steps whose pending tile
                        // is a gA_0 tile (layer 1, m >= 1) or gA_{depth-1}'s (first step of a tile)
                        static_assert(!GA0C || BHN_GA0C_DIST == 4, "the store counts below are written for a window of three steps");
                        // The first two steps of a tile also have the tile top's operations inside their window -- two encoded-input DMA
                        // pieces, the dout store, and with a next tile the MW relu-bit words fetched at the end of the previous tile's last
                        // layer and the next tile's prefetch (MW relu-bit words, e): counting them keeps the HBM latency of those loads out
                        // of the step end (every count is a lower bound)
                        constexpr int TOP = 3, TOPN = TOP + 2 * MW + 1;
                        if (l == a.depth - 1) {
                            if (m == 0) { if (has_next) rs.template step_end<TOPN>(); else rs.template step_end<TOP>(); }
                            else if (m == 1) { if (has_next) rs.template step_end<ES + TOPN>(); else rs.template step_end<ES + TOP>(); }
                            else if (m == 2) rs.template step_end<2 * ES>();
                            else rs.template step_end<3 * ES>();
                        } else if (l == 1) {
                            if (m == 0) rs.template step_end<3 * ES>();
                            else if (m == 1) rs.template step_end<2 * ES>();
                            else if (m == 2) rs.template step_end<ES>();
                            else rs.template step_end<0>();
                        } else rs.template step_end<3 * ES>();
                    }
                    else if (A.t.drop_ga && l == a.depth - 1 && m <= 4) rs.template step_end<YS_L1>();
                    else if (A.t.drop_ga && l == a.depth - 1) rs.template step_end<(KS >= 16 ? YS : 0)>();
                    else rs.template step_end<YS>();
                    pend = acc;
                    pnd_layer = l - 1;
                    pnd_mask = mwf[0][m >> 1] >> ((m & 1) * 16);       // word (l-1, m/2) of the FIFO's head layer
                    if constexpr (T8) {
                        if (m == 0) {       // the last tile of gA_l has been posted: its layer's |.|max to the thread's slot, on to gA_{l-1}
                            t8_flush(l, t8_amax);
                            t8_amax = 0u;
                            t8_sc = t8_lds[l - 1];
                        }
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KS - 2; ++ks) dl[ks] = next[ks];
                // the FIFO of relu-bit words moves on by one layer; its tail takes the layer MWL positions ahead of the one just finished
#pragma unroll
                for (int k = 0; k + 1 < MWL; ++k)
#pragma unroll
                    for (int i = 0; i < MW; ++i) mwf[k][i] = mwf[k + 1][i];
                layer_words(tile, (a.depth - 1 - l) + MWL, mwf[MWL - 1]);
            }
            if (a.depth > 1) {           // flush the last tile of gA_{LEND-1} (no further step to hide it behind)
                TapePost<Pol, false, GA0C> post(pend, dl[KS - 2], dl[KS - 1], pnd_mask, em,
                                              A.tape + ga_lin + (LEND - 1) * lin_stride + (qs * MT + MT - 1) * TT, nullptr, nullptr, no_acc, T8 && ((MT - 1) & 1), false, GA0C ? (edbg | 2) : edbg,
                                              t8_sc, t8_amax);
                if constexpr (GA0C) { post.stage = gaS + ((MT - 1) & 1) * STG + wvu * TB; post.stage_off = stage_off; }     // consumed in the second step of the next tile
                post.all();
                if constexpr (T8) t8_flush(LEND - 1, t8_amax);
            }
            if constexpr (GA0C) { tpar ^= 1; have_prev = true; }
        }   // MODE_CHAIN
    }
    if constexpr (GA0C) {
        // the last tile's gA_0 tiles MT-2, MT-1 have no next tile to be consumed in: one barrier (publishes tile MT-1's staging),
        // then waves MT-2, MT-1 add them; every wave flushes its accumulator tile to the workgroup's dW_0 slab
        if (have_prev) {
            lds_barrier();
            if (wvu == (MT - 2) % NCW || wvu == (MT - 1) % NCW) {       // tiles MT-2, MT-1: the second tile of their owners
                cons.ga = gaS + (wvu & 1) * STG; cons.enc = encS + (tpar ^ 1) * STG;
                cons.run(cacc1, 2 * MT);
            }
        }
        if (wvu < NCW) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float *tp = A.slab0 + ((long long)blockIdx.x * MT + wvu + k * NCW) * 1024;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) v[e4] = k ? cacc1[4 * g4 + e4] : cacc0[4 * g4 + e4];
                    f32x4 *dst = reinterpret_cast<f32x4 *>(tp + g4 * 256 + lane * 4);
                    if (A.accumulate) {
                        const f32x4 old = *dst;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) v[e4] += old[e4];
                    }
                    *dst = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
    clock_stamp(a.clk, MODE == MODE_CHAIN ? BHN_CLK_CHAIN : BHN_CLK_FWD_TRAIN, 1);
    if constexpr (T8 && MODE == MODE_CHAIN) {
        // the workgroup's largest |gA_l| per recorded layer -> the state block (t8_update_kernel turns it into the next call's scale)
        __syncthreads();
        if (tid < a.depth) {                  // (layer depth-1: the largest |dout|)
            float v = 0.f;
            for (int w = 0; w < Pol::NWAVES; ++w) v = __builtin_fmaxf(v, t8_lds[8 + tid * Pol::NWAVES + w]);
            atomicMax(reinterpret_cast<unsigned *>(A.t8) + 8 + tid, __float_as_uint(v));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dW kernel
// ---------------------------------------------------------------------------------------------
// Job types of the dW kernel (compile-time so that the streaming loop is straight-line code)
enum { JT_FIRST = 0, JT_HIDDEN = 1, JT_SKIP = 2, JT_OUT = 3, JT_HIDDEN1 = 4, JT_OUTSKIP = 5 };   // HIDDEN1: layer 1 with h_1 recomputed from the
                                                     // encoded inputs; OUTSKIP: output layer fed by concat[h, enc] (odd depths with do_skip)

// The tape stream of one dW job: every group is NPJ pieces of 1 KiB (one wave-wide 16-byte LDS-DMA) out of up to four
// regions -- [A tiles][h tiles][enc tile][dout piece] -- and wave w issues pieces NW*i + w.  A ROW of NW consecutive
// pieces that lies inside one region needs no per-piece address arithmetic: one buffer resource per region and group
// (base = the wave's first piece), the row as a compile-time scalar offset.  Round 2 ISA census of the f32 kernel: with
// 17 pieces per wave, a run-time region select and a 64-bit multiply-add per piece, the issue was ~330 SALU instructions
// in front of every group's MFMAs (one wave per SIMD: nothing hides them).  Rows that straddle regions (bf16: the 2-piece
// enc tile + dout piece) keep the generic per-piece form.
template <int NW, int PA, int PH, int PE, int PD, int OFF_H, int OFF_E, int OFF_D>
struct TapeStream {
    static constexpr int NPJ = PA + PH + PE + PD, PPW = (NPJ + NW - 1) / NW;
    static constexpr int region_of(int piece) { return piece < PA ? 0 : piece < PA + PH ? 1 : piece < PA + PH + PE ? 2 : 3; }
    static constexpr int region_start(int r) { return r == 0 ? 0 : r == 1 ? PA : r == 2 ? PA + PH : PA + PH + PE; }
    static constexpr int region_lds(int r) { return r == 0 ? 0 : r == 1 ? OFF_H : r == 2 ? OFF_E : OFF_D; }
    static constexpr bool uniform_row(int i) {
        return i >= 0 && NW * i + NW <= NPJ && region_of(NW * i) == region_of(NW * i + NW - 1);
    }
    static constexpr int mixed_before(int i) { int n = 0; for (int k = 0; k < i; ++k) n += uniform_row(k) ? 0 : 1; return n; }
    static constexpr int NMIX = mixed_before(PPW) > 0 ? mixed_before(PPW) : 1;
    // Every member is indexed by compile-time constants only (a run-time index would put the object into scratch memory):
    // the region of this wave's piece of a mixed row is resolved once, here, on the constructor's arguments.
    const char *src[4];          // region bases
    long long stride[4];         // bytes per group
    const char *msrc[NMIX];      // mixed rows: base of this wave's piece, its bytes per group, its LDS offset
    long long mstride[NMIX];
    int mlds[NMIX];
    int wvu;
    unsigned voffA;              // global-side lane offset of the region-0 pieces (default: lane-linear)
    DEVI TapeStream(const char *a, long long sa, const char *h, long long sh, const char *e, long long se, const char *d, long long sd, int wave,
                    int voff_a = -1) {
        wvu = wave;
        voffA = voff_a >= 0 ? (unsigned)voff_a : (threadIdx.x & 63) * 16;
        src[0] = a; src[1] = h; src[2] = e; src[3] = d;
        stride[0] = sa; stride[1] = sh; stride[2] = se; stride[3] = sd;
        int k = 0;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (uniform_row(i)) continue;
            int piece = wave + NW * i;
            piece = piece < NPJ ? piece : NPJ - 1;                    // tail waves re-issue the last piece (equal vmcnt for all waves)
            const int r = piece < PA ? 0 : piece < PA + PH ? 1 : piece < PA + PH + PE ? 2 : 3;
            const int st = r == 0 ? 0 : r == 1 ? PA : r == 2 ? PA + PH : PA + PH + PE;
            const int ld = r == 0 ? 0 : r == 1 ? OFF_H : r == 2 ? OFF_E : OFF_D;
            const unsigned long long base = reinterpret_cast<unsigned long long>(r == 0 ? a : r == 1 ? h : r == 2 ? e : d);
            msrc[k] = reinterpret_cast<const char *>(base + (unsigned long long)((piece - st) * 1024));
            mstride[k] = r == 0 ? sa : r == 1 ? sh : r == 2 ? se : sd;
            mlds[k] = ld + (piece - st) * 1024;
            ++k;
        }
    }
    static DEVI u32x4 rsrc(const char *p) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return u32x4{lo, hi & 0xffffu, 0x7fffffffu, 0x00020000u};
    }
    // voff: the lane's byte offset on the GLOBAL side (the LDS side of a piece is lane-linear).  Any permutation of the
    // 64 16-byte slots of a piece can be had for free by permuting these offsets (voffA, region 0).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (only: "inline asm clobber list contains reserved registers: M0")
    template <int POLICY>
    static DEVI void dma(const u32x4 &rs, unsigned soff, unsigned m, unsigned voff) {
        if constexpr (POLICY == 1)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
        else if constexpr (POLICY == 2)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" ::"s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
    }
#pragma clang diagnostic pop
    template <int POLICY, int I>
    DEVI void row(long long q, unsigned lb, u32x4 (&rs)[4]) const {
        constexpr int p0 = NW * I;
        if constexpr (uniform_row(I)) {                               // a whole row inside one region
            constexpr int r = region_of(p0), rel = (p0 - region_start(r)) * 1024;
            if constexpr (!(uniform_row(I - 1) && region_of(NW * (I - 1)) == r))      // first such row: the region's resource
                rs[r] = rsrc(src[r] + wvu * 1024 + q * stride[r]);
            dma<POLICY>(rs[r], (unsigned)rel, lb + (unsigned)(wvu * 1024 + region_lds(r) + rel), r == 0 ? voffA : (threadIdx.x & 63) * 16);
        } else {                                                      // a row across regions / the ragged tail
            constexpr int k = mixed_before(I);
            dma<POLICY>(rsrc(msrc[k] + q * mstride[k]), 0u, lb + (unsigned)mlds[k], (threadIdx.x & 63) * 16);
        }
    }
    template <int POLICY, int... I>
    DEVI void rows(long long q, unsigned lb, u32x4 (&rs)[4], std::integer_sequence<int, I...>) const {
        (row<POLICY, I>(q, lb, rs), ...);
    }
    // group q -> LDS image at `buf`
    template <int POLICY>
    DEVI void issue(long long q, char *buf) const {
        const unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(buf));
        u32x4 rs[4];                                                  // (indexed by compile-time constants only)
        rows<POLICY>(q, lb, rs, std::make_integer_sequence<int, PPW>{});
    }
};

// dW job body of the f32 policy (all jobs) and of the bf16 output-layer job at depths < 3 (deeper bf16 networks: dw_body2).
template <int W, class Pol, int JT>
DEVI void dw_body(const BwdArgs &A, int job, char *smem) {
    using BG = BwdGeom<W, Pol>;
    using frag = typename Pol::frag;
    static_assert(JT != JT_HIDDEN1, "h_1 recompute: dw_body2");
    constexpr int MT = BG::MT, TB = BG::TILE_BYTES;
    constexpr int OFF_H = MT * TB, OFF_E = 2 * MT * TB;               // LDS group image [A][h][enc]
    constexpr int GB = BG::GROUP_BYTES;
    constexpr bool out_job = JT == JT_OUT || JT == JT_OUTSKIP, has_h = JT != JT_FIRST, has_enc = (JT == JT_FIRST || JT == JT_SKIP || JT == JT_OUTSKIP);
    constexpr int mtA = out_job ? 1 : MT;                              // A tiles (gA rows; dout is 1 row)
    constexpr int nH = has_h ? MT : 0, nB = nH + (has_enc ? 1 : 0);    // B tiles; slab tile nB holds the bias column,
    constexpr int NT = nB;                                             // summed by VALU from the A fragments (Pol::sum8)
    constexpr bool TR = Pol::ELEM_BYTES == 2;                          // tape tiles are point-on-lane images: transposed LDS reads
    constexpr bool AGPR = Pol::NWAVES == 4;                            // one wave per SIMD (f32): accumulators in AGPRs, one sweep
    const int trl = tr_lane_off();
    // wave grid: rows = A tiles, columns = B tiles; the output job has ONE A tile (dout), so all waves go to the columns
    // (with the hidden jobs' grid only a quarter of its waves had work: 4x the MFMA time per group of an f32 job)
    constexpr int WRR = out_job ? 1 : BG::WRR, WCC = Pol::NWAVES / WRR;
    constexpr int MPW = (mtA + WRR - 1) / WRR;                         // A tiles per wave (1 for the output job)
    constexpr int NPWJ = (NT + WCC - 1) / WCC;                         // B tiles owned by one wave
    constexpr int NPASS = (NPWJ + BG::SWEEP - 1) / BG::SWEEP, NPW = (NPWJ + NPASS - 1) / NPASS;
    constexpr int NPIN = 256 / (16 * MPW);                             // accumulator tiles per A tile that fit the 256 AGPRs
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: wave-uniform branches
    const int nwg = A.wg_begin[job + 1] - A.wg_begin[job];
    const int kb = blockIdx.x - A.wg_begin[job];
    const long long q0 = uniform64(A.t.NQ * kb / nwg), q1 = uniform64(A.t.NQ * (kb + 1) / nwg);
    const char *srcA = out_job ? A.tape + A.t.dout_off : A.tape + A.t.ga_off[out_job ? 0 : job];
    const long long strideA = out_job ? A.t.dout_stride : (long long)MT * TB;       // dout is one tile per group
    const char *srcH = has_h ? A.tape + A.t.h_off[job] : nullptr;
    const char *srcE = A.tape + A.t.enc_off;
    const int wr = wv % WRR, wc = wv / WRR;
    const bool wave_works = wr * MPW < mtA;                            // output job: only the wr == 0 waves
    float *slab = A.f.slabs + (long long)blockIdx.x * BG::SLAB_FLOATS;
    // pieces (1 KiB = one wave-wide DMA) this job really needs: [A tiles | dout][h tiles][enc tile]
    constexpr int PA = out_job ? TB / 1024 : MT * TB / 1024, PH = has_h ? MT * TB / 1024 : 0, PE = has_enc ? TB / 1024 : 0;
    using Stream = TapeStream<Pol::NWAVES, PA, PH, PE, 0, OFF_H, OFF_E, 0>;
    constexpr int PPW = Stream::PPW;
    const Stream stream(srcA, strideA, srcH, (long long)MT * TB, srcE, TB, nullptr, 0, wv);

    for (int pass = 0; pass < NPASS; ++pass) {
        const int nbase = wc * NPWJ + pass * NPW;
        // per-tile LDS offsets, fixed for the whole stream (no branches in the loop); a wave column whose share of the
        // B tiles ends early computes on a duplicate of a real tile and drops the result at the flush
        int boff[NPW];
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) {
            const int n = nbase + ni;
            boff[ni] = (n < nH) ? OFF_H + n * TB : (has_enc ? OFF_E : OFF_H);
        }
        const bool has_tiles = nbase < nB;
        const bool bias_rows = wc == 0 && pass == 0;          // this wave sums the bias column of its A tiles
        float bsum[MPW];
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) bsum[mi] = 0.f;
        f32x16 acc[MPW][NPW];
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) {
                if constexpr (AGPR) {
                    // zeroed by an MFMA with a literal-zero SrcC, straight into the AGPRs: 256 zeros initialised through
                    // VGPRs made hipcc spill the stream's state, and every reload inside the loop is an
                    // `s_waitcnt vmcnt(0)` that also waits for the group just issued
                    const f32x16 z = {};
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, z, 0, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
                }
            }
        __syncthreads();
        // B fragments are fetched AHEAD MFMA groups ahead of their use (counted lgkmcnt instead of a full drain after
        // every read).  A fragments of both k-steps are loaded up front.
        auto load_b = [&](const char *gp, int t) -> frag {
            if constexpr (TR) return tr_frag(gp + boff[t % NPW], t / NPW, trl);
            else return Pol::lds_frag(gp + boff[t % NPW], t / NPW, lane);
        };
        auto load_a = [&](const char *tile, int s2) -> frag {
            if constexpr (TR) return tr_frag(tile, s2, trl);
            else return Pol::lds_frag(tile, s2, lane);
        };
        auto compute_group = [&](const char *gp) {
            constexpr int NTOT = 2 * NPW, AHEAD = (Pol::ELEM_BYTES == 2) ? 2 : 1;
            if constexpr (AGPR) {
                // the accumulators stay in AGPRs, in place (the MFMA takes SrcC / vDst from either file).  Left to itself
                // hipcc keeps the loop-carried tiles in VGPRs and copies them to AGPRs and back around every group
                // (v_accvgpr_write / _read: 256 VALU per group behind the MFMA results) -- the f32 dW kernel ran at 64 %
                // matrix-pipe occupancy.  The empty asm ties the tiles to the AGPR class across the back edge.
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NPW && ni < NPIN; ++ni) asm volatile("" : "+a"(acc[mi][ni]));
            }
            frag af[2][MPW];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) {
                    af[s][mi] = load_a(gp + (wr * MPW + mi) * TB, s);
                    if (bias_rows) bsum[mi] = Pol::sum8(af[s][mi], bsum[mi]);
                }
            frag bq[AHEAD + 1];
#pragma unroll
            for (int t = 0; t < AHEAD && t < NTOT; ++t) bq[t] = load_b(gp, t);
#pragma unroll
            for (int t = 0; t < NTOT; ++t) {
                if (t + AHEAD < NTOT) bq[(t + AHEAD) % (AHEAD + 1)] = load_b(gp, t + AHEAD);
                __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of this step's MFMAs
                const int s = t / NPW, ni = t % NPW;
                const frag bnow = bq[t % (AHEAD + 1)];
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) acc[mi][ni] = Pol::mma(af[s][mi], bnow, acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            // LDS-DMA ring: group q+NBUF-1 is issued right after the barrier that proves buffer (q-1)%NBUF has been
            // consumed; the counted wait leaves NBUF-2 younger groups in flight.  (f32: 2 buffers of 68 KB -- the DMA of
            // group q+1 runs under the MFMAs of group q, which are several times longer than an HBM round trip)
            constexpr int NBUF = BG::NBUF;
            auto issue = [&](long long q, char *buf) {
                q = q < q1 ? q : q1 - 1;
                if (BHN_DBG(A.wrap)) q %= A.wrap;
                if (BHN_DBG(A.policy == 1)) stream.template issue<0>(q, buf);
                else if (BHN_DBG(A.policy == 2)) stream.template issue<2>(q, buf);
                else stream.template issue<1>(q, buf);
            };
            constexpr int INFLIGHT = NBUF - 2;
            if (q0 < q1) {
#pragma unroll
                for (int j = 0; j < NBUF - 1; ++j) issue(q0 + j, smem + j * GB);
                int it = 0;
                for (long long q = q0; q < q1; ++q) {
                    if (!BHN_DBG(A.debug & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT * PPW) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");      // the raw barrier is not a compiler fence: keep the
                                                         // DMA issue and the ds_reads below it
                    const int nx = (it == 0) ? NBUF - 1 : it - 1;
                    if (!BHN_DBG(A.debug & 2)) issue(q + NBUF - 1, smem + nx * GB);
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (!BHN_DBG(A.debug & 1) && wave_works && has_tiles) compute_group(smem + it * GB);
                    it = (it == NBUF - 1) ? 0 : it + 1;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        // ---- flush this pass's partial dW^T tiles: slab[(m*NTMAX+n)][r/4][lane][r%4] ------------
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NPW; ++ni) {
                const int m = wr * MPW + mi, n = nbase + ni;
                if (m >= mtA || n >= nB || n >= (wc + 1) * NPWJ) continue;
                // (AGPR: the tile stays in its AGPRs until here -- hipcc otherwise copies all tiles to VGPRs at the loop exit,
                //  256 live VGPRs whose pressure spills the loop's state)
                if constexpr (AGPR) { if (ni < NPIN) asm volatile("" : "+a"(acc[mi][ni])); }
                float *tp = slab + (long long)(m * BG::NTMAX + n) * 1024;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g4 + e];
                    f32x4 *dst = reinterpret_cast<f32x4 *>(tp + g4 * 256 + lane * 4);
                    if (A.accumulate) {
                        const f32x4 old = *dst;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += old[e];
                    }
                    *dst = v;
                }
                __builtin_amdgcn_sched_barrier(0);      // one tile at a time
            }
        // bias column (slab tile nB, column 0: what reduce_kernel reads): row i of A tile m is held by lanes i and i + 32
        if (bias_rows) {
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                const int m = wr * MPW + mi;
                float v = bsum[mi] + __shfl_xor(bsum[mi], 32, 64);
                if (m < mtA && lane < 32) {
                    const int hh = (lane >> 2) & 1, r = (lane & 3) + 4 * (lane >> 3);
                    float *dst = slab + (long long)(m * BG::NTMAX + nB) * 1024 + (r >> 2) * 256 + (32 * hh) * 4 + (r & 3);
                    if (A.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// dW job body, bf16, software-pipelined over the 32-point groups (round 2).
//
// The first version (dw_body above, still used by the f32 policy) ran per group: vmcnt wait -> barrier -> DMA issue ->
// A-fragment LDS reads -> wait -> MFMAs; with the two waves of a SIMD in barrier lockstep everything in front of the
// MFMAs was dead time for the matrix pipe (about 2600 cycles per group for 1024-1280 cycles of MFMA work; the
// kernel took the same time whether its tape came from HBM or from an L2-resident window, tools/dbg_wrap.py).  Here
// group q+1 is published one barrier EARLY (the wait at the top of iteration q is for group q+1), so that while the
// MFMAs of group q run, the A fragments of group q+1 are read from LDS and prepared (layer depth-1: gA rebuilt from
// h_depth, the output layer's row; all: bias sums) and its first two B fragments are fetched.  Ring of 4 group buffers:
// q consumed, q+1 landed, q+2 and q+3 in flight.  State is double-buffered in registers by unrolling the loop twice.
//
// Other changes against dw_body: the skip layer's encoded-input tile is one EXTRA accumulator tile per wave (wave
// (wr, wc) pairs it with its A tile mi == wc) instead of a ninth column tile that left half of a 2x5 tile grid idle;
// layer depth-1 (LAST) makes the output layer's row and bias with v_dot2c / adds from the h_depth fragments and the f32
// dout it already holds (no dout tile on the tape, no 1-row MFMAs, 32 accumulator registers fewer).
// ---------------------------------------------------------------------------------------------
// LBITS (round 5, TapeLayout::lbits): the LAST job takes relu'(a_{depth-1}) from the relu-bit words the forward records anyway
// (1 KiB per group instead of the MT h_depth tiles: 16) and does not make the output layer's row from h_depth at all -- at the
// flush it forms  sum_k K[k][f] G[k][f] + b[f] g[f]  from its own (un-folded) accumulators, which IS that row because
// h_depth = relu(a) = relu'(a) a and a = K^T h_{depth-1} + b (bwd_common.h, drop_hd).
template <int W, class Pol, int JT, bool LAST = false, bool OUTENC = false, bool LBITS = false>     // OUTENC: the output layer riding on LAST takes concat[h, enc]
DEVI void dw_body2(const BwdArgs &A, int job, char *smem) {
    using BG = BwdGeom<W, Pol>;
    using frag = typename Pol::frag;
    static_assert(Pol::ELEM_BYTES == 2 && JT != JT_OUT && JT != JT_OUTSKIP, "bf16 jobs of layers 0 .. depth-1");
    static_assert(!LAST || JT == JT_HIDDEN || JT == JT_SKIP, "LAST: hidden / skip job");
    static_assert(!OUTENC || LAST, "OUTENC: a LAST job");
    static_assert(!LBITS || (LAST && !Pol::TAPE8 && JT == JT_HIDDEN), "LBITS: the bf16 LAST job of a plain hidden layer");
    static_assert(BG::NBUF == 4, "ring of four group buffers");
    constexpr int MT = BG::MT, TB = BG::TILE_BYTES, TT = BG::TAPE_TILE;
    constexpr bool T8 = Pol::TAPE8;                                    // 8-bit h / gA tape tiles (tr8_read / tr8_widen)
    constexpr bool has_h = JT != JT_FIRST, make_h = JT == JT_HIDDEN1, enc_extra = JT == JT_SKIP;
    static_assert(!T8 || !OUTENC, "8-bit tape: even depths / no skip into the output layer");
    // LDS group image [A][h][enc][f32 dout piece]; 8-bit tape: A and streamed h tiles are 1 KiB, a recomputed h_1 tile 2 KiB
    constexpr int OFF_H = MT * TT, OFF_E = T8 ? OFF_H + (make_h ? MT * TB : MT * TT) : 2 * MT * TB, OFF_D32 = OFF_E + TB;
    constexpr int GB = OFF_D32 + (LAST ? 1024 : 0);
    static_assert(T8 || (OFF_D32 == BG::GROUP_BYTES && GB == (LAST ? BG::GROUP_BYTES_LAST2 : BG::GROUP_BYTES)), "group image");
    static_assert(GB <= BG::GROUP_BYTES + 1024, "group image larger than the host's LDS budget");
#ifndef BHN_T8_NBUF
#define BHN_T8_NBUF 4            // ring depth of the 8-bit tape's dW jobs (8 = the bf16 jobs' bytes in flight: measured no faster, the jobs are VALU-bound)
#endif
    constexpr int NB = (T8 && !make_h) ? BHN_T8_NBUF : 4;              // group buffers of the ring (a power of two)
    static_assert((NB & (NB - 1)) == 0 && NB >= 4 && NB * GB <= 160 * 1024, "ring size");
    constexpr int nH = has_h ? MT : 0, nB = nH + ((JT == JT_FIRST || JT == JT_SKIP) ? 1 : 0);   // slab tile nB: the bias column
    constexpr int nBr = has_h ? MT : 1;                                // B tiles of the regular tile grid
    constexpr int WRR = BG::WRR, WCC = BG::WCC;
    constexpr int MPW = (MT + WRR - 1) / WRR, NPW = (nBr + WCC - 1) / WCC;
    static_assert(NPW <= 5, "one sweep");
    constexpr bool B8 = T8 && has_h && !make_h;                       // the B tiles are 8-bit tape tiles too
#ifndef BHN_T8_ABL
#define BHN_T8_ABL 0             // measurement builds (results wrong): 1 no e4m3 MFMAs, 2 no A preparation, 4 no B sorting in the e4m3 jobs, 8 the tape stream only (no MFMA phase in any 8-bit job)
#endif
#ifndef BHN_T8_F8MFMA
#define BHN_T8_F8MFMA 1          // 0 (A/B builds): widen both operands to bf16 in front of every MFMA (the first version)
#endif
    // both operands 8-bit: the weight-gradient tiles are accumulated by v_mfma_f32_32x32x16_fp8_fp8 straight from the sorted
    // bytes (gA / scale times h; the scale is applied at the flush); bf16 is made only where bf16 code needs it (bias sums,
    // the encoded-input tile, the output layer's row)
    constexpr bool F8 = B8 && BHN_T8_F8MFMA != 0;
    static_assert(!T8 || MPW == 2, "8-bit tape: the A tiles come in pairs (width 256)");
    static_assert(!B8 || NPW % 2 == 0, "8-bit tape: the B tiles come in pairs (width 256)");
    constexpr int ME = (MPW + WCC - 1) / WCC;                          // A tiles a wave pairs with the enc tile / the output row
    constexpr int NTOT = 2 * NPW;                                      // MFMA steps per group (k-step major)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: wave-uniform branches
    const int trl = tr_lane_off();
    const int tr8 = tr8_lane_off();
    // 8-bit tape: what the A bytes are multiplied by on their way to bf16 (gA_l: the layer's scale; h_depth of the LAST job: 1)
    float scaleA = 1.f;
    if constexpr (T8 && !LAST) scaleA = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, A.t8[job])));
    if constexpr (F8 && LAST) {          // the scale of dout (= of gA_{depth-1} without its W_out factor): the A bytes are (h != 0) e4m3(dout / scaleA)
        const float v = A.t8[job];
        scaleA = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (v > 0.f && v < __builtin_inff()) ? v : 1.f)));
    }
    const int nwg = A.wg_begin[job + 1] - A.wg_begin[job];
    const int kb = blockIdx.x - A.wg_begin[job];
    const long long q0 = uniform64(A.t.NQ * kb / nwg), q1 = uniform64(A.t.NQ * (kb + 1) / nwg);
    constexpr int MWB = ((MT + 1) / 2) * 256;                           // bytes of one layer's relu-bit words of a group
    const char *srcA = LBITS ? A.tape + A.t.mask_off + (long long)(A.f.depth - 1) * MWB
                             : LAST ? A.tape + A.t.h_off[job + 1] : A.tape + A.t.ga_off[job];
    const char *srcD = A.tape + A.t.dout_off;
    const char *srcH = has_h ? A.tape + A.t.h_off[job] : nullptr;
    const char *srcE = A.tape + (make_h ? A.t.encp_off : A.t.enc_off);
    const int wr = wv % WRR, wc = wv / WRR;
    const int nbase = wc * NPW;
    const bool works = wr * MPW < MT && nbase < nBr;                   // this wave owns accumulator tiles
    // HIDDEN1: W_0 and b_0 behind the ring (see dw_body)
    char *w0_lds = smem + NB * GB;
    float *b0_lds = reinterpret_cast<float *>(w0_lds + 2 * MT * Pol::FRAG_BYTES);
    if constexpr (make_h) {
        const char *w0 = A.f.packed + A.f.fwd_off;
        for (int i = tid; i < 2 * MT * Pol::FRAG_BYTES / 16; i += Pol::NTHREADS)
            reinterpret_cast<u32x4 *>(w0_lds)[i] = reinterpret_cast<const u32x4 *>(w0)[i];
        for (int i = tid; i < W; i += Pol::NTHREADS) b0_lds[i] = reinterpret_cast<const float *>(A.f.packed + A.f.bias_off)[i];
    }
    struct HIn { frag e0, e1, w0, w1; float b; };
    auto make_h_read = [&](const char *gp) {
        HIn in;
        in.e0 = Pol::lds_frag(gp + OFF_E, 0, lane); in.e1 = Pol::lds_frag(gp + OFF_E, 1, lane);
        const int t = wv < MT ? wv : 0;
        in.w0 = Pol::lds_frag(w0_lds, 2 * t, lane); in.w1 = Pol::lds_frag(w0_lds, 2 * t + 1, lane);
        in.b = b0_lds[32 * t + (lane & 31)];
        return in;
    };
    auto make_h_mma = [&](const HIn &in) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = in.b;
        acc = Pol::mma(in.e0, in.w0, acc);
        acc = Pol::mma(in.e1, in.w1, acc);
        return acc;
    };
    auto make_h_write = [&](char *gp, const f32x16 &acc) {
        if (wv < MT) {
            frag o[2];
            unsigned unused = 0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) Pol::relu_pair(o[r >> 3], (r & 7) >> 1, r >> 1, acc[r], acc[r + 1], unused);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) *reinterpret_cast<frag *>(gp + OFF_H + wv * TB + s2 * Pol::FRAG_BYTES + lane * 16) = o[s2];
        }
    };
    // LAST: W_out of this lane's feature in each of the wave's A tiles
    float wout_r[MPW];
    if constexpr (LAST) {
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            const int f = T8 ? t8_feature(wr * MPW + mi, lane & 31) : 32 * (wr * MPW + mi) + (lane & 31);
            wout_r[mi] = f < W ? reinterpret_cast<const float *>(A.f.packed + A.f.wout_off)[f] : 0.f;
        }
    }
    float *slab = A.f.slabs + (long long)blockIdx.x * BG::SLAB_FLOATS;

    int boff[NPW];
#pragma unroll
    for (int ni = 0; ni < NPW; ++ni) {
        const int n = nbase + ni;
        boff[ni] = has_h ? OFF_H + (n < nH ? n : 0) * (B8 ? TT : TB) : OFF_E;        // a column share that ends early repeats tile 0 (dropped at the flush)
    }
    // LBITS: where this lane's feature (row lane & 31 of A tile T) sits in the forward's relu-bit words
    int lb_word[MPW];
    unsigned lb_pos[MPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int T = wr * MPW + mi, i = lane & 31, hf = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
        lb_word[mi] = (T >> 1) * 64 + 32 * hf + 4 * (lane >> 5);
        lb_pos[mi] = 16u * (T & 1) + ((r & 1) ? 8u + (r >> 1) : (unsigned)(r >> 1));
    }
    const bool bias_rows = wc == 0;                                    // this wave sums the bias column of its A tiles
    const bool out_bias_wave = LAST && wr == 0 && wc == 0;             // ... and this one the output layer's bias
    f32x16 acc[MPW][NPW], acc_e[ME];
    float bsum[MPW], orow[ME], bout = 0.f, oenc = 0.f;
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        bsum[mi] = 0.f;
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < ME; ++e) {
        orow[e] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_e[e][r] = 0.f;
    }

    // ---- per-group A state: fragments of the wave's A tiles for both k-steps (+ LAST: the f32 dout of the lane's points)
    struct AState { frag af[2][MPW]; f32x4 da[2], db[2]; frag ef[2]; u32x2 a8[2][MPW]; };      // ef: encoded-input fragments (OUTENC, one wave); a8: e4m3 fragments (F8)
    auto load_b = [&](const char *gp, int t) -> frag {
        if constexpr (make_h) return Pol::lds_frag(gp + boff[t % NPW], t / NPW, lane);     // written by make_h_write in fragment order
        else return tr_frag(gp + boff[t % NPW], t / NPW, trl);
    };
    // 8-bit B tiles: raw read of pair u = (k-step u / (NPW/2), tiles 2 (u % (NPW/2)), +1 of the wave's share)
    constexpr int NPH = NPW / 2 > 0 ? NPW / 2 : 1;                    // (one B tile per wave: the 8-bit read is never instantiated)
    auto load_braw = [&](const char *gp, int u) -> Raw8 { return tr8_read(gp + boff[2 * (u % NPH)], u / NPH, tr8); };
    auto a_load = [&](const char *gp, AState &st) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                if constexpr (T8) {
                    // the raw pair read (4 dwords) rides in af[s2][0] until a_prep widens it into af[s2][0], af[s2][1]
                    if (mi == 0) {
                        const Raw8 raw = tr8_read(gp + (wr * MPW) * TT, s2, tr8);
                        st.af[s2][0] = __builtin_bit_cast(frag, (u32x4){raw.lo[0], raw.lo[1], raw.hi[0], raw.hi[1]});
                    }
                } else if constexpr (LBITS) {
                    // the lane's feature i of A tile T was accumulator element r of lane half hf in the forward: bit `lb_pos[mi]` of
                    // the word [T >> 1][point + 32 hf]; the lane's eight points of k-step s2 are two runs of four words
                    const unsigned *wp = reinterpret_cast<const unsigned *>(gp) + lb_word[mi] + 16 * s2;
                    const u32x4 wa = *reinterpret_cast<const u32x4 *>(wp), wb = *reinterpret_cast<const u32x4 *>(wp + 8);
                    u32x4 on;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned we = d < 2 ? wa[2 * d] : wb[2 * d - 4], wo = d < 2 ? wa[2 * d + 1] : wb[2 * d - 3];
                        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)we, lb_pos[mi], 1u), hi = (unsigned)__builtin_amdgcn_sbfe((int)wo, lb_pos[mi], 1u);
                        on[d] = (lo & 0xffffu) | (hi & 0xffff0000u);
                    }
                    st.af[s2][mi] = __builtin_bit_cast(frag, on);
                } else st.af[s2][mi] = tr_frag(gp + (wr * MPW + mi) * TB, s2, trl);
            }
            if constexpr (LAST) {
                // dout of this lane's eight points of k-step s2: tape point order p = (j&3) + 8(j>>2) + 16 s2 + 4 (lane>>5)
                const float *d32 = reinterpret_cast<const float *>(gp + OFF_D32) + 16 * s2 + 4 * (lane >> 5);
                st.da[s2] = *reinterpret_cast<const f32x4 *>(d32);
                st.db[s2] = *reinterpret_cast<const f32x4 *>(d32 + 8);
                if constexpr (OUTENC) st.ef[s2] = tr_frag(gp + OFF_E, s2, trl);       // lane = encoded-input slot, 8 points
            }
        }
    };
    // piece k = (k-step, A tile) of the preparation of a group's A fragments; `live`: the group exists (the state of
    // the group behind the last one is loaded but must not be accumulated)
    auto a_prep = [&](AState &st, int k, bool live) {
        const int s2 = k / MPW, mi = k % MPW;
        if constexpr (F8) {
#if (BHN_T8_ABL & 2)
            return;
#endif
            if (mi == 0) {
                const u32x4 rw = __builtin_bit_cast(u32x4, st.af[s2][0]);
                Raw8 raw;
                raw.lo = (u32x2){rw[0], rw[1]}; raw.hi = (u32x2){rw[2], rw[3]};
                const Pair8 pr = tr8_sort(raw);
                st.a8[s2][0] = pr.a; st.a8[s2][1] = pr.b;
            }
            if constexpr (!LAST) {
                // bf16 copies only where bf16 code reads them: the bias sums (wave column 0) and the encoded-input MFMA
                if (bias_rows || (enc_extra && (mi % WCC) == wc)) st.af[s2][mi] = widen8(st.a8[s2][mi], scaleA);
            } else {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                const f32x4 da = st.da[s2], db = st.db[s2];
                const u32x2 h8 = st.a8[s2][mi];                      // h_depth of the lane's feature, 8 points
                const bool my_row = (mi % WCC) == wc;
                // e4m3(dout / scaleA) of the lane's eight points (the chain limited dout to the range when it recorded it)
                u32x2 d8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 d4 = i == 0 ? da : db;
                    s16x2 r = {0, 0};
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, d4[0], d4[1], scaleA, false);
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, d4[2], d4[3], scaleA, true);
                    d8[i] = __builtin_bit_cast(unsigned, r);
                }
                if (my_row) {        // the output layer's row: dW_out[f] += dout_p h_depth[p][f] (bf16 operands, v_dot2c)
                    const frag hb = widen8(h8, 1.f);
                    float o = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2 dd = {i < 2 ? da[2 * i] : db[2 * i - 4], i < 2 ? da[2 * i + 1] : db[2 * i - 3]};
                        const typename Pol::bf16x2 dpk = {(__bf16)dd[0], (__bf16)dd[1]};
                        const typename Pol::bf16x2 hp = {hb[2 * i], hb[2 * i + 1]};
                        o = __builtin_amdgcn_fdot2_f32_bf16(hp, dpk, o, false);
                    }
                    if (live) orow[mi / WCC] += o;
                }
                // A bytes = (h != 0) ? e4m3(dout / scaleA) : 0.  h bytes are <= 0x7e: + 0x7f sets bit 7 of every nonzero byte, no carries
                u32x2 g8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned t = (h8[i] + 0x7f7f7f7fu) & 0x80808080u;
                    g8[i] = d8[i] & (t | (t - (t >> 7)));
                }
                st.a8[s2][mi] = g8;
                if (bias_rows || (enc_extra && (mi % WCC) == wc)) st.af[s2][mi] = widen8(g8, scaleA);      // (the bias of layer depth-1, summed below)
                if (live && out_bias_wave && mi == 0) bout += (da[0] + da[1]) + (da[2] + da[3]) + (db[0] + db[1]) + (db[2] + db[3]);
            }
            if (live && bias_rows) bsum[mi] = Pol::sum8(st.af[s2][mi], bsum[mi]);
            return;
        } else if constexpr (T8) {
            if (mi == 0) {
                const u32x4 rw = __builtin_bit_cast(u32x4, st.af[s2][0]);
                Raw8 raw;
                raw.lo = (u32x2){rw[0], rw[1]}; raw.hi = (u32x2){rw[2], rw[3]};
                tr8_widen(raw, scaleA, st.af[s2][0], st.af[s2][1]);
            }
        }
        if constexpr (LAST) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef short i16x2 __attribute__((ext_vector_type(2)));
            const f32x4 da = st.da[s2], db = st.db[s2];
            const frag rawf = st.af[s2][mi];
            const u32x4 raw = __builtin_bit_cast(u32x4, rawf);
            u32x4 ga;
            float o = 0.f;
            const bool my_row = !LBITS && (mi % WCC) == wc;        // (scalar) this wave makes the output layer's row for this A tile
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 dd = {i < 2 ? da[2 * i] : db[2 * i - 4], i < 2 ? da[2 * i + 1] : db[2 * i - 3]};
                // the output layer's row: dW_out[f] += dout_p h_depth[p][f], bf16 operands, f32 accumulation (v_dot2c)
                const typename Pol::bf16x2 dpk = {(__bf16)dd[0], (__bf16)dd[1]};
                const typename Pol::bf16x2 hp = {rawf[2 * i], rawf[2 * i + 1]};
                if (my_row) o = __builtin_amdgcn_fdot2_f32_bf16(hp, dpk, o, false);
                if constexpr (OUTENC) {      // ... and its encoded-input part dW_out[W + slot] += dout_p enc[p][slot] (one wave, once per k-step)
                    if (live && out_bias_wave && mi == 0) {
                        const typename Pol::bf16x2 ep = {st.ef[s2][2 * i], st.ef[s2][2 * i + 1]};
                        oenc = __builtin_amdgcn_fdot2_f32_bf16(ep, dpk, oenc, false);
                    }
                }
                // gA_{depth-1}[p][f] = (h_depth[p][f] != 0) W_out[f] dout_p.  W_out[f] is constant along the sum over points,
                // so the A operand is only (h != 0) bf16(dout_p) -- three VALU per two points instead of seven (this job was
                // VALU-bound: 237 VALU against 20 MFMAs per group) -- and the rows of dW_{depth-1} (and its bias) are scaled
                // by W_out[f] in f32 at the flush, which is also closer to the exact product than rounding it per point.
                if constexpr (LBITS) ga[i] = __builtin_bit_cast(unsigned, dpk) & raw[i];      // (a_load left the relu masks of the pair here)
                else {
                    const unsigned sgn = raw[i] + 0x7fff7fffu;          // bf16 h >= 0: sets the half's sign bit iff h != 0, no carry
                    const i16x2 on = __builtin_bit_cast(i16x2, sgn) >> (i16x2){15, 15};
                    ga[i] = __builtin_bit_cast(unsigned, dpk) & __builtin_bit_cast(unsigned, on);
                }
            }
            st.af[s2][mi] = __builtin_bit_cast(frag, ga);
            if (live && my_row) orow[mi / WCC] += o;
            if (live && out_bias_wave && mi == 0) bout += (da[0] + da[1]) + (da[2] + da[3]) + (db[0] + db[1]) + (db[2] + db[3]);
        }
        if (live && bias_rows) bsum[mi] = Pol::sum8(st.af[s2][mi], bsum[mi]);
    };
    constexpr int NPREP = 2 * MPW;
    constexpr int T_PREP = NTOT > 3 ? 3 : NTOT - 1;
    // MFMA phase of the group in `gp` (A fragments `cur`, first B fragments `bc`), with the loads / preparation of the
    // next group's A state `nx` and first B fragments `bn` from `gnext` folded into its steps
    auto mma_phase = [&](const char *gp, const AState &cur, const frag (&bc)[2], const char *gnext, AState &nx, frag (&bn)[2],
                         bool live_next) {
        frag bq[3], benc;
        if constexpr (F8) {
            // e4m3 operands: pair u of the wave's B tiles arrives sorted in ONE register quad (bc[0] / bn[0]: fragments of tile 2k
            // in dwords 0-1, of tile 2k+1 in dwords 2-3)
            constexpr int NPAIR = NTOT / 2;
            u32x4 bcur = __builtin_bit_cast(u32x4, bc[0]), bnx = bcur;
            Raw8 rnext, rawn;
            if (NPAIR > 1) rnext = load_braw(gp, 1);
#pragma unroll
            for (int t = 0; t < NTOT; ++t) {
                const int u = t >> 1, e = t & 1;
                if (e == 0 && u + 1 < NPAIR) {
#if (BHN_T8_ABL & 4)
                    const Pair8 pr = {rnext.lo, rnext.hi};
#else
                    const Pair8 pr = tr8_sort(rnext);
#endif
                    bnx = (u32x4){pr.a[0], pr.a[1], pr.b[0], pr.b[1]};
                    if (u + 2 < NPAIR) rnext = load_braw(gp, u + 2);
                }
                if (enc_extra && (t % NPW) == (NPW >= 2 ? NPW - 2 : 0)) benc = tr_frag(gp + OFF_E, t / NPW, trl);
                if (t == 0) a_load(gnext, nx);
                __builtin_amdgcn_sched_barrier(0);
                const int s2 = t / NPW, ni = t % NPW;
                const u32x2 bnow = e == 0 ? (u32x2){bcur[0], bcur[1]} : (u32x2){bcur[2], bcur[3]};
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) {
#if (BHN_T8_ABL & 1)
                    asm volatile("" :: "v"(cur.a8[s2][mi]), "v"(bnow));
#else
                    acc[mi][ni] = mma8(cur.a8[s2][mi], bnow, acc[mi][ni]);
#endif
                }
                if constexpr (enc_extra) {
                    if (ni == NPW - 1) {
#pragma unroll
                        for (int mi = 0; mi < MPW; ++mi)
                            if ((mi % WCC) == wc) acc_e[mi / WCC] = Pol::mma(cur.af[s2][mi], benc, acc_e[mi / WCC]);
                    }
                }
#pragma unroll
                for (int k = 0; k < NPREP; ++k)
                    if (t == (T_PREP + k < NTOT ? T_PREP + k : NTOT - 1)) a_prep(nx, k, live_next);
                if (t == (NTOT >= 2 ? NTOT - 2 : 0)) rawn = load_braw(gnext, 0);
                if (t == NTOT - 1) {
                    const Pair8 pr = tr8_sort(rawn);
                    bn[0] = __builtin_bit_cast(frag, (u32x4){pr.a[0], pr.a[1], pr.b[0], pr.b[1]});
                }
                __builtin_amdgcn_sched_barrier(0);
                if (e == 1) bcur = bnx;
            }
            return;
        }
        if constexpr (B8) {
            // 8-bit B tiles come in pairs (tiles 2k, 2k+1 of the wave's share, one k-step = MFMA steps 2u, 2u+1): the raw
            // read of pair u+1 is in flight while pair u is widened and used
            constexpr int NPAIR = NTOT / 2;
            frag bcur[2] = {bc[0], bc[1]}, bnx[2];
            Raw8 rnext, rawn;
            if (NPAIR > 1) rnext = load_braw(gp, 1);
#pragma unroll
            for (int t = 0; t < NTOT; ++t) {
                const int u = t >> 1, e = t & 1;
                if (e == 0 && u + 1 < NPAIR) {
                    tr8_widen(rnext, 1.f, bnx[0], bnx[1]);
                    if (u + 2 < NPAIR) rnext = load_braw(gp, u + 2);
                }
                if (enc_extra && (t % NPW) == (NPW >= 2 ? NPW - 2 : 0)) benc = tr_frag(gp + OFF_E, t / NPW, trl);
                if (t == 0) a_load(gnext, nx);
                __builtin_amdgcn_sched_barrier(0);
                const int s2 = t / NPW, ni = t % NPW;
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) acc[mi][ni] = Pol::mma(cur.af[s2][mi], bcur[e], acc[mi][ni]);
                if constexpr (enc_extra) {
                    if (ni == NPW - 1) {
#pragma unroll
                        for (int mi = 0; mi < MPW; ++mi)
                            if ((mi % WCC) == wc) acc_e[mi / WCC] = Pol::mma(cur.af[s2][mi], benc, acc_e[mi / WCC]);
                    }
                }
#pragma unroll
                for (int k = 0; k < NPREP; ++k)
                    if (t == (T_PREP + k < NTOT ? T_PREP + k : NTOT - 1)) a_prep(nx, k, live_next);
                if (t == (NTOT >= 2 ? NTOT - 2 : 0)) rawn = load_braw(gnext, 0);
                if (t == NTOT - 1) tr8_widen(rawn, 1.f, bn[0], bn[1]);
                __builtin_amdgcn_sched_barrier(0);
                if (e == 1) { bcur[0] = bnx[0]; bcur[1] = bnx[1]; }
            }
            return;
        }
        if constexpr (make_h) {
            bq[0] = load_b(gp, 0);
            if (NTOT > 1) bq[1] = load_b(gp, 1);
        } else {
            bq[0] = bc[0]; bq[1] = bc[1];
        }
#ifndef BHN_DW_ABL_READS
#define BHN_DW_ABL_READS 0       // measurement builds (dW wrong): 1 = every second B fragment is not read (the previous one is used again):
#endif                           // 8 instead of 12 transposed fragment reads per 16 MFMAs -- the LDS traffic of 4 x 4 register blocking
#pragma unroll
        for (int t = 0; t < NTOT; ++t) {
            if (t + 2 < NTOT) {
                if (BHN_DW_ABL_READS && ((t + 2) & 1)) bq[(t + 2) % 3] = bq[(t + 1) % 3];
                else bq[(t + 2) % 3] = load_b(gp, t + 2);
            }
            if (enc_extra && (t % NPW) == (NPW >= 2 ? NPW - 2 : 0)) benc = tr_frag(gp + OFF_E, t / NPW, trl);     // used at ni == NPW-1
            if (t == 0) a_load(gnext, nx);
            __builtin_amdgcn_sched_barrier(0);
            const int s2 = t / NPW, ni = t % NPW;
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) acc[mi][ni] = Pol::mma(cur.af[s2][mi], bq[t % 3], acc[mi][ni]);
            if constexpr (enc_extra) {
                if (ni == NPW - 1) {
#pragma unroll
                    for (int mi = 0; mi < MPW; ++mi)
                        if ((mi % WCC) == wc) acc_e[mi / WCC] = Pol::mma(cur.af[s2][mi], benc, acc_e[mi / WCC]);
                }
            }
#pragma unroll
            for (int k = 0; k < NPREP; ++k)
                if (t == (T_PREP + k < NTOT ? T_PREP + k : NTOT - 1)) a_prep(nx, k, live_next);
            if constexpr (!make_h) {
                if (t == (NTOT >= 2 ? NTOT - 2 : 0)) bn[0] = load_b(gnext, 0);
                if (t == NTOT - 1) bn[1] = BHN_DW_ABL_READS ? bn[0] : load_b(gnext, NTOT > 1 ? 1 : 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the stream: pieces (1 KiB = one wave-wide DMA) this job needs: [A tiles][h tiles][enc tile][f32 dout piece]
    constexpr int PA = LBITS ? 1 : MT * TT / 1024, PH = (has_h && !make_h) ? MT * TT / 1024 : 0,
                  PE = (JT == JT_FIRST || JT == JT_SKIP || make_h || OUTENC) ? TB / 1024 : 0, PD = LAST ? 1 : 0;
    using Stream = TapeStream<Pol::NWAVES, PA, PH, PE, PD, OFF_H, OFF_E, OFF_D32>;
    constexpr int PPW = Stream::PPW;
    const Stream stream(srcA, LBITS ? (long long)A.f.depth * MWB : (long long)MT * TT, srcH, (long long)MT * TT, srcE, TB, srcD, A.t.dout_stride, wv);
    auto issue = [&](long long q, char *buf) {
        q = q < q1 ? q : q1 - 1;
        if (BHN_DBG(A.wrap)) q %= A.wrap;
        if (BHN_DBG(A.policy == 1)) stream.template issue<0>(q, buf);
        else if (BHN_DBG(A.policy == 2)) stream.template issue<2>(q, buf);
        else stream.template issue<1>(q, buf);
    };
    __syncthreads();                                            // W_0 / b_0 visible (HIDDEN1)
    if (q0 < q1) {
        AState sa, sb;
        frag ba[2], bb[2];
#pragma unroll
        for (int j = 0; j < NB - 1; ++j) issue(q0 + j, smem + j * GB);
        if (!BHN_DBG(A.debug & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (make_h) make_h_write(smem, make_h_mma(make_h_read(smem)));     // published by the first loop barrier
        if (works) {
            a_load(smem, sa);
#pragma unroll
            for (int k = 0; k < NPREP; ++k) a_prep(sa, k, true);
            if constexpr (F8) {
                const Pair8 pr = tr8_sort(load_braw(smem, 0));
                ba[0] = __builtin_bit_cast(frag, (u32x4){pr.a[0], pr.a[1], pr.b[0], pr.b[1]});
            } else if constexpr (B8) tr8_widen(load_braw(smem, 0), 1.f, ba[0], ba[1]);
            else if constexpr (!make_h) { ba[0] = load_b(smem, 0); ba[1] = load_b(smem, NTOT > 1 ? 1 : 0); }
        }
        auto body = [&](AState &cur, AState &nx, frag (&bc)[2], frag (&bn)[2], long long q, int it) {
            // group q+1 has landed for this wave (q+2 .. q+NB-2 may still be in flight), then it is published to the workgroup
            if (!BHN_DBG(A.debug & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 3) * PPW) : "memory");
            if constexpr (make_h) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's h-tile writes
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");      // the raw barrier is not a compiler fence
            // every wave has finished group q-1: its buffer takes group q+NB-1
            if (!BHN_DBG(A.debug & 2)) issue(q + NB - 1, smem + ((it + NB - 1) & (NB - 1)) * GB);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char *gp = smem + it * GB;
            char *gnext = smem + ((it + 1) & (NB - 1)) * GB;
            const bool live_next = q + 1 < q1;
            f32x16 hacc = {};
            if constexpr (make_h) hacc = make_h_mma(make_h_read(gnext));
            if (!BHN_DBG(A.debug & 1) && works && !(T8 && (BHN_T8_ABL & 8))) mma_phase(gp, cur, bc, gnext, nx, bn, live_next);
            if constexpr (make_h) {
                if (live_next) make_h_write(gnext, hacc);
            }
        };
        int it = 0;
        for (long long q = q0; q < q1;) {
            body(sa, sb, ba, bb, q, it);
            ++q; it = (it + 1) & (NB - 1);
            if (q >= q1) break;
            body(sb, sa, bb, ba, q, it);
            ++q; it = (it + 1) & (NB - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- flush: slab[(m*NTMAX+n)][r/4][lane][r%4] ---------------------------------------------
    // LAST: row i of A tile m carries the factor W_out[32 m + i] (a_prep); accumulator register r of a lane holds row
    // (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const float *wout_g = reinterpret_cast<const float *>(A.f.packed + A.f.wout_off);
    auto flush_tile = [&](int m, int n, const f32x16 &t, bool enc_tile = false) {
        float *tp = slab + (long long)(m * BG::NTMAX + n) * 1024;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t[4 * g4 + e];
            if constexpr (LAST && T8) {      // rows are virtual (t8_feature); F8: the tile was accumulated on dout / scaleA
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (F8 && n < nBr && !enc_tile ? scaleA : 1.f) * wout_g[t8_feature(m, 8 * g4 + 4 * (lane >> 5) + e)];
            } else if constexpr (F8) {       // accumulated on gA / scaleA (the encoded-input tile: on the widened copy, true scale)
                if (!enc_tile) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= scaleA;
                }
            } else if constexpr (LAST) {
                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wout_g + 32 * m + 8 * g4 + 4 * (lane >> 5));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= w4[e];
            }
            f32x4 *dst = reinterpret_cast<f32x4 *>(tp + g4 * 256 + lane * 4);
            if (A.accumulate) {
                const f32x4 old = *dst;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += old[e];
            }
            *dst = v;
        }
    };
    // one float of row i (= lane & 31, halves added) of slab tile (m, n), column 0
    auto flush_column0 = [&](int m, int n, float v, bool on) {
        v += __shfl_xor(v, 32, 64);
        if (on && lane < 32) {
            const int hh = (lane >> 2) & 1, r = (lane & 3) + 4 * (lane >> 3);
            float *dst = slab + (long long)(m * BG::NTMAX + n) * 1024 + (r >> 2) * 256 + (32 * hh) * 4 + (r & 3);
            if (A.accumulate) v += *dst;
            *dst = v;
        }
    };
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int m = wr * MPW + mi;
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) {
            const int n = nbase + ni;
            if (m < MT && n < nBr) flush_tile(m, n, acc[mi][ni]);
        }
        if (enc_extra && (mi % WCC) == wc && m < MT) flush_tile(m, nH, acc_e[mi / WCC], true);
        flush_column0(m, nB, LAST ? bsum[mi] * wout_r[mi] : bsum[mi], bias_rows && m < MT);
        if constexpr (LAST && !LBITS) {
            // the output layer's row: slab row MT, tile m, row 0, column f = lane & 31 (where reduce_kernel reads dW_out[32 m + f])
            float v = orow[mi / WCC] + __shfl_xor(orow[mi / WCC], 32, 64);
            if ((mi % WCC) == wc && m < MT && lane < 32) {
                float *dst = slab + (long long)(MT * BG::NTMAX + m) * 1024 + lane * 4;
                if (A.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
    if constexpr (LBITS) {
        // The output layer's row from this workgroup's own sums (see the template comment): every lane weights its accumulator
        // elements (row f = output unit of layer depth-1, column k = its input unit) with the bf16 weight K[k][f] the forward
        // used, the 32 columns of a row meet through LDS, the bias term b[f] g[f] joins on wave column 0, and wave column wc leaves
        // its share in float wc of the row's slot (slab row MT, tile m, float4 of lane f): reduce_kernel adds the four floats.
        static_assert(WCC <= 8, "one float of the slot (two float4: lanes f and f + 32) per wave column");
        using PKf = Pack<W, Pol>;
        __syncthreads();                                            // every wave is out of the group ring
        float *stg = reinterpret_cast<float *>(smem) + wv * (MPW * 32 * 33);
        const int col = lane & 31, hh = lane >> 5;
        if (works) {
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                const int m = wr * MPW + mi;
                const char *wimg = A.f.packed + A.f.fwd_off + (size_t)(1 + (job - 1) * MT + (m < MT ? m : 0)) * PKf::CHUNK_BYTES;
                float sr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[r] = 0.f;
#pragma unroll
                for (int ni = 0; ni < NPW; ++ni) {
                    const int n = nbase + ni, k = 32 * n + col;
                    if (n < nBr) {
                        const int fr = k >> 4, ph = k & 15, h2 = (ph >> 2) & 1, jj = (ph & 3) + 4 * (ph >> 3);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int fl = (r & 3) + 8 * (r >> 2) + 4 * hh;
                            const float wq = (float)reinterpret_cast<const __bf16 *>(wimg + fr * Pol::FRAG_BYTES + (fl + 32 * h2) * 16)[jj];
                            sr[r] += wq * acc[mi][ni][r];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + col] = sr[r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the wave reads back what its own lanes wrote)
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            const int m = wr * MPW + mi;
            float tot = 0.f;
            if (works && lane < 32) {
                for (int c = 0; c < 32; ++c) tot += stg[(mi * 32 + lane) * 33 + c];
            }
            const float g = bsum[mi] + __shfl_xor(bsum[mi], 32, 64);           // un-folded bias gradient of feature lane & 31 (wave column 0)
            if (bias_rows && lane < 32 && m < MT) tot += reinterpret_cast<const float *>(A.f.packed + A.f.bias_off)[job * W + 32 * m + lane] * g;
            if (works && lane < 32 && m < MT) {
                float *dst = slab + (long long)(MT * BG::NTMAX + m) * 1024 + (lane + 32 * (wc >> 2)) * 4;
                if (A.accumulate) tot += dst[wc & 3];
                dst[wc & 3] = tot;
                if (wc == 0) {                  // the slots no wave column owns, or whose column has no B tiles (narrow networks)
#pragma unroll
                    for (int e = 1; e < (WCC > 4 ? 8 : 4); ++e)
                        if (e >= WCC || e * NPW >= nBr) slab[(long long)(MT * BG::NTMAX + m) * 1024 + (lane + 32 * (e >> 2)) * 4 + (e & 3)] = 0.f;
                }
            }
        }
    }
    if constexpr (LAST) {          // output layer: [tiles 0..MT-1: h part][tile MT: encoded-input part (OUTENC)][bias]
        float v = bout + __shfl_xor(bout, 32, 64);
        if (out_bias_wave && lane == 0) {
            float *dst = slab + (long long)(MT * BG::NTMAX + MT + (OUTENC ? 1 : 0)) * 1024;
            if (A.accumulate) v += *dst;
            *dst = v;
        }
        if constexpr (OUTENC) {
            float ve = oenc + __shfl_xor(oenc, 32, 64);
            if (out_bias_wave && lane < 32) {
                float *dst = slab + (long long)(MT * BG::NTMAX + MT) * 1024 + lane * 4;      // row 0, column = slot
                if (A.accumulate) ve += *dst;
                *dst = ve;
            }
        }
    }
}

template <int W, class Pol>
__global__ __launch_bounds__(Pol::NTHREADS) __attribute__((amdgpu_waves_per_eu(Pol::NWAVES == 4 ? 1 : (W <= 128 ? 4 : 2)))) void dw_kernel(BwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // NBUF x GROUP_BYTES
    const int depth = A.f.depth;
    clock_stamp(A.f.clk, BHN_CLK_DW, 0);
    int job = 0;
    while (job < depth && (int)blockIdx.x >= A.wg_begin[job + 1]) ++job;
    if (BHN_DBG(A.debug >> 2) && (A.debug >> 2) - 1 != job) return;
#ifdef BHN_T8_ONLY_JOB          // measurement builds: only this job of the 8-bit tape's dW kernel runs (the release build has no run-time switches)
    if (Pol::TAPE8 && job != BHN_T8_ONLY_JOB) return;
#endif
    if constexpr (Pol::ELEM_BYTES == 2) {           // bf16: software-pipelined bodies; the output layer rides on job depth-1
        const bool out_skip = (A.f.skip_mask >> depth) & 1;             // odd depths with do_skip
        if (job == depth) {                                               // (depth < 3 only)
            if (out_skip) dw_body<W, Pol, JT_OUTSKIP>(A, job, smem);
            else dw_body<W, Pol, JT_OUT>(A, job, smem);
        } else if (job == 0) dw_body2<W, Pol, JT_FIRST>(A, job, smem); else if (job == depth - 1 && A.t.drop_ga) {
            if constexpr (Pol::TAPE8) {           // (no skip into the output layer: checked by the host)
                if ((A.f.skip_mask >> job) & 1) dw_body2<W, Pol, JT_SKIP, true>(A, job, smem);
                else dw_body2<W, Pol, JT_HIDDEN, true>(A, job, smem);
            } else if ((A.f.skip_mask >> job) & 1) {
                if (out_skip) dw_body2<W, Pol, JT_SKIP, true, true>(A, job, smem);
                else dw_body2<W, Pol, JT_SKIP, true>(A, job, smem);
            } else if (A.t.lbits) {
                if (out_skip) dw_body2<W, Pol, JT_HIDDEN, true, true, true>(A, job, smem);
                else dw_body2<W, Pol, JT_HIDDEN, true, false, true>(A, job, smem);
            } else {
                if (out_skip) dw_body2<W, Pol, JT_HIDDEN, true, true>(A, job, smem);
                else dw_body2<W, Pol, JT_HIDDEN, true>(A, job, smem);
            }
        } else if ((A.f.skip_mask >> job) & 1) dw_body2<W, Pol, JT_SKIP>(A, job, smem);
        else if (job == 1 && A.t.drop_h1) dw_body2<W, Pol, JT_HIDDEN1>(A, job, smem);
        else dw_body2<W, Pol, JT_HIDDEN>(A, job, smem);
    } else {
        if (job == depth) {
            if ((A.f.skip_mask >> depth) & 1) dw_body<W, Pol, JT_OUTSKIP>(A, job, smem);
            else dw_body<W, Pol, JT_OUT>(A, job, smem);
        } else if (job == 0) dw_body<W, Pol, JT_FIRST>(A, job, smem);
        else if ((A.f.skip_mask >> job) & 1) dw_body<W, Pol, JT_SKIP>(A, job, smem);
        else dw_body<W, Pol, JT_HIDDEN>(A, job, smem);
    }
    clock_stamp(A.f.clk, BHN_CLK_DW, 1);
}

// ---------------------------------------------------------------------------------------------
// slab reduction -> flat dparams (flax tree order)
// ---------------------------------------------------------------------------------------------
// One block per slab tile (layer, A-tile row, B tile | bias tile), one float4 per thread and slab: the sum over the job's
// workgroups runs over CONTIGUOUS 4 KiB of every slab (round 1 walked the flat parameters and gathered single floats out
// of the tiles: 330 MB of sectors fetched for 94 MB of slabs, 0.117 ms), in workgroup order (bitwise reproducible), and
// the result is scattered to the flat parameters (flax tree order) by inverting the tile layout
//   idx = (m NTMAX + n) 1024 + (r >> 2) 256 + (col + 32 hh) 4 + (r & 3),  row = (r & 3) + 4 hh + 8 (r >> 2).
template <int W, class Pol>
struct ReduceGeom {
    using BG = BwdGeom<W, Pol>;
    // tiles of layer l: rows x (B tiles + bias tile); the output layer is one row (slab row MT when it rides on job depth-1)
    static __host__ __device__ int n_b(int l, unsigned skip_mask) { return (l >= 1 ? BG::MT : 0) + ((l == 0 || ((skip_mask >> l) & 1)) ? 1 : 0); }
    static __host__ __device__ int rows(int l, int depth) { return l == depth ? 1 : BG::MT; }
    static __host__ __device__ int blocks(int depth, unsigned skip_mask) {
        int n = 0;
        for (int l = 0; l <= depth; ++l) n += rows(l, depth) * (n_b(l, skip_mask) + 1);
        return n;
    }
};

// Stage 1 of the reduction of the delta chain's dW_0 slabs (TapeLayout::ga0_chain): block (tile, part) adds the slabs
// [part * per, (part + 1) * per) of its tile in order and leaves the sum in the first slab of its range (no block reads what
// another one writes); reduce_kernel then adds the `parts` partial sums.  A serial sum over 256 slabs in 8 blocks is latency-bound.
__global__ __launch_bounds__(256) void chain_slab_stage1(BwdArgs A, int mt, int per) {
    const int tile = blockIdx.x, s0 = blockIdx.y * per, s1 = (s0 + per < A.n_chain_wg) ? s0 + per : A.n_chain_wg;
    if (s0 >= s1) return;
    float *base = A.slab0 + (long long)tile * 1024 + threadIdx.x * 4;
    f32x4 sum = *reinterpret_cast<const f32x4 *>(base + (long long)s0 * mt * 1024);
    for (int wg = s0 + 1; wg < s1; ++wg) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long long)wg * mt * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
    *reinterpret_cast<f32x4 *>(base + (long long)s0 * mt * 1024) = sum;
}

template <int W, class Pol>
__global__ __launch_bounds__(256) void reduce_kernel(BwdArgs A) {
    using BG = BwdGeom<W, Pol>;
    using RG = ReduceGeom<W, Pol>;
    constexpr int MT = BG::MT;
    const int depth = A.f.depth, WT = A.width_true;
    int b = blockIdx.x, l = 0;
    for (; l <= depth; ++l) {
        const int cnt = RG::rows(l, depth) * (RG::n_b(l, A.f.skip_mask) + 1);
        if (b < cnt) break;
        b -= cnt;
    }
    if (l > depth) return;
    const bool has_h = l >= 1, has_enc = (l == 0) || ((A.f.skip_mask >> l) & 1);
    const int nH = has_h ? MT : 0, nB = nH + (has_enc ? 1 : 0);
    const int mi = b / (nB + 1), n = b % (nB + 1);
    const bool out = l == depth, rides = out && A.t.drop_ga;
    const int jl = rides ? depth - 1 : l;
    const int mrow = out ? (rides ? MT : 0) : mi;
    if (!out && 32 * mi >= WT) return;                              // padding rows of a narrower model
    const int tid = threadIdx.x, g4 = tid >> 6, lane = tid & 63, hh = lane >> 5, col = lane & 31;
    const float *src = A.f.slabs + (long long)(mrow * BG::NTMAX + n) * 1024 + g4 * 256 + lane * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const bool from_chain = l == 0 && A.t.ga0_chain;                // dW_0 slabs of the delta chain's workgroups: [wg][tile mi][1024],
    if (from_chain) {                                                // encoded-input tile with the bias in column 31
        if (n != 0) return;
        // (the slabs that hold the stage-1 sums of chain_slab_stage1: every A.chain_step-th)
        const float *s0 = A.slab0 + (long long)mi * 1024 + g4 * 256 + lane * 4;
        for (int wg = 0; wg < A.n_chain_wg; wg += A.chain_step) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(s0 + (long long)wg * MT * 1024);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] += v[e];
        }
    } else
    for (int wg = A.wg_begin[jl]; wg < A.wg_begin[jl + 1]; ++wg) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (long long)wg * BG::SLAB_FLOATS);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
    // input feature of this column
    long long kin = -1;
    bool is_bias = false;
    // 8-bit tape: rows of every job and the columns that came from 8-bit h tiles (all but the recomputed h_1 of layer 1) are
    // virtual (t8_feature); the output layer's row rides on job depth-1 with ITS row index in the column position
    if (n < nH) {
        const int c = (Pol::TAPE8 && !(l == 1 && A.t.drop_h1)) ? t8_feature(n, col) : 32 * n + col;
        if (c < WT) kin = c;
    }
    else if (n < nB) {
        const int fe = bhn_enc_slot_feature(col, A.f.deg);
        if (fe >= 0) kin = (has_h ? WT : 0) + fe;
        else if (from_chain && col == 31) is_bias = true;           // slot 31 of the recorded inputs is 1
    }
    else is_bias = col == 0;
    float lb_tot = (sum[0] + sum[1]) + (sum[2] + sum[3]);
    if constexpr (BG::WCC > 4) lb_tot += __shfl_xor(lb_tot, 32, 64);      // (wave columns 4 .. 7: the float4 of lane f + 32)
    if (kin < 0 && !is_bias) return;
    const int outw = out ? 1 : WT;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e, row = (r & 3) + 4 * hh + 8 * (r >> 2);
        const int o = out ? 0 : (Pol::TAPE8 && !from_chain) ? t8_feature(mi, row) : 32 * mi + row;      // (the chain's dW_0 tiles: bf16 staging, natural rows)
        if (out ? row != 0 : o >= WT) continue;
        float val = sum[e];
        if (rides && A.t.lbits && n < nH) val = lb_tot;                 // the shares of the dW job's wave columns (dw_body2 LBITS)
        A.dparams[is_bias ? A.bias_off[l] + o : A.kernel_off[l] + kin * outw + o] = val;
    }
}

// 8-bit tape state (BwdArgs::t8; floats): [0..7] the power-of-two scale gA_l is stored with in THIS call, [8..15] the largest
// |gA_l| this call's delta chain saw (bit patterns, atomicMax), [16..23] ratio_l = |gA_l|max / |dimages|max of the previous
// call, [24] |dimages|max of this call ([25]: its partial maxima in flight).  The delta chain is linear in d(loss)/d(images), so the ratios depend on the weights
// and on WHERE the residuals are, not on how large they are: they move slowly from step to step, while the loss scale may
// jump by orders of magnitude (a new batch, a restart, a caller's loss weights) -- that part is measured, not predicted.
//   t8_dmax + t8_prepare  (start of every backward call)  |dimages|max; scale_l = the power of two that puts 16 ratio_l |dimages|max
//               at or below 448, e4m3's largest value: four binades of head room, thirteen and a half below the maximum
//               before values flush to zero (tools/exp_fp8_tape_accuracy.py: the dW error does not notice)
//   t8_update   (end of the call)  ratio_l from the maxima the chain just saw; maxima cleared
//   t8_open     (BHN_T8_CALIBRATE: in front of a chain pass whose tape output is discarded)  scales so large that
//               nothing is limited; only the maxima of that pass are used
// |dimages|max in two kernels: partial maxima of up to 64 blocks meet in st[25] (atomicMax on the bit patterns of non-negative
// floats; zero between calls: t8_prepare resets it), one thread then turns it into the scales.  (One block over all of
// dimages took 26 us at config 2 and would take 0.3 ms at config 3's 1.5 M pixels.)
__global__ __launch_bounds__(1024) void t8_dmax_kernel(float *st, const float *dimages, long long n) {
    __shared__ float red[16];
    float m = 0.f;
    for (long long i = blockIdx.x * 1024ll + threadIdx.x; i < n; i += (long long)gridDim.x * 1024) m = __builtin_fmaxf(m, __builtin_fabsf(dimages[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = 0.f;
        for (int i = 0; i < 16; ++i) d = __builtin_fmaxf(d, red[i]);
        if (!(d < __builtin_inff())) d = __builtin_inff();             // (inf / NaN in d(loss)/d(images): t8_prepare leaves the scales alone)
        atomicMax(reinterpret_cast<unsigned *>(st) + 25, __float_as_uint(d));
    }
}
__global__ void t8_prepare_kernel(float *st, int nl, int fresh) {
    if (threadIdx.x != 0) return;
    float d = st[24];
    if (fresh) {                                                        // this call's |dimages|max: from t8_dmax_kernel
        d = st[25];
        st[25] = 0.f;
        if (!(d < __builtin_inff())) d = 0.f;
        st[24] = d;
    }
    for (int l = 0; l < nl; ++l) {
        const float a = st[16 + l] * d;                                   // expected largest |gA_l|
        if (a > 0.f && a < __builtin_inff()) {
            int e;
            const float mant = frexpf(a * (16.f / 448.f), &e);            // a 16 / 448 = mant 2^e, mant in [0.5, 1)
            if (mant == 0.5f) --e;
            e = e < -100 ? -100 : e > 100 ? 100 : e;
            st[l] = ldexpf(1.f, e);
        } else if (!(st[l] > 0.f && st[l] < __builtin_inff())) st[l] = 1.f;
    }
}
__global__ void t8_update_kernel(float *st, int nl) {
    const int l = threadIdx.x;
    if (l >= nl) return;
    unsigned *bits = reinterpret_cast<unsigned *>(st);
    const float a = __uint_as_float(bits[8 + l]), d = st[24];
    if (a > 0.f && a < __builtin_inff() && d > 0.f) st[16 + l] = a / d;
    bits[8 + l] = 0u;
}
__global__ void t8_open_kernel(float *st) {
    if (threadIdx.x < 8) { st[threadIdx.x] = 0x1p60f; reinterpret_cast<unsigned *>(st)[8 + threadIdx.x] = 0u; }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
int fused_fill_args(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                    const bhn_frames *fr, bool need_w, FusedArgs *a, MlpShape *s, int nwaves);

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#ifndef BHN_DEBUG
static constexpr int g_bwd_stages = 7, g_bwd_debug = 0;
#else
// measurement build only: bit 0 chain kernel, bit 1 dW kernel, bit 2 reduce kernel (default all)
static thread_local int g_bwd_stages = 7;
static thread_local int g_bwd_debug = 0;
static int dbg_env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
extern "C" int bhn_debug_set_bwd_stages(int32_t mask) {
    g_bwd_stages = mask & 7;
    g_bwd_debug = (mask >> 3) & 0xFFF;    // bit 12: ring-step time stamps of one tile (tools/dbg_chain_steps.py); bit 3: dW kernel without MFMA work, bit 4: without tape loads,
                                         // bits 5-8: run only dW job (value-1); bit 9: emit without global stores; bit 10: no emit
    return BHN_OK;
}
#endif

// the delta chain accumulates dW_0 itself (TapeLayout::ga0_chain): bf16, one gA_0 tile per wave (width 256),
// depth >= 3 (the chain then starts from the folded W_out image: bhn_folds_wout)
template <int W, class Pol>
static constexpr bool ga0_chain_ok(int depth) {
    // (not the 8-bit tape mode: its delta chain has 16 registers less to spare and spills with the consumer's state)
    return BHN_GA0_CHAIN != 0 && Pol::ELEM_BYTES == 2 && !Pol::TAPE8 &&
           W / 32 == Pol::NWAVES && depth >= 3;
}

template <int W, class Pol>
static void tape_layout(int depth, bool layer1_takes_enc, bool last_takes_enc, long long NQ, TapeLayout *t) {
    using BG = BwdGeom<W, Pol>;
    memset(t, 0, sizeof(*t));
    t->NQ = NQ;
    t->drop_h1 = Pol::ELEM_BYTES == 2 && depth >= 2 && !layer1_takes_enc;
    long long off = 0;
    const long long per_tensor = NQ * BG::MT * (long long)BG::TAPE_TILE;
    // (lbits / drop_hd: decided before the h tensors are laid out)
    const bool lbits = BHN_LBITS != 0 && Pol::ELEM_BYTES == 2 && !Pol::TAPE8 && bhn_folds_wout(Pol::MODE, depth) && !last_takes_enc;
    t->drop_hd = BHN_DROP_HD != 0 && lbits;
    for (int l = 1; l <= depth; ++l) {
        if ((l == 1 && t->drop_h1) || (l == depth && t->drop_hd)) { t->h_off[l] = -1; continue; }
        t->h_off[l] = off; off += per_tensor;
    }
    if (t->drop_h1) { t->encp_off = off; off += NQ * (long long)BG::TILE_BYTES; }
    t->drop_ga = bhn_folds_wout(Pol::MODE, depth);
    t->ga0_chain = ga0_chain_ok<W, Pol>(depth) && t->drop_ga;
    t->lbits = BHN_LBITS != 0 && Pol::ELEM_BYTES == 2 && !Pol::TAPE8 && t->drop_ga && !last_takes_enc;
    for (int l = 0; l < depth; ++l) {
        if ((l == depth - 1 && t->drop_ga) || (l == 0 && t->ga0_chain)) { t->ga_off[l] = -1; continue; }
        t->ga_off[l] = off; off += per_tensor;
    }
    t->lin_stride = per_tensor;
    {
        const int lmin = t->drop_h1 ? 2 : 1;
        t->h_lin = (lmin <= depth ? t->h_off[lmin] : 0) - lmin * per_tensor;
        t->ga_lin = t->ga0_chain ? t->ga_off[1] - per_tensor : t->ga_off[0];
    }
    t->enc_off = off; off += NQ * (long long)BG::TILE_BYTES;
    t->dout_stride = t->drop_ga ? 128 : BG::TILE_BYTES;         // 32 f32 dout per group, or dout as an A tile (row 0 = dout)
    t->dout_off = off; off += NQ * t->dout_stride;
    t->mask_off = off; off += NQ * (long long)depth * ((BG::MT + 1) / 2) * 256;
    t->e_off = off; off += NQ * 128;
    off = (long long)align_up((size_t)off + 1024, 256);          // +1 KiB: the last dout piece is DMA'd as a full KiB
    if (t->drop_hd) { t->scratch_off = off; off += 4096 * 64; }  // where the ring kernels' place-holder stores go (chain_kernel, no_hd): 4096 lines
    t->total = off;
}

template <int W, class Pol>
static long long bytes_per_group(int depth) {
    using BG = BwdGeom<W, Pol>;
    return (long long)(2 * depth * BG::MT + 1) * BG::TILE_BYTES + 128;
}

// bhn_tape_info: the bytes of tape each kernel of the training step moves per 32-point group, from the SAME flags the layout
// and the job table are built from (bench.py's `tape_stream` figures; round 5 re-derived them in Python and got depths != 4 wrong)
template <int W, class Pol>
static void tape_traffic(const MlpShape &s, const TapeLayout &t, int64_t *o) {
    using BG = BwdGeom<W, Pol>;
    const long long TT = BG::TAPE_TILE, TB = BG::TILE_BYTES, MT = BG::MT, MW = (MT + 1) / 2, depth = s.depth;
    long long fw = 0, cw = 0, cr = 0, dr = 0;
    if (t.fused128) {
        fw = (depth - 2) * MT * TB + MW * 256 + TB + 128;      // h_2 .. h_{depth-1}, relu bits of the last hidden layer, encoded inputs, e
        dr = fw + 128;                                         // the fused backward reads all of it + dout (dout128_kernel: 128 read + 128 written)
        cw = 128; cr = 128;
    } else {
        fw = TB + (t.drop_h1 ? TB : 0) + depth * MW * 256 + 128;
        for (int l = 1; l <= depth; ++l) if (t.h_off[l] >= 0) fw += MT * TT;
        for (int l = 0; l < depth; ++l) if (t.ga_off[l] >= 0) cw += MT * TT;
        cw += t.dout_stride;
        cr = depth * MW * 256 + 128 + (t.ga0_chain ? TB : 0);
        const int last_job = t.drop_ga ? (int)depth - 1 : (int)depth;
        for (int l = 0; l <= last_job; ++l) {
            if (l == 0) { if (!t.ga0_chain) dr += MT * TT + TB; continue; }
            if (l == depth) { dr += t.dout_stride + MT * TT + (s.skip_in[l] ? TB : 0); continue; }
            const bool last = l == depth - 1 && t.drop_ga;
            dr += last ? (t.lbits ? MW * 256 : MT * TT) + 1024 : MT * TT;            // A: gA_l, or h_depth (its relu bits) + the KiB that starts with dout
            dr += (l == 1 && t.drop_h1) ? TB : MT * TT;                              // B: h_l (layer 1: the encoded inputs it is recomputed from)
            if (s.skip_in[l] || (last && s.skip_in[depth])) dr += TB;
        }
    }
    o[0] = fw; o[1] = cw; o[2] = cr; o[3] = dr;
    o[4] = (t.drop_h1 ? 1 : 0) | (t.drop_ga ? 2 : 0) | (t.ga0_chain ? 4 : 0) | (t.fused128 ? 8 : 0) | (t.drop_hd ? 16 : 0) | (t.lbits ? 32 : 0);
}

enum { RUN_QUERY = 0, RUN_RECOMPUTE = 1, RUN_FWD_TRAIN = 2, RUN_BWD_TAPE = 3, RUN_INFO = 4 };

template <int W, class Pol>
static int bwd_run(int what, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                   const bhn_frames *fr, const float *dimages, float *images, float *dparams, void *workspace,
                   size_t workspace_bytes, hipStream_t st, size_t *query_bytes, int query_B, long long query_P,
                   int device, void *const *events = nullptr, int n_events = 0) {
    using BG = BwdGeom<W, Pol>;
    using PK = Pack<W, Pol>;
    BHN_CHECK_DEVICE(device);
    const bool t8_cal = Pol::TAPE8 && (mode & BHN_T8_CALIBRATE);
    mode = bhn_norm_mode(mode);
    constexpr size_t t8_bytes = Pol::TAPE8 ? 256 : 0;                   // the 8-bit tape's state block, in front of the tape
    const int ncu = bhn_num_cus(device);
#ifdef BHN_DEBUG
    static const int grid_override = dbg_env_int("BHN_DEBUG_DW_GRID", 0);
    static const int job1_w = dbg_env_int("BHN_DEBUG_JOB1_W", BHN_JOB1_W), jobl_w = dbg_env_int("BHN_DEBUG_JOBL_W", BHN_JOBL_W);
    static const int joblb_w = dbg_env_int("BHN_DEBUG_JOBLB_W", BHN_JOBLB_W);
#else
    constexpr int grid_override = 0, job1_w = BHN_JOB1_W, jobl_w = BHN_JOBL_W, joblb_w = BHN_JOBLB_W;
#endif
    const int grid_dw = grid_override > 0 ? grid_override : ncu;        // one dW workgroup per CU
    MlpShape s;
    {
        const int rcq = bhn_mlp_shape(m, &s);
        if (rcq != BHN_OK) return rcq;
    }
    // width 128, bf16, depth 4 (the reference's default network): delta chain and weight gradients fused in one kernel, the
    // gradient accumulated on chip, a tape of h_l / enc / e only (fused_bwd128.hip)
    const bool f128 = bwd128_supported(Pol::MODE, W, s.depth);
    // (ga0_chain: + one dW_0 slab of MT tiles per delta-chain workgroup, behind the dW kernel's slabs)
    const bool ga0c = !f128 && ga0_chain_ok<W, Pol>(s.depth) && bhn_folds_wout(Pol::MODE, s.depth);
    const size_t slab_dw_bytes = align_up(f128 ? bwd128_slab_bytes(ncu) : (size_t)grid_dw * BG::SLAB_FLOATS * 4, 256);
    const size_t slab_bytes = slab_dw_bytes + (ga0c ? align_up((size_t)ncu * BG::MT * 4096, 256) : 0);
    auto layout = [&](long long NQ, TapeLayout *t) {
        if (f128) bwd128_tape_layout(s.depth, NQ, t);
        else tape_layout<W, Pol>(s.depth, s.depth >= 2 && s.skip_in[1], s.depth >= 2 && s.skip_in[s.depth - 1], NQ, t);
    };
    // the fused 4x128 path's training forward runs on 12-wave workgroups (PolBF16X, fused_common.h): 12 groups per tile
    constexpr bool CAN_X = W == 128 && Pol::ELEM_BYTES == 2 && !Pol::TAPE8;
    using FPol = std::conditional_t<CAN_X, PolBF16X, Pol>;
    const bool x12 = CAN_X && f128 && bhn_fwd_w12(Pol::MODE, W, s.depth, what == RUN_QUERY ? (query_P + 31) / 32 : bhn_groups_per_frame(geom));
    const int nwf = x12 ? FPol::NWAVES : Pol::NWAVES;
    if (what == RUN_QUERY) {
        // (the caller's P may be the dense point count of a ray set that is walked compacted, or the other way round: room for either tile size)
        size_t need = 0;
        for (int nw : {(int)Pol::NWAVES, CAN_X && f128 ? (int)FPol::NWAVES : (int)Pol::NWAVES}) {
            const long long tiles = (query_P + nw * 32 - 1) / (nw * 32) * query_B;
            TapeLayout t;
            layout(tiles * nw, &t);
            if ((size_t)t.total > need) need = (size_t)t.total;
        }
        *query_bytes = slab_bytes + t8_bytes + need;
        return BHN_OK;
    }
    if (what == RUN_INFO) {        // query_bytes: int64_t out[8] (bhn_tape_info); query_P: 32-point groups per frame
        TapeLayout t;
        layout(query_P > 0 ? query_P : 1, &t);
        int64_t *o = reinterpret_cast<int64_t *>(query_bytes);
        tape_traffic<W, Pol>(s, t, o);
        o[5] = bhn_fwd_w12(Pol::MODE, W, s.depth, query_P) && CAN_X && f128 ? (int)FPol::NWAVES : (int)Pol::NWAVES;
        return BHN_OK;
    }
    BwdArgs A;
    memset(&A, 0, sizeof(A));
    int rc = fused_fill_args(m, mode, packed, geom, fr, true, &A.f, &s, nwf);
    A.fwd_nw = nwf;
    if (rc != BHN_OK) return rc;
    BHN_CHECK_ARG(workspace, "null workspace");
    BHN_CHECK_ARG(what == RUN_FWD_TRAIN ? images != nullptr : (dimages && dparams), "null pointer");
    const int depth = s.depth;
    // frames per pass so that the tape fits the workspace (same layout function as the size query)
    const long long groups_per_frame = (long long)A.f.tiles_per_frame * nwf;
    TapeLayout t1;
    layout(groups_per_frame, &t1);
    if (workspace_bytes < slab_bytes + t8_bytes + (size_t)t1.total) {
        bhn_set_error("render_bwd workspace too small: %zu bytes, need >= %zu (slabs %zu + one frame of tape %lld)",
                      workspace_bytes, slab_bytes + t8_bytes + (size_t)t1.total, slab_bytes, t1.total);
        return BHN_EWORKSPACE;
    }
    long long fpp = 1;
    while (fpp < A.f.B) {
        TapeLayout tn;
        layout(groups_per_frame * (fpp + 1), &tn);
        if (slab_bytes + t8_bytes + (size_t)tn.total > workspace_bytes) break;
        ++fpp;
    }
    if (what != RUN_RECOMPUTE && fpp < A.f.B) {
        bhn_set_error("the recorded-tape path needs a workspace for all %d frames at once (%zu bytes given); "
                      "use bhn_render_fwd + bhn_render_bwd, which iterate over frame groups", A.f.B, workspace_bytes);
        return BHN_EWORKSPACE;
    }
    A.f.slabs = reinterpret_cast<float *>(workspace);
    A.slab0 = ga0c ? reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + slab_dw_bytes) : nullptr;
    A.tape = reinterpret_cast<char *>(workspace) + slab_bytes + t8_bytes;
    A.t8 = Pol::TAPE8 ? reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + slab_bytes) : nullptr;
    if constexpr (Pol::TAPE8) {
        // what the 8-bit kernels are built for (everything the reference's own drivers use at this width)
        BHN_CHECK_ARG(t1.drop_h1 && t1.drop_ga && !s.skip_in[depth] && s.width_true == W,
                      "BHN_BF16_T8: depth >= 3, no skip into layer 1 or into the output layer, net_width == %d", W);
    }
    A.f.slab_floats = BG::SLAB_FLOATS;
    A.f.dimages = dimages;
    A.f.images = images;
    A.dparams = dparams;
    A.nparams = s.nparams;
    A.F = s.F;
    A.width_true = s.width_true;
    for (int l = 0; l <= depth; ++l) { A.kernel_off[l] = s.kernel_off[l]; A.bias_off[l] = s.bias_off[l]; A.in_dim[l] = s.in_dim[l]; }
    A.kernel_off[depth + 1] = s.nparams;
    // dW jobs: every layer gets workgroups in proportion to the tiles it streams per 32-point group (A + B), with
    // measured corrections for the two jobs that compute more than they stream (layer 1: recompute of h_1; layer depth-1:
    // rebuild of gA, output row).  Balancing the jobs so that each takes the same time when it runs ALONE
    // (tools/dbg_dw.py) measured slower (5.0 vs 4.75 ms): run together they share the HBM stream, and the light
    // layer-0 job finishing early leaves its bandwidth to the others.
    {
        double work[BHN_MAX_LAYERS + 1], tot = 0;
        const int last_job = t1.drop_ga ? depth - 1 : depth;       // drop_ga: the output row rides on layer depth-1's job
        for (int l = 0; l <= depth; ++l) {
            const int mtA = (l == depth) ? 0 : BG::MT;
            int nB = (l >= 1 ? BG::MT : 0) + ((l == 0 || s.skip_in[l]) ? 1 : 0);
            if (l == 1 && t1.drop_h1) nB = job1_w * BG::MT / 8;   // reads only the encoded inputs instead of h_1 but has the
                                                                  // same MFMA work + the recompute: not byte-bound any more
            work[l] = (double)(mtA + nB) + 0.5;
            // + the rebuild of gA and the output row (8-bit tape: the byte masks of that job are its long pole; 12 measured 2-3 % faster than 8)
            if (l == depth - 1 && t1.drop_ga) work[l] += (Pol::TAPE8 ? 12 : jobl_w) * BG::MT / 8.0;
            // LBITS: that job streams 1 KiB of relu bits in place of its MT A tiles
            if (l == depth - 1 && t1.lbits) work[l] = (double)nB + 1.0 + joblb_w * BG::MT / 8.0;
            if constexpr (Pol::ELEM_BYTES == 4) {
                // f32: the jobs are MFMA-bound (a 32x32x2 MFMA is 64 cycles; one 32x32 tile product over a 32-point
                // group = 16 of them = 0.55 us at the observed 1.87 GHz) unless they stream more than ~34 GB/s per
                // workgroup (one 68 KB group in flight per ~2 us round trip = 0.075 tile products per KiB: measured 26 GB/s for the layer-0 job alone)
                const int a_tiles = (l == depth) ? 1 : BG::MT;           // the output job's A operand is the dout tile
                const int wrr = (l == depth) ? 1 : BG::WRR, wcc = Pol::NWAVES / wrr;
                const double tp = (double)((a_tiles + wrr - 1) / wrr) * ((nB + wcc - 1) / wcc);
                const double kib = (double)(a_tiles + nB) * BG::TILE_BYTES / 1024.0;
                work[l] = (tp > 0.075 * kib ? tp : 0.075 * kib) + 0.1;
            }
            if (l > last_job) work[l] = 0;
            if (t1.ga0_chain && l == 0) work[l] = 0;               // dW_0 comes out of the delta chain: no layer-0 job
            tot += work[l];
        }
        int used = 0;
        A.wg_begin[0] = 0;
        for (int l = 0; l <= depth; ++l) {
            int n = (int)(grid_dw * work[l] / tot);
            if (n < 1) n = 1;
            if (l == last_job) n = grid_dw - used;
            if (n < 1) n = 1;
            if (l > last_job || (t1.ga0_chain && l == 0)) n = 0;
            used += n;
            A.wg_begin[l + 1] = used;
        }
        if (used > grid_dw) {
            bhn_set_error("internal: dW job split overflow (%d > %d)", used, grid_dw);
            return BHN_EINVAL;
        }
    }
    // ring + bias rows + zero row + output weights + identity fragments
    const size_t lds_fixed = ((size_t)depth * W + 32) * 4 + 128 + W * 4 + (Pol::ELEM_BYTES == 2 ? 0 : 2 * Pol::FRAG_BYTES) + RaySum<Pol::NWAVES>::bytes(A.f.Sx);
    const size_t lds_taped = (size_t)(BG::RING_DIST_TAPED + 1) * PK::CHUNK_BYTES + lds_fixed;
    // training forward with the resident encoded-input block (EncBlock): ring buffers of the KS hidden fragments + the block
    const size_t lds_fwd_encr = (size_t)(BG::RING_DIST_TAPED + 1) * PK::KS * Pol::FRAG_BYTES + lds_fixed + EncBlock<W, Pol>::BYTES;
    size_t lds_dw = (size_t)BG::NBUF * BG::GROUP_BYTES + (t1.drop_h1 ? (size_t)2 * BG::MT * Pol::FRAG_BYTES + W * 4 : 0);
    if (t1.drop_ga && (size_t)BG::NBUF * BG::GROUP_BYTES_LAST2 > lds_dw) lds_dw = (size_t)BG::NBUF * BG::GROUP_BYTES_LAST2;
    if constexpr (Pol::TAPE8) {          // the 8-bit jobs other than layer 1's run a deeper ring of smaller group images (dw_body2: NB)
        const size_t deep = (size_t)BHN_T8_NBUF * (2 * BG::MT * BG::TAPE_TILE + BG::TILE_BYTES + 1024);
        if (deep > lds_dw) lds_dw = deep;
    }
#ifndef BHN_RESIDENT
#define BHN_RESIDENT 1           // 0: never keep the weight images resident in LDS (A/B builds)
#endif
    // small networks: the training forward / the delta chain keep their whole chunk sequence in LDS and run without the
    // per-chunk barrier (ResidentRing), each when its own sequence fits
    const size_t res_fwd = (size_t)PK::fwd_chunks(depth) * PK::CHUNK_BYTES + lds_fixed, res_chn = (size_t)PK::bwd_chunks(depth) * PK::CHUNK_BYTES + lds_fixed;
    constexpr bool CAN_RES = BHN_RESIDENT != 0 && W <= 128 && BHN_CHAIN_STAMPS == 0;     // (width 256: no second instantiation)
    const bool rf = CAN_RES && res_fwd <= 160 * 1024, rch = CAN_RES && res_chn <= 160 * 1024;
    auto k_fwd = rf ? chain_kernel<W, Pol, 3, MODE_FWD_TRAIN, CAN_RES> : chain_kernel<W, Pol, 3, MODE_FWD_TRAIN, false>;
    // (x12: the resident image + the ray-sum scratch of 12 groups: 154 KB at four Stokes planes)
    const size_t res_fwd_x = res_fwd - RaySum<Pol::NWAVES>::bytes(A.f.Sx) + RaySum<FPol::NWAVES>::bytes(A.f.Sx);
    if constexpr (CAN_X) {
        if (x12) {
            BHN_CHECK_ARG(CAN_RES && res_fwd_x <= 160 * 1024, "internal: the 12-wave training forward needs its weights resident (%zu bytes)", res_fwd_x);
            k_fwd = chain_kernel<W, FPol, 3, MODE_FWD_TRAIN, CAN_RES>;
        }
    }
    const unsigned nthr_fwd = (unsigned)nwf * 64u;
    constexpr bool CAN_GA0C = ga0_chain_ok<W, Pol>(3);                // (compile-time part of the condition: which widths instantiate it)
    auto k_chn = rch ? chain_kernel<W, Pol, 3, MODE_CHAIN, CAN_RES> : chain_kernel<W, Pol, 3, MODE_CHAIN, false>;
    if (ga0c) k_chn = chain_kernel<W, Pol, 3, MODE_CHAIN, false, CAN_GA0C>;
    // ga0_chain: a ring of BHN_GA0C_DIST + 1 buffers of the KS fragments the chain streams, the fixed part, 4 staging images of one tile per wave
    const size_t lds_ga0c = (size_t)(BHN_GA0C_DIST + 1) * PK::KS * Pol::FRAG_BYTES + lds_fixed + (size_t)4 * Pol::NWAVES * BG::TILE_BYTES;
    const size_t lds_fwd = x12 ? res_fwd_x : rf ? res_fwd : (EncBlock<W, Pol>::ON ? lds_fwd_encr : lds_taped), lds_chn = ga0c ? lds_ga0c : rch ? res_chn : lds_taped;
    auto kdw = dw_kernel<W, Pol>;
    static DeviceOnce once;                 // per template instantiation and device
    BHN_HIP(once.run(device, [&](int &) {
        for (const void *k : {(const void *)chain_kernel<W, FPol, 3, MODE_FWD_TRAIN, CAN_RES>, (const void *)chain_kernel<W, Pol, 3, MODE_FWD_TRAIN, CAN_RES>, (const void *)chain_kernel<W, Pol, 3, MODE_FWD_TRAIN, false>,
                              (const void *)chain_kernel<W, Pol, 3, MODE_CHAIN, CAN_RES>, (const void *)chain_kernel<W, Pol, 3, MODE_CHAIN, false>,
                              (const void *)chain_kernel<W, Pol, 3, MODE_CHAIN, false, CAN_GA0C>, (const void *)kdw}) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }));
    BHN_CHECK_ARG(lds_taped <= 160 * 1024 && lds_dw <= 160 * 1024 && lds_chn <= 160 * 1024, "LDS budget exceeded (chain %zu / %zu, dw %zu)", lds_taped, lds_chn, lds_dw);
    const int B_total = A.f.B;
    const double *tM0 = A.f.tM0;
    int nslabs128 = 0;
    if constexpr (Pol::TAPE8) {
        if (what != RUN_FWD_TRAIN) {         // this call's tape scales: the stored ratios times the size of THIS d(loss)/d(images)
            const long long npx = (long long)B_total * A.f.Sx * A.f.R;
            // a calibrating call starts from a clean state block: the workspace may never have been used, and t8_update keeps
            // a layer's OLD ratio when the chain saw only zeros there -- all-zero ratios fall back to scale 1 (t8_prepare)
            if (t8_cal) BHN_HIP(hipMemsetAsync(A.t8, 0, t8_bytes, st));
            hipLaunchKernelGGL(t8_dmax_kernel, dim3((unsigned)(npx >= 65536 ? 64 : (npx + 1023) / 1024)), dim3(1024), 0, st, A.t8, dimages, npx);
            hipLaunchKernelGGL(t8_prepare_kernel, dim3(1), dim3(64), 0, st, A.t8, depth, 1);
            BHN_HIP(hipGetLastError());
        }
    }
    if (what == RUN_FWD_TRAIN)
        BHN_HIP(hipMemsetAsync(images, 0, sizeof(float) * (size_t)B_total * A.f.Sx * A.f.R, st));
    for (int b0 = 0, pass = 0; b0 < B_total; b0 += (int)fpp, ++pass) {
        const int nb = (b0 + fpp <= B_total) ? (int)fpp : B_total - b0;
        A.f.B = nb;
        A.f.tM0 = tM0 + b0;
        A.f.dimages = dimages ? dimages + (long long)b0 * A.f.Sx * A.f.R : nullptr;
        A.f.total_tiles = (long long)A.f.tiles_per_frame * nb;
        layout(A.f.total_tiles * nwf, &A.t);
        A.accumulate = pass > 0;
        A.debug = g_bwd_debug;
#ifdef BHN_DEBUG
        A.ts_buf = (g_bwd_debug & 512) ? reinterpret_cast<long long *>(bhn_debug_buffer()) : nullptr;
        A.wrap = dbg_env_int("BHN_DEBUG_WRAP", 0);
        A.policy = dbg_env_int("BHN_DEBUG_POLICY", 0);
#endif
        // (ga0_chain: later passes ACCUMULATE onto the dW_0 slabs of the first: never more workgroups than the first pass had)
        const long long grid = bhn_balanced_grid(A.f.total_tiles, (pass > 0 && A.n_chain_wg > 0 && A.n_chain_wg < ncu) ? A.n_chain_wg : ncu);
        if (pass == 0) A.n_chain_wg = (int)grid;
        if (what == RUN_FWD_TRAIN) {
            hipLaunchKernelGGL(k_fwd, dim3((unsigned)grid), dim3(nthr_fwd), lds_fwd, st, A);
            BHN_HIP(hipGetLastError());
            continue;
        }
        // bhn_render_bwd_tape_timed: event i is recorded behind kernel i - 1 (single-pass calls only)
        auto mark = [&](int i) -> hipError_t {
            return (events && i < n_events && events[i] && pass == 0) ? hipEventRecord((hipEvent_t)events[i], st) : hipSuccess;
        };
        BHN_HIP(mark(0));
        if (f128) {                          // kernel slot 0 = the fused chain + dW kernel, slot 1 empty
            if (what == RUN_RECOMPUTE) {
                hipLaunchKernelGGL(k_fwd, dim3((unsigned)grid), dim3(nthr_fwd), lds_fwd, st, A);
                BHN_HIP(hipGetLastError());
            }
            // (later passes ACCUMULATE onto the slabs of the first: never more workgroups than the first pass had)
            const long long g128 = bhn_balanced_grid(A.t.NQ / 4, (pass > 0 && nslabs128 > 0 && nslabs128 < ncu) ? nslabs128 : ncu);
            if ((int)g128 > nslabs128) nslabs128 = (int)g128;
            const int rc128 = bwd128_launch(A, depth, (int)g128, st);
            if (rc128 != BHN_OK) return rc128;
            BHN_HIP(mark(1));
            BHN_HIP(mark(2));
            continue;
        }
        if (g_bwd_stages & 1) {
            if (what == RUN_RECOMPUTE) {     // forward again (tape only: A.f.images is null), then the chain
                hipLaunchKernelGGL(k_fwd, dim3((unsigned)grid), dim3(nthr_fwd), lds_fwd, st, A);
                BHN_HIP(hipGetLastError());
            }
            if constexpr (Pol::TAPE8) {
                if (t8_cal && pass == 0) {
                    // calibration: the delta chain once with nothing limited (its tape output is overwritten below); the
                    // ratios of its |gA_l| maxima to |dimages|max give this call's scales
                    hipLaunchKernelGGL(t8_open_kernel, dim3(1), dim3(64), 0, st, A.t8);
                    hipLaunchKernelGGL(k_chn, dim3((unsigned)grid), dim3(Pol::NTHREADS), lds_chn, st, A);
                    hipLaunchKernelGGL(t8_update_kernel, dim3(1), dim3(64), 0, st, A.t8, depth);
                    hipLaunchKernelGGL(t8_prepare_kernel, dim3(1), dim3(64), 0, st, A.t8, depth, 0);
                    BHN_HIP(hipGetLastError());
                }
            }
            hipLaunchKernelGGL(k_chn, dim3((unsigned)grid), dim3(Pol::NTHREADS), lds_chn, st, A);
        }
        BHN_HIP(hipGetLastError());
        BHN_HIP(mark(1));
        if (g_bwd_stages & 2) hipLaunchKernelGGL(kdw, dim3((unsigned)A.wg_begin[depth + 1]), dim3(Pol::NTHREADS), lds_dw, st, A);
        BHN_HIP(hipGetLastError());
        BHN_HIP(mark(2));
    }
    if (what != RUN_FWD_TRAIN && f128) {
        const int rcr = reduce128_launch(A, depth, nslabs128, st);
        if (rcr != BHN_OK) return rcr;
    } else if (what != RUN_FWD_TRAIN && (g_bwd_stages & 4)) {
        A.chain_step = 1;
        if (ga0c && A.n_chain_wg > 16) {
            const int parts = 16;
            A.chain_step = (A.n_chain_wg + parts - 1) / parts;
            hipLaunchKernelGGL(chain_slab_stage1, dim3(BG::MT, parts), dim3(256), 0, st, A, BG::MT, A.chain_step);
        }
        hipLaunchKernelGGL((reduce_kernel<W, Pol>), dim3((unsigned)ReduceGeom<W, Pol>::blocks(depth, (unsigned)A.f.skip_mask)), dim3(256), 0, st, A);
    }
    if constexpr (Pol::TAPE8) {
        if (what != RUN_FWD_TRAIN) hipLaunchKernelGGL(t8_update_kernel, dim3(1), dim3(64), 0, st, A.t8, depth);      // the next call's ratios
    }
    BHN_HIP(hipGetLastError());
    if (events && n_events > 3 && events[3]) BHN_HIP(hipEventRecord((hipEvent_t)events[3], st));
    return BHN_OK;
}

template <class Pol>
static int bwd_dispatch(int what, int width, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                        const bhn_frames *fr, const float *dimages, float *images, float *dparams, void *ws, size_t wsb,
                        hipStream_t st, size_t *qb, int qB, long long qP, int device, void *const *ev = nullptr, int nev = 0) {
    switch (width) {
        case 32: return bwd_run<32, Pol>(what, m, mode, packed, geom, fr, dimages, images, dparams, ws, wsb, st, qb, qB, qP, device, ev, nev);
        case 64: return bwd_run<64, Pol>(what, m, mode, packed, geom, fr, dimages, images, dparams, ws, wsb, st, qb, qB, qP, device, ev, nev);
        case 128: return bwd_run<128, Pol>(what, m, mode, packed, geom, fr, dimages, images, dparams, ws, wsb, st, qb, qB, qP, device, ev, nev);
        case 256: return bwd_run<256, Pol>(what, m, mode, packed, geom, fr, dimages, images, dparams, ws, wsb, st, qb, qB, qP, device, ev, nev);
        default:
            bhn_set_error("net_width %d: fused kernels are built for 32, 64, 128, 256", width);
            return BHN_EUNSUPPORTED;
    }
}

static int bwd_entry(int what, const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                     const bhn_frames *fr, const float *dimages, float *images, float *dparams, void *workspace,
                     size_t workspace_bytes, void *stream, void *const *ev = nullptr, int nev = 0) {
    BHN_CHECK_ARG(m, "null model");
    const bool t8 = (mode & 0xff) == BHN_BF16_T8 && bhn_norm_mode(mode) == BHN_BF16;
    BHN_CHECK_ARG(mode == BHN_F32 || mode == BHN_BF16 || t8, "bad mode %d", mode);
    int dev = 0;
    BHN_HIP(hipGetDevice(&dev));
    MlpShape shape;
    const int rcs = bhn_mlp_shape(m, &shape);
    if (rcs != BHN_OK) return rcs;
    const int kernel_width = shape.width;
    if (shape.general) {
        // shapes outside the fused kernels (general_mlp.hip, f32 in both modes): the training forward records the tape when the
        // workspace holds it, bhn_render_bwd recomputes chunk by chunk
        if (t8) { bhn_set_error("BHN_BF16_T8 (8-bit tape) is built for net_width 256, posenc_deg <= 4; use BHN_BF16"); return BHN_EUNSUPPORTED; }
        BHN_CHECK_ARG(!ev, "per-kernel events are not available for posenc_deg > 4 / net_width > 256");
        if (what == RUN_FWD_TRAIN) {
            BHN_CHECK_ARG(images, "null pointer");
            return gen_forward(true, m, mode, packed, geom, fr, images, (hipStream_t)stream, workspace, workspace_bytes);
        }
        return gen_backward(what == RUN_BWD_TAPE, m, mode, packed, geom, fr, dimages, dparams, workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (t8) {
        if (kernel_width != 256 || shape.depth < 3) {
            bhn_set_error("BHN_BF16_T8 (8-bit tape) is built for net_width 256 and net_depth >= 3 (got %d x %d); use BHN_BF16", shape.depth, shape.width_true);
            return BHN_EUNSUPPORTED;
        }
        return bwd_run<256, PolBF16T8>(what, m, mode, packed, geom, fr, dimages, images, dparams, workspace, workspace_bytes,
                                       (hipStream_t)stream, nullptr, 0, 0, dev, ev, nev);
    }
    return (mode == BHN_BF16)
               ? bwd_dispatch<PolBF16>(what, kernel_width, m, mode, packed, geom, fr, dimages, images, dparams, workspace,
                                       workspace_bytes, (hipStream_t)stream, nullptr, 0, 0, dev, ev, nev)
               : bwd_dispatch<PolF32>(what, kernel_width, m, mode, packed, geom, fr, dimages, images, dparams, workspace,
                                      workspace_bytes, (hipStream_t)stream, nullptr, 0, 0, dev, ev, nev);
}

extern "C" size_t bhn_render_bwd_workspace_bytes(const bhn_model *m, int32_t mode, int32_t B, int64_t P, int32_t device) {
    MlpShape s;
    if (bhn_mlp_shape(m, &s) != BHN_OK || B <= 0 || P <= 0) return 0;
    size_t q = 0;
    if (s.general && (mode & 0xff) != BHN_BF16_T8) return gen_bwd_workspace_bytes(s, mode, B, P);
    if ((mode & 0xff) == BHN_BF16_T8 && bhn_norm_mode(mode) == BHN_BF16) {
        if (s.general || s.width != 256 || s.depth < 3) {
            bhn_set_error("BHN_BF16_T8 (8-bit tape) is built for net_width 256 and net_depth >= 3 (got %d x %d); use BHN_BF16", s.depth, s.width_true);
            return 0;
        }
        return bwd_run<256, PolBF16T8>(RUN_QUERY, m, mode, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &q, B, P, device) == BHN_OK ? q : 0;
    }
    int rc = (mode == BHN_BF16)
                 ? bwd_dispatch<PolBF16>(RUN_QUERY, s.width, m, mode, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &q, B, P, device)
                 : bwd_dispatch<PolF32>(RUN_QUERY, s.width, m, mode, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &q, B, P, device);
    return rc == BHN_OK ? q : 0;
}

extern "C" int bhn_tape_info(const bhn_model *m, int32_t mode, int64_t groups_per_frame, int64_t *info, int32_t n_info) {
    BHN_CHECK_ARG(m && info && n_info >= BHN_TAPE_INFO_N, "bhn_tape_info: info must hold %d entries", BHN_TAPE_INFO_N);
    MlpShape s;
    const int rcs = bhn_mlp_shape(m, &s);
    if (rcs != BHN_OK) return rcs;
    for (int i = 0; i < BHN_TAPE_INFO_N; ++i) info[i] = 0;
    const bool t8 = (mode & 0xff) == BHN_BF16_T8 && bhn_norm_mode(mode) == BHN_BF16;
    BHN_CHECK_ARG(mode == BHN_F32 || mode == BHN_BF16 || t8, "bad mode %d", mode);
    if (s.general) { info[4] = 64; info[5] = 1; return BHN_OK; }       // (the general path: one 32-point group per workgroup, an f32 tape in chunks)
    static_assert(sizeof(size_t) == sizeof(int64_t), "RUN_INFO passes the output array through the size query's pointer");
    size_t *q = reinterpret_cast<size_t *>(info);
    if (t8) {
        if (s.width != 256 || s.depth < 3) { bhn_set_error("BHN_BF16_T8: net_width 256, net_depth >= 3"); return BHN_EUNSUPPORTED; }
        return bwd_run<256, PolBF16T8>(RUN_INFO, m, mode, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, q, 1, groups_per_frame, 0);
    }
    return (mode == BHN_BF16)
               ? bwd_dispatch<PolBF16>(RUN_INFO, s.width, m, mode, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, q, 1, groups_per_frame, 0)
               : bwd_dispatch<PolF32>(RUN_INFO, s.width, m, mode, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, q, 1, groups_per_frame, 0);
}

extern "C" int bhn_render_bwd(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                              const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                              size_t workspace_bytes, void *stream) {
    return bwd_entry(RUN_RECOMPUTE, m, mode, packed, geom, fr, dimages, nullptr, dparams, workspace, workspace_bytes, stream);
}

extern "C" int bhn_render_fwd_train(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                                    const bhn_frames *fr, float *images, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    return bwd_entry(RUN_FWD_TRAIN, m, mode, packed, geom, fr, nullptr, images, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int bhn_render_bwd_tape(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                                   const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    return bwd_entry(RUN_BWD_TAPE, m, mode, packed, geom, fr, dimages, nullptr, dparams, workspace, workspace_bytes, stream);
}

extern "C" int bhn_render_bwd_tape_timed(const bhn_model *m, int32_t mode, const void *packed, const bhn_geom *geom,
                                         const bhn_frames *fr, const float *dimages, float *dparams, void *workspace,
                                         size_t workspace_bytes, void *stream, void *const *events, int32_t n_events) {
    BHN_CHECK_ARG(events && n_events >= 1 && n_events <= BHN_BWD_TAPE_KERNELS + 1, "events: 1..%d HIP events", BHN_BWD_TAPE_KERNELS + 1);
    return bwd_entry(RUN_BWD_TAPE, m, mode, packed, geom, fr, dimages, nullptr, dparams, workspace, workspace_bytes, stream,
                     events, n_events);
}

extern "C" const char *bhn_render_bwd_tape_kernel_name(int32_t i);
extern "C" const char *bhn_render_bwd_tape_kernel_name_for(const bhn_model *m, int32_t mode, int32_t i) {
    MlpShape s;
    if (!m || bhn_mlp_shape(m, &s) != BHN_OK) return nullptr;
    if (s.general) {
        static const char *const names[BHN_BWD_TAPE_KERNELS] = {"gen_mlp_kernel<GEN_CHAIN>", "gen_dw_kernel", "gen_reduce_kernel"};
        return (i >= 0 && i < BHN_BWD_TAPE_KERNELS) ? names[i] : nullptr;
    }
    if (bwd128_supported(bhn_norm_mode(mode), s.width, s.depth)) {
        static const char *const names[BHN_BWD_TAPE_KERNELS] = {"bwd128_kernel", "-", "reduce128_kernel"};
        return (i >= 0 && i < BHN_BWD_TAPE_KERNELS) ? names[i] : nullptr;
    }
    return bhn_render_bwd_tape_kernel_name(i);
}

extern "C" const char *bhn_render_bwd_tape_kernel_name(int32_t i) {
    static const char *const names[BHN_BWD_TAPE_KERNELS] = {"chain_kernel<MODE_CHAIN>", "dw_kernel", "reduce_kernel"};
    return (i >= 0 && i < BHN_BWD_TAPE_KERNELS) ? names[i] : nullptr;
}

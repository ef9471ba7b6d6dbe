// Building blocks of the fused predictor kernels (gfx950 / CDNA4).
//
// Data flow (DESIGN.md "fused predictor"): one wave owns 32 points.  Activations are kept
// TRANSPOSED, H^T [features x points]: the point is the MFMA column (lane & 31), features live in
// registers.  A layer is  H_{l+1}^T = W_l^T . H_l^T  with  A = W_l^T (LDS, pre-packed in fragment
// order) and  B = H_l^T (registers).  The 32x32 f32 accumulator of v_mfma_f32_32x32x16_bf16 has
// its column on the lane and rows (r&3)+8*(r>>2)+4*(lane>>5) in its 16 registers, so registers
// 8s..8s+7 of an accumulator ARE the B fragment of k-step s of the next layer: activations never
// leave registers between layers.
//
// Canonical register order of a 32-feature block: element j (0..7) of k-step s (0..1) of lane-half
// h is feature 16*s + 8*(j>>2) + 4*h + (j&3).   (phi below)
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DEVI __device__ __forceinline__

// feature index inside a 16-feature k-step for (lane half h, element j)
DEVI constexpr int phi16(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }

// ---------------------------------------------------------------------------------------------
// Precision policies
// ---------------------------------------------------------------------------------------------
struct PolBF16 {
    static constexpr int MODE = BHN_BF16;
    static constexpr int NWAVES = 8;            // 2 waves per SIMD, <=256 VGPRs each
    static constexpr int NTHREADS = NWAVES * 64;
    static constexpr int WPE = 2;               // waves per SIMD the kernels are built for (register budget 512 / WPE)
#ifndef BHN_FWD_DIST
#define BHN_FWD_DIST 4           // weight chunks in flight in the inference forward (6 measured 6 % slower here)
#endif
    static constexpr int FWD_DIST = BHN_FWD_DIST;
    static constexpr int ELEM_BYTES = 2;
    static constexpr int FRAG_BYTES = 1024;     // 64 lanes x 8 bf16
    static constexpr bool TAPE8 = false;        // (PolBF16T8: the h / gA tape tiles in 8 bits)
#ifndef BHN_LDS_PF
#define BHN_LDS_PF 4
#endif
    static constexpr int LDS_PREFETCH = BHN_LDS_PF;      // A fragments in flight + 1 (ring_step); 6 / 8 measured no faster
    static constexpr bool PHASE_LAG = false;    // RingState LAG: measured 5 % slower in the render kernel, off
    using frag = bf16x8;
    static DEVI frag zero() { frag f; for (int j = 0; j < 8; ++j) f[j] = (__bf16)0.f; return f; }
    static DEVI frag lds_frag(const char *chunk, int f, int lane) {
        return *reinterpret_cast<const frag *>(chunk + f * FRAG_BYTES + lane * 16);
    }
    static DEVI f32x16 mma(const frag &a, const frag &b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static DEVI void set(frag &f, int j, float v) { f[j] = (__bf16)v; }
    // one instruction (v_med3_f32; the plain C form gets a canonicalising second v_max from hipcc).  NOT inline asm:
    // the hazard recogniser must see this VALU read of an MFMA result to insert the wait states gfx9 needs
    static DEVI float relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }
    // ReLU bits of a 32x32 tile, per lane a 16-bit code: bit k = element 2k is active, bit 8+k = element 2k+1
    // (k = 0..7: the k-th packed dword of the tile's two B fragments).  While a tile is being packed the bits are
    // kept "spread" (bit k and bit 16+k), which is what two-at-a-time operations on the packed dword produce:
    //   relu_pair : dword = max_i16x2(pack(a, b), 0); spread |= min_u16x2(dword, 1) << k     (4 VALU per pair)
    //   mask_pair : dword = pack(a, b) & pk_ashr15(spread << (15 - k))                       (4 VALU per pair)
    // A bf16 result of +0 counts as inactive (an f32 pre-activation below 2^-133 would be active in exact arithmetic).
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    // 1 in each 16-bit half of u that is nonzero, for halves that are non-negative as signed integers (clamped bf16:
    // <= 0x7f80): a SIGNED packed min with 1 = one v_pk_min_i16, compiler-visible.  (The unsigned form of the same
    // builtin is turned into two compares, two selects and a v_perm; the inline-asm v_pk_min_u16 used until round 2 was
    // opaque to the hazard recogniser -- after the pack was pipelined it landed in a SrcA register two instructions behind
    // its MFMA, which tools/check_asm_hazard.py caught -- and cost an s_nop behind every clamp.)
    // The (1, 1) operand is made opaque (an SGPR behind an empty asm): knowing both that the halves are clamped and that
    // the other operand is 1, hipcc folds clamp + min into per-half compares, selects and a v_perm (100 VALU per tile).
    static DEVI unsigned nonzero_halves(unsigned u) {
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        unsigned one = 0x00010001u;
        asm("" : "+s"(one));
        return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(i16x2, u), __builtin_bit_cast(i16x2, one)));
    }
    static DEVI void put_dword(frag &f, int i, unsigned u) {
        u32x4 w = __builtin_bit_cast(u32x4, f);
        w[i] = u;
        f = __builtin_bit_cast(frag, w);
    }
    static DEVI void relu_pair(frag &f, int i, int k, float a, float b, unsigned &spread) {
        // round first, then clamp the two bf16 halves as signed 16-bit integers (sign bit set -> 0, -0 -> +0): one
        // v_cvt_pk + one v_pk_max_i16 instead of two v_max per element (hipcc canonicalises before every max).
        // Inline asm only on the VALU result of the convert, never on an MFMA accumulator (hazard, see relu()).
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 ab = {a, b};           // ONE vector conversion = one v_cvt_pk_bf16_f32 (two scalar ones: two converts + v_perm)
        const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bf16x2));
        // v_pk_max_i16 through the vector builtin, not as an inline-asm instruction: its result replaces fragment
        // registers that an MFMA issued just before may still be reading, and only compiler-visible writes are
        // covered by the hazard recogniser (tools/check_asm_hazard.py)
        const unsigned u = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), (i16x2){0, 0}));
        put_dword(f, i, u);
        spread |= nonzero_halves(u) << k;
    }
    // relu_pair in three phases for the software-pipelined pack of ring_step (one phase per k-step and pair): hipcc puts
    // an `s_nop 0` between a VALU instruction and an inline-asm statement that reads its result in the next cycle, and
    // relu_pair was convert -> (asm pin) -> clamp -> (asm v_pk_min_u16 / asm use marker): 16-20 s_nop per ring step
    // (10 % of its instructions, round-2 ISA census).  With the phases one k-step apart the neighbours are independent.
    static DEVI unsigned pack_a(float a, float b) {                       // round (ONE vector conversion = v_cvt_pk_bf16_f32;
        typedef float f32x2 __attribute__((ext_vector_type(2)));          //  two scalar ones become two converts + v_perm)
        const f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static DEVI void pack_b(frag &f, int i, int, float, float, unsigned w, unsigned &) {       // clamp
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        // (no asm pin on w here: rounded one k-step earlier, behind a sched_barrier, the convert stays one v_cvt_pk_bf16_f32)
        put_dword(f, i, __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), (i16x2){0, 0})));
    }
    static DEVI void pack_c(const frag &f, int i, int k, unsigned &spread) {                  // relu bits
        spread |= nonzero_halves(__builtin_bit_cast(u32x4, f)[i]) << k;
    }
    static DEVI void mask_pair(frag &f, int i, int k, float a, float b, unsigned spread) {
        // bits k and 16+k of `spread` moved to the sign bits of the two halves, spread by a packed arithmetic >> 15
        // (v_lshlrev_b32 + v_pk_ashrrev_i16: one VALU less than mask-and-multiply)
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 t = {(__bf16)a, (__bf16)b};
        const i16x2 on = __builtin_bit_cast(i16x2, spread << (15 - k)) >> (i16x2){15, 15};
        put_dword(f, i, __builtin_bit_cast(unsigned, t) & __builtin_bit_cast(unsigned, on));
    }
    static DEVI unsigned mask_code(unsigned spread) { return (spread & 0xffu) | ((spread >> 8) & 0xff00u); }
    static DEVI unsigned mask_spread(unsigned code) { return (code & 0xffu) | ((code & 0xff00u) << 8); }
    static DEVI float get(const frag &f, int j) { return (float)f[j]; }
    // acc + the sum of the fragment's eight values (v_dot2c_f32_bf16 against (1, 1): the bias column of a dW GEMM)
    // (the pairs are formed from ELEMENTS of the fragment: with the dwords of a u32x4 bit-cast to bf16x2 hipcc 7.2 feeds
    //  dword 0 to all four dot products -- reproduced in ten lines, found in the ISA)
    static DEVI float sum8(const frag &f, float acc) {
        const bf16x2 one = {(__bf16)1.f, (__bf16)1.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x2 p = {f[2 * i], f[2 * i + 1]};
            acc = __builtin_amdgcn_fdot2_f32_bf16(p, one, acc, false);
        }
        return acc;
    }
    static DEVI float fsin_rev(float rev) { return __builtin_amdgcn_sinf(rev); }   // sin(2*pi*rev)
    static DEVI float fcos_rev(float rev) { return __builtin_amdgcn_cosf(rev); }
    static DEVI float fexp(float x) { return __expf(x); }
    static constexpr bool FAST_TRIG = true;
};

// bf16 arithmetic everywhere, but the backward's tape keeps the two operands of the weight-gradient GEMMs -- the layer inputs
// h_l and the pre-activation gradients gA_l -- as OCP e4m3 bytes (BHN_BF16_T8; fused_bwd.hip "8-bit tape").  Forward, delta
// chain and every accumulation are those of PolBF16.
struct PolBF16T8 : PolBF16 {
    static constexpr bool TAPE8 = true;
};

// PolBF16 on 12-wave workgroups (three per SIMD) for the two forward kernels of the fused 4x128 path (round 5): at width 128 the
// forward is bound by per-point vector work and its latencies, not by the matrix pipe -- a third wave per SIMD
// hides more of them (one box, profiles/r5_ab_twelve_waves_width128.txt: inference forward 0.92 -> 0.85 ms, -7 %; training forward
// 1.43 -> 1.25 ms, -12 %; 146 / 167 of the 170 registers a wave may have at that occupancy).
// The training forward and the inference forward share the tile size: `render` and `render_train` give bit-identical images.
#ifndef BHN_W12
#define BHN_W12 1
#endif
struct PolBF16X : PolBF16 {
    static constexpr int NWAVES = 12;
    static constexpr int NTHREADS = NWAVES * 64;
    static constexpr int WPE = 3;
};
// bf16, kernel width 128, depth 4 (= bwd128_supported) and at least one round of 12-group tiles per frame on a 256-CU device: the
// problems whose forward kernels run on PolBF16X (a small ray set -- config 5: 1,460 groups per frame -- is better off with more,
// smaller tiles: 0.253 against 0.256 ms per step).  A function of the MODEL and the RAY SET only, never of the batch: the
// inference forward and the training forward of one problem always agree on the tile size.
__host__ __device__ static inline bool bhn_fwd_w12(int mode, int kernel_width, int depth, long long groups_per_frame) {
#ifdef BHN_NO_FUSED128
    return false;
#else
    return BHN_W12 != 0 && mode == BHN_BF16 && kernel_width == 128 && depth == 4 && groups_per_frame >= 12 * 256;
#endif
}
// 32-point groups per frame of a ray set as the fused kernels walk it (fused_fill_args: n_groups)
static inline long long bhn_groups_per_frame(const bhn_geom *geom) {
    if (!geom) return 0;
    if (geom->groups) return geom->n_groups;
    const long long P = geom->ray_idx ? geom->n_points : geom->R * geom->G;
    return (P + 31) / 32;
}

struct PolF32 {
    static constexpr int MODE = BHN_F32;
    static constexpr int NWAVES = 4;            // 1 wave per SIMD, 512 registers each
    static constexpr int NTHREADS = NWAVES * 64;
    static constexpr int WPE = 1;
    static constexpr int FWD_DIST = 3;
    static constexpr int ELEM_BYTES = 4;
    static constexpr int FRAG_BYTES = 2048;     // 2 halves x 64 lanes x 4 f32
    static constexpr bool TAPE8 = false;
    static constexpr int LDS_PREFETCH = 2;
    static constexpr bool PHASE_LAG = false;    // one wave per SIMD
    using frag = f32x8;
    static DEVI frag zero() { frag f; for (int j = 0; j < 8; ++j) f[j] = 0.f; return f; }
    static DEVI frag lds_frag(const char *chunk, int f, int lane) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(chunk + f * FRAG_BYTES + lane * 16);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(chunk + f * FRAG_BYTES + 1024 + lane * 16);
        frag r;
        for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
        return r;
    }
    // v_mfma_f32_32x32x2_f32: k = lane>>5, so element j of both operands pairs features
    // (8*(j>>2)+(j&3)) [h=0] and (+4) [h=1] -- the same phi16 map as the bf16 fragment.
    static DEVI f32x16 mma(const frag &a, const frag &b, f32x16 c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
    static DEVI void set(frag &f, int j, float v) { f[j] = v; }
    static DEVI float relu(float v) { return v > 0.f ? v : 0.f; }
    // same 16-bit ReLU code as PolBF16 (bit k = element 2k, bit 8+k = element 2k+1), exact `> 0`
    static DEVI void relu_pair(frag &f, int i, int k, float a, float b, unsigned &code) {
        f[2 * i] = relu(a); f[2 * i + 1] = relu(b);
        code |= (a > 0.f ? 1u << k : 0u) | (b > 0.f ? 1u << (8 + k) : 0u);
    }
    static DEVI unsigned pack_a(float, float) { return 0u; }             // (phases of the pipelined pack: see PolBF16)
    static DEVI void pack_b(frag &f, int i, int k, float a, float b, unsigned, unsigned &code) { relu_pair(f, i, k, a, b, code); }
    static DEVI void pack_c(const frag &, int, int, unsigned &) {}
    static DEVI void mask_pair(frag &f, int i, int k, float a, float b, unsigned code) {
        f[2 * i] = ((code >> k) & 1) ? a : 0.f; f[2 * i + 1] = ((code >> (8 + k)) & 1) ? b : 0.f;
    }
    static DEVI unsigned mask_code(unsigned code) { return code; }
    static DEVI unsigned mask_spread(unsigned code) { return code; }
    static DEVI float get(const frag &f, int j) { return f[j]; }
    static DEVI float sum8(const frag &f, float acc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f[j];
        return acc;
    }
    static DEVI float fexp(float x) { return expf(x); }
    static constexpr bool FAST_TRIG = false;
};

// ---------------------------------------------------------------------------------------------
// Division of a 32-bit unsigned by a run-time constant (Granlund & Montgomery): the divisor's multiplier and shifts
// come from the host.  hipcc expands `long long / long long` into ~100 instructions (there is no integer divide);
// the tile -> (frame, group) map and the point -> ray map did that twice per tile and once per point: a fifth of the
// instructions between the output layer of one tile and layer 0 of the next (round-2 ISA census).
// ---------------------------------------------------------------------------------------------
struct FastDiv {
    unsigned d, m, sh1, sh2;
    __host__ __device__ static FastDiv make(unsigned d) {
        FastDiv f;
        f.d = d;
        unsigned l = 0;
        while (l < 32 && (1ull << l) < d) ++l;                         // ceil(log2 d)
        f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
        f.sh1 = l < 1 ? l : 1;
        f.sh2 = l > 1 ? l - 1 : 0;
        return f;
    }
    __device__ __forceinline__ unsigned div(unsigned n) const {
        const unsigned t = __umulhi(m, n);
        return (t + ((n - t) >> sh1)) >> sh2;
    }
};

// ---------------------------------------------------------------------------------------------
// Kernel arguments (passed by value)
// ---------------------------------------------------------------------------------------------
struct FusedArgs {
    // model
    int depth;            // hidden layers
    int skip_mask;        // bit l set: layer l consumes concat[h, enc]
    float scale, inv_scale;   // inv_scale: bf16 policy only (one multiply instead of a ~12-instruction f32 division per coordinate)
    // geometry
    const float *x, *y, *z, *Omega, *t_geo, *w;
    const uint8_t *dom;
    const int *groups;    // active 32-point groups (NULL = all)
    const int *ray_idx;   // point-level compaction: ray of point p (NULL = p / G)
    long long n_groups;
    long long P, G, R;
    int Sx;               // max(S,1)
    // frames
    const double *tM0;
    int B;
    // packed weights
    const char *packed;
    unsigned fwd_off, bwd_off, bias_off, wout_off;   // byte offsets inside `packed`
    // outputs / inputs of the individual kernels
    float *emission;      // predict: (B,P)
    float *images;        // render fwd: (B,Sx,R)
    const float *dimages; // render bwd
    float *slabs;         // render bwd: per-workgroup dW slabs
    long long slab_floats;
    // tiling
    int tiles_per_frame;
    long long total_tiles;          // < 2^32 (checked on the host), as is P: the maps below use 32-bit FastDiv
    FastDiv fd_tpf, fd_G;
    int ray_direct;                 // every ray touches at most two 32-point wave tiles (dense: from G; compacted: bhn_geom.ray_span): one atomic per (tile, ray) is already
                                    // order-independent (RaySum::direct), no combine through LDS needed
    int debug;            // measurement builds only
    int deg;              // posenc degree 0..BHN_DEG_MAX (run time: only the prologue and the weight packing depend on it)
    long long *clk;       // bhn_frames.clock_probe (NULL: off): clock stamps of workgroup 0, four per kernel slot (clock_stamp)
};

// bhn_frames.clock_probe: thread 0 of workgroup 0 stamps the shader clock counter (s_memtime: counts GPU core cycles) and the
// constant 100-MHz counter (s_memrealtime) at the start (`end` = 0) and the end (1) of the kernel into clk[4 slot ..]: the ratio
// of the two differences is the clock the kernel sustained (bench.py: sustained_clock_mhz).  Slots: BHN_CLK_* of bhnerf_hip.h.
DEVI void clock_stamp(long long *clk, int slot, int end) {
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[4 * slot + 2 * end] = (long long)__builtin_amdgcn_s_memtime();
        clk[4 * slot + 2 * end + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// Packed-weight geometry for hidden width W
template <int W, class Pol>
struct Pack {
    static constexpr int MT = W / 32;                 // 32-row output tiles per hidden layer
    static constexpr int KS = W / 16;                 // 16-feature k-steps of a hidden input
    static constexpr int CH = KS + 2;                 // fragments per chunk (hidden + enc block)
    static constexpr int CHUNK_BYTES = CH * Pol::FRAG_BYTES;
    static_assert(2 * MT <= CH, "layer-0 chunk must hold all its fragments");
    // forward image: chunk 0 = layer 0 (frag m*2+ks), chunks 1+(l-1)*MT+m = hidden layer l tile m
    // (frag ks, enc block at KS,KS+1), last chunk = output layer (frag ks, row 0 real).
    __host__ __device__ static constexpr int fwd_chunks(int depth) { return 1 + (depth - 1) * MT + 1; }
    // transposed image (delta chain through hidden layer l>=1): chunk (l-1)*MT+m, frag ks:
    //   element = kernel_l[32m+i][16ks+phi]
    __host__ __device__ static constexpr int bwd_chunks(int depth) { return (depth - 1) * MT; }
};

// ---------------------------------------------------------------------------------------------
// Per-point prologue: velocity warp (emission.py:200-210) + positional encoding (network.py:118-122)
// ---------------------------------------------------------------------------------------------
struct PointState {
    bool live;        // contributes: in range, inside the domain, after injection, finite
    long long p;      // flat point index inside the frame
};

// tile -> (frame, point) mapping shared by all fused kernels: wave `wv` of tile `tile` owns the 32-point group
// number (tile % tiles_per_frame) * NWAVES + wv of the (compacted) group list
template <int NWAVES>
DEVI void tile_point(const FusedArgs &a, long long tile, int wv, int pl, int &b, long long &p, bool &inb) {
    const unsigned tq = a.fd_tpf.div((unsigned)tile);
    b = (int)tq;
    const long long gi = (long long)((unsigned)tile - tq * (unsigned)a.tiles_per_frame) * NWAVES + wv;
    const bool gok = gi < a.n_groups;
    const long long grp = gok ? (a.groups ? (long long)a.groups[gi] : gi) : 0;
    p = grp * 32 + pl;
    inb = gok && p < a.P;
}

// the same with the groups per tile as a run-time number (kernels that walk the tape of a forward of another policy)
DEVI void tile_point_nw(const FusedArgs &a, long long tile, int wv, int pl, int nw, int &b, long long &p, bool &inb) {
    const unsigned tq = a.fd_tpf.div((unsigned)tile);
    b = (int)tq;
    const long long gi = (long long)((unsigned)tile - tq * (unsigned)a.tiles_per_frame) * nw + wv;
    const bool gok = gi < a.n_groups;
    const long long grp = gok ? (a.groups ? (long long)a.groups[gi] : gi) : 0;
    p = grp * 32 + pl;
    inb = gok && p < a.P;
}

// Raw per-point inputs of one tile.  They are loaded one tile AHEAD (the loads of tile i+1 are issued while
// tile i runs its layers), so the ~2 us dependent-load latency at every tile start is hidden.
struct PointIn {
    int b;
    long long p;
    bool inb;
    float x, y, z, om, tg, tm0;   // tm0 is unused; the frame offset stays in double (tM0d)
    double tM0d;
    bool dom;
};

template <int NG>
DEVI PointIn load_point(const FusedArgs &a, long long tile, int gslot, int pl) {
    PointIn q;
    q.x = q.y = q.z = q.om = q.tg = q.tm0 = 0.f;
    q.tM0d = 0.0;
    q.dom = false;
    q.b = 0; q.p = 0; q.inb = false;
    if (tile < a.total_tiles) {
        tile_point<NG>(a, tile, gslot, pl, q.b, q.p, q.inb);
        q.tM0d = a.tM0[q.b];
        if (q.inb) {
            q.x = a.x[q.p]; q.y = a.y[q.p]; q.z = a.z[q.p]; q.om = a.Omega[q.p]; q.tg = a.t_geo[q.p];
            q.dom = a.dom[q.p] != 0;
        }
    }
    return q;
}

template <class Pol, int DEG>
DEVI void point_prologue(const FusedArgs &a, const PointIn &in, typename Pol::frag (&enc)[2], bool &live) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const bool inb = in.inb;
    const float x = in.x, y = in.y, z = in.z, om = in.om, tg = in.tg;
    const bool dom = in.dom;
    // t_M = (t_frame - t_start_obs)/GM_c3 + t_geo - t_injection in double: the f32 reference loses
    // ~6e-5 here (|t_geo| ~ 1e3), the oracle is float64 (DESIGN.md "numerics").
    const double tM = in.tM0d + (double)tg;
    const bool pre = tM < 0.0;                                  // emission.py:204-205 -> NaN
    const double rev_d = tM * (double)om * 0.15915494309189535; // theta / (2 pi)
    const double fr = rev_d - floor(rev_d);
    float s, c;
    if (Pol::FAST_TRIG) {
        s = __builtin_amdgcn_sinf((float)fr);
        c = __builtin_amdgcn_cosf((float)fr);
    } else {
        sincosf((float)(fr * 6.283185307179586), &s, &c);
    }
    // rot_z(-theta) (utils.py:126-132 with angle=-theta): x' = c x + s y, y' = -s x + c y, z' = z
    float u[3];
    u[0] = c * x + s * y;
    u[1] = c * y - s * x;
    u[2] = z;
    const bool finite_theta = !pre && (fr == fr);
    bool valid[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        valid[k] = finite_theta && isfinite(u[k]);               // network.py:226
        u[k] = valid[k] ? (Pol::FAST_TRIG ? u[k] * a.inv_scale : u[k] / a.scale) : 0.f;     // network.py:227,229 (f32 mode: the division itself)
    }
    live = inb && dom && valid[0];                               // emission.py:370-373, network.py:232
    // encoded features in the kernel's slot layout (common.h): register n = 8 ks + j of this lane holds u_n (n < 3, lane half 0),
    // sin(2^i u_k) on half 0 / cos(2^i u_k) on half 1 (n = 3 + 3 i + k, i < deg); bhn_pack_weights puts the reference's rows
    // [u | sin block | cos block], network.py:118-122, on these slots, unused slots meet zero weight rows.  The degree is a
    // run-time argument: a wave-uniform branch per octave.
#if BHN_ENC_PAIRS
    float reg[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) reg[n] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) reg[k] = h ? 0.f : u[k];
    if (Pol::FAST_TRIG) {
        // ONE transcendental per register: cos(x) = sin(x + a quarter revolution) -- v_sin_f32 takes revolutions
        const float qh = h ? 0.25f : 0.f;
        float rv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) rv[k] = u[k] * 0.15915494309189535f;
#pragma unroll
        for (int i = 0; i < BHN_DEG_MAX; ++i) {
            if (i < a.deg) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    reg[3 + 3 * i + k] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rv[k] * (float)(1 << i)) + qh);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < BHN_DEG_MAX; ++i) {
            if (i < a.deg) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float sv, cv;
                    sincosf(u[k] * (float)(1 << i), &sv, &cv);
                    reg[3 + 3 * i + k] = h ? cv : sv;
                }
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) Pol::set(enc[ks], j, reg[8 * ks + j]);
#else
    float feat[BHN_ENC_PAD];
#pragma unroll
    for (int q = 0; q < BHN_ENC_PAD; ++q) feat[q] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) feat[k] = u[k];
#pragma unroll
    for (int i = 0; i < BHN_DEG_MAX; ++i) {
        if (i < a.deg) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float arg = u[k] * (float)(1 << i);
                float sv, cv;
                if (Pol::FAST_TRIG) {
                    const float rev = __builtin_amdgcn_fractf(arg * 0.15915494309189535f);
                    sv = __builtin_amdgcn_sinf(rev);
                    cv = __builtin_amdgcn_cosf(rev);
                } else {
                    sincosf(arg, &sv, &cv);
                }
                feat[3 + 3 * i + k] = sv;
                feat[3 + 3 * BHN_DEG_MAX + 3 * i + k] = cv;
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v0 = feat[16 * ks + phi16(0, j)];
            const float v1 = feat[16 * ks + phi16(1, j)];
            Pol::set(enc[ks], j, h ? v1 : v0);
        }
#endif
}

// bias rows of output tile m as the initial accumulator: acc[r] = bias[32m + (r&3)+8(r>>2)+4h]
DEVI f32x16 bias_acc(const float *bias_lds, int m, int h) {
    f32x16 acc;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bias_lds + 32 * m + 8 * g4 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = v[e];
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined ring steps (fused_fwd_kernel).  Three things that used to sit between the MFMA chains
// of consecutive steps are folded INTO the chains, because the workgroup barrier keeps all eight waves in
// lockstep and the SIMD's other wave is then in the same non-MFMA phase:
//   * the first PF-1 A fragments of the NEXT chunk are read before the barrier (`ap`), so the first MFMA
//     after it does not wait for LDS;
//   * relu + repack of the PREVIOUS output tile (`pend`) runs in the shadow of this tile's first MFMAs;
//   * the DMA issue for chunk c+DIST (scalar address math + buffer_load...lds) sits after MFMA 3.
// The fences (sched_barrier 0) pin that order; hipcc otherwise sinks every LDS read to its use and hoists
// the whole pack to the end of the layer.
// ---------------------------------------------------------------------------------------------
template <class Pol, int R0, int N>
DEVI void pack_elems(const f32x16 &acc, typename Pol::frag &d0, typename Pol::frag &d1, unsigned &mask) {
    static_assert(R0 % 2 == 0 && N % 2 == 0, "elements are packed in pairs");
    if constexpr (Pol::ELEM_BYTES == 2 && N >= 4) {
        // phase by phase over the pairs (all rounded, then all clamped, then all relu bits): pair by pair every clamp sits
        // right behind its convert and every bit extraction right behind its clamp, an s_nop each (hipcc 7.2, gfx950)
        unsigned w[N / 2];
#pragma unroll
        for (int p = 0; p < N / 2; ++p) w[p] = Pol::pack_a(acc[R0 + 2 * p], acc[R0 + 2 * p + 1]);
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            const int r = R0 + 2 * p;
            Pol::pack_b(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, acc[r], acc[r + 1], w[p], mask);
        }
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            const int r = R0 + 2 * p;
            Pol::pack_c(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, mask);
        }
    } else {
#pragma unroll
        for (int r = R0; r < R0 + N; r += 2)       // mask: spread ReLU bits (Pol::mask_code), dead code unless recorded
            Pol::relu_pair(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, acc[r], acc[r + 1], mask);
    }
    // the packed registers are "used" here: without this the machine sinker moves the whole pack down to the
    // next layer's first read of the fragment, i.e. out of the MFMA shadow it was placed in
    if (R0 < 8) asm volatile("" : "+v"(d0));
    if (R0 + N > 8) asm volatile("" : "+v"(d1));
}

// Software-pipelined relu + repack of a pending tile over k-steps 0..9 of a ring step (pair p = elements 2p, 2p+1):
// k-step t rounds pair t, clamps pair t-1 into its fragment dword and takes the relu bits of pair t-2 (Pol::pack_a/b/c).
// The fragments are complete after k-step 8, the relu bits after k-step 9.
template <class Pol, bool BITS>
DEVI void pack_pipe(int t, const f32x16 &pend, typename Pol::frag &d0, typename Pol::frag &d1, unsigned &w, unsigned &mask) {
    if (t >= 1 && t <= 8) {
        const int r = 2 * (t - 1);
        Pol::pack_b(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, pend[r], pend[r + 1], w, mask);
    }
    if (t <= 7) w = Pol::pack_a(pend[2 * t], pend[2 * t + 1]);
    if (t >= 2 && t <= 9) {
        const int r = 2 * (t - 2);
        Pol::pack_c(r < 8 ? d0 : d1, (r & 7) >> 1, r >> 1, mask);
    }
    // the packed registers are "used" once, a k-step after the last write (see pack_elems; an empty asm right behind the
    // VALU instruction that writes a packed 16-bit result costs an s_nop: hipcc's inline-asm hazard rule)
    // (BITS, the recording kernels: the relu bits as well -- else the bit extraction of an even tile sinks into the next step)
    if (t == 10) {
        if constexpr (BITS) asm volatile("" : "+v"(d0), "+v"(d1), "+v"(mask));
        else asm volatile("" : "+v"(d0), "+v"(d1));
    }
}

template <class Pol>
struct APipe {                                  // fragments 0..PF-2 of the chunk about to be consumed
    static constexpr int N = Pol::LDS_PREFETCH - 1;
    typename Pol::frag f[N];
    f32x16 bias;                                // bias rows of the upcoming output tile (its initial accumulator)
    DEVI void prime(const char *ch, const float *bias_tile) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = Pol::lds_frag(ch, i, lane);
        bias = bias_acc(bias_tile, 0, lane >> 5);
    }
};

// fragment `idx` of the stream "NF fragments of chunk ch, then chunk chn from its fragment 0"
template <class Pol, int NF>
DEVI typename Pol::frag stream_frag(const char *ch, const char *chn, int idx, int lane) {
    return idx < NF ? Pol::lds_frag(ch, idx, lane) : Pol::lds_frag(chn, idx - NF, lane);
}

// Chunk to start copying in the middle of a step.  The source is an OFFSET from the base of one buffer resource that the
// ring builds once per kernel (RingState::start): issuing a piece then costs an s_add for the scalar offset and the M0
// set-up instead of rebuilding a 128-bit resource from a 64-bit pointer per piece (~40 SALU instructions per ring step,
// a quarter of its instruction count: round-2 ISA census).
struct DmaJob {
    bool on;
    unsigned soff;                              // byte offset of the chunk from the ring's source base
    char *dst;
    __amdgpu_buffer_rsrc_t rs;
    int wvu;                                    // the issuing wave's index, held in an SGPR by the ring (hipcc otherwise
                                                // re-derives it from threadIdx with v_readfirstlane + shifts in every step)
    u32x4 rsa;                                  // the same resource as four dwords (DmaRing<.., ASM = true>)
};

// Work folded into the MFMA shadows of a ring step ("Post" objects): at(t) is called right after MFMA t of the
// step (t is a constant after unrolling) when the layer has >= 16 k-steps, all() before the first MFMA otherwise.

// relu + repack of the pending output tile into its two B fragments (k-steps 0..7: two elements each).
// d0/d1 may be the src[KS-2], src[KS-1] of the running step (layer boundary): complete before k-step KS-2.
template <class Pol>
struct PackPost {
    const f32x16 &pend;
    typename Pol::frag &d0, &d1;
    unsigned mask;                              // relu bits of the pending tile (bit r: element r > 0)
    DEVI PackPost(const f32x16 &p, typename Pol::frag &a, typename Pol::frag &b) : pend(p), d0(a), d1(b), mask(0) {}
    unsigned w = 0;                             // the pair rounded in the previous k-step (pack_pipe)
    DEVI void at(int t) {
        if constexpr (Pol::ELEM_BYTES == 2) pack_pipe<Pol, false>(t, pend, d0, d1, w, mask);
        else {                                  // f32: no packed converts, nothing to pipeline
            if (t == 0) pack_elems<Pol, 0, 2>(pend, d0, d1, mask);
            if (t == 1) pack_elems<Pol, 2, 2>(pend, d0, d1, mask);
            if (t == 2) pack_elems<Pol, 4, 2>(pend, d0, d1, mask);
            if (t == 3) pack_elems<Pol, 6, 2>(pend, d0, d1, mask);
            if (t == 4) pack_elems<Pol, 8, 2>(pend, d0, d1, mask);
            if (t == 5) pack_elems<Pol, 10, 2>(pend, d0, d1, mask);
            if (t == 6) pack_elems<Pol, 12, 2>(pend, d0, d1, mask);
            if (t == 7) pack_elems<Pol, 14, 2>(pend, d0, d1, mask);
        }
    }
    DEVI void all() { pack_elems<Pol, 0, 16>(pend, d0, d1, mask); }
    DEVI void finish() {}                       // (behind the last k-step of the ring step)
};

// EXPERIMENT (-DBHN_PRIO_MODE=n, round 3): issue priority of the two waves of a SIMD inside a ring step.  With equal
// priority the older wave of a SIMD wins every arbitration: per-step stamps show it finishing its MFMAs ~500 cycles before
// its partner and then waiting at the barrier.  1: static s_setprio 1 for waves NW/2.. (set once, RingState::start);
// 2 / 3 / 4: the two halves swap priority every 1 / 2 / 4 k-steps.
#ifndef BHN_PRIO_MODE
#define BHN_PRIO_MODE 0
#endif
DEVI void prio_flip(int t, int wvu) {
    if constexpr (BHN_PRIO_MODE >= 2) {
        constexpr int P = BHN_PRIO_MODE == 2 ? 1 : BHN_PRIO_MODE == 3 ? 2 : 4;
        if (t % P == 0) {
            const bool up = ((t / P) & 1) != 0;
            if (wvu >= 4) { if (up) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            else { if (up) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
        }
    }
}

// One output tile of a hidden / output / delta-chain layer: acc = ap.bias + sum_ks A[ks] . src[ks] (+ the enc
// block when with_enc), A streamed from the ring chunk `ch`; reads the head of the next chunk `chn` and the bias
// rows `bias_next` of the next tile before returning (both consumed after the barrier).
// NFR: fragments of a chunk the step streams -- KS + 2 (hidden fragments + the encoded-input block), or KS for chunk
// sequences that never use the encoded-input block (the transposed image of the delta chain: `with_enc` must be false)
// ZB: every tile of the sequence starts from a ZERO accumulator (the delta chain): no bias rows are read and ap.bias -- sixteen
// registers that would hold zeros across the step boundary -- is not used
template <int W, class Pol, class RG, class Post, int NFR = W / 16 + 2, bool ZB = false>
// `encw` (NFR == KS with an encoded-input block, i.e. the forward kernels whose ring copies only the hidden fragments of a chunk):
// the tile's two encoded-input weight fragments in the RESIDENT block the kernel filled at its start (EncBlock below)
DEVI f32x16 ring_step(const char *ch, const char *chn, APipe<Pol> &ap, const typename Pol::frag (&src)[W / 16],
                      const typename Pol::frag (&enc)[2], bool with_enc, const float *bias_next, Post &post, DmaJob dma,
                      int dbg = 0, const char *encw = nullptr) {
    const int lane = threadIdx.x & 63;
    constexpr int KS = W / 16, NF = NFR, PF = Pol::LDS_PREFETCH;
    static_assert(NF == KS + 2 || (NF == KS && KS >= PF - 1), "chunk fragments (the prefetch reaches PF - 1 fragments into the next chunk)");
    typename Pol::frag ew0 = Pol::zero(), ew1 = Pol::zero();
    if constexpr (NF == KS) {
        if (with_enc) { ew0 = Pol::lds_frag(encw, 0, lane); ew1 = Pol::lds_frag(encw, 1, lane); }     // (read here: landed long before the last k-step)
    }
    typename Pol::frag a[PF];
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) a[i] = ap.f[i];
    f32x16 acc;
    if constexpr (ZB) { const f32x16 z = {}; acc = z; } else acc = ap.bias;
    const bool do_post = !(dbg & 2), do_mma = !(dbg & 1);
    if (do_post && KS < 16) post.all();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        prio_flip(t, dma.wvu);
        a[(t + PF - 1) % PF] = stream_frag<Pol, NF>(ch, chn, t + PF - 1, lane);
        if (do_mma) acc = Pol::mma(a[t % PF], src[t], acc);
        if (do_post && KS >= 16) post.at(t);
        if (t == (KS >= 16 ? 9 : 0) && dma.on) RG::issue(dma);
        if constexpr (!ZB) { if (t == (KS >= 16 ? 13 : 0)) ap.bias = bias_acc(bias_next, 0, lane >> 5); }   // next tile's bias, before the barrier
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = KS; t < NF; ++t) {
        a[(t + PF - 1) % PF] = stream_frag<Pol, NF>(ch, chn, t + PF - 1, lane);
        if (with_enc && do_mma) acc = Pol::mma(a[t % PF], enc[t - KS], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NF == KS) {
        if (with_enc && do_mma) { acc = Pol::mma(ew0, enc[0], acc); acc = Pol::mma(ew1, enc[1], acc); }
    }
    if (do_post) post.finish();
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) ap.f[i] = a[(NF + i) % PF];
    return acc;
}

// Resident block of encoded-input weight fragments (forward kernels, bf16, KS >= 8, weight ring -- not the resident image): a chunk
// of the packed forward image is KS hidden fragments + the 2-fragment encoded-input block, which only the ONE hidden layer with a
// skip input (network.py:59-61: at most one per depth 2..8) and, at odd depths, the output layer use.  The ring copies the KS
// hidden fragments only (two instead of three DMA issues per wave and step, 11 % fewer L2 -> LDS bytes; the delta chain got 2 %
// from the same cut); the 2 x (MT + 1) encoded-input fragments are copied ONCE per workgroup into this block:
// fragments 2 m, 2 m + 1 = tile m of the hidden skip layer, 2 MT, 2 MT + 1 = the output layer.
template <int W, class Pol>
struct EncBlock {
    static constexpr int KS = W / 16, MT = W / 32, CB = (KS + 2) * Pol::FRAG_BYTES;
    static constexpr int BYTES = 2 * (MT + 1) * Pol::FRAG_BYTES;
    static constexpr bool ON = Pol::ELEM_BYTES == 2 && KS >= 8;
    static DEVI void fill(char *blk, const char *fwd_image, int depth, int skip_mask) {      // all threads, before the first barrier
        int ls = 0;
        for (int l = 1; l < depth; ++l) if ((skip_mask >> l) & 1) { ls = l; break; }
        constexpr int VPT = 2 * Pol::FRAG_BYTES / 16;                                          // 16-byte vectors per tile
        for (int v = threadIdx.x; v < (MT + 1) * VPT; v += blockDim.x) {
            const int m = v / VPT, r = v - m * VPT;
            const int chunk = m < MT ? 1 + (ls - 1) * MT + m : 1 + (depth - 1) * MT;           // (ls == 0: never read)
            u32x4 val = {0u, 0u, 0u, 0u};
            if (m == MT || ls > 0) val = reinterpret_cast<const u32x4 *>(fwd_image + (size_t)chunk * CB + KS * Pol::FRAG_BYTES)[r];
            reinterpret_cast<u32x4 *>(blk)[v] = val;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// LDS-DMA weight ring: chunk c+2 is copied global -> LDS (buffer_load_dwordx4 ... lds, no registers)
// while chunk c is consumed; three buffers.  Every wave issues exactly PPW 1-KiB pieces per chunk
// (tail waves re-issue the last piece) so that the counted vmcnt below is the same for all waves.
// ---------------------------------------------------------------------------------------------
// One wave copies 1 KiB global -> LDS without registers: lane i's 16 bytes land at dst + 16 i (src, dst wave-
// uniform).  The MUBUF form (buffer_load_dwordx4 ... lds) is used on purpose: the compiler's waitcnt pass books
// the FLAT form (global_load_lds_dwordx4) as a pending flat access and then degrades every later LDS wait to
// lgkmcnt(0), which serialises the software-pipelined A-fragment reads of ring_step.
// STREAM: cache policy `nt` for bytes that one CU reads exactly once (the tape in the dW kernel); never for the
// weight ring, which every CU re-reads from L2.
template <bool STREAM = false>
DEVI void dma_1k(const char *src, char *dst) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(src);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void *us = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(us, 0, 1 << 20, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)dst, 16, (int)(threadIdx.x & 63) * 16, 0, 0, STREAM ? 2 : 0);
}

// The same copy as inline asm, for kernels that read the DMA'd LDS image with ds_read_b64_tr_b16 (dW kernel): the
// transposed-read builtin carries no memory operand, so the compiler's waitcnt pass makes it wait for EVERY LDS-DMA
// it knows to be in flight -- s_waitcnt vmcnt(0) in front of the first read of each group, i.e. no prefetch at all
// (measured: dW kernel 5.3 -> 7.5 ms).  Issued from inline asm the DMA is invisible to that pass; its completion is
// ordered by the kernel's own counted vmcnt waits + barrier, exactly as for the builtin form.  M0 = LDS byte address
// of the wave's first lane; M0 is declared clobbered (the builtin LDS-DMA form and movrel / gpr_idx indexing use M0 too:
// without the clobber LLVM may hoist or merge its own M0 initialisations across this statement).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (only: "inline asm clobber list contains reserved registers: M0")
template <int POLICY = 0>      // 0 default, 1 nt (bytes one CU reads once from HBM), 2 sc1 (bytes another CU has just written)
DEVI void dma_1k_asm(const char *src, char *dst) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(src);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const u32x4 rs = {lo, hi & 0xffffu, 1u << 20, 0x00020000u};
    // LDS byte address = low 32 bits of the flat address (the shared aperture base sits in the high half); no
    // addrspacecast: its null check (v_cmp against src_shared_base) fails hipcc 7.2's machine verifier in some contexts
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(dst));
    const unsigned voff = (threadIdx.x & 63) * 16;
    if constexpr (POLICY == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(m), "v"(voff), "s"(rs) : "memory", "m0");
    else if constexpr (POLICY == 2)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds" ::"s"(m), "v"(voff), "s"(rs) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m), "v"(voff), "s"(rs) : "memory", "m0");
}
#pragma clang diagnostic pop

// ASM: the pieces are issued from inline asm (dma_1k_asm's reason: kernels that also read LDS with ds_read_b64_tr_b16 -- the
// builtin carries no memory operand and hipcc's waitcnt pass then drains EVERY LDS-DMA it knows about in front of each such read)
template <int CHUNK_BYTES, int NWAVES, bool ASM_ = false>
struct DmaRing {
    static constexpr bool ASM = ASM_;
    static constexpr int NPIECE = CHUNK_BYTES / 1024;
    static constexpr int PPW = (NPIECE + NWAVES - 1) / NWAVES;
    static_assert(CHUNK_BYTES % 1024 == 0, "chunks are whole KiB");
    // buffer resource over the whole packed-weight buffer from `base` on (raw buffer, no range limit that matters)
    static DEVI __amdgpu_buffer_rsrc_t resource(const char *base) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *us = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(us, 0, 0x7fffffff, 0x00020000);
    }
    static DEVI u32x4 resource_raw(const char *base) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return u32x4{lo, hi & 0xffffu, 0x7fffffffu, 0x00020000u};
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (only: "inline asm clobber list contains reserved registers: M0")
    static DEVI void issue(const DmaJob &j) {
        const int wvu = j.wvu;
        const int voff = (int)(threadIdx.x & 63) * 16;
        [[maybe_unused]] const unsigned m_base = ASM ? __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(j.dst)) : 0u;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int piece = wvu + NWAVES * i;
            piece = piece < NPIECE ? piece : NPIECE - 1;        // tail waves re-issue the last piece (equal vmcnt for all waves)
            if constexpr (ASM) {
                const unsigned m = m_base + (unsigned)piece * 1024u, so = j.soff + (unsigned)piece * 1024u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(voff), "s"(j.rsa), "s"(so) : "memory", "m0");
            } else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(j.rs, (__attribute__((address_space(3))) void *)(j.dst + piece * 1024), 16, voff,
                                                         (int)(j.soff + piece * 1024), 0, 0);
        }
    }
#pragma clang diagnostic pop
    // vmcnt retires in issue order and counts loads, stores and LDS-DMA alike: a chunk has landed once at
    // most the operations issued AFTER its last piece are pending.  A smaller count than the true one is
    // always safe (it only waits longer).
    template <int YOUNGER>
    static DEVI void wait_younger() {   // returns when at most YOUNGER of this wave's vector-memory ops are pending
        static_assert(YOUNGER >= 0 && YOUNGER <= 63, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    }
};

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain vmcnt, so
// global stores (tape tiles) and loads (weight prefetch) stay in flight across it.
DEVI void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Ring bookkeeping of the pipelined steps: chunk c is consumed from buffer `cur` while chunks c+1 .. c+DIST-1
// are resident or in flight; step_end() waits for this wave's pieces of chunk c+2 (NOT c+1: the A-fragment
// prefetch of the next step reads chunk c+2 before that step's barrier) and synchronises the workgroup.
// ---------------------------------------------------------------------------------------------
// CB: bytes between consecutive chunks of the packed image (the source stride).  RG copies RG::NPIECE KiB of each chunk and
// the ring buffers are RG::NPIECE KiB apart -- normally the whole chunk; the delta chain leaves out the two encoded-input
// fragments its transposed image never uses (16 instead of 18 pieces at width 256: two instead of three DMA issues per wave
// and step).
template <class RG, int CB, int DIST, bool LAG, int MT = 1, bool STAMPS = false>
struct RingState {
    static constexpr int CBL = RG::NPIECE * 1024;          // bytes copied per chunk = stride of the ring buffers in LDS
    static_assert(CBL <= CB, "the ring copies at most a whole chunk");
    // LAG (measured, not used): the second wave of every SIMD (waves NWAVES/2..) consumes the ring ONE STEP BEHIND
    // the first, so that the two waves of a SIMD are never in their per-tile VALU phases at the same time.  Costs one
    // more resident chunk and one idle step per wave; 5 % slower in the render kernel.
    static constexpr int NB = DIST + (LAG ? 2 : 1);
    static_assert(DIST >= 2, "ring geometry");
    __host__ __device__ static constexpr size_t lds_bytes(int) { return (size_t)NB * CBL; }
    char *ring;
    // chunk sequence of one tile, consumed cyclically: NCA forward chunks (img_a, in order), then the transposed
    // chunks of the delta chain (img_b): hidden layers nlb .. 1, MT chunks each, stored layer-major ascending
    const char *img_a;
    unsigned off_b;         // byte offset of the transposed image from img_a (both live in the packed-weight buffer)
    __amdgpu_buffer_rsrc_t rs;
    u32x4 rsa;
    int NC, NCA, nlb, cur, issue_c, dbg, lag;
    long long *ts;          // STAMPS (measurement builds): per-step time stamps (compute done, barrier passed)
    static DEVI int wrap(int i) { return i < 0 ? i + NB : (i >= NB ? i - NB : i); }
    // The buffer offset is made opaque to the compiler: when the number of ring buffers divides the steps of a layer it
    // proves the ring position of every unrolled step, folds it into the LDS address of each fragment read and, LDS being
    // larger than the 16-bit offset field of ds_read, pays one v_add per read (16 per step, round-2 ISA census).  From
    // an SGPR base it is one v_add per step and immediate offsets.
    static DEVI int opaque(int v) { asm volatile("" : "+s"(v)); return v; }
    // Without LAG the three buffer offsets a step needs (consumed, next, the one just freed) are carried as byte offsets
    // and rotated at the step end (3 SALU) instead of being derived from `cur` each time (three wrap()s and multiplies:
    // ~25 of the 140-190 instructions of a ring step, round-2 ISA census).
    int o_cur, o_nxt, o_prv, wvu;
    DEVI const char *ch() const { return ring + opaque(LAG ? wrap(cur - lag) * CBL : o_cur); }
    DEVI const char *chn() const { return ring + opaque(LAG ? wrap(cur - lag + 1) * CBL : o_nxt); }
    DEVI unsigned next_src() {
        unsigned src;
        if (issue_c < NCA) src = (unsigned)issue_c * CB;
        else {
            const int i = issue_c - NCA;
            src = off_b + (unsigned)((nlb - 1 - i / MT) * MT + i % MT) * CB;
        }
        issue_c = (issue_c + 1 == NC) ? 0 : issue_c + 1;
        return src;
    }
    DEVI DmaJob job() {
        if (dbg & 4) return DmaJob{false, 0u, nullptr, rs, wvu, rsa};
        return DmaJob{true, next_src(), ring + (LAG ? wrap(cur - 2) * CBL : o_prv), rs, wvu, rsa};
    }
    // STORES: global stores this wave is GUARANTEED to have issued after the DMA pieces of chunk c+2 (issued in the
    // middle of step c-2) other than the two younger chunks: vmcnt retires in order and counts stores, so they may
    // stay in flight across the wait.  0 is always safe (it only waits for more).
    template <int STORES = 0>
    DEVI void step_end() {
        long long t1 = 0;
        if (STAMPS && ts) t1 = __builtin_readcyclecounter();
        if (!(dbg & 4)) RG::template wait_younger<RG::PPW * (DIST - 2) + STORES>();
        if (!(dbg & 8)) lds_barrier();
        if (STAMPS && ts) {
            const long long t3 = __builtin_readcyclecounter();
            if ((threadIdx.x & 63) == 0) { ts[0] = t1; ts[1] = t3; }
            ts += 2;
        }
        if constexpr (LAG) cur = (cur == NB - 1) ? 0 : cur + 1;
        else {
            o_prv = o_cur;
            o_cur = o_nxt;
            o_nxt = (o_nxt == (NB - 1) * CBL) ? 0 : o_nxt + CBL;
        }
    }
    DEVI void idle_step() {      // a step in which this wave consumes nothing (lagging waves: first; the others: last)
        const DmaJob j = job();
        if (j.on) RG::issue(j);
        step_end();
    }
    DEVI void start(char *ring_, const char *a_, int nca, const char *b_, int nlb_, int dbg_, int lag_) {
        ring = ring_; img_a = a_; NCA = nca; nlb = nlb_; NC = nca + nlb_ * MT;
        off_b = b_ ? (unsigned)(b_ - a_) : 0u;                  // (the transposed image follows the forward image)
        rs = RG::resource(a_);
        rsa = RG::resource_raw(a_);
        dbg = dbg_; cur = 0; issue_c = 0; lag = LAG ? lag_ : 0; ts = nullptr;
        o_cur = 0; o_nxt = CBL; o_prv = (NB - 1) * CBL;
        wvu = opaque(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
        if constexpr (BHN_PRIO_MODE == 1) { if (wvu >= 4) __builtin_amdgcn_s_setprio(1); }
#pragma unroll
        for (int j = 0; j < DIST; ++j) RG::issue(DmaJob{true, next_src(), ring + j * CBL, rs, wvu, rsa});
        RG::template wait_younger<RG::PPW * (DIST - 2)>();
        lds_barrier();
    }
};

// ---------------------------------------------------------------------------------------------
// The same interface with the WHOLE chunk sequence resident in LDS (small networks: 4x128 bf16 is 14 chunks = 140 KB): copied
// once at kernel start, no DMA, no counted waits and -- the point -- NO per-chunk workgroup barrier: the eight waves run
// their tiles independently instead of in lockstep (at width 128 a chunk is 10 MFMAs per wave: the barrier cadence, the DMA
// issue and the waits were a third of the kernel).  Each wave walks the chunks with its own cursor.
// ---------------------------------------------------------------------------------------------
template <class RG, int CB, int MT = 1>
struct ResidentRing {
    static constexpr int NB = 0;                 // (no ring buffers: lds_bytes(nc) is the footprint)
    char *ring;
    int NC, dbg, lag, o_cur, o_nxt, wvu;
    long long *ts;
    static DEVI int opaque(int v) { asm volatile("" : "+s"(v)); return v; }
    __host__ __device__ static constexpr size_t lds_bytes(int nc) { return (size_t)nc * CB; }
    DEVI const char *ch() const { return ring + opaque(o_cur); }
    DEVI const char *chn() const { return ring + opaque(o_nxt); }
    DEVI DmaJob job() { return DmaJob{false, 0u, nullptr, __amdgpu_buffer_rsrc_t(), wvu, u32x4{0u, 0u, 0u, 0u}}; }
    template <int STORES = 0>
    DEVI void step_end() {
        o_cur = o_nxt;
        o_nxt = (o_nxt + CB == NC * CB) ? 0 : o_nxt + CB;
    }
    DEVI void idle_step() {}
    // chunk j of the consumption order: forward chunks 0..nca-1 of img_a, then the transposed chunks of hidden layers
    // nlb .. 1 (stored layer-major ascending in img_b), MT each -- RingState::next_src's order
    DEVI void start(char *ring_, const char *a_, int nca, const char *b_, int nlb_, int dbg_, int) {
        ring = ring_; NC = nca + nlb_ * MT; dbg = dbg_; lag = 0; ts = nullptr;
        o_cur = 0; o_nxt = NC > 1 ? CB : 0;
        wvu = opaque(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
        // the whole sequence by LDS-DMA, every piece in flight at once (round 5: through registers -- 12 dependent load / store
        // rounds of 512 threads for a 100-KB image -- the copy took ~10 us per workgroup, a third of the training forward of a
        // config-5-sized problem, where a workgroup has only five or six tiles to spread it over)
        constexpr int PPC = CB / 1024;           // 1-KiB pieces per chunk
        const int nw = (int)(blockDim.x >> 6);
        for (int pc = wvu; pc < NC * PPC; pc += nw) {
            const int j = pc / PPC, r = pc - j * PPC;
            const char *src;
            if (j < nca) src = a_ + (size_t)j * CB;
            else { const int i = j - nca; src = b_ + (size_t)((nlb_ - 1 - i / MT) * MT + i % MT) * CB; }
            dma_1k<false>(src + r * 1024, ring + (size_t)pc * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
};

// layer 0 (chunk 0: fragment 2m+ks, B = enc[ks]); tile m-1 is finished (l0.tile: relu + pack into act, and in
// the training kernels mask + tape emission) behind the two MFMAs of tile m; the last tile is left pending for
// the first hidden step.
template <class Pol>
struct PackTile0 {
    DEVI void tile(int, const f32x16 &acc, typename Pol::frag &d0, typename Pol::frag &d1) {
        unsigned mask = 0;
        pack_elems<Pol, 0, 16>(acc, d0, d1, mask);
    }
};

template <int W, class Pol, class RG, int STORES, class RS, class L0, int NFR = W / 16 + 2>
DEVI void layer0_step(RS &rs, APipe<Pol> &ap, const typename Pol::frag (&enc)[2], typename Pol::frag (&act)[W / 16],
                      const float *bias_lds, int h, f32x16 &pend, L0 &l0) {
    const int lane = threadIdx.x & 63;
    constexpr int KS = W / 16, MT = W / 32, NF = NFR, PF = Pol::LDS_PREFETCH;      // every chunk is a stream of NFR (KS + 2, or KS) fragments;
    const char *ch = rs.ch(), *chn = rs.chn();                                     // layer 0 uses the first KS of them
    const DmaJob dj = rs.job();
    typename Pol::frag a[PF];
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) a[i] = ap.f[i];
    f32x16 prev = {};
    f32x16 acc = ap.bias;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f32x16 nb = bias_acc(bias_lds + 32 * (m + 1), 0, h);       // tile m+1 (m = MT-1: first tile of layer 1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int t = 2 * m + ks;
            a[(t + PF - 1) % PF] = stream_frag<Pol, NF>(ch, chn, t + PF - 1, lane);
            acc = Pol::mma(a[t % PF], enc[ks], acc);
        }
        if (m > 0) l0.tile(m - 1, prev, act[2 * (m > 0 ? m - 1 : 0)], act[2 * (m > 0 ? m - 1 : 0) + 1]);
        if (m == (MT > 1 ? 1 : 0) && dj.on) RG::issue(dj);
        __builtin_amdgcn_sched_barrier(0);
        prev = acc;
        acc = nb;
    }
#pragma unroll
    for (int t = KS; t < NF; ++t) a[(t + PF - 1) % PF] = stream_frag<Pol, NF>(ch, chn, t + PF - 1, lane);
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) ap.f[i] = a[(NF + i) % PF];
    ap.bias = acc;
    pend = prev;
    rs.template step_end<STORES>();
}

// hidden layer l: src -> dst.  On entry `pend` is the last tile of the previous layer (destination src[KS-2],
// src[KS-1]); on exit it is this layer's last tile (destination dst[KS-2], dst[KS-1]).  bl = this layer's bias
// rows; the rows of the next layer (or of the output layer) follow them at bl + W.
template <int W, class Pol, class RG, class RS, int NFR = W / 16 + 2>
DEVI void hidden_layer(RS &rs, APipe<Pol> &ap, typename Pol::frag (&src)[W / 16], typename Pol::frag (&dst)[W / 16],
                       const typename Pol::frag (&enc)[2], bool sk, const float *bl, f32x16 &pend, const char *encblk) {
    constexpr int KS = W / 16, MT = W / 32;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const char *ch = rs.ch(), *chn = rs.chn();
        const DmaJob dj = rs.job();
        PackPost<Pol> post(pend, m == 0 ? src[KS - 2] : dst[2 * (m > 0 ? m - 1 : 0)], m == 0 ? src[KS - 1] : dst[2 * (m > 0 ? m - 1 : 0) + 1]);
        const char *encw = nullptr;                     // (only the KS-fragment streams read the resident encoded-input block)
        if constexpr (NFR == KS) encw = encblk + 2 * m * Pol::FRAG_BYTES;
        const f32x16 acc = ring_step<W, Pol, RG, PackPost<Pol>, NFR>(ch, chn, ap, src, enc, sk, bl + 32 * (m + 1), post, dj, rs.dbg, encw);
        rs.step_end();
        pend = acc;
    }
}

// wave-level sum over the 32 lanes of each half (lanes 0-31 and 32-63 independently)
DEVI float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Render epilogue: x J g^2 dtau Sigma and the sum over the ray (network.py:415-419, kgeo.py:621).
// A 32-point wave tile may hold pieces ("segments") of several rays, and a ray may run over several wave tiles.  Round 1
// added every segment to its pixel with a float atomic: more than two adds per pixel (rays of > 64 samples, masked
// domains) made images -- and with them gradients -- depend on the arrival order.  Now every wave leaves its segment
// sums in LDS, and after ONE workgroup barrier the lane that owns the first segment of a ray within the workgroup tile
// (32 NW consecutive points) adds the ray's following segments in wave order and issues a single atomic.  A pixel then
// receives one add per workgroup tile its ray touches, ceil((G - 1) / (32 NW)) + 1: two for G <= 32 NW + 1 (257 samples
// in bf16 mode, 129 in f32 mode) -- two commutative adds onto zero, bitwise reproducible.
// LDS: SEG_BYTES<NW> at `lds`; rewritten a whole tile (>= 3 barriers) later, so one barrier suffices.
// ---------------------------------------------------------------------------------------------
template <int NW>
struct RaySum {
    static constexpr int SMAX = 4;                                   // Stokes planes (bhn_geom.S <= 4)
    static constexpr int BYTES = NW * 32 * 4 + NW * SMAX * 32 * 4 + NW * 4;                      // at Sx = 4
    // the scratch is sized for the Stokes planes actually rendered (the f32 8x256 training forward has 1.9 KB left)
    __host__ __device__ static constexpr int bytes(int sx) { return NW * 32 * 4 + NW * sx * 32 * 4 + NW * 4; }
    // Dense layouts whose rays start on a 32-point boundary and are at most 64 samples long (G = 32, 64), or at most 33
    // samples anywhere: a pixel gets at most two adds from per-tile atomics as well -- round 1's epilogue, kept for these
    // (BASELINE config 2 is G = 64): the LDS combine and its barrier cost 3 % of the inference forward.
    // the point's contribution to every Stokes plane, e w_s (network.py:417), loaded ONCE in front of the segment loop (round 5:
    // inside it -- one dependent global load per (segment, plane) -- a compacted polarised ray set, three or four ray segments
    // per 32-point group and three planes, spent 4 of the 14.5 us of a width-128 tile waiting for them)
    static DEVI void weighted(const FusedArgs &a, long long p, bool on, float e, float w0, bool have_w0, float (&we)[SMAX]) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            float w = 0.f;
            if (on && e != 0.f && s < a.Sx) w = (s == 0 && have_w0) ? w0 : a.w[(long long)s * a.P + p];
            we[s] = w * e;
        }
    }
    static DEVI void direct(const FusedArgs &a, int b, long long p, bool inb, float e, float w0, bool have_w0) {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const long long ray = inb ? (a.ray_idx ? (long long)a.ray_idx[p] : (long long)a.fd_G.div((unsigned)p)) : -1;
        float we[SMAX];
        weighted(a, p, h == 0 && inb, e, w0, have_w0, we);
        unsigned long long rem = __ballot(h == 0 && inb);
        while (rem) {
            const int first = __ffsll((long long)rem) - 1;
            const long long r0 = __shfl(ray, first, 64);
            const bool mine = (h == 0) && inb && (ray == r0);
#pragma unroll
            for (int s = 0; s < SMAX; ++s)
                if (s < a.Sx) {
                    float v = mine ? we[s] : 0.f;
                    v = half_wave_sum(v);
                    if (lane == first) atomicAdd(a.images + ((long long)b * a.Sx + s) * a.R + r0, v);
                }
            rem &= ~__ballot(mine);
        }
    }
    // Segment sums of the 32-point tile of (virtual) wave vw -> LDS; with ray_direct: straight to the pixels.
    static DEVI void put(const FusedArgs &a, char *lds, int vw, long long p, bool inb, float e, float w0, bool have_w0, int b = -1) {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        int *seg_ray = reinterpret_cast<int *>(lds);                                  // [NW][32]
        float *seg_val = reinterpret_cast<float *>(lds + NW * 32 * 4);                // [NW][Sx][32]
        int *seg_n = reinterpret_cast<int *>(lds + NW * 32 * 4 + NW * a.Sx * 32 * 4); // [NW]
        const long long ray = inb ? (a.ray_idx ? (long long)a.ray_idx[p] : (long long)a.fd_G.div((unsigned)p)) : -1;
        float we[SMAX];
        weighted(a, p, h == 0 && inb, e, w0, have_w0, we);
        unsigned long long rem = __ballot(h == 0 && inb);
        int k = 0;
        while (rem) {
            const int first = __ffsll((long long)rem) - 1;
            const long long r0 = __shfl(ray, first, 64);
            const bool mine = (h == 0) && inb && (ray == r0);
#pragma unroll
            for (int s = 0; s < SMAX; ++s)
                if (s < a.Sx) {
                    float v = mine ? we[s] : 0.f;
                    v = half_wave_sum(v);
                    if (lane == first) {
                        if (a.ray_direct) atomicAdd(a.images + ((long long)b * a.Sx + s) * a.R + r0, v);
                        else seg_val[(vw * a.Sx + s) * 32 + k] = v;
                    }
                }
            if (lane == first && !a.ray_direct) seg_ray[vw * 32 + k] = (int)r0;
            ++k;
            rem &= ~__ballot(mine);
        }
        if (lane == 0 && !a.ray_direct) seg_n[vw] = k;
    }
    // after a workgroup barrier: the lane that owns the first segment of a ray within the workgroup tile adds the ray's
    // following segments in (virtual) wave order and issues one atomic
    static DEVI void combine(const FusedArgs &a, char *lds, int vw, int b) {
        const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
        const int *seg_ray = reinterpret_cast<const int *>(lds);
        const float *seg_val = reinterpret_cast<const float *>(lds + NW * 32 * 4);
        const int *seg_n = reinterpret_cast<const int *>(lds + NW * 32 * 4 + NW * a.Sx * 32 * 4);
        if (h == 0 && pl < seg_n[vw]) {
            const int r = seg_ray[vw * 32 + pl];
            bool owner = true;                      // the ray starts in this workgroup tile with this segment
            if (pl == 0 && vw > 0) {
                const int np = seg_n[vw - 1];
                owner = !(np > 0 && seg_ray[(vw - 1) * 32 + np - 1] == r);
            }
            if (owner) {
                for (int s = 0; s < a.Sx; ++s) {
                    float sum = seg_val[(vw * a.Sx + s) * 32 + pl];
                    if (pl == seg_n[vw] - 1) {      // the wave's last segment may continue in the following waves
                        for (int w2 = vw + 1; w2 < NW; ++w2) {
                            if (seg_n[w2] == 0 || seg_ray[w2 * 32] != r) break;
                            sum += seg_val[(w2 * a.Sx + s) * 32];
                            if (seg_n[w2] != 1) break;
                        }
                    }
                    atomicAdd(a.images + ((long long)b * a.Sx + s) * a.R + r, sum);
                }
            }
        }
    }
    static DEVI void run(const FusedArgs &a, char *lds, int b, long long p, bool inb, float e, float w0, bool have_w0) {
        if (a.ray_direct) return direct(a, b, p, inb, e, w0, have_w0);
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        put(a, lds, wv, p, inb, e, w0, have_w0);
        lds_barrier();
        combine(a, lds, wv, b);
    }
};

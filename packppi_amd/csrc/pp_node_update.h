// The matrix-pipe node update (split-f16 MFMA) as a device function + the kernel that wraps it.  Included by pp_node.hip (the
// launches of the node level) and by pp_edge_f16.hip (the fused layer-0 launch: node message + node update in one kernel).
#pragma once
#include "pp_internal.h"

// (x + pi) % (2 pi) - pi with torch.remainder semantics in fp32
__device__ __forceinline__ float wrap_pi(float x) {
    const float PIf = 3.14159274101257324f, TWO_PIf = 6.28318548202514648f;
    float y = x + PIf;
    float r = fmodf(y, TWO_PIf);
    if (r != 0.f && r < 0.f) r += TWO_PIf;
    return r - PIf;
}

// ==================================================================================================================
// k_node_update on the matrix pipe (split-f16 MFMA, fp32-level accuracy)
//
// A 512-thread workgroup owns a tile of 16 consecutive residues for the whole chain
//     S -> W_out -> LN(h_V + .) -> FFN 128 -> 512 -> 128 -> LN -> mask -> {W_A, W_C, points} x 2        (layers 0, 1)
//                                                             ... -> decoder -> reverse step -> embedding -> {W_A, W_C, points}   (layer 2)
// computed transposed, Y^T[feature][residue] = W[feature][k] X^T[k][residue], with v_mfma_f32_16x16x32_f16: wave w owns
// output-feature tile w (16 features) of every 128-wide layer and reads only its own weight rows (A operand, straight from
// global memory into registers) but all input features (B operand) from an LDS image [residue][feature] of the previous
// layer's output, stored as split f16 (hi, lo * 2^11) in two planes whose row stride (features + 16 halves) makes the
// 16-byte operand reads bank-conflict free.  D: lane (r = lane & 15, g = lane >> 4), register i <-> feature 16 w + 4 g + i
// of residue r.
//
// Arithmetic: x = hi + lo with hi = f16(x) (round to nearest) and lo' = f16((x - hi) * 2^11); a product is three MFMAs,
// Wh xh into one accumulator and Wh xl' + Wl' xh into a second one that is folded in with 2^-11 at the end.  The scaling keeps
// lo' a normal f16 number for every |x| >= 2^-25 (unscaled, lo is subnormal below |x| = 2^-3 and carries an absolute error
// of 3e-8); the dropped Wl xl term is 2^-22 relative.
//
// What bounds the kernel: every workgroup needs the layer's whole weight set (0.88 MB) through its CU's vector memory
// path (64 B/clk); N/16 workgroups run, so for one complex most CUs idle and the launch takes as long as ONE CU needs to
// stream 0.88 MB.  The stream is therefore decoupled from the dependent chain: each wave's slots are contiguous in global
// memory in consumption order and are fetched PP_NU_DEPTH stages ahead into a register ring, through barriers and
// LayerNorms (the compiler counts these ordinary loads; __syncthreads() waits for LDS traffic only), so the loads never
// stop while the chain (about 2 us of MFMA + LDS latency) runs underneath.
// ==================================================================================================================
typedef _Float16 nh8 __attribute__((ext_vector_type(8)));
typedef float nf4 __attribute__((ext_vector_type(4)));
typedef float nf2 __attribute__((ext_vector_type(2)));
typedef unsigned nu2 __attribute__((ext_vector_type(2)));

#ifndef PP_NU_DEPTH
#define PP_NU_DEPTH 8      // weight stages in flight per wave when a launch is one round of workgroups (<= one tile per CU)
#endif
#define PP_NU_DEPTH_MULTI 5   // ... when there are more tiles than CUs: 128 VGPRs or fewer, so that two workgroups share a CU
#define NU_S128 144        // halves per residue row of a 128-feature operand image
#define NU_S512 528        // ... of the 512-feature one
#define NU_S32 48          // ... of the 32-feature one (node embedding inputs)
#define NU_INV_LO (1.0f / PP_NU_LO_SCALE)

struct AOpN {
    nh8 hi, lo;
};

struct SmemU {
    _Float16 a_hi[16 * NU_S128], a_lo[16 * NU_S128];     // S, later h2
    _Float16 b_hi[16 * NU_S128], b_lo[16 * NU_S128];     // h1, later the next step's embedded h_V
    _Float16 c_hi[16 * NU_S512], c_lo[16 * NU_S512];     // FFN hidden; decoder activations in columns 0..127
    float stats[3][8][16][2];                            // LayerNorm partials (per wave: mean, centred sum of squares)
    float pts[16][48];                                   // local points of the tile
    _Float16 e_hi[16 * NU_S32], e_lo[16 * NU_S32];       // layer 2: dense inputs of the next step's node embedding (30 of 32)
    float fr[16][12];                                    // backbone frames of the tile's residues
    float par[NU_P_LAST_TOTAL];                          // parameter block (layers 0, 1 use the first NU_P_MID_TOTAL)
};

// per-step scalars and the next step's time embedding travel as kernel arguments (no device buffer to fill, so pp_score /
// pp_sample never wait for the stream)
struct StepScalars {
    float c_ode, w, c_drift, c_diff;
};
struct TimeEmb {
    float v[16];
};

struct NUpdArgs {
    int N;
    const float *rmask;          // [N]
    const int64_t *rtype;        // [N]
    const float *bb_sincos;      // [N][6]
    const float *sc_mask;        // [N][4]
    const uint8_t *m1pi, *m2pi;  // [N][4]
    const float *frames;         // [N][12]
    const float *embT;           // [51][128]
    const float *wstream, *params;
    const float *hV;             // h_V of the previous layer (input)
    float *hV_out;               // where the new h_V goes: the same buffer, or the context's alternate one for split launches
    const float *S, *msum;
    float *ptsN, *PAn, *PCn, *ptsE, *PAe, *PCe, *score;
    unsigned *sat;               // the context's sticky saturation word (bit 1: node kernels)
};
NUpdArgs pp_node_update_args(pp_ctx *c, int layer);      // pp_node.hip


template <bool LAST>
__device__ constexpr int nu_slot_waves(int s) {       // how many waves (0 .. n-1) own slot s; see pp_internal.h
    if (!LAST) return s < 52 ? 8 : 3;
    return s < 36 ? 8 : s < 40 ? 4 : s < 46 ? 1 : s < 55 ? 8 : 2;
}

// SPLIT launches of a middle layer (CL = 2 or 4 workgroups per tile, see k_node_update): workgroup q of a tile runs the common
// part (slots 0..35) and then only ITS 4 / CL of the four 128-wide projections, workgroup 0 the local points as well.  Logical
// slot s of such a workgroup -> slot of the packed stream, and the waves that own a logical slot.
template <int CL>
__device__ __forceinline__ int nu_phys_slot(int s, int q) {
    if constexpr (CL == 1) return s;
    constexpr int NPS = 16 / CL;                 // projection slots per workgroup
    return s < 36 ? s : s < 36 + NPS ? s + NPS * q : s + (16 - NPS);
}
template <bool LAST, int CL>
__device__ constexpr int nu_logical_waves(int s) {
    if (CL == 1) return nu_slot_waves<LAST>(s);
    return s < 36 + 16 / CL ? 8 : 3;
}
__device__ __forceinline__ void gload_N(const nh8 *__restrict__ wq, int slot, AOpN &a) {
    int off = slot * 128;                     // nh8 units per 2 KB slot
    asm volatile("" : "+s"(off));             // opaque: the fetch is issued where it is written, not hoisted to the top
    const nh8 *p = wq + off;
    a.hi = p[0];
    a.lo = p[64];
}
#define MFMA_N(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ void mm3(const AOpN &a, const nh8 &bh, const nh8 &bl, nf4 &cH, nf4 &cL) {
#ifdef PP_X_NU_NOMFMA        /* timing experiment (wrong results): the weight stream and the barriers without the matrix work */
    asm volatile("" ::"v"(a.hi), "v"(a.lo), "v"(bh), "v"(bl));
    return;
#endif
    cH = MFMA_N(a.hi, bh, cH);
    cL = MFMA_N(a.hi, bl, cL);
    cL = MFMA_N(a.lo, bh, cL);
}
__device__ __forceinline__ nf4 fold(const nf4 &cH, const nf4 &cL) {
    return nf4{fmaf(cL[0], NU_INV_LO, cH[0]), fmaf(cL[1], NU_INV_LO, cH[1]), fmaf(cL[2], NU_INV_LO, cH[2]), fmaf(cL[3], NU_INV_LO, cH[3])};
}
// B operand of k-step ks: features 32 ks + 8 g .. + 7 of residue r
__device__ __forceinline__ void ldB(const _Float16 *hi, const _Float16 *lo, int rowoff, int ks, nh8 &bh, nh8 &bl) {
    bh = *reinterpret_cast<const nh8 *>(hi + rowoff + 32 * ks);
    bl = *reinterpret_cast<const nh8 *>(lo + rowoff + 32 * ks);
}
// two fp32 values -> packed (hi, hi), (lo', lo').  Scalar round-to-nearest conversions + v_pack: gfx950's packed
// v_cvt_pk_f16_f32 (what a plain cast of a pair compiles to) is the instruction DESIGN.md section 4 found unreliable with more
// than one wave per SIMD.
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hp, unsigned &lp) {
    unsigned a, b, c, d;
    float fa, fb;
    PP_RANGE(x0) PP_RANGE(x1)
    asm("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(x0));
    asm("v_cvt_f16_f32 %0, %1" : "=v"(b) : "v"(x1));
    asm("v_cvt_f32_f16 %0, %1" : "=v"(fa) : "v"(a));
    asm("v_cvt_f32_f16 %0, %1" : "=v"(fb) : "v"(b));
    const float d0 = (x0 - fa) * PP_NU_LO_SCALE, d1 = (x1 - fb) * PP_NU_LO_SCALE;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(c) : "v"(d0));
    asm("v_cvt_f16_f32 %0, %1" : "=v"(d) : "v"(d1));
    asm("v_pack_b32_f16 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
    asm("v_pack_b32_f16 %0, %1, %2" : "=v"(lp) : "v"(c), "v"(d));
}
// four consecutive features of one residue into an operand image
__device__ __forceinline__ void publish4(_Float16 *hi, _Float16 *lo, int off, const nf4 &v) {
    nu2 h, l;
    unsigned h0, l0, h1, l1;
    split2(v[0], v[1], h0, l0);
    split2(v[2], v[3], h1, l1);
    h[0] = h0; h[1] = h1; l[0] = l0; l[1] = l1;
    *reinterpret_cast<nu2 *>(hi + off) = h;
    *reinterpret_cast<nu2 *>(lo + off) = l;
}
// hidden activations: ReLU, saturated at the f16 maximum (one v_med3)
// `satm` keeps the largest pre-clamp value this lane has seen (sticky saturation flag, pp_internal.h).  A NaN is NOT caught
// (v_max returns the other operand, v_med3 then yields a finite value): with finite weights (pp_plan_create checks them) and
// f16-range operands an fp32 accumulator cannot overflow (65504^2 x 512 << 3.4e38), so a NaN can only enter through the
// caller's batch tensors, which this flag is not about
__device__ __forceinline__ nf4 relu_sat(const nf4 &v, float &satm) {
    satm = __builtin_fmaxf(__builtin_fmaxf(satm, __builtin_fmaxf(v[0], v[1])), __builtin_fmaxf(v[2], v[3]));
    return nf4{__builtin_amdgcn_fmed3f(v[0], 0.f, 65504.f), __builtin_amdgcn_fmed3f(v[1], 0.f, 65504.f),
               __builtin_amdgcn_fmed3f(v[2], 0.f, 65504.f), __builtin_amdgcn_fmed3f(v[3], 0.f, 65504.f)};
}
__device__ __forceinline__ float xsum_g(float v) {       // sum over the four lane groups g (lanes r, r+16, r+32, r+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// LayerNorm over the 128 features of residue r (16 here in 4 lanes x 4 registers, the rest in the other waves): per-wave
// mean and centred sum of squares, the eight partials meet once in LDS and merge with Chan's update for equal counts.
__device__ __forceinline__ nf4 ln128(float (*st)[16][2], int wv, int r, int g, const nf4 &x, const nf4 &gain, const nf4 &beta) {
    const float mw = xsum_g((x[0] + x[1]) + (x[2] + x[3])) * (1.f / 16.f);
    const nf4 d = x - mw;
    const float qw = xsum_g(fmaf(d[0], d[0], d[1] * d[1]) + fmaf(d[2], d[2], d[3] * d[3]));
    if (g == 0) *reinterpret_cast<nf2 *>(st[wv][r]) = nf2{mw, qw};
    __syncthreads();
    float m8[8], msum = 0.f, qsum = 0.f;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const nf2 t = *reinterpret_cast<const nf2 *>(st[v][r]);
        m8[v] = t[0];
        msum += t[0];
        qsum += t[1];
    }
    const float mean = msum * 0.125f;
    float dm = 0.f;
#pragma unroll
    for (int v = 0; v < 8; v++) dm = fmaf(m8[v] - mean, m8[v] - mean, dm);
    const float var = fmaf(16.f, dm, qsum) * (1.f / 128.f);
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    return (x - mean) * rstd * gain + beta;
}

// stage k: fetch slot k + NU_ND (if this wave owns it), then BODY on the operands of slot k (AK).  The fetch condition is
// the slot's owner set only: a run-time condition such as embed_next here makes the compiler's vmcnt bookkeeping assume
// the path without the later fetches, and every wait after it drains the ring (the last step of a sampling run fetches 12
// slots per wave it never uses; s_endpgm waits for them).  The scheduling barrier
// keeps fetches and MFMAs in their stage; the empty asm on the accumulator keeps the (pure) MFMAs from sinking.
#define NSTAGE_IF(k, OWN, ACCV, BODY)                                                                                  \
    {                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if constexpr ((k) + NU_ND < NLOAD && !NU_X_NOLOAD) {                                                           \
            constexpr int nw_ = nu_logical_waves<LAST, CL>((k) + NU_ND);                                               \
            if (nw_ == 8 || wv < nw_)                                                                                  \
                gload_N(wq, nu_phys_slot<CL>((k) + NU_ND, clq), AR[((k) + NU_ND) % NU_NRING]);                         \
        }                                                                                                              \
        if (OWN) {                                                                                                     \
            const AOpN &AK = AR[(k) % NU_NRING];                                                                       \
            BODY;                                                                                                      \
            asm volatile("" ::"v"(ACCV[0]));                                                                           \
        }                                                                                                              \
    }
#define NSTAGE(k, ACCV, BODY) NSTAGE_IF(k, true, ACCV, BODY)
// four stages = one 16-feature tile over a 128-deep input held in bh / bl
#define NTILE4_IF(k0, OWN, cH, cL)                                     \
    NSTAGE_IF((k0) + 0, OWN, cH, mm3(AK, bh[0], bl[0], cH, cL))        \
    NSTAGE_IF((k0) + 1, OWN, cH, mm3(AK, bh[1], bl[1], cH, cL))        \
    NSTAGE_IF((k0) + 2, OWN, cH, mm3(AK, bh[2], bl[2], cH, cL))        \
    NSTAGE_IF((k0) + 3, OWN, cH, mm3(AK, bh[3], bl[3], cH, cL))
#define NTILE4(k0, cH, cL) NTILE4_IF(k0, true, cH, cL)
#define LDB4(HI, LO)                                                                  \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ks_++) ldB(HI, LO, r * NU_S128 + 8 * g, ks_, bh[ks_], bl[ks_]);
// FFN-out stage s: B operand of k-step s + 1 is requested first
#define FOSTAGE(s)                                                                                   \
    NSTAGE(20 + (s), cH, {                                                                           \
        if constexpr ((s) + 1 < 16) ldB(sm.c_hi, sm.c_lo, r * NU_S512 + 8 * g, (s) + 1, fh[((s) + 1) & 1], fl[((s) + 1) & 1]); \
        mm3(AK, fh[(s) & 1], fl[(s) & 1], cH, cL);                                                   \
    })

#ifdef PP_X_NU_NOLOAD        /* timing experiment (wrong results): no weight fetches after the prologue's */
#define NU_X_NOLOAD true
#else
#define NU_X_NOLOAD false
#endif
// CL > 1 (middle layers, when CL x tiles workgroups still fit the chip in one round): CL workgroups per 16-residue tile.  A
// launch lasts as long as ONE CU needs for its workgroup's weight stream, and most CUs idle (47 tiles at T1124); so the tile's
// common part (W_out, FFN, both LayerNorms: 36 slots) is computed redundantly by CL workgroups on CL CUs, and the 20 slots of
// projections behind it are dealt out among them -- no exchange between workgroups, identical arithmetic per output.
// TILE_R < 16 (the fused layer-0 launch of pp_edge_f16.hip): the workgroup owns TILE_R consecutive residues; operand rows
// TILE_R .. 15 mirror row 0 and store nothing.  Every residue's arithmetic is independent of its row: same bits.
// `bidx` = blockIdx.x of the plain kernel; `smem_raw` = the workgroup's dynamic LDS (sizeof(SmemU) bytes).
// `between` (the fused layer-0 launch): work that runs after this body has requested its S-independent inputs and started its
// weight stream, and before it reads S / msum -- the node message that PRODUCES them.  It may use the whole LDS block: nothing
// is staged before it returns.  The plain kernel passes NuNothing (S and msum are then requested first, as before: loads retire
// in order, and waiting for an input must not mean waiting for eight weight stages).
struct NuNothing {
    __device__ __forceinline__ void operator()() const {}
};
template <int MODE, int NU_ND, int CL = 1, int TILE_R = 16, class Between = NuNothing>
__device__ __forceinline__ void node_update_body(const NUpdArgs &A, float *chi, int step, int sde, const float *noise, int embed_next,
                                                 const StepScalars &sp, const TimeEmb &te_next, char *smem_raw, const int bidx,
                                                 const Between &between = Between()) {
    constexpr bool DEFER_S = !__is_same(Between, NuNothing);
    static_assert(CL == 1 || MODE == PP_NU_MID, "only the middle layers have a split form");
    static_assert(TILE_R == 16 || (MODE == PP_NU_MID && CL == 1), "short tiles: unsplit middle layers only");
#define NU_ROW(x) ((TILE_R == 16 || (x) < TILE_R) ? (x) : 0)
    constexpr int NU_NRING = NU_ND + 1;
    constexpr bool LAST = MODE != PP_NU_MID;
    constexpr int NSLOT = LAST ? PP_NU_SLOTS_LAST : PP_NU_SLOTS_MID;
    constexpr int NLOAD = MODE == PP_NU_SCORE ? 46 : CL > 1 ? 36 + 16 / CL + 4 : NSLOT;       // (logical) slots this instance ever fetches
    const int clq = CL > 1 ? (int)(bidx % CL) : 0;          // which of the tile's workgroups this is
    constexpr int NPAR = LAST ? NU_P_LAST_TOTAL : NU_P_MID_TOTAL;
    SmemU &sm = *reinterpret_cast<SmemU *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4, N = A.N, n0 = (int)(bidx / CL) * TILE_R;
    const int n = n0 + NU_ROW(r), nc = n < N ? n : N - 1;
    const bool live = (TILE_R == 16 || r < TILE_R) && n < N;
    const int fc = 16 * wv + 4 * g;            // first of this lane's four features in a 128-wide vector
    const float *par = sm.par;

    // ---- inputs first: they are waited for before the weight stream's loads (vmcnt retires in order) ------------
    const int srow = tid >> 5, scol = (tid & 31) * 4, sn = n0 + NU_ROW(srow) < N ? n0 + NU_ROW(srow) : N - 1;
    nf4 s4 = {0.f, 0.f, 0.f, 0.f};
    float ms = 0.f;
    if constexpr (!DEFER_S) {
        s4 = *reinterpret_cast<const nf4 *>(A.S + (size_t)sn * 128 + scol);
        ms = A.msum[nc];
    }
    const nf4 hv4 = *reinterpret_cast<const nf4 *>(A.hV + (size_t)nc * 128 + fc);
    const float rm = A.rmask[nc];
    const nf4 zero4i = {0.f, 0.f, 0.f, 0.f};
    // Every load of this prologue is UNCONDITIONAL and every comparison on a loaded value is left to its use: a load inside a
    // run-time branch (wave 0 only, `embed_next`, thread ranges) makes the compiler wait for it INSIDE the branch, with a vmcnt
    // that covers every input requested before it -- and the weight stream below would only start a memory round trip later
    // (ISA of the layer-2 instance, round 4: two such waits in front of the first weight fetch).  The waves that do not need a
    // value read the same addresses as the one that does.
    float chi1 = 0.f, scm1 = 0.f, nz1 = 0.f, nz2 = 0.f;       // wave 0: lane (r, g) steps chi g of residue r
    unsigned char m1raw = 0, m2raw = 0;
    int rt = 0;
    nf4 spv = zero4i;                          // c_ode, w, c_drift, c_diff of this step
    if constexpr (MODE == PP_NU_STEP) {
        spv = nf4{sp.c_ode, sp.w, sp.c_drift, sp.c_diff};
        chi1 = chi[(size_t)nc * 4 + g];
        scm1 = A.sc_mask[(size_t)nc * 4 + g];
        m1raw = A.m1pi[(size_t)nc * 4 + g];
        m2raw = A.m2pi[(size_t)nc * 4 + g];
        if (sde) {           // (the noise tensor only exists in sde mode)
            const size_t NN = (size_t)N * 4;
            const float *nz = noise + (size_t)step * 2 * NN + (size_t)nc * 4 + g;
            nz1 = nz[0];
            nz2 = nz[NN];
        }
        rt = (int)A.rtype[nc];
    }
    // small inputs of the kernel's tail, staged in LDS now (a dependent fetch there would sit on the critical path): the
    // tile's backbone frames (threads 0..47) and, in layer 2, the chi-independent dense inputs of the next step's node
    // embedding as MFMA operand rows (encoder.py:218-242: 6 backbone sin / cos by threads 128..143, the 16-d time embedding
    // by threads 192..207; the 8 chi sin / cos follow after the reverse step)
    nf4 tailv = {0.f, 0.f, 0.f, 0.f};
    nf4 ev[4] = {tailv, tailv, tailv, tailv};
    const int erow = tid & 15, ern = n0 + NU_ROW(erow) < N ? n0 + NU_ROW(erow) : N - 1;
    const bool e_bb = MODE == PP_NU_STEP && tid >= 128 && tid < 144, e_te = MODE == PP_NU_STEP && tid >= 192 && tid < 208;
    {       // frames of the tile: threads 0..47 need them, every thread reads (thread t the quad t mod 48 reads)
        const int t48 = tid % 48, row = t48 / 3, rn = n0 + NU_ROW(row) < N ? n0 + NU_ROW(row) : N - 1;
        tailv = *reinterpret_cast<const nf4 *>(A.frames + (size_t)rn * 12 + 4 * (t48 - 3 * row));
    }
    nf2 bb0 = {0.f, 0.f}, bb1 = bb0, bb2 = bb0;      // backbone sin / cos of row `ern` (threads 128..143 use them, after the stream start)
    if constexpr (MODE == PP_NU_STEP) {
        const nf2 *bp = reinterpret_cast<const nf2 *>(A.bb_sincos + (size_t)ern * 6);
        bb0 = bp[0]; bb1 = bp[1]; bb2 = bp[2];
    }
    constexpr int NPV = (NPAR / 4 + 511) / 512;
    nf4 pv[NPV];
#pragma unroll
    for (int i = 0; i < NPV; i++) {
        const int q = tid + 512 * i;
        pv[i] = reinterpret_cast<const nf4 *>(A.params)[q < NPAR / 4 ? q : 0];
    }
    // ---- start the weight stream ---------------------------------------------------------------------------------
    const nh8 *wq = reinterpret_cast<const nh8 *>(A.wstream) + (size_t)wv * NSLOT * 128 + lane;
    AOpN AR[NU_NRING];
#pragma unroll
    for (int k = 0; k < NU_ND; k++) gload_N(wq, k, AR[k]);
    __builtin_amdgcn_sched_barrier(0);       // the stream is on its way before anything waits for the inputs
    if constexpr (DEFER_S) {
        between();
        s4 = *reinterpret_cast<const nf4 *>(A.S + (size_t)sn * 128 + scol);
        ms = A.msum[nc];
    }
#pragma unroll
    for (int i = 0; i < NPV; i++) {
        const int q = tid + 512 * i;
        if (q < NPAR / 4) reinterpret_cast<nf4 *>(sm.par)[q] = pv[i];
    }
    // next step's embedding: the one-hot column is fetched now, used at the very end
    nf4 oh4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == PP_NU_STEP) {
        oh4 = *reinterpret_cast<const nf4 *>(A.embT + (size_t)rt * 128 + fc);      // (unconditional, as rt above)
    }
    if (e_te) {
#pragma unroll
        for (int k = 0; k < 16; k++) ev[k >> 2][k & 3] = te_next.v[k];
    } else {
        ev[0] = nf4{bb0[0], bb0[1], bb1[0], bb1[1]};
        ev[1] = nf4{bb2[0], bb2[1], 0.f, 0.f};
    }
    if (tid < 48) reinterpret_cast<nf4 *>(&sm.fr[0][0])[tid] = tailv;
    else if (e_bb) {                       // features 0..5 (6, 7 are rewritten with chi_0's sin / cos later)
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32, ev[0]);
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32 + 4, ev[1]);
    } else if (e_te) {                     // features 14..29, zeros in 30, 31; 12..13 are rewritten later
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32 + 12, nf4{0.f, 0.f, ev[0][0], ev[0][1]});
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32 + 16, nf4{ev[0][2], ev[0][3], ev[1][0], ev[1][1]});
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32 + 20, nf4{ev[1][2], ev[1][3], ev[2][0], ev[2][1]});
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32 + 24, nf4{ev[2][2], ev[2][3], ev[3][0], ev[3][1]});
        publish4(sm.e_hi, sm.e_lo, erow * NU_S32 + 28, nf4{ev[3][2], ev[3][3], 0.f, 0.f});
    }
    publish4(sm.a_hi, sm.a_lo, srow * NU_S128 + scol, s4);
    __syncthreads();

#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 1     /* timing experiment: stop here */
    return;
#endif
    nh8 bh[4], bl[4];
    nf4 cH, cL;
    float satm = 0.f;
    const nf4 zero4 = zero4i;
    // ---- W_out on the masked mean S: mean_j mask_j (W_out y_j + b) = W_out mean_j(mask_j y_j) + b mean_j(mask_j) -----
    LDB4(sm.a_hi, sm.a_lo)
    cH = zero4; cL = zero4;
    NTILE4(0, cH, cL)
    nf4 x = fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_OUTB + fc) * ms + hv4;
    const nf4 h1 = ln128(sm.stats[0], wv, r, g, x, *reinterpret_cast<const nf4 *>(par + NU_P_G0 + fc),
                         *reinterpret_cast<const nf4 *>(par + NU_P_B0 + fc));
    publish4(sm.b_hi, sm.b_lo, r * NU_S128 + fc, h1);
    __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 2     /* timing experiment: stop here */
    return;
#endif
    // ---- FFN 128 -> 512: hidden tiles 4 w .. 4 w + 3 ----------------------------------------------------------------
    LDB4(sm.b_hi, sm.b_lo)
#define FFN_IN_TILE(c)                                                                                     \
    cH = zero4; cL = zero4;                                                                                \
    NTILE4(4 + 4 * (c), cH, cL)                                                                            \
    publish4(sm.c_hi, sm.c_lo, r * NU_S512 + 16 * (4 * wv + (c)) + 4 * g,                                  \
             relu_sat(fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_FIB + 16 * (4 * wv + (c)) + 4 * g), satm));
    FFN_IN_TILE(0) FFN_IN_TILE(1) FFN_IN_TILE(2) FFN_IN_TILE(3)
    if (!(satm < 65504.f)) atomicOr(A.sat, 2u);
    __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 3     /* timing experiment: stop here */
    return;
#endif
    // ---- FFN 512 -> 128, LayerNorm, mask ------------------------------------------------------------------------------
    nh8 fh[2], fl[2];
    ldB(sm.c_hi, sm.c_lo, r * NU_S512 + 8 * g, 0, fh[0], fl[0]);
    cH = zero4; cL = zero4;
    FOSTAGE(0) FOSTAGE(1) FOSTAGE(2) FOSTAGE(3) FOSTAGE(4) FOSTAGE(5) FOSTAGE(6) FOSTAGE(7)
    FOSTAGE(8) FOSTAGE(9) FOSTAGE(10) FOSTAGE(11) FOSTAGE(12) FOSTAGE(13) FOSTAGE(14) FOSTAGE(15)
    x = fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_FOB + fc) + h1;
    const nf4 h2 = ln128(sm.stats[1], wv, r, g, x, *reinterpret_cast<const nf4 *>(par + NU_P_G1 + fc),
                         *reinterpret_cast<const nf4 *>(par + NU_P_B1 + fc)) * rm;
    if (live && clq == 0 && (MODE != PP_NU_STEP || !embed_next)) *reinterpret_cast<nf4 *>(A.hV_out + (size_t)n * 128 + fc) = h2;
    publish4(sm.a_hi, sm.a_lo, r * NU_S128 + fc, h2);
    __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 4     /* timing experiment: stop here */
    return;
#endif
    LDB4(sm.a_hi, sm.a_lo)

    if constexpr (!LAST) {
        // ---- inputs of this layer's edge message and of the next layer's node message ----------------------------------
        // projection pj = 0..3: PAe (+ bias), PCe, PAn (+ bias), PCn; this workgroup computes pj = NPJ clq .. NPJ clq + NPJ - 1
        constexpr int NPJ = 4 / CL;
#define NU_PROJ(i)                                                                                                     \
        if constexpr ((i) < NPJ) {                                                                                     \
            cH = zero4; cL = zero4;                                                                                    \
            NTILE4(36 + 4 * (i), cH, cL)                                                                               \
            const int pj = NPJ * clq + (i);                                                                            \
            float *dst = pj == 0 ? A.PAe : pj == 1 ? A.PCe : pj == 2 ? A.PAn : A.PCn;                                  \
            nf4 v = fold(cH, cL);                                                                                      \
            if (pj == 0) v = v + *reinterpret_cast<const nf4 *>(par + NU_P_PAE_B + fc);                                \
            else if (pj == 2) v = v + *reinterpret_cast<const nf4 *>(par + NU_P_PAN_B + fc);                           \
            if (live) *reinterpret_cast<nf4 *>(dst + (size_t)n * 128 + fc) = v;                                        \
        }
        NU_PROJ(0) NU_PROJ(1) NU_PROJ(2) NU_PROJ(3)
#undef NU_PROJ
        if (CL > 1 && clq != 0) return;        // (uniform per workgroup) the local points are workgroup 0's
        cH = zero4; cL = zero4;
        NTILE4_IF(36 + 4 * NPJ, wv < 3, cH, cL)
        if (wv < 3) {              // local points: features 0..23 edge message, 24..47 next node message
            const nf4 p = fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_PTS_B + fc);
            *reinterpret_cast<nf4 *>(&sm.pts[r][fc]) = p;
            if (live) {
                if (fc < 24) *reinterpret_cast<nf4 *>(A.ptsE + (size_t)n * 48 + fc) = p;
                else *reinterpret_cast<nf4 *>(A.ptsN + (size_t)n * 48 + fc - 24) = p;
            }
        }
        __syncthreads();
        if (tid < 256) {           // (message m, point q, residue i): p_glob = R p_loc + t
            const int mm = tid >> 7, q = (tid >> 4) & 7, i = tid & 15, ni = n0 + i;
            if ((TILE_R == 16 || i < TILE_R) && ni < N) {
                float *pts = mm == 0 ? A.ptsE : A.ptsN;
                const float *fr = sm.fr[i];
                const float px = sm.pts[i][24 * mm + 3 * q], py = sm.pts[i][24 * mm + 3 * q + 1], pz = sm.pts[i][24 * mm + 3 * q + 2];
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
                    pts[(size_t)ni * 48 + 24 + 3 * q + rr] = (fr[3 * rr] * px + fr[3 * rr + 1] * py + fr[3 * rr + 2] * pz) + fr[9 + rr];
            }
        }
        return;
    } else {
        // ---- decoder 128 -> 64 -> 32 -> relu -> 16 -> 4 (TorsionalDiffusion.py:105-109) --------------------------------
        cH = zero4; cL = zero4;
        NTILE4_IF(36, wv < 4, cH, cL)
        if (wv < 4) {
            publish4(sm.c_hi, sm.c_lo, r * NU_S512 + fc, relu_sat(fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_DB0 + fc), satm));
        }
        __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 5     /* timing experiment: stop here */
        return;
#endif
        // the rest of the decoder is one wave's work: 64 -> 32 (two tiles), 32 -> 16, 16 -> 4; activations go through
        // columns 64..127 of the same image (a wave's LDS operations execute in order; the asm is the compiler fence).
        // The other waves only keep their weight stream going (the stages' fetches) and wait at the next barrier.
        const bool w0 = wv == 0;
        {
            nh8 dh[2], dl[2];
            if (w0) {
                ldB(sm.c_hi, sm.c_lo, r * NU_S512 + 8 * g, 0, dh[0], dl[0]);
                ldB(sm.c_hi, sm.c_lo, r * NU_S512 + 8 * g, 1, dh[1], dl[1]);
            }
            nf4 eH = zero4, eL = zero4;
            cH = zero4; cL = zero4;
            NSTAGE_IF(40, w0, cH, mm3(AK, dh[0], dl[0], cH, cL))
            NSTAGE_IF(41, w0, cH, mm3(AK, dh[1], dl[1], cH, cL))
            NSTAGE_IF(42, w0, eH, mm3(AK, dh[0], dl[0], eH, eL))
            NSTAGE_IF(43, w0, eH, mm3(AK, dh[1], dl[1], eH, eL))
            if (w0) {
                publish4(sm.c_hi, sm.c_lo, r * NU_S512 + 64 + 4 * g, relu_sat(fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_DB1 + 4 * g), satm));
                publish4(sm.c_hi, sm.c_lo, r * NU_S512 + 80 + 4 * g, relu_sat(fold(eH, eL) + *reinterpret_cast<const nf4 *>(par + NU_P_DB1 + 16 + 4 * g), satm));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ldB(sm.c_hi, sm.c_lo, r * NU_S512 + 8 * g, 2, dh[0], dl[0]);
            }
            cH = zero4; cL = zero4;
            NSTAGE_IF(44, w0, cH, mm3(AK, dh[0], dl[0], cH, cL))
            if (w0) {
                publish4(sm.c_hi, sm.c_lo, r * NU_S512 + 96 + 4 * g, relu_sat(fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_DB2 + 4 * g), satm));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ldB(sm.c_hi, sm.c_lo, r * NU_S512 + 8 * g, 3, dh[0], dl[0]);      // columns 112..127: stale but finite, zero weights
            }
            cH = zero4; cL = zero4;
            NSTAGE_IF(45, w0, cH, mm3(AK, dh[0], dl[0], cH, cL))
        }
        if (w0) {
            // registers 0..3 of lane group 0 = the four scores of residue r
            if (!(satm < 65504.f)) atomicOr(A.sat, 2u);          // decoder hidden layers (the FFN's were reported above)
            const nf4 sc = fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_DB3);
            if (g == 0 && live) *reinterpret_cast<nf4 *>(A.score + (size_t)n * 4) = sc;
            if constexpr (MODE == PP_NU_STEP) {
                // reverse step (schedule.py:198-235 with the two periodicity masks, TorsionalDiffusion.py:268-280): one
                // (residue, chi) per lane -- lane (r, g) takes score g from lane (r, 0)
                const float s0 = __shfl(sc[0], r), s1 = __shfl(sc[1], r), s2 = __shfl(sc[2], r), s3 = __shfl(sc[3], r);
                const float sg = g == 0 ? s0 : g == 1 ? s1 : g == 2 ? s2 : s3;
                const float sp_c_ode = spv[0], sp_w = spv[1], sp_c_drift = spv[2], sp_c_diff = spv[3];
                const float sw = sg * sp_w;
                float yk = chi1;
                if (!sde) {
                    if (m1raw != 0 || m2raw != 0) yk = chi1 + sp_c_ode * sw;
                } else {
                    if (m1raw != 0) yk = chi1 + (sp_c_drift * sw + sp_c_diff * nz1);
                    if (m2raw != 0) yk = yk + (sp_c_drift * sw + sp_c_diff * nz2);
                }
                const float y = wrap_pi(yk) * scm1;
                if (live) chi[(size_t)n * 4 + g] = y;
                if (embed_next) {          // features 6 + 2 g, 7 + 2 g of the embedding operand
                    unsigned hp, lp;
                    split2(sinf(y) * scm1, cosf(y) * scm1, hp, lp);
                    *reinterpret_cast<unsigned *>(sm.e_hi + r * NU_S32 + 6 + 2 * g) = hp;
                    *reinterpret_cast<unsigned *>(sm.e_lo + r * NU_S32 + 6 + 2 * g) = lp;
                }
            }
        }
        if constexpr (MODE != PP_NU_STEP) return;
        if (!embed_next) return;
        __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 6     /* timing experiment: stop here */
        return;
#endif
        // ---- next step's node embedding (encoder.py:218-242) and the layer-0 node-message inputs ---------------------
        // bias + one-hot column + W[:, 21:51] . (30 dense inputs) as one MFMA k-step (slot 46)
        nh8 eh, el;
        eh = *reinterpret_cast<const nh8 *>(sm.e_hi + r * NU_S32 + 8 * g);
        el = *reinterpret_cast<const nh8 *>(sm.e_lo + r * NU_S32 + 8 * g);
        cH = zero4; cL = zero4;
        NSTAGE(46, cH, mm3(AK, eh, el, cH, cL))
        const nf4 e0 = fold(cH, cL) + (*reinterpret_cast<const nf4 *>(par + NU_P_EMB_B + fc) + oh4);
        const nf4 h0 = ln128(sm.stats[2], wv, r, g, e0, *reinterpret_cast<const nf4 *>(par + NU_P_EMB_G + fc),
                             *reinterpret_cast<const nf4 *>(par + NU_P_EMB_BETA + fc));
        if (live) *reinterpret_cast<nf4 *>(A.hV_out + (size_t)n * 128 + fc) = h0;
        publish4(sm.b_hi, sm.b_lo, r * NU_S128 + fc, h0);
        __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 7     /* timing experiment: stop here */
        return;
#endif
        LDB4(sm.b_hi, sm.b_lo)
        cH = zero4; cL = zero4;
        NTILE4(47, cH, cL)
        if (live) *reinterpret_cast<nf4 *>(A.PAn + (size_t)n * 128 + fc) = fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_PAN0_B + fc);
        cH = zero4; cL = zero4;
        NTILE4(51, cH, cL)
        if (live) *reinterpret_cast<nf4 *>(A.PCn + (size_t)n * 128 + fc) = fold(cH, cL);
        if (wv < 2) {
            cH = zero4; cL = zero4;
            NTILE4(55, cH, cL)
            if (fc < 24) {
                const nf4 p = fold(cH, cL) + *reinterpret_cast<const nf4 *>(par + NU_P_PTS0_B + fc);
                *reinterpret_cast<nf4 *>(&sm.pts[r][fc]) = p;
                if (live) *reinterpret_cast<nf4 *>(A.ptsN + (size_t)n * 48 + fc) = p;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int q = tid >> 4, i = tid & 15, ni = n0 + i;
            if ((TILE_R == 16 || i < TILE_R) && ni < N) {
                const float *fr = sm.fr[i];
                const float px = sm.pts[i][3 * q], py = sm.pts[i][3 * q + 1], pz = sm.pts[i][3 * q + 2];
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
                    A.ptsN[(size_t)ni * 48 + 24 + 3 * q + rr] = (fr[3 * rr] * px + fr[3 * rr + 1] * py + fr[3 * rr + 2] * pz) + fr[9 + rr];
            }
        }
    }
}
#undef NU_ROW

template <int MODE, int NU_ND, int CL = 1>
__global__ void __launch_bounds__(512)
k_node_update(NUpdArgs A, float *chi, int step, int sde, const float *noise, int embed_next, StepScalars sp, TimeEmb te_next) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    node_update_body<MODE, NU_ND, CL>(A, chi, step, sde, noise, embed_next, sp, te_next, smem_raw, (int)blockIdx.x);
}

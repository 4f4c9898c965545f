// Node-level stages of the score network (everything that is [B*L, 128]-shaped).
//
//   k_node_embed  : node features -> Linear(51,128) -> LN              (encoder.py:218-242, layers.py:257-268)
//                   + inputs of the layer-0 node message
//   k_node_update : mean message -> W_out -> LN -> FFN -> LN -> mask   (layers.py:124-132)
//                   + inputs of this layer's edge message and the next layer's node message,
//                   or (last layer) decoder (TorsionalDiffusion.py:105-109), reverse step
//                   (schedule.py:198-235, TorsionalDiffusion.py:268-280) and the next embedding.
//
// This work is ~3% of the FLOPs but a long dependent chain over only B*L rows, so it is laid out for
// latency: a block of 512 threads owns NB = 4 NG consecutive residues; thread (f = tid & 127, ks = tid >> 7) owns
// output feature f for all of them over the ks-th quarter of the reduction dimension; the four partial sums meet
// in LDS.  Weights are read from transposed copies ([in][out]) so a wave's loads are contiguous; activations
// sit in LDS as float4 groups (one component per residue).  The kernel is bound by one block's dependent chain, not by
// bytes (same time for 16 and 256 blocks), so k_node_update fetches every phase's weights a phase or two ahead.
// The cheap epilogues (LayerNorm, bias, ReLU) are replicated in the four ks-groups to keep control flow uniform.
#include <string.h>

#include "pp_internal.h"
#include <utility>

PP_RANGE_COUNTER
PP_RANGE_READER(pp_node_range_hits)

#define NT 512
#ifndef PP_NODE_GROUPS
#define PP_NODE_GROUPS 1
#endif
#define NG PP_NODE_GROUPS
#define NB (4 * NG)

struct NodeArgs {
    int N;
    const float *rmask;          // [N]
    const int64_t *rtype;        // [N]
    const float *bb_sincos;      // [N][6]
    const float *sc_mask;        // [N][4]
    const uint8_t *m1pi, *m2pi;  // [N][4]
    const float *frames;         // [N][12]
    const float *embT, *emb_b, *emb_g, *emb_beta;
    float *hV, *S, *msum;
    float *ptsN, *PAn, *PCn, *ptsE, *PAe, *PCe;
    float *score;
};

struct PreW {            // weights feeding one message function
    const float *ptsT, *pts_b;      // [128][24], [24]
    const float *AT, *CT, *in_b;    // [128][128] x2, [128]
};


// one feature's values for the NB residues of the block (ext vector: fma on it selects v_pk_fma_f32)
typedef float f4v __attribute__((ext_vector_type(4)));
struct VN {
    f4v g[NG];
};

struct Smem {
    VN part[2][4 * 576];       // ping-pong partial sums (4 K-slices x up to 576 columns)
    VN a[512];                 // wide activation vector (FFN hidden, decoder scratch)
    VN h[128];                 // current node vector
    VN p[48];                  // local points / small inputs
    VN red[2][16];             // LayerNorm partials (8 wave means, 8 wave sums of squares), ping-pong
};

__device__ __forceinline__ f4v f4(float v) { return f4v{v, v, v, v}; }
__device__ __forceinline__ float comp(f4v v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

#define VN_FOR _Pragma("unroll") for (int gi = 0; gi < NG; gi++)
__device__ __forceinline__ VN vn(float v) { VN r; VN_FOR r.g[gi] = f4(v); return r; }
__device__ __forceinline__ VN vfma(float w, const VN &a, VN c) {
    VN_FOR c.g[gi] = __builtin_elementwise_fma(f4(w), a.g[gi], c.g[gi]);
    return c;
}
__device__ __forceinline__ VN vadd(VN a, const VN &b) {
    VN_FOR a.g[gi] = a.g[gi] + b.g[gi];
    return a;
}
__device__ __forceinline__ VN vsub(VN a, const VN &b) {
    VN_FOR a.g[gi] = a.g[gi] - b.g[gi];
    return a;
}
__device__ __forceinline__ VN vmul(VN a, const VN &b) {
    VN_FOR a.g[gi] = a.g[gi] * b.g[gi];
    return a;
}
__device__ __forceinline__ VN vscale(VN a, float s) {
    VN_FOR a.g[gi] = a.g[gi] * s;
    return a;
}
__device__ __forceinline__ VN vrelu(VN a) {
    VN_FOR a.g[gi] = f4v{fmaxf(a.g[gi].x, 0.f), fmaxf(a.g[gi].y, 0.f), fmaxf(a.g[gi].z, 0.f), fmaxf(a.g[gi].w, 0.f)};
    return a;
}
__device__ __forceinline__ float vcomp(const VN &v, int i) { return comp(v.g[i >> 2], i & 3); }   // i: residue in block

// partial sum over the ks-th quarter of the reduction dimension; WT is k-quad interleaved [in / 4][ldo][4] (put_T4)
template <int KIN>
__device__ __forceinline__ VN dense_slice(const float *__restrict__ WT, int ldo, int col, const VN *act, int ks) {
    constexpr int KL = KIN / 4;
    const float4 *w = reinterpret_cast<const float4 *>(WT) + (size_t)(ks * (KL / 4)) * ldo + col;
    const VN *a = act + ks * KL;
    VN acc = vn(0.f);
#pragma unroll 4
    for (int i = 0; i < KL / 4; i++) {
        const float4 q = w[(size_t)i * ldo];
        acc = vfma(q.x, a[4 * i], acc);
        acc = vfma(q.y, a[4 * i + 1], acc);
        acc = vfma(q.z, a[4 * i + 2], acc);
        acc = vfma(q.w, a[4 * i + 3], acc);
    }
    return acc;
}

// meet the four K-slices: every thread returns the full sum for its column (one barrier, ping-pong buffer)
__device__ __forceinline__ VN meet(Smem &sm, int &flip, const VN &partial, int stride, int col, int ks) {
    VN *buf = sm.part[flip];
    flip ^= 1;
    buf[ks * stride + col] = partial;
    __syncthreads();
    return vadd(vadd(buf[col], buf[stride + col]), vadd(buf[2 * stride + col], buf[3 * stride + col]));
}

// all-lanes sum over a wave: four DPP butterfly steps inside each row of 16 lanes (VALU speed), then the four row
// totals through SGPRs -- no LDS-pipeline permutes (ds_bpermute x 6 levels was ~0.4 us per call)
__device__ __forceinline__ float wave_sum1(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));  // row_mirror
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ VN wave_sum(VN v) {
    VN_FOR v.g[gi] = f4v{wave_sum1(v.g[gi].x), wave_sum1(v.g[gi].y), wave_sum1(v.g[gi].z), wave_sum1(v.g[gi].w)};
    return v;
}

// LayerNorm over 128 features (one per thread of a ks-group = 2 waves), NB residues at once, eps 1e-5, two-pass
// LayerNorm over the 128 features held by a pair of waves (grp, grp + 1): each wave reduces its own 64 features to
// (mean, centred sum of squares), the pair meets ONCE in LDS and merges with Chan's update for equal counts --
// one barrier per LayerNorm instead of two (mean, then variance), and as stable as the two-pass form.
__device__ __forceinline__ VN layernorm(Smem &sm, int &rflip, const VN &v, float g, float b) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, grp = wid & ~1;
    const VN mw = vscale(wave_sum(v), 1.f / 64.f);
    const VN dw = vsub(v, mw);
    const VN qw = wave_sum(vmul(dw, dw));
    VN *r = sm.red[rflip];
    rflip ^= 1;
    if (lane == 0) { r[wid] = mw; r[8 + wid] = qw; }
    __syncthreads();
    const VN ma = r[grp], mb = r[grp + 1];
    const VN mean = vscale(vadd(ma, mb), 0.5f);
    const VN dm = vsub(ma, mb);
    const VN var = vscale(vadd(vadd(r[8 + grp], r[8 + grp + 1]), vscale(vmul(dm, dm), 32.f)), 1.f / 128.f);
    const VN d = vsub(v, mean);
    VN o;
    VN_FOR o.g[gi] = f4v{d.g[gi].x * (1.f / sqrtf(var.g[gi].x + 1e-5f)) * g + b,
                                 d.g[gi].y * (1.f / sqrtf(var.g[gi].y + 1e-5f)) * g + b,
                                 d.g[gi].z * (1.f / sqrtf(var.g[gi].z + 1e-5f)) * g + b,
                                 d.g[gi].w * (1.f / sqrtf(var.g[gi].w + 1e-5f)) * g + b};
    return o;
}

__device__ __forceinline__ void store_rows(float *dst, int ld, int n0, int N, int col, const VN &v) {
#pragma unroll
    for (int i = 0; i < NB; i++)
        if (n0 + i < N) dst[(size_t)(n0 + i) * ld + col] = vcomp(v, i);
}
__device__ __forceinline__ VN load_rows(const float *src, int ld, int n0, int N, int col) {
    VN v;
    VN_FOR {
        const int b = n0 + 4 * gi;
        v.g[gi] = f4v{b + 0 < N ? src[(size_t)(b + 0) * ld + col] : 0.f, b + 1 < N ? src[(size_t)(b + 1) * ld + col] : 0.f,
                              b + 2 < N ? src[(size_t)(b + 2) * ld + col] : 0.f, b + 3 < N ? src[(size_t)(b + 3) * ld + col] : 0.f};
    }
    return v;
}

// Inputs of one message function from h = sm.h: points (local + global), W_A h + b, W_C h.
// One meeting round for all 128 + 128 + 24 outputs.
__device__ void message_inputs(Smem &sm, int &flip, const PreW &w, const float *frames, int n0, int N,
                               float *pts, float *PA, float *PC) {
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7;
    VN pa = dense_slice<128>(w.AT, 128, f, sm.h, ks);
    VN pc = dense_slice<128>(w.CT, 128, f, sm.h, ks);
    VN *buf = sm.part[flip];
    flip ^= 1;
    buf[ks * 384 + f] = pa;
    buf[ks * 384 + 128 + f] = pc;
    if (f < 24) buf[ks * 384 + 256 + f] = dense_slice<128>(w.ptsT, 24, f, sm.h, ks);
    __syncthreads();
    if (ks == 0) {
        VN a = vadd(vadd(buf[f], buf[384 + f]), vadd(buf[768 + f], buf[1152 + f]));
        store_rows(PA, 128, n0, N, f, vadd(a, vn(w.in_b[f])));
    } else if (ks == 1) {
        VN c = vadd(vadd(buf[128 + f], buf[512 + f]), vadd(buf[896 + f], buf[1280 + f]));
        store_rows(PC, 128, n0, N, f, c);
    } else if (ks == 2 && f < 24) {
        VN p = vadd(vadd(buf[256 + f], buf[640 + f]), vadd(buf[1024 + f], buf[1408 + f]));
        p = vadd(p, vn(w.pts_b[f]));
        sm.p[f] = p;
        store_rows(pts, 48, n0, N, f, p);
    }
    __syncthreads();
    if (threadIdx.x < 8 * NB) {            // (point q, residue i): p_glob = R p_loc + t
        int q = threadIdx.x / NB, i = threadIdx.x % NB;
        int n = n0 + i;
        if (n < N) {
            const float *fr = frames + (size_t)n * 12;
            float x = vcomp(sm.p[3 * q], i), y = vcomp(sm.p[3 * q + 1], i), z = vcomp(sm.p[3 * q + 2], i);
            for (int r = 0; r < 3; r++)
                pts[(size_t)n * 48 + 24 + 3 * q + r] = (fr[3 * r] * x + fr[3 * r + 1] * y + fr[3 * r + 2] * z) + fr[9 + r];
        }
    }
    __syncthreads();
}

// Node embedding for the block's residues -> value (before LN) of feature f (all ks-groups compute the same thing)
__device__ __forceinline__ VN embed_pre(Smem &sm, const NodeArgs &A, const float *chi, const StepParams &sp, int n0) {
    const int t = threadIdx.x, f = t & 127;
    if (t < 6) sm.p[t] = load_rows(A.bb_sincos, 6, n0, A.N, t);
    else if (t < 14) {
        int k = (t - 6) >> 1, sc = (t - 6) & 1;
        VN x = load_rows(chi, 4, n0, A.N, k), m = load_rows(A.sc_mask, 4, n0, A.N, k), v;
        VN_FOR v.g[gi] = sc ? f4v{cosf(x.g[gi].x), cosf(x.g[gi].y), cosf(x.g[gi].z), cosf(x.g[gi].w)}
                            : f4v{sinf(x.g[gi].x), sinf(x.g[gi].y), sinf(x.g[gi].z), sinf(x.g[gi].w)};
        sm.p[t] = vmul(v, m);
    }
    __syncthreads();
    VN acc = vn(A.emb_b[f]);
    VN_FOR {
        const int b = n0 + 4 * gi;
        const int t0 = b + 0 < A.N ? (int)A.rtype[b + 0] : 0, t1 = b + 1 < A.N ? (int)A.rtype[b + 1] : 0;
        const int t2 = b + 2 < A.N ? (int)A.rtype[b + 2] : 0, t3 = b + 3 < A.N ? (int)A.rtype[b + 3] : 0;
        acc.g[gi].x += A.embT[t0 * 128 + f]; acc.g[gi].y += A.embT[t1 * 128 + f];
        acc.g[gi].z += A.embT[t2 * 128 + f]; acc.g[gi].w += A.embT[t3 * 128 + f];
    }
#pragma unroll
    for (int k = 0; k < 14; k++) acc = vfma(A.embT[(21 + k) * 128 + f], sm.p[k], acc);
    const float *te = sp.temb;
    float tacc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) tacc = fmaf(A.embT[(35 + k) * 128 + f], te[k], tacc);
    return vadd(acc, vn(tacc));
}

__global__ void __launch_bounds__(NT)
k_node_embed(NodeArgs A, PreW pre0, const float *chi, StepParams sp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    int flip = 0, rflip = 0;
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7, n0 = blockIdx.x * NB;
    VN v = embed_pre(sm, A, chi, sp, n0);
    VN h = layernorm(sm, rflip, v, A.emb_g[f], A.emb_beta[f]);
    if (ks == 0) {
        store_rows(A.hV, 128, n0, A.N, f, h);
        sm.h[f] = h;
    }
    __syncthreads();
    message_inputs(sm, flip, pre0, A.frames, n0, A.N, A.ptsN, A.PAn, A.PCn);
}

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
        if constexpr ((k) + NU_ND < NLOAD) {                                                           \
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

// CL > 1 (middle layers, when CL x tiles workgroups still fit the chip in one round): CL workgroups per 16-residue tile.  A
// launch lasts as long as ONE CU needs for its workgroup's weight stream, and most CUs idle (47 tiles at T1124); so the tile's
// common part (W_out, FFN, both LayerNorms: 36 slots) is computed redundantly by CL workgroups on CL CUs, and the 20 slots of
// projections behind it are dealt out among them -- no exchange between workgroups, identical arithmetic per output.
template <int MODE, int NU_ND, int CL = 1>
__global__ void __launch_bounds__(512)
k_node_update(NUpdArgs A, float *chi, int step, int sde, const float *noise, int embed_next, StepScalars sp, TimeEmb te_next) {
    static_assert(CL == 1 || MODE == PP_NU_MID, "only the middle layers have a split form");
    constexpr int NU_NRING = NU_ND + 1;
    constexpr bool LAST = MODE != PP_NU_MID;
    constexpr int NSLOT = LAST ? PP_NU_SLOTS_LAST : PP_NU_SLOTS_MID;
    constexpr int NLOAD = MODE == PP_NU_SCORE ? 46 : CL > 1 ? 36 + 16 / CL + 4 : NSLOT;       // (logical) slots this instance ever fetches
    const int clq = CL > 1 ? (int)(blockIdx.x % CL) : 0;          // which of the tile's workgroups this is
    constexpr int NPAR = LAST ? NU_P_LAST_TOTAL : NU_P_MID_TOTAL;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SmemU &sm = *reinterpret_cast<SmemU *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4, N = A.N, n0 = (int)(blockIdx.x / CL) * 16;
    const int n = n0 + r, nc = n < N ? n : N - 1;
    const bool live = n < N;
    const int fc = 16 * wv + 4 * g;            // first of this lane's four features in a 128-wide vector
    const float *par = sm.par;

    // ---- inputs first: they are waited for before the weight stream's loads (vmcnt retires in order) ------------
    const int srow = tid >> 5, scol = (tid & 31) * 4, sn = n0 + srow < N ? n0 + srow : N - 1;
    const nf4 s4 = *reinterpret_cast<const nf4 *>(A.S + (size_t)sn * 128 + scol);
    const nf4 hv4 = *reinterpret_cast<const nf4 *>(A.hV + (size_t)nc * 128 + fc);
    const float ms = A.msum[nc], rm = A.rmask[nc];
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
    const int erow = tid & 15, ern = n0 + erow < N ? n0 + erow : N - 1;
    const bool e_bb = MODE == PP_NU_STEP && tid >= 128 && tid < 144, e_te = MODE == PP_NU_STEP && tid >= 192 && tid < 208;
    {       // frames of the tile: threads 0..47 need them, every thread reads (thread t the quad t mod 48 reads)
        const int t48 = tid % 48, row = t48 / 3, rn = n0 + row < N ? n0 + row : N - 1;
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
            if (ni < N) {
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
            if (ni < N) {
                const float *fr = sm.fr[i];
                const float px = sm.pts[i][3 * q], py = sm.pts[i][3 * q + 1], pz = sm.pts[i][3 * q + 2];
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
                    A.ptsN[(size_t)ni * 48 + 24 + 3 * q + rr] = (fr[3 * rr] * px + fr[3 * rr + 1] * py + fr[3 * rr + 2] * pz) + fr[9 + rr];
            }
        }
    }
}

// ==================================================================================================================
// k_node_update in plain fp32 on the VALU -- the node update of the EXACT-FP32 library (libpackppi_hip.f32.so, built without
// PP_EDGE_F16): round 1's kernel (software-pipelined weight fetches, K split over four thread groups) with the per-step scalars
// as kernel arguments.  19.5 / 25.6 us per launch at T1124 against 12.5 / 14 for the matrix-pipe kernel above; no f16 operand
// anywhere, so a checkpoint whose activations leave the f16 range (python -m packppi_amd.rangecheck) has a library that computes
// every dense layer of the path in fp32.
// ==================================================================================================================
#ifndef PP_EDGE_F16
struct UpdW {
    const float *outT, *out_b;      // node_message_fn.W_out^T [128][128], bias
    const float *g0, *b0, *g1, *b1; // norm.0 / norm.1
    const float *ffn_inT, *ffn_in_b, *ffn_outT, *ffn_out_b;   // [128][512],[512],[512][128],[128]
    PreW pre_edge, pre_next;
    const float *d0_inT, *d0_in_b, *d0_outT, *d0_out_b, *d2_inT, *d2_in_b, *d2_outT, *d2_out_b;
};
// small dense layer used by the decoder: width <= 128 outputs, K split over the four ks-groups
template <int KIN>
__device__ __forceinline__ VN dense_small(Smem &sm, int &flip, const float *WT, int width, const VN *act,
                                          const float *bias) {
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7;
    VN p = f < width ? dense_slice<KIN>(WT, width, f, act, ks) : vn(0.f);
    VN r = meet(sm, flip, p, 128, f, ks);
    return f < width ? vadd(r, vn(bias[f])) : vn(0.f);
}

// ---- k_node_update: the same arithmetic as the helpers above, software-pipelined -------------------------------
// Measured (tools/debug/time_vs_n.py): the kernel takes the same 24 us for 16 and for 256 blocks -- it is one block's
// dependent chain of ~30 L2 round trips (every dense phase used to fetch its 32-128 weights per thread right before
// using them).  Weights do not depend on activations, so here every phase's weights are fetched into registers one or
// two phases ahead (five register sets of 32), and the per-feature vectors (biases, LayerNorm gains) at kernel start.
struct WSet {
    float v[32];
};
// rows k0 .. k0+KL of column col of a transposed weight [in][ldo]
// `after`: a value the previous user of this register set produced.  The weights are read-only kernel arguments, so
// the compiler would otherwise hoist every fetch of the kernel to its top (and spill ~1900 registers); making the
// offset opaque behind an empty asm that consumes `after` pins the fetch between that value and its first use.
template <int KL, int DST0 = 0>
__device__ __forceinline__ void wload(WSet &w, const float *__restrict__ WT, int ldo, int col, int k0, float after) {
    int off = (k0 >> 2) * ldo + col;          // in float4 units of the k-quad interleaved layout (put_T4)
    asm volatile("" : "+v"(off) : "v"(after));
    const float4 *w4 = reinterpret_cast<const float4 *>(WT);
#pragma unroll
    for (int i = 0; i < KL / 4; i++) {
        const float4 q = w4[off + i * ldo];
        w.v[DST0 + 4 * i] = q.x; w.v[DST0 + 4 * i + 1] = q.y; w.v[DST0 + 4 * i + 2] = q.z; w.v[DST0 + 4 * i + 3] = q.w;
    }
}
template <int KL, int SRC0 = 0>
__device__ __forceinline__ VN wdot(const WSet &w, const VN *act, VN acc) {
#pragma unroll
    for (int i = 0; i < KL; i++) {
        acc = vfma(w.v[SRC0 + i], act[i], acc);
        // keep the scheduler from running the x/y chains of a whole slice ahead of the z/w chains (it then parks
        // half of every activation read in scratch)
        if ((i & 7) == 7) VN_FOR asm volatile("" : "+v"(acc.g[gi].x), "+v"(acc.g[gi].y), "+v"(acc.g[gi].z), "+v"(acc.g[gi].w));
    }
    return acc;
}

// four dot products over the same activation slice in one pass (one LDS read feeds four outputs)
#define ACC_PIN(a) VN_FOR asm volatile("" : "+v"(a.g[gi].x), "+v"(a.g[gi].y), "+v"(a.g[gi].z), "+v"(a.g[gi].w))
__device__ __forceinline__ void wdot4(const WSet &w0, const WSet &w1, const WSet &w2, const WSet &w3, const VN *act,
                                      VN &a0, VN &a1, VN &a2, VN &a3) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const VN x = act[i];
        a0 = vfma(w0.v[i], x, a0);
        a1 = vfma(w1.v[i], x, a1);
        a2 = vfma(w2.v[i], x, a2);
        a3 = vfma(w3.v[i], x, a3);
        if ((i & 3) == 3) { ACC_PIN(a0); ACC_PIN(a1); ACC_PIN(a2); ACC_PIN(a3); }
    }
}

// message_inputs with the weights already in registers (wA, wC: this thread's K-quarter of column f; wP: of column f < 24).
// refill_ptsT, if not null: the point weights of the NEXT call, fetched into wP as soon as it has been consumed.
__device__ __forceinline__ void message_inputs_pre(Smem &sm, int &flip, const WSet &wA, const WSet &wC, WSet &wP,
                                                   const float *refill_ptsT, float in_b, float pts_b, const float *frames,
                                                   int n0, int N, float *pts, float *PA, float *PC) {
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7;
    const VN *h = sm.h + ks * 32;
    VN *buf = sm.part[flip];
    flip ^= 1;
    buf[ks * 384 + f] = wdot<32>(wA, h, vn(0.f));
    buf[ks * 384 + 128 + f] = wdot<32>(wC, h, vn(0.f));
    if (f < 24) {
        const VN up = wdot<32>(wP, h, vn(0.f));
        buf[ks * 384 + 256 + f] = up;
        if (refill_ptsT) wload<32>(wP, refill_ptsT, 24, f, ks * 32, up.g[0].x);
    }
    __syncthreads();
    if (ks == 0) {
        VN a = vadd(vadd(buf[f], buf[384 + f]), vadd(buf[768 + f], buf[1152 + f]));
        store_rows(PA, 128, n0, N, f, vadd(a, vn(in_b)));
    } else if (ks == 1) {
        VN c = vadd(vadd(buf[128 + f], buf[512 + f]), vadd(buf[896 + f], buf[1280 + f]));
        store_rows(PC, 128, n0, N, f, c);
    } else if (ks == 2 && f < 24) {
        VN p = vadd(vadd(buf[256 + f], buf[640 + f]), vadd(buf[1024 + f], buf[1408 + f]));
        p = vadd(p, vn(pts_b));
        sm.p[f] = p;
        store_rows(pts, 48, n0, N, f, p);
    }
    __syncthreads();
    if (threadIdx.x < 8 * NB) {            // (point q, residue i): p_glob = R p_loc + t
        int q = threadIdx.x / NB, i = threadIdx.x % NB;
        int n = n0 + i;
        if (n < N) {
            const float *fr = frames + (size_t)n * 12;
            float x = vcomp(sm.p[3 * q], i), y = vcomp(sm.p[3 * q + 1], i), z = vcomp(sm.p[3 * q + 2], i);
            for (int r = 0; r < 3; r++)
                pts[(size_t)n * 48 + 24 + 3 * q + r] = (fr[3 * r] * x + fr[3 * r + 1] * y + fr[3 * r + 2] * z) + fr[9 + r];
        }
    }
    __syncthreads();
}

// LAST_MODE is a template parameter so that the middle-layer variant (two of three launches) gets its own register
// allocation: as one function the decoder / step / embedding tail cost it ~30 spilled registers.
template <int LAST_MODE>
__global__ void __launch_bounds__(NT)
k_node_update_valu(NodeArgs A, UpdW W, float *chi, int step, int sde, const float *noise, int embed_next, PreW pre0,
                   StepScalars sp, TimeEmb te_next) {
    constexpr int last_mode = LAST_MODE;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    int flip = 0, rflip = 0;
    const int t = threadIdx.x, f = t & 127, ks = t >> 7, n0 = blockIdx.x * NB, N = A.N;
    const int kq = ks * 32;                  // this thread's quarter of a 128-deep reduction
    constexpr bool mid = last_mode == PP_NU_MID;
    WSet wa, wb, wc, wd, we;
    wload<32>(wa, W.outT, 128, f, kq, 0.f);
    wload<32>(wb, W.ffn_inT, 512, f, kq, 0.f);            // FFN-in, half 0: units f and f + 128
    wload<32>(wc, W.ffn_inT, 512, f + 128, kq, 0.f);
    // per-feature vectors and row inputs, all up front
    const float out_b = W.out_b[f], g0 = W.g0[f], b0 = W.b0[f], g1 = W.g1[f], b1 = W.b1[f], ffn_out_b = W.ffn_out_b[f];
    const float fib = W.ffn_in_b[t];
    const VN ms = load_rows(A.msum, 1, n0, N, 0);
    const VN hv = load_rows(A.hV, 128, n0, N, f);
    const VN rm = load_rows(A.rmask, 1, n0, N, 0);
    if (ks == 0) sm.a[f] = load_rows(A.S, 128, n0, N, f);
    __syncthreads();
    // mean_j mask_j (W_out y_j + b) = W_out mean_j(mask_j y_j) + b mean_j(mask_j)
    VN part = wdot<32>(wa, sm.a + kq, vn(0.f));
    wload<32>(wa, W.ffn_inT, 512, 256 + f, kq, part.g[0].x);           // FFN-in, units f + 256, f + 384
    wload<32>(wd, W.ffn_inT, 512, 256 + f + 128, kq, part.g[0].x);
    VN m = meet(sm, flip, part, 128, f, ks);
    m = vadd(m, vscale(ms, out_b));
    VN h1 = layernorm(sm, rflip, vadd(hv, m), g0, b0);
    if (ks == 0) sm.h[f] = h1;
    __syncthreads();
    // FFN 128 -> 512: thread (f, ks) builds the ks-th K-quarter of hidden units f, f+128, f+256, f+384 in one pass
    // over h1, then every thread owns one hidden unit
    {
        VN *buf = sm.part[flip];
        flip ^= 1;
        VN u0 = vn(0.f), u1 = vn(0.f), u2 = vn(0.f), u3 = vn(0.f);
        wdot4(wb, wc, wa, wd, sm.h + kq, u0, u1, u2, u3);
        buf[ks * 512 + f] = u0;
        buf[ks * 512 + f + 128] = u1;
        buf[ks * 512 + f + 256] = u2;
        buf[ks * 512 + f + 384] = u3;
        wload<32>(wb, W.ffn_outT, 128, f, ks * 128, u0.g[0].x);       // FFN-out: this thread's 128 inputs in four sets
        wload<32>(wc, W.ffn_outT, 128, f, ks * 128 + 32, u1.g[0].x);
        wload<32>(wa, W.ffn_outT, 128, f, ks * 128 + 64, u2.g[0].x);
        wload<32>(wd, W.ffn_outT, 128, f, ks * 128 + 96, u3.g[0].x);
        __syncthreads();
        VN hd = vadd(vadd(buf[t], buf[512 + t]), vadd(buf[1024 + t], buf[1536 + t]));
        sm.a[t] = vrelu(vadd(hd, vn(fib)));
    }
    __syncthreads();
    part = wdot<32>(wb, sm.a + ks * 128, vn(0.f));
    part = wdot<32>(wc, sm.a + ks * 128 + 32, part);
    part = wdot<32>(wa, sm.a + ks * 128 + 64, part);
    part = wdot<32>(wd, sm.a + ks * 128 + 96, part);
    // next phase's weights: the two message functions (middle layers) / decoder + next embedding (last layer)
    float in_b_1, pts_b_1 = 0.f, in_b_2 = 0.f, pts_b_2 = 0.f;
    float db0 = 0.f, db1 = 0.f, db2 = 0.f, db3 = 0.f;
    if constexpr (mid) {
        const float tk = part.g[0].x;
        wload<32>(wb, W.pre_edge.AT, 128, f, kq, tk);
        wload<32>(wc, W.pre_edge.CT, 128, f, kq, tk);
        // point weights: threads f < 24 take the edge message's column f, threads 32 <= f < 56 the next node message's
        // column f - 32, so one register set serves both and nothing is fetched between the two dot products
        if (f < 24) { wload<32>(we, W.pre_edge.ptsT, 24, f, kq, tk); pts_b_1 = W.pre_edge.pts_b[f]; pts_b_2 = W.pre_next.pts_b[f]; }
        else if (f >= 32 && f < 56) wload<32>(we, W.pre_next.ptsT, 24, f - 32, kq, tk);
        wload<32>(wa, W.pre_next.AT, 128, f, kq, tk);
        wload<32>(wd, W.pre_next.CT, 128, f, kq, tk);
        in_b_1 = W.pre_edge.in_b[f];
        in_b_2 = W.pre_next.in_b[f];
    } else {
        // decoder 128 -> 64 -> 32 -> 16 -> 4: K-quarters of 32 / 16 / 8 / 4 inputs
        const float tk = part.g[0].x;
        if (f < 64) { wload<32>(wb, W.d0_inT, 64, f, kq, tk); db0 = W.d0_in_b[f]; }
        if (f < 32) { wload<16, 0>(wc, W.d0_outT, 32, f, ks * 16, tk); db1 = W.d0_out_b[f]; }
        if (f < 16) { wload<8, 16>(wc, W.d2_inT, 16, f, ks * 8, tk); db2 = W.d2_in_b[f]; }
        if (f < 4) { wload<4, 24>(wc, W.d2_outT, 4, f, ks * 4, tk); db3 = W.d2_out_b[f]; }
        wload<32>(wa, pre0.AT, 128, f, kq, tk);
        wload<32>(wd, pre0.CT, 128, f, kq, tk);
        if (f < 24) { wload<32>(we, pre0.ptsT, 24, f, kq, tk); pts_b_1 = pre0.pts_b[f]; }
        in_b_1 = pre0.in_b[f];
    }
    VN o = meet(sm, flip, part, 128, f, ks);
    o = vadd(o, vn(ffn_out_b));
    VN h2 = layernorm(sm, rflip, vadd(h1, o), g1, b1);
    h2 = vmul(h2, rm);
    if (ks == 0) {
        store_rows(A.hV, 128, n0, N, f, h2);
        sm.h[f] = h2;
    }
    __syncthreads();
    if constexpr (mid) {
        // inputs of this layer's edge message and of the next layer's node message in one pass over h2:
        // columns PAe 0..127 | PCe 128..255 | PAn 256..383 | PCn 384..511 | ptsE 512..535 | ptsN 536..559
        VN *buf = sm.part[flip];
        flip ^= 1;
        VN u0 = vn(0.f), u1 = vn(0.f), u2 = vn(0.f), u3 = vn(0.f);
        wdot4(wb, wc, wa, wd, sm.h + kq, u0, u1, u2, u3);
        buf[ks * 576 + f] = u0;
        buf[ks * 576 + 128 + f] = u1;
        buf[ks * 576 + 256 + f] = u2;
        buf[ks * 576 + 384 + f] = u3;
        if (f < 24) buf[ks * 576 + 512 + f] = wdot<32>(we, sm.h + kq, vn(0.f));
        else if (f >= 32 && f < 56) buf[ks * 576 + 536 + (f - 32)] = wdot<32>(we, sm.h + kq, vn(0.f));
        __syncthreads();
        {
            const int c = ks * 128 + f;           // ks-group 0: PAe, 1: PCe, 2: PAn, 3: PCn
            VN a = vadd(vadd(buf[c], buf[576 + c]), vadd(buf[1152 + c], buf[1728 + c]));
            if (ks == 0) store_rows(A.PAe, 128, n0, N, f, vadd(a, vn(in_b_1)));
            else if (ks == 1) store_rows(A.PCe, 128, n0, N, f, a);
            else if (ks == 2) store_rows(A.PAn, 128, n0, N, f, vadd(a, vn(in_b_2)));
            else store_rows(A.PCn, 128, n0, N, f, a);
            if (ks < 2 && f < 24) {               // local points: ks-group 0 -> edge message, 1 -> next node message
                const int pc = 512 + 24 * ks + f;
                VN pl = vadd(vadd(buf[pc], buf[576 + pc]), vadd(buf[1152 + pc], buf[1728 + pc]));
                pl = vadd(pl, vn(ks == 0 ? pts_b_1 : pts_b_2));
                sm.p[24 * ks + f] = pl;
                store_rows(ks == 0 ? A.ptsE : A.ptsN, 48, n0, N, f, pl);
            }
        }
        __syncthreads();
        if (t < 16 * NB) {                        // (message m, point q, residue i): p_glob = R p_loc + t
            const int mm = t / (8 * NB), q = (t / NB) & 7, i = t % NB;
            const int n = n0 + i;
            if (n < N) {
                float *pts = mm == 0 ? A.ptsE : A.ptsN;
                const float *fr = A.frames + (size_t)n * 12;
                const float x = vcomp(sm.p[24 * mm + 3 * q], i), y = vcomp(sm.p[24 * mm + 3 * q + 1], i),
                            z = vcomp(sm.p[24 * mm + 3 * q + 2], i);
                for (int r = 0; r < 3; r++)
                    pts[(size_t)n * 48 + 24 + 3 * q + r] = (fr[3 * r] * x + fr[3 * r + 1] * y + fr[3 * r + 2] * z) + fr[9 + r];
            }
        }
        return;
    }
    // decoder: 128 -> 64 -> 32 -> relu -> 16 -> 4 (weights in wb / wc)
    VN v;
    {
        VN pp = f < 64 ? wdot<32>(wb, sm.h + kq, vn(0.f)) : vn(0.f);
        // wb is free: the next embedding's 30 dense rows (14 angle features + 16 time features)
        if (embed_next) {
            int off = 21 * 128 + f;
            asm volatile("" : "+v"(off) : "v"(pp.g[0].x));
#pragma unroll
            for (int i = 0; i < 30; i++) wb.v[i] = A.embT[off + i * 128];
            wb.v[30] = A.emb_b[f];
        }
        VN r = meet(sm, flip, pp, 128, f, ks);
        v = f < 64 ? vrelu(vadd(r, vn(db0))) : vn(0.f);
    }
    if (ks == 0 && f < 64) sm.a[f] = v;
    __syncthreads();
    {
        VN pp = f < 32 ? wdot<16, 0>(wc, sm.a + ks * 16, vn(0.f)) : vn(0.f);
        VN r = meet(sm, flip, pp, 128, f, ks);
        v = f < 32 ? vrelu(vadd(r, vn(db1))) : vn(0.f);
    }
    if (ks == 0 && f < 32) sm.a[64 + f] = v;
    __syncthreads();
    {
        VN pp = f < 16 ? wdot<8, 16>(wc, sm.a + 64 + ks * 8, vn(0.f)) : vn(0.f);
        VN r = meet(sm, flip, pp, 128, f, ks);
        v = f < 16 ? vrelu(vadd(r, vn(db2))) : vn(0.f);
    }
    if (ks == 0 && f < 16) sm.a[96 + f] = v;
    __syncthreads();
    {
        VN pp = f < 4 ? wdot<4, 24>(wc, sm.a + 96 + ks * 4, vn(0.f)) : vn(0.f);
        VN r = meet(sm, flip, pp, 128, f, ks);
        v = f < 4 ? vadd(r, vn(db3)) : vn(0.f);
    }
    if (ks == 0 && f < 4) {
        sm.a[112 + f] = v;
        store_rows(A.score, 4, n0, N, f, v);
    }
    __syncthreads();
    if constexpr (last_mode != PP_NU_STEP) return;
    // reverse step on (residue i, chi k) = 4 NB threads
    if (t < 4 * NB) {
        int i = t >> 2, k = t & 3, n = n0 + i;
        if (n < N) {
            float x = chi[(size_t)n * 4 + k];
            float sw = vcomp(sm.a[112 + k], i) * sp.w;
            bool m1 = A.m1pi[(size_t)n * 4 + k] != 0, m2 = A.m2pi[(size_t)n * 4 + k] != 0;
            float y = x;
            if (!sde) {
                if (m1 || m2) y = x + sp.c_ode * sw;
            } else {
                size_t NN = (size_t)N * 4;
                const float *nz = noise + (size_t)step * 2 * NN;
                if (m1) y = x + (sp.c_drift * sw + sp.c_diff * nz[(size_t)n * 4 + k]);
                if (m2) y = y + (sp.c_drift * sw + sp.c_diff * nz[NN + (size_t)n * 4 + k]);
            }
            y = wrap_pi(y) * A.sc_mask[(size_t)n * 4 + k];
            chi[(size_t)n * 4 + k] = y;
        }
    }
    __syncthreads();
    if (!embed_next) return;
    // next step's node embedding (embed_pre with the dense rows already in wb)
    {
        if (t < 6) sm.p[t] = load_rows(A.bb_sincos, 6, n0, N, t);
        else if (t < 14) {
            int k = (t - 6) >> 1, sc = (t - 6) & 1;
            VN x = load_rows(chi, 4, n0, N, k), mk = load_rows(A.sc_mask, 4, n0, N, k), sv;
            VN_FOR sv.g[gi] = sc ? f4v{cosf(x.g[gi].x), cosf(x.g[gi].y), cosf(x.g[gi].z), cosf(x.g[gi].w)}
                                 : f4v{sinf(x.g[gi].x), sinf(x.g[gi].y), sinf(x.g[gi].z), sinf(x.g[gi].w)};
            sm.p[t] = vmul(sv, mk);
        }
        VN acc = vn(wb.v[30]);
        VN_FOR {
            const int b = n0 + 4 * gi;
            const int t0 = b + 0 < N ? (int)A.rtype[b + 0] : 0, t1 = b + 1 < N ? (int)A.rtype[b + 1] : 0;
            const int t2 = b + 2 < N ? (int)A.rtype[b + 2] : 0, t3 = b + 3 < N ? (int)A.rtype[b + 3] : 0;
            acc.g[gi].x += A.embT[t0 * 128 + f]; acc.g[gi].y += A.embT[t1 * 128 + f];
            acc.g[gi].z += A.embT[t2 * 128 + f]; acc.g[gi].w += A.embT[t3 * 128 + f];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 14; k++) acc = vfma(wb.v[k], sm.p[k], acc);
        const float *te = te_next.v;
        float tacc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) tacc = fmaf(wb.v[14 + k], te[k], tacc);
        VN e = vadd(acc, vn(tacc));
        VN h = layernorm(sm, rflip, e, A.emb_g[f], A.emb_beta[f]);
        if (ks == 0) {
            store_rows(A.hV, 128, n0, N, f, h);
            sm.h[f] = h;
        }
        __syncthreads();
    }
    message_inputs_pre(sm, flip, wa, wd, we, nullptr, in_b_1, pts_b_1, A.frames, n0, N, A.ptsN, A.PAn, A.PCn);
}

#endif      // !PP_EDGE_F16

// ---------------------------------------------------------------------------------------------
static NodeArgs make_args(pp_ctx *c) {
    const pp_plan *p = c->plan;
    NodeArgs A;
    A.N = c->N;
    A.rmask = c->b.residue_mask;
    A.rtype = c->b.residue_type;
    A.bb_sincos = c->b.BB_D_sincos;
    A.sc_mask = c->b.SC_D_mask;
    A.m1pi = c->b.chi_1pi_periodic_mask;
    A.m2pi = c->b.chi_2pi_periodic_mask;
    A.frames = c->frames;
    A.embT = p->node_emb_T;
    A.emb_b = p->w + p->off.node_emb_b;
    A.emb_g = p->w + p->off.norm_nodes_g;
    A.emb_beta = p->w + p->off.norm_nodes_b;
    A.hV = c->hV; A.S = c->S; A.msum = c->msum;
    A.ptsN = c->ptsN; A.PAn = c->PAn; A.PCn = c->PCn;
    A.ptsE = c->ptsE; A.PAe = c->PAe; A.PCe = c->PCe;
    A.score = c->score;
    return A;
}
static PreW make_pre(const pp_plan *p, int layer, bool edge) {
    const LayerOff &o = p->off.layer[layer];
    const LayerT &t = p->lt[layer];
    PreW w;
    w.ptsT = edge ? t.pts_edge_wT : t.pts_node_wT;
    w.pts_b = p->w + (edge ? o.pts_edge_b : o.pts_node_b);
    w.AT = edge ? t.em_A_T : t.nm_A_T;
    w.CT = edge ? t.em_C_T : t.nm_C_T;
    w.in_b = p->w + (edge ? o.em_in_b : o.nm_in_b);
    return w;
}

typedef void (*nu_kernel_t)(NUpdArgs, float *, int, int, const float *, int, StepScalars, TimeEmb);
// mode 0 / 1 / 2 = PP_NU_MID / PP_NU_STEP / PP_NU_SCORE; multi: more tiles than CUs (shallower ring, two workgroups per CU)
static nu_kernel_t nu_kernel(int mode, bool multi) {
    if (multi)
        return mode == 0 ? k_node_update<PP_NU_MID, PP_NU_DEPTH_MULTI> : mode == 1 ? k_node_update<PP_NU_STEP, PP_NU_DEPTH_MULTI>
                                                                                    : k_node_update<PP_NU_SCORE, PP_NU_DEPTH_MULTI>;
    return mode == 0 ? k_node_update<PP_NU_MID, PP_NU_DEPTH> : mode == 1 ? k_node_update<PP_NU_STEP, PP_NU_DEPTH>
                                                                          : k_node_update<PP_NU_SCORE, PP_NU_DEPTH>;
}
static nu_kernel_t nu_kernel_split(int cl) {
    return cl == 4 ? k_node_update<PP_NU_MID, PP_NU_DEPTH, 4> : k_node_update<PP_NU_MID, PP_NU_DEPTH, 2>;
}
static int g_nu_cus = 0;
static pp_status node_attrs() {
    static bool done = false;
    if (!done) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_embed),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)));
        for (int multi = 0; multi < 2; multi++)
            for (int mode = 0; mode < 3; mode++)
                PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nu_kernel(mode, multi != 0)),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemU)));
        for (int cl = 2; cl <= 4; cl += 2)
            PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nu_kernel_split(cl)),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemU)));
#ifndef PP_EDGE_F16
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_update_valu<PP_NU_MID>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)));
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_update_valu<PP_NU_STEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)));
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_update_valu<PP_NU_SCORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)));
#endif
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_nu_cus = prop.multiProcessorCount;
        if (g_nu_cus <= 0) g_nu_cus = 256;
        done = true;
    }
    return PP_OK;
}

pp_status pp_launch_node_embed(pp_ctx *c, const float *chi, const StepParams &sp, hipStream_t s) {
    pp_status st = node_attrs();
    if (st != PP_OK) return st;
    NodeArgs A = make_args(c);
    PreW pre0 = make_pre(c->plan, 0, false);
    hipLaunchKernelGGL(k_node_embed, dim3((c->N + NB - 1) / NB), dim3(NT), sizeof(Smem), s, A, pre0, chi, sp);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_node_update(pp_ctx *c, int layer, int last_mode, float *chi, int step, int mode,
                                const float *noise, const StepParams *cur, const StepParams *next, hipStream_t s) {
    const bool embed_next_step = next != nullptr;
    pp_status st0 = node_attrs();
    if (st0 != PP_OK) return st0;
    if ((last_mode == PP_NU_MID) != (layer < 2)) {
        pp_set_error("pp_launch_node_update: layers 0 and 1 are middle layers, layer 2 is the last one");
        return PP_ERR_INVALID;
    }
    const pp_plan *p = c->plan;
    const LayerT &t = p->lt[layer];
    NUpdArgs A;
    A.N = c->N;
    A.rmask = c->b.residue_mask;
    A.rtype = c->b.residue_type;
    A.bb_sincos = c->b.BB_D_sincos;
    A.sc_mask = c->b.SC_D_mask;
    A.m1pi = c->b.chi_1pi_periodic_mask;
    A.m2pi = c->b.chi_2pi_periodic_mask;
    A.frames = c->frames;
    A.embT = p->node_emb_T;
    A.wstream = t.nu_stream;
    A.params = t.nu_params;
    A.hV = c->hV; A.hV_out = c->hV; A.S = c->S; A.msum = c->msum;
    A.ptsN = c->ptsN; A.PAn = c->PAn; A.PCn = c->PCn;
    A.ptsE = c->ptsE; A.PAe = c->PAe; A.PCe = c->PCe;
    A.score = c->score;
    A.sat = c->sat;
    int embed_next = (last_mode == PP_NU_STEP && embed_next_step) ? 1 : 0;    // node embedding for step + 1 afterwards
    const dim3 grid((c->N + 15) / 16), block(512);
    const int sde = mode == PP_MODE_SDE ? 1 : 0;
    StepScalars sp = {0.f, 0.f, 0.f, 0.f};
    TimeEmb te = {};
    if (cur) sp = {cur->c_ode, cur->w, cur->c_drift, cur->c_diff};
    if (next) memcpy(te.v, next->temb, sizeof(te.v));
#ifndef PP_EDGE_F16
    {       // exact-fp32 library: the VALU kernel (PP_NODE_F16=1 in the environment runs the matrix-pipe kernel here too, for A/B runs)
        static const bool f16_node = PP_GETENV("PP_NODE_F16") != nullptr;
        if (!f16_node) {
            const LayerOff &o = p->off.layer[layer];
            NodeArgs NA = make_args(c);
            UpdW W;
            W.outT = t.nm_out_T; W.out_b = p->w + o.nm_out_b;
            W.g0 = p->w + o.norm_g[0]; W.b0 = p->w + o.norm_b[0];
            W.g1 = p->w + o.norm_g[1]; W.b1 = p->w + o.norm_b[1];
            W.ffn_inT = t.nd_in_T; W.ffn_in_b = p->w + o.nd_in_b;
            W.ffn_outT = t.nd_out_T; W.ffn_out_b = p->w + o.nd_out_b;
            W.pre_edge = make_pre(p, layer, true);
            W.pre_next = make_pre(p, layer < 2 ? layer + 1 : 0, false);
            W.d0_inT = p->d0_in_T; W.d0_in_b = p->w + p->off.d0_in_b;
            W.d0_outT = p->d0_out_T; W.d0_out_b = p->w + p->off.d0_out_b;
            W.d2_inT = p->d2_in_T; W.d2_in_b = p->w + p->off.d2_in_b;
            W.d2_outT = p->d2_out_T; W.d2_out_b = p->w + p->off.d2_out_b;
            const PreW pre0 = make_pre(p, 0, false);
            const dim3 vgrid((c->N + NB - 1) / NB), vblock(NT);
            if (last_mode == PP_NU_MID)
                PP_LAUNCH(c, k_node_update_valu<PP_NU_MID>, vgrid, vblock, sizeof(Smem), s, NA, W, chi, step, sde, noise, embed_next, pre0, sp, te);
            else if (last_mode == PP_NU_STEP)
                PP_LAUNCH(c, k_node_update_valu<PP_NU_STEP>, vgrid, vblock, sizeof(Smem), s, NA, W, chi, step, sde, noise, embed_next, pre0, sp, te);
            else
                PP_LAUNCH(c, k_node_update_valu<PP_NU_SCORE>, vgrid, vblock, sizeof(Smem), s, NA, W, chi, step, sde, noise, embed_next, pre0, sp, te);
            PP_HIP_CHECK(hipGetLastError());
            return PP_OK;
        }
    }
#endif
    const bool multi = (int)grid.x > g_nu_cus;
    // middle layers of a launch that leaves most CUs idle: 4 (or 2) workgroups per tile share out the projections
    static const char *split_env = PP_GETENV("PP_NU_SPLIT");
    static const int split_max = split_env ? atoi(split_env) : 4;     // measurement aid: 1 = never
    const int tiles = (int)grid.x;
    const int cl = last_mode != PP_NU_MID ? 1 : (split_max >= 4 && 4 * tiles <= g_nu_cus) ? 4 : (split_max >= 2 && 2 * tiles <= g_nu_cus) ? 2 : 1;
    if (cl > 1) {
        // the tile's workgroups all read the old h_V and one of them writes the new one: into the other buffer, so that a
        // workgroup that starts late (a shared GPU, a busy chip) still reads what it must; the context's pointers swap
        A.hV_out = c->hV_alt;
        std::swap(c->hV, c->hV_alt);
        PP_LAUNCH(c, nu_kernel_split(cl), dim3(tiles * cl), block, sizeof(SmemU), s, A, chi, step, sde, noise, embed_next, sp, te);
        PP_HIP_CHECK(hipGetLastError());
        return PP_OK;
    }
    const nu_kernel_t kern = nu_kernel(last_mode == PP_NU_MID ? 0 : last_mode == PP_NU_STEP ? 1 : 2, multi);
    PP_LAUNCH(c, kern, grid, block, sizeof(SmemU), s, A, chi, step, sde, noise, embed_next, sp, te);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

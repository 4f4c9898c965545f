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

#include "pp_node_update.h"      // wrap_pi, the matrix-pipe node update (node_update_body, k_node_update) and its helpers

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
#ifdef PP_X_NOACT             // timing-only ablation: no activation reads from LDS
        acc = vfma(w.v[SRC0 + i], acc, acc);
#else
        acc = vfma(w.v[SRC0 + i], act[i], acc);
#endif
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
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 1
    return;
#endif
    VN h1 = layernorm(sm, rflip, vadd(hv, m), g0, b0);
    if (ks == 0) sm.h[f] = h1;
    __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 2
    return;
#endif
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
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 3
    return;
#endif
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
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 4
    return;
#endif
    VN h2 = layernorm(sm, rflip, vadd(h1, o), g1, b1);
    h2 = vmul(h2, rm);
    if (ks == 0) {
        store_rows(A.hV, 128, n0, N, f, h2);
        sm.h[f] = h2;
    }
    __syncthreads();
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 5
    return;
#endif
#if defined(PP_X_NU_STOP) && PP_X_NU_STOP == 6
    if (!mid) return;
#endif
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

// the kernel's view of the context (in-place h_V: split launches redirect hV_out)
NUpdArgs pp_node_update_args(pp_ctx *c, int layer) {
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
    return A;
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
    NUpdArgs A = pp_node_update_args(c, layer);
    int embed_next = (last_mode == PP_NU_STEP && embed_next_step) ? 1 : 0;    // node embedding for step + 1 afterwards
#ifdef PP_X_NU_EMBED_LAUNCH      /* experiment: the next step's embedding as its own launch (k_node_embed) */
    const bool embed_after = embed_next != 0;
    embed_next = 0;
#endif
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
#ifdef PP_X_NU_EMBED_LAUNCH
    if (embed_after) return pp_launch_node_embed(c, chi, *next, s);
#endif
    return PP_OK;
}

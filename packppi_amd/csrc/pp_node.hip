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
// latency: a block of 512 threads owns 4 consecutive nodes; thread (f = tid & 127, ks = tid >> 7) owns
// output feature f for all four nodes over the ks-th quarter of the reduction dimension; the four
// partial sums meet in LDS.  Weights are read from transposed copies ([in][out]) so a wave's loads are
// contiguous; activations sit in LDS as float4 (one component per node).  The cheap epilogues
// (LayerNorm, bias, ReLU) are replicated in the four ks-groups to keep control flow uniform.
#include "pp_internal.h"

#define NT 512
#define NB 4

struct NodeArgs {
    int N;
    const float *rmask;          // [N]
    const int64_t *rtype;        // [N]
    const float *bb_sincos;      // [N][6]
    const float *sc_mask;        // [N][4]
    const uint8_t *m1pi, *m2pi;  // [N][4]
    const float *frames;         // [N][12]
    const StepParams *steps;
    const float *embT, *emb_b, *emb_g, *emb_beta;
    float *hV, *S, *msum;
    float *ptsN, *PAn, *PCn, *ptsE, *PAe, *PCe;
    float *score;
};

struct PreW {            // weights feeding one message function
    const float *ptsT, *pts_b;      // [128][24], [24]
    const float *AT, *CT, *in_b;    // [128][128] x2, [128]
};

struct UpdW {
    const float *outT, *out_b;      // node_message_fn.W_out^T [128][128], bias
    const float *g0, *b0, *g1, *b1; // norm.0 / norm.1
    const float *ffn_inT, *ffn_in_b, *ffn_outT, *ffn_out_b;   // [128][512],[512],[512][128],[128]
    PreW pre_edge, pre_next;
    const float *d0_inT, *d0_in_b, *d0_outT, *d0_out_b, *d2_inT, *d2_in_b, *d2_outT, *d2_out_b;
};

struct Smem {
    float4 part[2][4 * 512];   // ping-pong partial sums  (64 KB)
    float4 a[512];             // wide activation vector (FFN hidden, decoder scratch)
    float4 h[128];             // current node vector
    float4 p[24];              // local points / small inputs
    float4 red[2][8];          // LayerNorm partials, ping-pong
    int flip, rflip;
};

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 fma4(float w, float4 a, float4 c) {
    return make_float4(fmaf(w, a.x, c.x), fmaf(w, a.y, c.y), fmaf(w, a.z, c.z), fmaf(w, a.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float comp(float4 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// partial sum over the ks-th quarter of the reduction dimension
template <int KIN>
__device__ __forceinline__ float4 dense_slice(const float *__restrict__ WT, int ldo, int col, const float4 *act, int ks) {
    constexpr int KL = KIN / 4;
    const float *w = WT + (size_t)(ks * KL) * ldo + col;
    const float4 *a = act + ks * KL;
    float4 acc = f4(0.f);
#pragma unroll 16
    for (int i = 0; i < KL; i++) acc = fma4(w[(size_t)i * ldo], a[i], acc);
    return acc;
}

// meet the four K-slices: every thread returns the full sum for its column (one barrier, ping-pong buffer)
__device__ __forceinline__ float4 meet(Smem &sm, int &flip, float4 partial, int stride, int col, int ks) {
    float4 *buf = sm.part[flip];
    flip ^= 1;
    buf[ks * stride + col] = partial;
    __syncthreads();
    return add4(add4(buf[col], buf[stride + col]), add4(buf[2 * stride + col], buf[3 * stride + col]));
}

__device__ __forceinline__ float4 wave_sum4(float4 v) {
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
    }
    return v;
}

// LayerNorm over 128 features (one per thread of a ks-group = 2 waves), four nodes at once, eps 1e-5, two-pass
__device__ __forceinline__ float4 layernorm4(Smem &sm, int &rflip, float4 v, float g, float b) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, grp = wid & ~1;
    float4 s = wave_sum4(v);
    float4 *r = sm.red[rflip];
    rflip ^= 1;
    if (lane == 0) r[wid] = s;
    __syncthreads();
    const float4 mean = scale4(add4(r[grp], r[grp + 1]), 1.f / 128.f);
    const float4 d = sub4(v, mean);
    float4 q = wave_sum4(mul4(d, d));
    r = sm.red[rflip];
    rflip ^= 1;
    if (lane == 0) r[wid] = q;
    __syncthreads();
    const float4 var = scale4(add4(r[grp], r[grp + 1]), 1.f / 128.f);
    const float4 rs = make_float4(1.f / sqrtf(var.x + 1e-5f), 1.f / sqrtf(var.y + 1e-5f), 1.f / sqrtf(var.z + 1e-5f),
                                  1.f / sqrtf(var.w + 1e-5f));
    return make_float4(d.x * rs.x * g + b, d.y * rs.y * g + b, d.z * rs.z * g + b, d.w * rs.w * g + b);
}

__device__ __forceinline__ void store_rows(float *dst, int ld, int n0, int N, int col, float4 v) {
    if (n0 + 0 < N) dst[(size_t)(n0 + 0) * ld + col] = v.x;
    if (n0 + 1 < N) dst[(size_t)(n0 + 1) * ld + col] = v.y;
    if (n0 + 2 < N) dst[(size_t)(n0 + 2) * ld + col] = v.z;
    if (n0 + 3 < N) dst[(size_t)(n0 + 3) * ld + col] = v.w;
}
__device__ __forceinline__ float4 load_rows(const float *src, int ld, int n0, int N, int col) {
    float4 v = f4(0.f);
    if (n0 + 0 < N) v.x = src[(size_t)(n0 + 0) * ld + col];
    if (n0 + 1 < N) v.y = src[(size_t)(n0 + 1) * ld + col];
    if (n0 + 2 < N) v.z = src[(size_t)(n0 + 2) * ld + col];
    if (n0 + 3 < N) v.w = src[(size_t)(n0 + 3) * ld + col];
    return v;
}

// Inputs of one message function from h = sm.h: points (local + global), W_A h + b, W_C h.
// One meeting round for all 128 + 128 + 24 outputs.
__device__ void message_inputs(Smem &sm, int &flip, const PreW &w, const float *frames, int n0, int N,
                               float *pts, float *PA, float *PC) {
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7;
    float4 pa = dense_slice<128>(w.AT, 128, f, sm.h, ks);
    float4 pc = dense_slice<128>(w.CT, 128, f, sm.h, ks);
    float4 pp = f < 24 ? dense_slice<128>(w.ptsT, 24, f, sm.h, ks) : f4(0.f);
    float4 *buf = sm.part[flip];
    flip ^= 1;
    buf[ks * 384 + f] = pa;
    buf[ks * 384 + 128 + f] = pc;
    if (f < 24) buf[ks * 384 + 256 + f] = pp;
    __syncthreads();
    if (ks == 0) {
        float4 a = add4(add4(buf[f], buf[384 + f]), add4(buf[768 + f], buf[1152 + f]));
        store_rows(PA, 128, n0, N, f, add4(a, f4(w.in_b[f])));
    } else if (ks == 1) {
        float4 c = add4(add4(buf[128 + f], buf[512 + f]), add4(buf[896 + f], buf[1280 + f]));
        store_rows(PC, 128, n0, N, f, c);
    } else if (ks == 2 && f < 24) {
        float4 p = add4(add4(buf[256 + f], buf[640 + f]), add4(buf[1024 + f], buf[1408 + f]));
        p = add4(p, f4(w.pts_b[f]));
        sm.p[f] = p;
        store_rows(pts, 48, n0, N, f, p);
    }
    __syncthreads();
    if (threadIdx.x < 32) {            // (point q, node i): p_glob = R p_loc + t
        int q = threadIdx.x >> 2, i = threadIdx.x & 3;
        int n = n0 + i;
        if (n < N) {
            const float *fr = frames + (size_t)n * 12;
            float x = comp(sm.p[3 * q], i), y = comp(sm.p[3 * q + 1], i), z = comp(sm.p[3 * q + 2], i);
            for (int r = 0; r < 3; r++)
                pts[(size_t)n * 48 + 24 + 3 * q + r] = (fr[3 * r] * x + fr[3 * r + 1] * y + fr[3 * r + 2] * z) + fr[9 + r];
        }
    }
    __syncthreads();
}

// Node embedding for 4 nodes -> float4 (before LN) for feature f (all ks-groups compute the same thing)
__device__ __forceinline__ float4 embed_pre(Smem &sm, const NodeArgs &A, const float *chi, int step, int n0) {
    const int t = threadIdx.x, f = t & 127;
    if (t < 6) sm.p[t] = load_rows(A.bb_sincos, 6, n0, A.N, t);
    else if (t < 14) {
        int k = (t - 6) >> 1, sc = (t - 6) & 1;
        float4 x = load_rows(chi, 4, n0, A.N, k), m = load_rows(A.sc_mask, 4, n0, A.N, k);
        float4 v = sc ? make_float4(cosf(x.x), cosf(x.y), cosf(x.z), cosf(x.w))
                      : make_float4(sinf(x.x), sinf(x.y), sinf(x.z), sinf(x.w));
        sm.p[t] = mul4(v, m);
    }
    __syncthreads();
    float4 acc = f4(A.emb_b[f]);
    int t0 = n0 + 0 < A.N ? (int)A.rtype[n0 + 0] : 0, t1 = n0 + 1 < A.N ? (int)A.rtype[n0 + 1] : 0;
    int t2 = n0 + 2 < A.N ? (int)A.rtype[n0 + 2] : 0, t3 = n0 + 3 < A.N ? (int)A.rtype[n0 + 3] : 0;
    acc = add4(acc, make_float4(A.embT[t0 * 128 + f], A.embT[t1 * 128 + f], A.embT[t2 * 128 + f], A.embT[t3 * 128 + f]));
#pragma unroll
    for (int k = 0; k < 14; k++) acc = fma4(A.embT[(21 + k) * 128 + f], sm.p[k], acc);
    const float *te = A.steps[step].temb;
    float tacc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) tacc = fmaf(A.embT[(35 + k) * 128 + f], te[k], tacc);
    return add4(acc, f4(tacc));
}

__global__ void __launch_bounds__(NT)
k_node_embed(NodeArgs A, PreW pre0, const float *chi, int step) {
    __shared__ Smem sm;
    int flip = 0, rflip = 0;
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7, n0 = blockIdx.x * NB;
    float4 v = embed_pre(sm, A, chi, step, n0);
    float4 h = layernorm4(sm, rflip, v, A.emb_g[f], A.emb_beta[f]);
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

// small dense layer used by the decoder: width <= 128 outputs, K split over the four ks-groups
template <int KIN>
__device__ __forceinline__ float4 dense_small(Smem &sm, int &flip, const float *WT, int width, const float4 *act,
                                              const float *bias) {
    const int f = threadIdx.x & 127, ks = threadIdx.x >> 7;
    float4 p = f < width ? dense_slice<KIN>(WT, width, f, act, ks) : f4(0.f);
    float4 r = meet(sm, flip, p, 128, f, ks);
    return f < width ? add4(r, f4(bias[f])) : f4(0.f);
}

__global__ void __launch_bounds__(NT)
k_node_update(NodeArgs A, UpdW W, int last_mode, float *chi, int step, int sde, const float *noise, int embed_next,
              PreW pre0) {
    __shared__ Smem sm;
    int flip = 0, rflip = 0;
    const int t = threadIdx.x, f = t & 127, ks = t >> 7, n0 = blockIdx.x * NB, N = A.N;
    if (ks == 0) sm.a[f] = load_rows(A.S, 128, n0, N, f);
    __syncthreads();
    float4 ms = f4(0.f);
    if (n0 + 0 < N) ms.x = A.msum[n0 + 0];
    if (n0 + 1 < N) ms.y = A.msum[n0 + 1];
    if (n0 + 2 < N) ms.z = A.msum[n0 + 2];
    if (n0 + 3 < N) ms.w = A.msum[n0 + 3];
    // mean_j mask_j (W_out y_j + b) = W_out mean_j(mask_j y_j) + b mean_j(mask_j)
    float4 m = meet(sm, flip, dense_slice<128>(W.outT, 128, f, sm.a, ks), 128, f, ks);
    m = add4(m, scale4(ms, W.out_b[f]));
    float4 h0 = load_rows(A.hV, 128, n0, N, f);
    float4 h1 = layernorm4(sm, rflip, add4(h0, m), W.g0[f], W.b0[f]);
    if (ks == 0) sm.h[f] = h1;
    __syncthreads();
    // FFN 128 -> 512: each thread builds the ks-th K-quarter of 4 hidden units, then owns hidden unit t
    {
        float4 *buf = sm.part[flip];
        flip ^= 1;
#pragma unroll
        for (int j = 0; j < 4; j++) buf[ks * 512 + f + 128 * j] = dense_slice<128>(W.ffn_inT, 512, f + 128 * j, sm.h, ks);
        __syncthreads();
        float4 hd = add4(add4(buf[t], buf[512 + t]), add4(buf[1024 + t], buf[1536 + t]));
        sm.a[t] = relu4(add4(hd, f4(W.ffn_in_b[t])));
        __syncthreads();
    }
    float4 o = meet(sm, flip, dense_slice<512>(W.ffn_outT, 128, f, sm.a, ks), 128, f, ks);
    o = add4(o, f4(W.ffn_out_b[f]));
    float4 h2 = layernorm4(sm, rflip, add4(h1, o), W.g1[f], W.b1[f]);
    h2 = mul4(h2, load_rows(A.rmask, 1, n0, N, 0));
    if (ks == 0) {
        store_rows(A.hV, 128, n0, N, f, h2);
        sm.h[f] = h2;
    }
    __syncthreads();
    if (last_mode == PP_NU_MID) {
        message_inputs(sm, flip, W.pre_edge, A.frames, n0, N, A.ptsE, A.PAe, A.PCe);
        message_inputs(sm, flip, W.pre_next, A.frames, n0, N, A.ptsN, A.PAn, A.PCn);
        return;
    }
    // decoder: 128 -> 64 -> 32 -> relu -> 16 -> 4
    float4 v = relu4(dense_small<128>(sm, flip, W.d0_inT, 64, sm.h, W.d0_in_b));
    if (ks == 0 && f < 64) sm.a[f] = v;
    __syncthreads();
    v = relu4(dense_small<64>(sm, flip, W.d0_outT, 32, sm.a, W.d0_out_b));
    if (ks == 0 && f < 32) sm.a[64 + f] = v;
    __syncthreads();
    v = relu4(dense_small<32>(sm, flip, W.d2_inT, 16, sm.a + 64, W.d2_in_b));
    if (ks == 0 && f < 16) sm.a[96 + f] = v;
    __syncthreads();
    v = dense_small<16>(sm, flip, W.d2_outT, 4, sm.a + 96, W.d2_out_b);
    if (ks == 0 && f < 4) {
        sm.a[112 + f] = v;
        store_rows(A.score, 4, n0, N, f, v);
    }
    __syncthreads();
    if (last_mode != PP_NU_STEP) return;
    // reverse step on (node i, chi k) = 16 threads
    if (t < 16) {
        int i = t >> 2, k = t & 3, n = n0 + i;
        if (n < N) {
            const StepParams sp = A.steps[step];
            float x = chi[(size_t)n * 4 + k];
            float sw = comp(sm.a[112 + k], i) * sp.w;
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
    float4 e = embed_pre(sm, A, chi, step + 1, n0);
    float4 h = layernorm4(sm, rflip, e, A.emb_g[f], A.emb_beta[f]);
    if (ks == 0) {
        store_rows(A.hV, 128, n0, N, f, h);
        sm.h[f] = h;
    }
    __syncthreads();
    message_inputs(sm, flip, pre0, A.frames, n0, N, A.ptsN, A.PAn, A.PCn);
}

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
    A.steps = c->steps;
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

pp_status pp_launch_node_embed(pp_ctx *c, const float *chi, int step, hipStream_t s) {
    NodeArgs A = make_args(c);
    PreW pre0 = make_pre(c->plan, 0, false);
    hipLaunchKernelGGL(k_node_embed, dim3((c->N + NB - 1) / NB), dim3(NT), 0, s, A, pre0, chi, step);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_node_update(pp_ctx *c, int layer, int last_mode, float *chi, int step, int mode,
                                const float *noise, hipStream_t s) {
    const pp_plan *p = c->plan;
    const LayerOff &o = p->off.layer[layer];
    const LayerT &t = p->lt[layer];
    NodeArgs A = make_args(c);
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
    PreW pre0 = make_pre(p, 0, false);
    // step >= 0: re-embed for step + 1 afterwards; step < 0 encodes "last step (-step-1), no re-embed"
    int embed_next = (last_mode == PP_NU_STEP && step >= 0) ? 1 : 0;
    int st = step;
    if (last_mode == PP_NU_STEP && step < 0) { st = -step - 1; embed_next = 0; }
    hipLaunchKernelGGL(k_node_update, dim3((c->N + NB - 1) / NB), dim3(NT), 0, s, A, W, last_mode, chi, st,
                       mode == PP_MODE_SDE ? 1 : 0, noise, embed_next, pre0);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

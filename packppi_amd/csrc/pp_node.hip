// Node-level stages of the score network (everything that is [B*L, 128]-shaped).
//
//   k_node_embed  : node features -> Linear(51,128) -> LN              (encoder.py:218-242, layers.py:257-268)
//                   + inputs of the layer-0 node message
//   k_node_update : mean message -> W_out -> LN -> FFN -> LN -> mask   (layers.py:124-132)
//                   + inputs of this layer's edge message and the next layer's node message,
//                   or (last layer) decoder (TorsionalDiffusion.py:105-109), reverse step
//                   (schedule.py:198-235, TorsionalDiffusion.py:268-280) and the next embedding.
//
// Blocks of 128 threads own 4 consecutive nodes; thread f owns output feature f for all four, so
// every weight (transposed copy, [in][out]) is read once per block, coalesced, and reused 4x.
// Activations sit in LDS as float4 (one component per node).
#include "pp_internal.h"

#define NT 128
#define NB 4

struct NodeArgs {
    int N;
    const float *rmask;          // [N]
    // embedding inputs
    const int64_t *rtype;        // [N]
    const float *bb_sincos;      // [N][6]
    const float *sc_mask;        // [N][4]
    const uint8_t *m1pi, *m2pi;  // [N][4]
    const float *frames;         // [N][12]
    const StepParams *steps;
    // weights
    const float *embT, *emb_b, *emb_g, *emb_beta;
    // state
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

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 fma4(float w, float4 a, float4 c) {
    return make_float4(fmaf(w, a.x, c.x), fmaf(w, a.y, c.y), fmaf(w, a.z, c.z), fmaf(w, a.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaf(0.f, 0.f, fmaxf(a.z, 0.f)), fmaxf(a.w, 0.f)); }

// acc += sum_k WT[k*ldo + col] * act[k]
template <int KIN>
__device__ __forceinline__ float4 dense4(const float *__restrict__ WT, int ldo, int col, const float4 *act, float4 acc) {
#pragma unroll 8
    for (int k = 0; k < KIN; k++) acc = fma4(WT[(size_t)k * ldo + col], act[k], acc);
    return acc;
}

__device__ __forceinline__ float4 wave_sum4(float4 v) {
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
    }
    return v;
}

// LayerNorm over the 128 features held one-per-thread, four nodes at once (eps 1e-5, biased variance).
__device__ __forceinline__ float4 layernorm4(float4 v, float g, float b, float4 *s_red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4 s = wave_sum4(v);
    if (lane == 0) s_red[wid] = s;
    __syncthreads();
    float4 mean = add4(s_red[0], s_red[1]);
    mean = make_float4(mean.x * (1.f / 128.f), mean.y * (1.f / 128.f), mean.z * (1.f / 128.f), mean.w * (1.f / 128.f));
    __syncthreads();
    float4 d = sub4(v, mean);
    float4 q = wave_sum4(mul4(d, d));
    if (lane == 0) s_red[wid] = q;
    __syncthreads();
    float4 var = add4(s_red[0], s_red[1]);
    __syncthreads();
    float4 r = make_float4(1.f / sqrtf(var.x * (1.f / 128.f) + 1e-5f), 1.f / sqrtf(var.y * (1.f / 128.f) + 1e-5f),
                           1.f / sqrtf(var.z * (1.f / 128.f) + 1e-5f), 1.f / sqrtf(var.w * (1.f / 128.f) + 1e-5f));
    return make_float4(d.x * r.x * g + b, d.y * r.y * g + b, d.z * r.z * g + b, d.w * r.w * g + b);
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
__device__ __forceinline__ float comp(float4 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// Inputs of one message function from h (LDS, float4[128]): points (local + global), W_A h + b, W_C h.
__device__ void message_inputs(const PreW &w, const float4 *s_h, float4 *s_p, const float *frames, int n0, int N,
                               float *pts, float *PA, float *PC) {
    const int f = threadIdx.x;
    float4 a = dense4<128>(w.AT, 128, f, s_h, f4(w.in_b[f]));
    float4 c = dense4<128>(w.CT, 128, f, s_h, f4(0.f));
    store_rows(PA, 128, n0, N, f, a);
    store_rows(PC, 128, n0, N, f, c);
    if (f < 24) {
        float4 p = dense4<128>(w.ptsT, 24, f, s_h, f4(w.pts_b[f]));
        s_p[f] = p;
        store_rows(pts, 48, n0, N, f, p);
    }
    __syncthreads();
    if (f < 32) {                      // (point q, node i): p_glob = R p_loc + t
        int q = f >> 2, i = f & 3;
        int n = n0 + i;
        if (n < N) {
            const float *fr = frames + (size_t)n * 12;
            float x = comp(s_p[3 * q], i), y = comp(s_p[3 * q + 1], i), z = comp(s_p[3 * q + 2], i);
            for (int r = 0; r < 3; r++)
                pts[(size_t)n * 48 + 24 + 3 * q + r] = (fr[3 * r] * x + fr[3 * r + 1] * y + fr[3 * r + 2] * z) + fr[9 + r];
        }
    }
    __syncthreads();
}

// Node embedding for 4 nodes -> float4 (before LN) for feature f.
__device__ __forceinline__ float4 embed_pre(const NodeArgs &A, const float *chi, int step, int n0, float4 *s_in) {
    const int f = threadIdx.x;
    // s_in[0..5] bb sincos, [6..13] sc sincos*mask, [14..17] residue type (as float)
    if (f < 6) s_in[f] = load_rows(A.bb_sincos, 6, n0, A.N, f);
    else if (f < 14) {
        int k = (f - 6) >> 1, sc = (f - 6) & 1;
        float4 x = load_rows(chi, 4, n0, A.N, k), m = load_rows(A.sc_mask, 4, n0, A.N, k);
        float4 v = sc ? make_float4(cosf(x.x), cosf(x.y), cosf(x.z), cosf(x.w))
                      : make_float4(sinf(x.x), sinf(x.y), sinf(x.z), sinf(x.w));
        s_in[f] = mul4(v, m);
    }
    __syncthreads();
    float4 acc = f4(A.emb_b[f]);
    int t0 = n0 + 0 < A.N ? (int)A.rtype[n0 + 0] : 0, t1 = n0 + 1 < A.N ? (int)A.rtype[n0 + 1] : 0;
    int t2 = n0 + 2 < A.N ? (int)A.rtype[n0 + 2] : 0, t3 = n0 + 3 < A.N ? (int)A.rtype[n0 + 3] : 0;
    acc = add4(acc, make_float4(A.embT[t0 * 128 + f], A.embT[t1 * 128 + f], A.embT[t2 * 128 + f], A.embT[t3 * 128 + f]));
#pragma unroll
    for (int k = 0; k < 14; k++) acc = fma4(A.embT[(21 + k) * 128 + f], s_in[k], acc);
    const float *te = A.steps[step].temb;
    float tacc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) tacc = fmaf(A.embT[(35 + k) * 128 + f], te[k], tacc);
    return add4(acc, f4(tacc));
}

__global__ void __launch_bounds__(NT)
k_node_embed(NodeArgs A, PreW pre0, const float *chi, int step) {
    __shared__ float4 s_h[128];
    __shared__ float4 s_in[24];
    __shared__ float4 s_red[2];
    const int f = threadIdx.x, n0 = blockIdx.x * NB;
    float4 v = embed_pre(A, chi, step, n0, s_in);
    float4 h = layernorm4(v, A.emb_g[f], A.emb_beta[f], s_red);
    store_rows(A.hV, 128, n0, A.N, f, h);
    s_h[f] = h;
    __syncthreads();
    message_inputs(pre0, s_h, s_in, A.frames, n0, A.N, A.ptsN, A.PAn, A.PCn);
}

// (x + pi) % (2 pi) - pi with torch.remainder semantics in fp32
__device__ __forceinline__ float wrap_pi(float x) {
    const float PIf = 3.14159274101257324f, TWO_PIf = 6.28318548202514648f;
    float y = x + PIf;
    float r = fmodf(y, TWO_PIf);
    if (r != 0.f && r < 0.f) r += TWO_PIf;
    return r - PIf;
}

__global__ void __launch_bounds__(NT)
k_node_update(NodeArgs A, UpdW W, int last_mode, float *chi, int step, int sde, const float *noise, int embed_next,
              PreW pre0) {
    __shared__ float4 s_a[512];
    __shared__ float4 s_h[128];
    __shared__ float4 s_p[24];
    __shared__ float4 s_red[2];
    const int f = threadIdx.x, n0 = blockIdx.x * NB, N = A.N;
    s_a[f] = load_rows(A.S, 128, n0, N, f);
    __syncthreads();
    float4 ms = f4(0.f);
    if (n0 + 0 < N) ms.x = A.msum[n0 + 0];
    if (n0 + 1 < N) ms.y = A.msum[n0 + 1];
    if (n0 + 2 < N) ms.z = A.msum[n0 + 2];
    if (n0 + 3 < N) ms.w = A.msum[n0 + 3];
    // mean_j mask_j (W_out y_j + b) = W_out mean_j(mask_j y_j) + b mean_j(mask_j)
    float bo = W.out_b[f];
    float4 m = dense4<128>(W.outT, 128, f, s_a, make_float4(bo * ms.x, bo * ms.y, bo * ms.z, bo * ms.w));
    float4 h0 = load_rows(A.hV, 128, n0, N, f);
    float4 h1 = layernorm4(add4(h0, m), W.g0[f], W.b0[f], s_red);
    s_h[f] = h1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float4 hd = dense4<128>(W.ffn_inT, 512, f + 128 * j, s_h, f4(W.ffn_in_b[f + 128 * j]));
        s_a[f + 128 * j] = make_float4(fmaxf(hd.x, 0.f), fmaxf(hd.y, 0.f), fmaxf(hd.z, 0.f), fmaxf(hd.w, 0.f));
    }
    __syncthreads();
    float4 o = dense4<512>(W.ffn_outT, 128, f, s_a, f4(W.ffn_out_b[f]));
    float4 h2 = layernorm4(add4(h1, o), W.g1[f], W.b1[f], s_red);
    h2 = mul4(h2, load_rows(A.rmask, 1, n0, N, 0));
    store_rows(A.hV, 128, n0, N, f, h2);
    __syncthreads();
    s_h[f] = h2;
    __syncthreads();
    if (last_mode == PP_NU_MID) {
        message_inputs(W.pre_edge, s_h, s_p, A.frames, n0, N, A.ptsE, A.PAe, A.PCe);
        message_inputs(W.pre_next, s_h, s_p, A.frames, n0, N, A.ptsN, A.PAn, A.PCn);
        return;
    }
    // decoder: 128 -> 64 -> 32 -> relu -> 16 -> 4
    if (f < 64) {
        float4 v = dense4<128>(W.d0_inT, 64, f, s_h, f4(W.d0_in_b[f]));
        s_a[f] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
    __syncthreads();
    if (f < 32) {
        float4 v = dense4<64>(W.d0_outT, 32, f, s_a, f4(W.d0_out_b[f]));
        s_a[64 + f] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
    __syncthreads();
    if (f < 16) {
        float4 v = dense4<32>(W.d2_inT, 16, f, s_a + 64, f4(W.d2_in_b[f]));
        s_a[96 + f] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
    __syncthreads();
    if (f < 4) {
        float4 v = dense4<16>(W.d2_outT, 4, f, s_a + 96, f4(W.d2_out_b[f]));
        s_a[112 + f] = v;
        store_rows(A.score, 4, n0, N, f, v);
    }
    __syncthreads();
    if (last_mode != PP_NU_STEP) return;
    // reverse step on (node i, chi k) = 16 threads
    if (f < 16) {
        int i = f >> 2, k = f & 3, n = n0 + i;
        if (n < N) {
            const StepParams sp = A.steps[step];
            float x = chi[(size_t)n * 4 + k];
            float sw = comp(s_a[112 + k], i) * sp.w;
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
    __threadfence_block();
    float4 v = embed_pre(A, chi, step + 1, n0, s_p);
    float4 h = layernorm4(v, A.emb_g[f], A.emb_beta[f], s_red);
    store_rows(A.hV, 128, n0, N, f, h);
    __syncthreads();
    s_h[f] = h;
    __syncthreads();
    message_inputs(pre0, s_h, s_p, A.frames, n0, N, A.ptsN, A.PAn, A.PCn);
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
    int embed_next = (last_mode == PP_NU_STEP && step >= 0) ? 1 : 0;
    int st = step;
    if (last_mode == PP_NU_STEP && step < 0) { st = -step - 1; embed_next = 0; }   // negative: last step, no re-embed
    hipLaunchKernelGGL(k_node_update, dim3((c->N + NB - 1) / NB), dim3(NT), 0, s, A, W, last_mode, chi, st,
                       mode == PP_MODE_SDE ? 1 : 0, noise, embed_next, pre0);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// Edge-level stages of InvariantPointMessagePassing (layers.py:65-148) as fused FP32-MFMA kernels.
//
// One wave64 owns one residue i and its K<=32 edges (i, j); a workgroup is 4 waves (4 residues).
// All per-edge activations stay in registers for the whole MLP chain, in the accumulator layout of
// v_mfma_f32_32x32x2_f32 computing  Y^T[feature][edge] = W[feature][k] * X^T[k][edge]:
//
//      lane l = (edge j = l & 31, half h = l >> 5),  register r of tile t  <->  feature
//      F(t, r, h) = 32 t + 8 (r >> 2) + 4 h + (r & 3).
//
// With that k-ordering the D registers of one layer ARE the B operands of the next layer (no LDS
// round trip, no shuffles), and the A operand of 4 consecutive k-steps is one float4 of a row of the
// nn.Linear weight in its native [out][in] layout.  Weights stream L2 -> registers -> LDS in
// [128 rows][32 (or 72) cols] chunks, double buffered, shared by the 4 waves of the workgroup.
//
// The 456-wide first layer is never materialised: W_in [h_V_i | h_E_ij | h_V_j | geom] =
// (W_A h_V_i + b) + W_C h_V_j  (node-level, precomputed per residue in pp_node.hip, gathered here)
// + W_B h_E_ij + W_G geom_ij (MFMA here).
#include "pp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define ET 256
#define LDS_BUF_FLOATS (128 * 76)

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct EdgeArgs {
    int N, K;
    float inv_K;
    const float *rmask;        // [N]
    const int32_t *eidx;       // [N][K]
    const float *mask_att;     // [N][32]
    const float *frames;       // [N][12]
    const float *pts;          // [N][48]   p_loc | p_glob of the message being computed
    const float *PA, *PC;      // [N][128]
    const float *hE_in;        // [N][K][128]
    float *hE_out;             // [N][K][128]   (edge kernel)
    float *S, *msum;           // node kernel outputs
    // weights, native layouts
    const float *w_in, *b_mid_dummy;
    const float *w_mid, *b_mid, *w_out, *b_out;
    const float *g2, *be2, *g3, *be3;
    const float *ffn_in, *ffn_in_b, *ffn_out, *ffn_out_b;
};

template <int NC>
struct ChunkRegs {
    f32x4v v[(128 * NC / 4) / ET];
};

template <int NC>
__device__ __forceinline__ void chunk_load(const float *__restrict__ g, int ld, ChunkRegs<NC> &r, int tid) {
    constexpr int PER_ROW = NC / 4, PER_T = (128 * PER_ROW) / ET;
#pragma unroll
    for (int m = 0; m < PER_T; m++) {
        int idx = tid + ET * m;
        int row = idx / PER_ROW, c4 = idx - row * PER_ROW;
        r.v[m] = *reinterpret_cast<const f32x4v *>(g + (size_t)row * ld + 4 * c4);
    }
}
template <int NC>
__device__ __forceinline__ void chunk_store(float *lds, const ChunkRegs<NC> &r, int tid) {
    constexpr int PER_ROW = NC / 4, PER_T = (128 * PER_ROW) / ET;
#pragma unroll
    for (int m = 0; m < PER_T; m++) {
        int idx = tid + ET * m;
        int row = idx / PER_ROW, c4 = idx - row * PER_ROW;
        *reinterpret_cast<f32x4v *>(lds + row * (NC + 4) + 4 * c4) = r.v[m];
    }
}

// acc[t] += W[32t.., chunk cols] * x   (SWAP: acc[t] += x * W^T, edges on rows / features on lanes)
template <bool SWAP>
__device__ __forceinline__ void mfma_chunk32(const float *lds, const f32x16 &x, f32x16 (&acc)[4], int lane) {
    const float *base = lds + (lane & 31) * 36 + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v a = *reinterpret_cast<const f32x4v *>(base + 32 * 36 * t + 8 * q);
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (SWAP) acc[t] = MFMA(x[4 * q + p], a[p], acc[t]);
                else acc[t] = MFMA(a[p], x[4 * q + p], acc[t]);
            }
        }
    }
}

// geometry block: 72 inputs = 36 k-steps; lane half h supplies input 36 h + m at step m
__device__ __forceinline__ void mfma_chunk72(const float *lds, const float (&g)[36], f32x16 (&acc)[4], int lane) {
    const float *base = lds + (lane & 31) * 76 + 36 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int q = 0; q < 9; q++) {
            f32x4v a = *reinterpret_cast<const f32x4v *>(base + 32 * 76 * t + 4 * q);
#pragma unroll
            for (int p = 0; p < 4; p++) acc[t] = MFMA(a[p], g[4 * q + p], acc[t]);
        }
    }
}

// row-major [128] vector <-> accumulator layout
__device__ __forceinline__ void load_dl(const float *__restrict__ row, int h, f32x16 (&d)[4]) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v a = *reinterpret_cast<const f32x4v *>(row + 32 * t + 8 * q + 4 * h);
            d[t][4 * q] = a[0]; d[t][4 * q + 1] = a[1]; d[t][4 * q + 2] = a[2]; d[t][4 * q + 3] = a[3];
        }
}
__device__ __forceinline__ void add_dl(const float *__restrict__ row, int h, f32x16 (&d)[4]) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v a = *reinterpret_cast<const f32x4v *>(row + 32 * t + 8 * q + 4 * h);
            d[t][4 * q] += a[0]; d[t][4 * q + 1] += a[1]; d[t][4 * q + 2] += a[2]; d[t][4 * q + 3] += a[3];
        }
}
__device__ __forceinline__ void store_dl(float *__restrict__ row, int h, const f32x16 (&d)[4]) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v a = {d[t][4 * q], d[t][4 * q + 1], d[t][4 * q + 2], d[t][4 * q + 3]};
            *reinterpret_cast<f32x4v *>(row + 32 * t + 8 * q + 4 * h) = a;
        }
}
__device__ __forceinline__ void relu_dl(f32x16 (&d)[4]) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) d[t][r] = fmaxf(d[t][r], 0.f);
}

// LayerNorm over the 128 features of this lane's edge (64 here, 64 in lane ^ 32), eps 1e-5.
__device__ __forceinline__ void layernorm_dl(f32x16 (&v)[4], const float *__restrict__ gamma,
                                             const float *__restrict__ beta, int h) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += v[t][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.f / 128.f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float d = v[t][r] - mean;
            v[t][r] = d;
            q = fmaf(d, d, q);
        }
    q += __shfl_xor(q, 32);
    const float rstd = 1.f / sqrtf(q * (1.f / 128.f) + 1e-5f);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            f32x4v g = *reinterpret_cast<const f32x4v *>(gamma + 32 * t + 8 * qq + 4 * h);
            f32x4v b = *reinterpret_cast<const f32x4v *>(beta + 32 * t + 8 * qq + 4 * h);
#pragma unroll
            for (int p = 0; p < 4; p++) v[t][4 * qq + p] = fmaf(v[t][4 * qq + p] * rstd, g[p], b[p]);
        }
}

// 72 invariant point features of edge (i, j); returns the 36 this lane half feeds to the MFMA.
__device__ __forceinline__ void edge_geometry(const float *__restrict__ pts_i, const float *__restrict__ fr,
                                              const float *__restrict__ pts_j, int h, float (&g)[36]) {
    float geom[72];
    float R[9], tr[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = fr[k];
#pragma unroll
    for (int k = 0; k < 3; k++) tr[k] = fr[9 + k];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        float lx = pts_i[3 * q], ly = pts_i[3 * q + 1], lz = pts_i[3 * q + 2];
        float gx = pts_i[24 + 3 * q], gy = pts_i[24 + 3 * q + 1], gz = pts_i[24 + 3 * q + 2];
        float jx = pts_j[24 + 3 * q], jy = pts_j[24 + 3 * q + 1], jz = pts_j[24 + 3 * q + 2];
        geom[3 * q] = lx; geom[3 * q + 1] = ly; geom[3 * q + 2] = lz;
        geom[24 + q] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
        float dx = jx - tr[0], dy = jy - tr[1], dz = jz - tr[2];
        float nx = R[0] * dx + R[3] * dy + R[6] * dz;
        float ny = R[1] * dx + R[4] * dy + R[7] * dz;
        float nz = R[2] * dx + R[5] * dy + R[8] * dz;
        geom[32 + 3 * q] = nx; geom[32 + 3 * q + 1] = ny; geom[32 + 3 * q + 2] = nz;
        geom[56 + q] = sqrtf(nx * nx + ny * ny + nz * nz + 1e-8f);
        float ex = gx - jx, ey = gy - jy, ez = gz - jz;
        geom[64 + q] = sqrtf(ex * ex + ey * ey + ez * ez + 1e-8f);
    }
#pragma unroll
    for (int m = 0; m < 36; m++) g[m] = h ? geom[36 + m] : geom[m];
}

// One pipeline stage: prefetch the next weight chunk into registers, compute on the current LDS
// buffer, then publish the prefetched chunk into the other buffer.
#define STAGE(COMPUTE, NEXT_NC, NEXT_PTR, NEXT_LD)                         \
    {                                                                      \
        ChunkRegs<NEXT_NC> _r;                                             \
        chunk_load<NEXT_NC>((NEXT_PTR), (NEXT_LD), _r, tid);               \
        if (active) { COMPUTE; }                                           \
        chunk_store<NEXT_NC>(cur ? buf0 : buf1, _r, tid);                  \
        __syncthreads();                                                   \
        cur ^= 1;                                                          \
    }
#define CURBUF (cur ? buf1 : buf0)

// ---------------------------------------------------------------------------------------------
// node message: S[i] = (1/K) sum_j mask_ij relu(W_mid relu(W_in [..]) + b), msum[i] = (1/K) sum_j mask_ij
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ET, 1)
k_node_message(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *buf0 = smem, *buf1 = smem + LDS_BUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x * 4 + wave;
    const bool active = (n < A.N) && (A.rmask[n < A.N ? n : 0] != 0.f);
    const int K = A.K;
    int cur = 0;

    f32x16 x[4], acc[4];
    float g[36];
    if (active) {
        const int jj = j < K ? j : K - 1;
        const int nbr = A.eidx[(size_t)n * K + jj];
        load_dl(A.hE_in + ((size_t)n * K + jj) * 128, h, x);
        load_dl(A.PA + (size_t)n * 128, h, acc);
        add_dl(A.PC + (size_t)nbr * 128, h, acc);
        edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g);
    }
    const float *wB = A.w_in + 128, *wG = A.w_in + 384;
    {   // prologue: chunk 0 of W_B
        ChunkRegs<32> r0;
        chunk_load<32>(wB, PP_MSG_IN, r0, tid);
        chunk_store<32>(buf0, r0, tid);
        __syncthreads();
    }
    STAGE(mfma_chunk32<false>(CURBUF, x[0], acc, lane), 32, wB + 32, PP_MSG_IN)
    STAGE(mfma_chunk32<false>(CURBUF, x[1], acc, lane), 32, wB + 64, PP_MSG_IN)
    STAGE(mfma_chunk32<false>(CURBUF, x[2], acc, lane), 32, wB + 96, PP_MSG_IN)
    STAGE(mfma_chunk32<false>(CURBUF, x[3], acc, lane), 72, wG, PP_MSG_IN)
    STAGE(mfma_chunk72(CURBUF, g, acc, lane), 32, A.w_mid, 128)
    if (active) {
        relu_dl(acc);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            x[t] = acc[t];
            const float b = A.b_mid[32 * t + j];          // SWAP form: feature on the lane
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = b;
        }
    }
    STAGE(mfma_chunk32<true>(CURBUF, x[0], acc, lane), 32, A.w_mid + 32, 128)
    STAGE(mfma_chunk32<true>(CURBUF, x[1], acc, lane), 32, A.w_mid + 64, 128)
    STAGE(mfma_chunk32<true>(CURBUF, x[2], acc, lane), 32, A.w_mid + 96, 128)
    if (active) {
        mfma_chunk32<true>(CURBUF, x[3], acc, lane);
        // rows (registers) are edges e = 8 (r>>2) + 4 h + (r&3); mask and reduce over them
        float m16[16];
        const float *mrow = A.mask_att + (size_t)n * 32;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v mm = *reinterpret_cast<const f32x4v *>(mrow + 8 * q + 4 * h);
            m16[4 * q] = mm[0]; m16[4 * q + 1] = mm[1]; m16[4 * q + 2] = mm[2]; m16[4 * q + 3] = mm[3];
        }
        float ms = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) ms += m16[r];
        ms += __shfl_xor(ms, 32);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) s = fmaf(fmaxf(acc[t][r], 0.f), m16[r], s);
            s += __shfl_xor(s, 32);
            if (h == 0) A.S[(size_t)n * 128 + 32 * t + j] = s * A.inv_K;
        }
        if (lane == 0) A.msum[n] = ms * A.inv_K;
    } else if (n < A.N) {
        for (int f = lane; f < 128; f += 64) A.S[(size_t)n * 128 + f] = 0.f;
        if (lane == 0) A.msum[n] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// edge update: h_E <- mask * LN3(x1 + FFN(x1)),  x1 = LN2(h_E + mask * MLP3([..]))
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ET, 1)
k_edge_update(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *buf0 = smem, *buf1 = smem + LDS_BUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x * 4 + wave;
    const bool active = (n < A.N) && (A.rmask[n < A.N ? n : 0] != 0.f);
    const int K = A.K;
    const int jj = j < K ? j : K - 1;
    int cur = 0;

    f32x16 x[4], acc[4], out[4];
    float g[36];
    float me = 0.f;
    const float *hrow = A.hE_in + ((size_t)(n < A.N ? n : 0) * K + jj) * 128;
    if (active) {
        const int nbr = A.eidx[(size_t)n * K + jj];
        me = A.mask_att[(size_t)n * 32 + j];
        load_dl(hrow, h, x);
        load_dl(A.PA + (size_t)n * 128, h, acc);
        add_dl(A.PC + (size_t)nbr * 128, h, acc);
        edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g);
    }
    const float *wB = A.w_in + 128, *wG = A.w_in + 384;
    {
        ChunkRegs<32> r0;
        chunk_load<32>(wB, PP_MSG_IN, r0, tid);
        chunk_store<32>(buf0, r0, tid);
        __syncthreads();
    }
    STAGE(mfma_chunk32<false>(CURBUF, x[0], acc, lane), 32, wB + 32, PP_MSG_IN)
    STAGE(mfma_chunk32<false>(CURBUF, x[1], acc, lane), 32, wB + 64, PP_MSG_IN)
    STAGE(mfma_chunk32<false>(CURBUF, x[2], acc, lane), 32, wB + 96, PP_MSG_IN)
    STAGE(mfma_chunk32<false>(CURBUF, x[3], acc, lane), 72, wG, PP_MSG_IN)
    STAGE(mfma_chunk72(CURBUF, g, acc, lane), 32, A.w_mid, 128)
    if (active) {
        relu_dl(acc);
#pragma unroll
        for (int t = 0; t < 4; t++) x[t] = acc[t];
        load_dl(A.b_mid, h, acc);
    }
    STAGE(mfma_chunk32<false>(CURBUF, x[0], acc, lane), 32, A.w_mid + 32, 128)
    STAGE(mfma_chunk32<false>(CURBUF, x[1], acc, lane), 32, A.w_mid + 64, 128)
    STAGE(mfma_chunk32<false>(CURBUF, x[2], acc, lane), 32, A.w_mid + 96, 128)
    STAGE(mfma_chunk32<false>(CURBUF, x[3], acc, lane), 32, A.w_out, 128)
    if (active) {
        relu_dl(acc);
#pragma unroll
        for (int t = 0; t < 4; t++) x[t] = acc[t];
        load_dl(A.b_out, h, acc);
    }
    STAGE(mfma_chunk32<false>(CURBUF, x[0], acc, lane), 32, A.w_out + 32, 128)
    STAGE(mfma_chunk32<false>(CURBUF, x[1], acc, lane), 32, A.w_out + 64, 128)
    STAGE(mfma_chunk32<false>(CURBUF, x[2], acc, lane), 32, A.w_out + 96, 128)
    STAGE(mfma_chunk32<false>(CURBUF, x[3], acc, lane), 32, A.ffn_in, 128)
    if (active) {
        // x1 = LN2(h_E + mask * m)
        load_dl(hrow, h, x);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) x[t][r] = fmaf(acc[t][r], me, x[t][r]);
        layernorm_dl(x, A.g2, A.be2, h);
        load_dl(A.ffn_out_b, h, out);
    }
    // FFN 128 -> 512 -> 128 in four hidden blocks of 128
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float *w1 = A.ffn_in + (size_t)(128 * c) * 128;      // rows 128c.. of [512][128]
        const float *w2 = A.ffn_out + 128 * c;                     // cols 128c.. of [128][512]
        if (active) load_dl(A.ffn_in_b + 128 * c, h, acc);
        STAGE(mfma_chunk32<false>(CURBUF, x[0], acc, lane), 32, w1 + 32, 128)
        STAGE(mfma_chunk32<false>(CURBUF, x[1], acc, lane), 32, w1 + 64, 128)
        STAGE(mfma_chunk32<false>(CURBUF, x[2], acc, lane), 32, w1 + 96, 128)
        STAGE(mfma_chunk32<false>(CURBUF, x[3], acc, lane), 32, w2, 512)
        if (active) relu_dl(acc);
        STAGE(mfma_chunk32<false>(CURBUF, acc[0], out, lane), 32, w2 + 32, 512)
        STAGE(mfma_chunk32<false>(CURBUF, acc[1], out, lane), 32, w2 + 64, 512)
        STAGE(mfma_chunk32<false>(CURBUF, acc[2], out, lane), 32, w2 + 96, 512)
        if (c < 3) {
            STAGE(mfma_chunk32<false>(CURBUF, acc[3], out, lane), 32, A.ffn_in + (size_t)(128 * (c + 1)) * 128, 128)
        } else {
            if (active) mfma_chunk32<false>(CURBUF, acc[3], out, lane);
        }
    }
    if (n < A.N && j < K) {
        float *orow = A.hE_out + ((size_t)n * K + j) * 128;
        if (active) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) out[t][r] += x[t][r];
            layernorm_dl(out, A.g3, A.be3, h);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) out[t][r] *= me;
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) out[t][r] = 0.f;
        }
        store_dl(orow, h, out);
    }
}

// ---------------------------------------------------------------------------------------------
static EdgeArgs edge_args(pp_ctx *c, int layer, bool edge) {
    const pp_plan *p = c->plan;
    const LayerOff &o = p->off.layer[layer];
    EdgeArgs A;
    A.N = c->N; A.K = c->K; A.inv_K = 1.0f / (float)c->K;
    A.rmask = c->b.residue_mask;
    A.eidx = c->eidx; A.mask_att = c->mask_att; A.frames = c->frames;
    A.pts = edge ? c->ptsE : c->ptsN;
    A.PA = edge ? c->PAe : c->PAn;
    A.PC = edge ? c->PCe : c->PCn;
    A.hE_in = layer == 0 ? c->hE0 : c->hE;
    A.hE_out = c->hE;
    A.S = c->S; A.msum = c->msum;
    const float *w = p->w;
    A.w_in = w + (edge ? o.em_in_w : o.nm_in_w);
    A.b_mid_dummy = nullptr;
    A.w_mid = w + (edge ? o.em_mid_w : o.nm_mid_w);
    A.b_mid = w + (edge ? o.em_mid_b : o.nm_mid_b);
    A.w_out = w + (edge ? o.em_out_w : o.nm_out_w);
    A.b_out = w + (edge ? o.em_out_b : o.nm_out_b);
    A.g2 = w + o.norm_g[2]; A.be2 = w + o.norm_b[2];
    A.g3 = w + o.norm_g[3]; A.be3 = w + o.norm_b[3];
    A.ffn_in = w + o.ed_in_w; A.ffn_in_b = w + o.ed_in_b;
    A.ffn_out = w + o.ed_out_w; A.ffn_out_b = w + o.ed_out_b;
    return A;
}

static const size_t EDGE_SMEM = 2 * LDS_BUF_FLOATS * sizeof(float);

pp_status pp_launch_node_message(pp_ctx *c, int layer, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_message),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE_SMEM));
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_update),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE_SMEM));
        attr_set = true;
    }
    EdgeArgs A = edge_args(c, layer, false);
    hipLaunchKernelGGL(k_node_message, dim3((c->N + 3) / 4), dim3(ET), EDGE_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_edge_update(pp_ctx *c, int layer, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_update),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE_SMEM));
        attr_set = true;
    }
    EdgeArgs A = edge_args(c, layer, true);
    hipLaunchKernelGGL(k_edge_update, dim3((c->N + 3) / 4), dim3(ET), EDGE_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// EXPERIMENT (not built): NW = 1..3 residues per workgroup sharing one weight stream and barrier.  Correct for all NW
// (GPU parity suite), but on MI355X at L=739: NW=1 104 us, NW=2 135 us, NW=3 127 us per edge-update launch -- the
// larger barrier domain costs more than the saved L2 traffic.  Needs pp_mfma.h from this directory.
// Edge-level stages of InvariantPointMessagePassing (layers.py:65-148) as fused FP32-MFMA kernels.
//
// A workgroup of 4 waves owns ONE residue i and its K<=32 edges (i, j).  Every activation tensor of the
// chain lives in the accumulator layout of v_mfma_f32_32x32x2_f32 for the transposed product
//      Y^T[feature][edge] = W[feature][k] * X^T[k][edge]:
//      lane l = (edge j = l & 31, half h = l >> 5),  register r of tile t  <->  feature
//      F(t, r, h) = 32 t + 8 (r >> 2) + 4 h + (r & 3).
// With that k-ordering the D registers of one layer ARE the B operands of the next, and the A operand of
// 4 consecutive k-steps is one float4 of a row of the nn.Linear weight ([out][in]: no transposition needed).
//
// N-split: wave w computes output tile w (32 of the 128 features; for the 512-wide FFN hidden layer,
// tile 4c + w of hidden block c).  A finished tile is published to a 16 KB LDS exchange buffer
// (same [tile][quad][lane] float4 layout it has in registers) and every wave reads back the full
// 128-vector it needs as B operands.  Weights stream L2 -> registers -> LDS in [128 rows][32 | 24 cols]
// chunks (pre-packed contiguously in consumption order), two chunks in flight, double-buffered LDS; each wave
// reads its own 32 rows of the shared chunk.  53 KB of LDS and <=168
// VGPRs per workgroup let up to 3 workgroups share a CU, so one workgroup's barrier / LDS latency is covered
// by another's MFMAs, and 739 residues x 4 waves spread evenly over the 1024 SIMDs.
//
// The 456-wide first layer is never materialised: W_in [h_V_i | h_E_ij | h_V_j | geom] =
// (W_A h_V_i + b) + W_C h_V_j  (node-level, precomputed per residue in pp_node.hip, gathered here)
// + W_B h_E_ij + W_G geom_ij (MFMA here; the 72 invariant-point features are built in registers).
#include <stdlib.h>

#include "pp_mfma.h"

// Weight pipeline, prefetch distance 2: at the start of stage k chunk k is visible in LDS slot k&1 and chunk k+1 is in
// flight into one register set.  The stage issues the loads of chunk k+2 into the other set, computes on chunk k,
// then publishes chunk k+1 into the other LDS slot (last read one stage ago) and barriers.  A load therefore has two
// stages of MFMA time to land.  RS / RL name the register sets stored / loaded in this stage (they alternate).
#ifdef PP_X_STAMP
#define STAMP(i)                                                                                        \
    {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long _t = __builtin_amdgcn_s_memtime();                                           \
        if (lane == 0 && stage_no < 64) A.dbg[(((size_t)n * 4 + wave) * 64 + stage_no) * 4 + (i)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }
#define NEXT_STAGE() stage_no++
#else
#define STAMP(i)
#define NEXT_STAGE()
#endif
#define STAGE2(COMPUTE, RS, RS_NC, RL, RL_NC, RL_PTR)                      \
    {                                                                      \
        STAMP(0)                                                           \
        chunk_load<RL_NC, NT>((RL_PTR), RL, tid);                          \
        if (active) { COMPUTE; }                                           \
        STAMP(1)                                                           \
        chunk_store<RS_NC, NT>(cur ? wbuf0 : wbuf1, RS, tid);              \
        STAMP(2)                                                           \
        STAGE_SYNC();                                                      \
        STAMP(3)                                                           \
        NEXT_STAGE();                                                      \
        cur ^= 1;                                                          \
    }
// last stages of a kernel: nothing further to load
#define STAGE2_NOLOAD(COMPUTE, RS, RS_NC)                                  \
    {                                                                      \
        if (active) { COMPUTE; }                                           \
        chunk_store<RS_NC, NT>(cur ? wbuf0 : wbuf1, RS, tid);              \
        STAGE_SYNC();                                                      \
        cur ^= 1;                                                          \
    }
#define CURBUF (cur ? wbuf1 : wbuf0)

// Stream offsets (floats) of chunk k: chunks 0..3 are 32 columns wide, 4..6 (geometry) 24, the rest 32.
#define CH32 (128 * 32)
#define CH24 (128 * 24)
#define CHUNK_OFF(k) ((k) < 4 ? (k) * CH32 : ((k) < 7 ? 4 * CH32 + ((k) - 4) * CH24 : 4 * CH32 + 3 * CH24 + ((k) - 7) * CH32))

// shared first layer (chunks 0..6 = W_B x4, W_G x3): acc (tile `wave`) = PA_i + PC_j + W_B h_E + W_G geom, ReLU,
// published to xbuf.  On exit: chunk 7 visible in LDS, chunk 8 in flight in RA.
#define FIRST_LAYER()                                                                                     \
    chunk_load<32, NT>(ws + CHUNK_OFF(0), RA, tid);                                                       \
    chunk_store<32, NT>(wbuf0, RA, tid);                                                                  \
    chunk_load<32, NT>(ws + CHUNK_OFF(1), RB, tid);                                                       \
    __syncthreads();                                                                                      \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[0], acc, lane), RB, 32, RA, 32, ws + CHUNK_OFF(2))          \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[1], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(3))          \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[2], acc, lane), RB, 32, RA, 24, ws + CHUNK_OFF(4))          \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[3], acc, lane), RA, 24, RB, 24, ws + CHUNK_OFF(5))          \
    /* x[] is dead from here to the exchange: build the 72 point features in its place */                \
    if (active) edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g); \
    STAGE2(mfma_tile24(CURBUF, wave, g[0], acc, lane), RB, 24, RA, 24, ws + CHUNK_OFF(6))                 \
    STAGE2(mfma_tile24(CURBUF, wave, g[1], acc, lane), RA, 24, RB, 32, ws + CHUNK_OFF(7))                 \
    STAGE2(mfma_tile24(CURBUF, wave, g[2], acc, lane); relu_tile(acc); xbuf_put(xbuf, wave, lane, acc),   \
           RB, 32, RA, 32, ws + CHUNK_OFF(8))

// ---------------------------------------------------------------------------------------------
// node message: S[i] = (1/K) sum_j mask_ij relu(W_mid relu(W_in [..]) + b), msum[i] = (1/K) sum_j mask_ij
// ---------------------------------------------------------------------------------------------
// NW residues per workgroup (4 NW waves): one shared weight stream and barrier, one exchange buffer per residue.
template <int NW>
__global__ void __launch_bounds__(ET * NW, NW == 1 ? 3 : NW)
k_node_message(EdgeArgs A) {
    constexpr int NT = ET * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + WBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wid & 3, slot = wid >> 2;
    float *xbuf = smem + 2 * WBUF_FLOATS + slot * XBUF_FLOATS;
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x * NW + slot;
    const int K = A.K;
    int cur = 0;
#ifdef PP_X_STAMP
    int stage_no = 0;
#endif
    // masked / padded / out-of-range residue: its waves keep staging weights and meeting barriers, nothing else
    const bool active = n < A.N && A.rmask[n < A.N ? n : 0] != 0.f;
    if (!active && n < A.N) {
        if (wave < 2) A.S[(size_t)n * 128 + 64 * wave + lane] = 0.f;
        if (wave == 0 && lane == 0) A.msum[n] = 0.f;
    }
    if (NW == 1 && !active) return;

    f32x16 x[4], acc;
    float g[3][12];
    const int jj = j < K ? j : K - 1;
    const int nbr = active ? A.eidx[(size_t)n * K + jj] : 0;
    if (active) {
        const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
#pragma unroll
        for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
    }
    const float *ws = A.wstream;          // chunks: W_B 0..3, W_G 4..6, W_mid 7..10
    WRegs RA, RB;
    FIRST_LAYER()
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        const float b = A.b_mid[32 * wave + j];           // SWAP form: feature on the lane
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = b;
    }
    STAGE2(mfma_tile32<true>(CURBUF, wave, x[0], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(9))
    STAGE2(mfma_tile32<true>(CURBUF, wave, x[1], acc, lane), RB, 32, RA, 32, ws + CHUNK_OFF(10))
    STAGE2_NOLOAD(mfma_tile32<true>(CURBUF, wave, x[2], acc, lane), RA, 32)
    if (active) {
        mfma_tile32<true>(CURBUF, wave, x[3], acc, lane);
        // rows (registers) are edges e = 8 (r>>2) + 4 h + (r&3); mask and reduce over them
        float m16[16];
        const float *mrow = A.mask_att + (size_t)n * 32;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v mm = *reinterpret_cast<const f32x4v *>(mrow + 8 * q + 4 * h);
            m16[4 * q] = mm[0]; m16[4 * q + 1] = mm[1]; m16[4 * q + 2] = mm[2]; m16[4 * q + 3] = mm[3];
        }
        float s = 0.f, ms = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s = fmaf(fmaxf(acc[r], 0.f), m16[r], s);
            ms += m16[r];
        }
        s += __shfl_xor(s, 32);
        ms += __shfl_xor(ms, 32);
        if (h == 0) A.S[(size_t)n * 128 + 32 * wave + j] = s * A.inv_K;
        if (wave == 0 && lane == 0) A.msum[n] = ms * A.inv_K;
    }
}

// ---------------------------------------------------------------------------------------------
// edge update: h_E <- mask * LN3(x1 + FFN(x1)),  x1 = LN2(h_E + mask * MLP3([..]))
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(ET * NW, NW == 1 ? 2 : NW)
k_edge_update(EdgeArgs A) {
    constexpr int NT = ET * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + WBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wid & 3, slot = wid >> 2;
    float *xbuf = smem + 2 * WBUF_FLOATS + slot * XBUF_FLOATS;
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x * NW + slot;
    const int K = A.K;
    const int jj = j < K ? j : K - 1;
    int cur = 0;
#ifdef PP_X_STAMP
    int stage_no = 0;
#endif
    const bool active = n < A.N && A.rmask[n < A.N ? n : 0] != 0.f;
    if (!active && n < A.N && j < K) {      // masked / padded residue: its edges are zero
        f32x4v z = {0.f, 0.f, 0.f, 0.f};
        float *orow = A.hE_out + ((size_t)n * K + j) * 128 + 32 * wave;
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4v *>(orow + 8 * q + 4 * h) = z;
    }
    if (NW == 1 && !active) return;

    f32x16 x[4], acc, out;
    float g[3][12];
    const float *hrow = A.hE_in + ((size_t)(active ? n : 0) * K + jj) * 128;
    const int nbr = active ? A.eidx[(size_t)n * K + jj] : 0;
    const float me = active ? A.mask_att[(size_t)n * 32 + j] : 0.f;
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
    }
    const float *ws = A.wstream;   // chunks: W_B 0..3, W_G 4..6, W_mid 7..10, W_out 11..14, then per c: W1 x4, W2 x4
    WRegs RA, RB;
    FIRST_LAYER()
    // ---- second layer (chunks 7..10) -------------------------------------------------------------
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        load_tile(A.b_mid + 32 * wave, h, acc);
    }
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[0], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(9))
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[1], acc, lane), RB, 32, RA, 32, ws + CHUNK_OFF(10))
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[2], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(11))
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[3], acc, lane); relu_tile(acc); xbuf_put(xbuf, wave, lane, acc),
           RB, 32, RA, 32, ws + CHUNK_OFF(12))
    // ---- third layer (chunks 11..14) --------------------------------------------------------------
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        load_tile(A.b_out + 32 * wave, h, acc);
    }
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[0], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(13))
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[1], acc, lane), RB, 32, RA, 32, ws + CHUNK_OFF(14))
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[2], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(15))
    // last chunk; then publish v = h_E + mask * m for the first LayerNorm
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[3], acc, lane); {
               f32x16 v;
               load_tile(hrow + 32 * wave, h, v);
               _Pragma("unroll") for (int r = 0; r < 16; r++) v[r] = fmaf(acc[r], me, v[r]);
               xbuf_put(xbuf, wave, lane, v);
           },
           RB, 32, RA, 32, ws + CHUNK_OFF(16))
    if (active) {
        // x1 = LN2(v): every wave normalises the full vector (it needs all of x1 as B operands)
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        float mean;
        const float rstd = ln_center(x, mean);
#pragma unroll
        for (int t = 0; t < 4; t++) ln_affine_tile(x[t], rstd, A.g2 + 32 * t, A.be2 + 32 * t, h);
        load_tile(A.ffn_out_b + 32 * wave, h, out);
    }
    // ---- FFN 128 -> 512 -> 128 in four hidden blocks of 128 (chunks 15 + 8c ..) ------------------------
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float *wc = ws + CHUNK_OFF(15 + 8 * c);             // this block's 8 chunks: W1 s=0..3, W2 s'=0..3
        if (active) load_tile(A.ffn_in_b + 128 * c + 32 * wave, h, acc);
        STAGE2(mfma_tile32<false>(CURBUF, wave, x[0], acc, lane), RA, 32, RB, 32, wc + 2 * CH32)
        STAGE2(mfma_tile32<false>(CURBUF, wave, x[1], acc, lane), RB, 32, RA, 32, wc + 3 * CH32)
        STAGE2(mfma_tile32<false>(CURBUF, wave, x[2], acc, lane), RA, 32, RB, 32, wc + 4 * CH32)
        STAGE2(mfma_tile32<false>(CURBUF, wave, x[3], acc, lane); relu_tile(acc); xbuf_put(xbuf, wave, lane, acc),
               RB, 32, RA, 32, wc + 5 * CH32)
        // second FFN layer over this hidden block: B operands come tile by tile from the exchange buffer
        STAGE2(xbuf_get(xbuf, 0, lane, acc); mfma_tile32<false>(CURBUF, wave, acc, out, lane), RA, 32, RB, 32, wc + 6 * CH32)
        STAGE2(xbuf_get(xbuf, 1, lane, acc); mfma_tile32<false>(CURBUF, wave, acc, out, lane), RB, 32, RA, 32, wc + 7 * CH32)
        if (c < 3) {
            STAGE2(xbuf_get(xbuf, 2, lane, acc); mfma_tile32<false>(CURBUF, wave, acc, out, lane), RA, 32, RB, 32, wc + 8 * CH32)
            STAGE2(xbuf_get(xbuf, 3, lane, acc); mfma_tile32<false>(CURBUF, wave, acc, out, lane), RB, 32, RA, 32,
                   wc + 9 * CH32)
        } else {
            STAGE2_NOLOAD(xbuf_get(xbuf, 2, lane, acc); mfma_tile32<false>(CURBUF, wave, acc, out, lane), RA, 32)
            if (active) {
                xbuf_get(xbuf, 3, lane, acc);
                mfma_tile32<false>(CURBUF, wave, acc, out, lane);
            }
            __syncthreads();          // every wave is done reading the hidden tiles before they are overwritten
        }
    }
    // ---- h_E = mask * LN3(x1 + ffn) ---------------------------------------------------------------------
    if (active) {
        // residual: this wave's tile of x1 (wave is scalar: four uniform branches, static register indices)
        if (wave == 0) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[0][r]; }
        else if (wave == 1) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[1][r]; }
        else if (wave == 2) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[2][r]; }
        else { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[3][r]; }
        xbuf_put(xbuf, wave, lane, out);
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        float mean3;
        const float rstd = ln_center(x, mean3);
#pragma unroll
        for (int r = 0; r < 16; r++) out[r] -= mean3;
        ln_affine_tile(out, rstd, A.g3 + 32 * wave, A.be3 + 32 * wave, h);
#pragma unroll
        for (int r = 0; r < 16; r++) out[r] *= me;
        if (j < K) store_tile(A.hE_out + ((size_t)n * K + j) * 128 + 32 * wave, h, out);
    }
}

// ---------------------------------------------------------------------------------------------
EdgeArgs pp_edge_args(pp_ctx *c, int layer, bool edge) {
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
    A.wstream = edge ? p->lt[layer].em_stream : p->lt[layer].nm_stream;
    A.b_mid = w + (edge ? o.em_mid_b : o.nm_mid_b);
    A.b_out = w + (edge ? o.em_out_b : o.nm_out_b);
    A.g2 = w + o.norm_g[2]; A.be2 = w + o.norm_b[2];
    A.g3 = w + o.norm_g[3]; A.be3 = w + o.norm_b[3];
    A.ffn_in_b = w + o.ed_in_b;
    A.ffn_out_b = w + o.ed_out_b;
    A.dbg = c->dbg;
    return A;
}

static size_t edge_smem(int nw) { return (2 * WBUF_FLOATS + (size_t)nw * XBUF_FLOATS) * sizeof(float); }

// Residues per workgroup.  A CU holds 12 of these waves (3 per SIMD); a workgroup of NW residues is 4 NW waves sharing
// one weight stream.  Up to one residue per CU -> NW = 1 (several small workgroups per CU), else 2, else 3 (one
// 12-wave workgroup per CU: a third of the L2 -> LDS weight traffic and, at T1124's 739 residues, exactly one round).
static int pick_nw(int N) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                  ? prop.multiProcessorCount : 256;
    }
    const char *force = getenv("PP_EDGE_NW");
    if (force && force[0] >= '1' && force[0] <= '3') return force[0] - '0';
    if (N <= cus) return 1;
    if (N <= 2 * cus) return 2;
    return 3;
}

template <int NW>
static pp_status launch_nm(const EdgeArgs &A, int N, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_message<NW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)edge_smem(NW)));
        attr = true;
    }
    hipLaunchKernelGGL(k_node_message<NW>, dim3((N + NW - 1) / NW), dim3(ET * NW), edge_smem(NW), s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
template <int NW>
static pp_status launch_eu(const EdgeArgs &A, int N, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_update<NW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)edge_smem(NW)));
        attr = true;
    }
    hipLaunchKernelGGL(k_edge_update<NW>, dim3((N + NW - 1) / NW), dim3(ET * NW), edge_smem(NW), s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// resident workgroups per CU the runtime predicts for the NW = 1 kernels (measurement aid)
void pp_edge_occupancy(int *node_msg, int *edge_upd) {
    hipOccupancyMaxActiveBlocksPerMultiprocessor(node_msg, reinterpret_cast<const void *>(k_node_message<1>), ET, edge_smem(1));
    hipOccupancyMaxActiveBlocksPerMultiprocessor(edge_upd, reinterpret_cast<const void *>(k_edge_update<1>), ET, edge_smem(1));
}

pp_status pp_launch_node_message(pp_ctx *c, int layer, hipStream_t s) {
    EdgeArgs A = pp_edge_args(c, layer, false);
    switch (pick_nw(c->N)) {
        case 1: return launch_nm<1>(A, c->N, s);
        case 2: return launch_nm<2>(A, c->N, s);
        default: return launch_nm<3>(A, c->N, s);
    }
}

pp_status pp_launch_edge_update(pp_ctx *c, int layer, hipStream_t s) {
    EdgeArgs A = pp_edge_args(c, layer, true);
    switch (pick_nw(c->N)) {
        case 1: return launch_eu<1>(A, c->N, s);
        case 2: return launch_eu<2>(A, c->N, s);
        default: return launch_eu<3>(A, c->N, s);
    }
}

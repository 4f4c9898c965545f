// EXPERIMENT (not built): two residues per wave sharing each weight chunk.  Correct (GPU parity suite passed) but
// slower than pp_edge.hip on MI355X: x[2][4] + geometry push it past 256 VGPRs (229 spills) -- kept for reference.
// Dual-residue variant of the fused FP32-MFMA edge kernels (same math and layouts as pp_edge.hip).
//
// A workgroup of 4 waves owns ONE or TWO residues; wave w computes output tile w (N-split) for each of them with
// the same A operands.  Every weight chunk staged in LDS therefore feeds 2 x 16 MFMAs per wave instead of 16:
// half the L2 -> LDS weight traffic and half the barriers per MFMA.  (Measured on MI355X: with one residue per
// workgroup the kernel is limited by the ~6 TB/s at which 739 workgroups can pull identical 16 KB chunks out of
// L2 with 32-96 KB in flight per CU, not by the matrix pipe.)
//
// Two workgroups of this kernel fit a CU (<= 256 VGPRs, 70 KB LDS).  The launcher mixes dual and single
// workgroups so that every CU gets the same number of residues: for N residues and S = 2 * #CU workgroup
// slots, N <= S -> all single; N <= 2S -> (N - S) dual + the rest single; else all dual.
#include "pp_mfma.h"

#define XBUF2_FLOATS (2 * XBUF_FLOATS)

// prefetch distance 1: load chunk k+1 into R at the top of stage k, publish it at the bottom
#define STAGE1(COMPUTE, NEXT_NC, NEXT_PTR)                                 \
    {                                                                      \
        chunk_load<NEXT_NC>((NEXT_PTR), R, tid);                           \
        { COMPUTE; }                                                       \
        chunk_store<NEXT_NC>(cur ? wbuf0 : wbuf1, R, tid);                 \
        __syncthreads();                                                   \
        cur ^= 1;                                                          \
    }
#define CURBUF (cur ? wbuf1 : wbuf0)
#define CH32 (128 * 32)
#define CH24 (128 * 24)
#define CHUNK_OFF(k) ((k) < 4 ? (k) * CH32 : ((k) < 7 ? 4 * CH32 + ((k) - 4) * CH24 : 4 * CH32 + 3 * CH24 + ((k) - 7) * CH32))

// run STMT for residue slot p = 0 and (when the workgroup has a second residue) p = 1
#define FOR_P(STMT)            \
    { constexpr int p = 0; STMT; } \
    if (two) { constexpr int p = 1; STMT; }

struct Edge2Args {
    EdgeArgs e;
    int n_dual;        // workgroups [0, n_dual) own residues (2b, 2b+1); workgroup b >= n_dual owns 2 n_dual + (b - n_dual)
};

// first layer for both residue slots (chunks 0..6); leaves chunk 7 visible
#define FIRST_LAYER2()                                                                                        \
    chunk_load<32>(ws + CHUNK_OFF(0), R, tid);                                                                \
    chunk_store<32>(wbuf0, R, tid);                                                                           \
    __syncthreads();                                                                                          \
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][0], acc[p], lane)), 32, ws + CHUNK_OFF(1))             \
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][1], acc[p], lane)), 32, ws + CHUNK_OFF(2))             \
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][2], acc[p], lane)), 32, ws + CHUNK_OFF(3))             \
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][3], acc[p], lane)), 24, ws + CHUNK_OFF(4))             \
    /* x[][] is dead until the exchange: the point features live in its place */                             \
    FOR_P(edge_geometry(A.pts + (size_t)n[p] * 48, A.frames + (size_t)n[p] * 12, A.pts + (size_t)nbr[p] * 48, h, g[p])) \
    STAGE1(FOR_P(mfma_tile24(CURBUF, wave, g[p][0], acc[p], lane)), 24, ws + CHUNK_OFF(5))                    \
    STAGE1(FOR_P(mfma_tile24(CURBUF, wave, g[p][1], acc[p], lane)), 24, ws + CHUNK_OFF(6))                    \
    STAGE1(FOR_P(mfma_tile24(CURBUF, wave, g[p][2], acc[p], lane); relu_tile(acc[p]);                         \
                 xbuf_put(xbuf + p * XBUF_FLOATS, wave, lane, acc[p])),                                       \
           32, ws + CHUNK_OFF(7))

__device__ __forceinline__ void wg_residues(const Edge2Args &B, int blk, int (&n)[2], bool &two) {
    if (blk < B.n_dual) { n[0] = 2 * blk; n[1] = 2 * blk + 1; two = true; }
    else { n[0] = 2 * B.n_dual + (blk - B.n_dual); n[1] = n[0]; two = false; }
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ET, 2)
k_node_message2(Edge2Args B) {
    const EdgeArgs &A = B.e;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + WBUF_FLOATS, *xbuf = smem + 2 * WBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int K = A.K;
    int n[2];
    bool two;
    wg_residues(B, blockIdx.x, n, two);
    // masked / padded residues produce zeros; a workgroup whose residues are all masked leaves at once
    const bool live0 = A.rmask[n[0]] != 0.f, live1 = two && A.rmask[n[1]] != 0.f;
    if (!live0) {
        if (tid < 128) A.S[(size_t)n[0] * 128 + tid] = 0.f;
        if (tid == 0) A.msum[n[0]] = 0.f;
    }
    if (two && !live1) {
        if (tid < 128) A.S[(size_t)n[1] * 128 + tid] = 0.f;
        if (tid == 0) A.msum[n[1]] = 0.f;
    }
    if (!live0 && !live1) return;
    if (!live0) { n[0] = n[1]; two = false; }          // only the second one is real: run it in slot 0
    else if (!live1) two = false;
    int cur = 0;

    f32x16 x[2][4], acc[2];
    float g[2][3][12];
    int nbr[2];
    const int jj = j < K ? j : K - 1;
    FOR_P(
        nbr[p] = A.eidx[(size_t)n[p] * K + jj];
        const float *hrow = A.hE_in + ((size_t)n[p] * K + jj) * 128;
        _Pragma("unroll") for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[p][t]);
        load_tile(A.PA + (size_t)n[p] * 128 + 32 * wave, h, acc[p]);
        add_tile(A.PC + (size_t)nbr[p] * 128 + 32 * wave, h, acc[p]))
    const float *ws = A.wstream;          // chunks: W_B 0..3, W_G 4..6, W_mid 7..10
    WRegs R;
    FIRST_LAYER2()
    {
        const float b = A.b_mid[32 * wave + j];           // SWAP form: feature on the lane
        FOR_P(
            _Pragma("unroll") for (int t = 0; t < 4; t++) xbuf_get(xbuf + p * XBUF_FLOATS, t, lane, x[p][t]);
            _Pragma("unroll") for (int r = 0; r < 16; r++) acc[p][r] = b)
    }
    STAGE1(FOR_P(mfma_tile32<true>(CURBUF, wave, x[p][0], acc[p], lane)), 32, ws + CHUNK_OFF(8))
    STAGE1(FOR_P(mfma_tile32<true>(CURBUF, wave, x[p][1], acc[p], lane)), 32, ws + CHUNK_OFF(9))
    STAGE1(FOR_P(mfma_tile32<true>(CURBUF, wave, x[p][2], acc[p], lane)), 32, ws + CHUNK_OFF(10))
    FOR_P(
        mfma_tile32<true>(CURBUF, wave, x[p][3], acc[p], lane);
        // rows (registers) are edges e = 8 (r>>2) + 4 h + (r&3); mask and reduce over them
        float m16[16];
        const float *mrow = A.mask_att + (size_t)n[p] * 32;
        _Pragma("unroll") for (int q = 0; q < 4; q++) {
            f32x4v mm = *reinterpret_cast<const f32x4v *>(mrow + 8 * q + 4 * h);
            m16[4 * q] = mm[0]; m16[4 * q + 1] = mm[1]; m16[4 * q + 2] = mm[2]; m16[4 * q + 3] = mm[3];
        }
        float s = 0.f; float ms = 0.f;
        _Pragma("unroll") for (int r = 0; r < 16; r++) {
            s = fmaf(fmaxf(acc[p][r], 0.f), m16[r], s);
            ms += m16[r];
        }
        s += __shfl_xor(s, 32);
        ms += __shfl_xor(ms, 32);
        if (h == 0) A.S[(size_t)n[p] * 128 + 32 * wave + j] = s * A.inv_K;
        if (tid == 0) A.msum[n[p]] = ms * A.inv_K)
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ET, 2)
k_edge_update2(Edge2Args B) {
    const EdgeArgs &A = B.e;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + WBUF_FLOATS, *xbuf = smem + 2 * WBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int K = A.K;
    const int jj = j < K ? j : K - 1;
    int n[2];
    bool two;
    wg_residues(B, blockIdx.x, n, two);
    const bool live0 = A.rmask[n[0]] != 0.f, live1 = two && A.rmask[n[1]] != 0.f;
    {   // masked / padded residues: their edges are zero
        const f32x4v z = {0.f, 0.f, 0.f, 0.f};
        if (!live0 && j < K) {
            float *orow = A.hE_out + ((size_t)n[0] * K + j) * 128 + 32 * wave;
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4v *>(orow + 8 * q + 4 * h) = z;
        }
        if (two && !live1 && j < K) {
            float *orow = A.hE_out + ((size_t)n[1] * K + j) * 128 + 32 * wave;
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4v *>(orow + 8 * q + 4 * h) = z;
        }
    }
    if (!live0 && !live1) return;
    if (!live0) { n[0] = n[1]; two = false; }
    else if (!live1) two = false;
    int cur = 0;

    f32x16 x[2][4], acc[2], out[2];
    float g[2][3][12];
    int nbr[2];
    float me[2];
    FOR_P(
        nbr[p] = A.eidx[(size_t)n[p] * K + jj];
        me[p] = A.mask_att[(size_t)n[p] * 32 + j];
        const float *hrow = A.hE_in + ((size_t)n[p] * K + jj) * 128;
        _Pragma("unroll") for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[p][t]);
        load_tile(A.PA + (size_t)n[p] * 128 + 32 * wave, h, acc[p]);
        add_tile(A.PC + (size_t)nbr[p] * 128 + 32 * wave, h, acc[p]))
    const float *ws = A.wstream;   // chunks: W_B 0..3, W_G 4..6, W_mid 7..10, W_out 11..14, then per c: W1 x4, W2 x4
    WRegs R;
    FIRST_LAYER2()
    // ---- second layer (chunks 7..10) -------------------------------------------------------------
    FOR_P(
        _Pragma("unroll") for (int t = 0; t < 4; t++) xbuf_get(xbuf + p * XBUF_FLOATS, t, lane, x[p][t]);
        load_tile(A.b_mid + 32 * wave, h, acc[p]))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][0], acc[p], lane)), 32, ws + CHUNK_OFF(8))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][1], acc[p], lane)), 32, ws + CHUNK_OFF(9))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][2], acc[p], lane)), 32, ws + CHUNK_OFF(10))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][3], acc[p], lane); relu_tile(acc[p]);
                 xbuf_put(xbuf + p * XBUF_FLOATS, wave, lane, acc[p])),
           32, ws + CHUNK_OFF(11))
    // ---- third layer (chunks 11..14) --------------------------------------------------------------
    FOR_P(
        _Pragma("unroll") for (int t = 0; t < 4; t++) xbuf_get(xbuf + p * XBUF_FLOATS, t, lane, x[p][t]);
        load_tile(A.b_out + 32 * wave, h, acc[p]))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][0], acc[p], lane)), 32, ws + CHUNK_OFF(12))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][1], acc[p], lane)), 32, ws + CHUNK_OFF(13))
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][2], acc[p], lane)), 32, ws + CHUNK_OFF(14))
    // last chunk; then publish v = h_E + mask * m for the first LayerNorm
    STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][3], acc[p], lane);
                 f32x16 v;
                 load_tile(A.hE_in + ((size_t)n[p] * K + jj) * 128 + 32 * wave, h, v);
                 _Pragma("unroll") for (int r = 0; r < 16; r++) v[r] = fmaf(acc[p][r], me[p], v[r]);
                 xbuf_put(xbuf + p * XBUF_FLOATS, wave, lane, v)),
           32, ws + CHUNK_OFF(15))
    // x1 = LN2(v): every wave normalises the full vector (it needs all of x1 as B operands)
    FOR_P(
        _Pragma("unroll") for (int t = 0; t < 4; t++) xbuf_get(xbuf + p * XBUF_FLOATS, t, lane, x[p][t]);
        float mean;
        const float rstd = ln_center(x[p], mean);
        _Pragma("unroll") for (int t = 0; t < 4; t++) ln_affine_tile(x[p][t], rstd, A.g2 + 32 * t, A.be2 + 32 * t, h);
        load_tile(A.ffn_out_b + 32 * wave, h, out[p]))
    // ---- FFN 128 -> 512 -> 128 in four hidden blocks of 128 (chunks 15 + 8c ..) ------------------------
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float *wc = ws + CHUNK_OFF(15 + 8 * c);             // this block's 8 chunks: W1 s=0..3, W2 s'=0..3
        FOR_P(load_tile(A.ffn_in_b + 128 * c + 32 * wave, h, acc[p]))
        STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][0], acc[p], lane)), 32, wc + 1 * CH32)
        STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][1], acc[p], lane)), 32, wc + 2 * CH32)
        STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][2], acc[p], lane)), 32, wc + 3 * CH32)
        STAGE1(FOR_P(mfma_tile32<false>(CURBUF, wave, x[p][3], acc[p], lane); relu_tile(acc[p]);
                     xbuf_put(xbuf + p * XBUF_FLOATS, wave, lane, acc[p])),
               32, wc + 4 * CH32)
        // second FFN layer over this hidden block: B operands come tile by tile from the exchange buffer
        STAGE1(FOR_P(xbuf_get(xbuf + p * XBUF_FLOATS, 0, lane, acc[p]); mfma_tile32<false>(CURBUF, wave, acc[p], out[p], lane)),
               32, wc + 5 * CH32)
        STAGE1(FOR_P(xbuf_get(xbuf + p * XBUF_FLOATS, 1, lane, acc[p]); mfma_tile32<false>(CURBUF, wave, acc[p], out[p], lane)),
               32, wc + 6 * CH32)
        STAGE1(FOR_P(xbuf_get(xbuf + p * XBUF_FLOATS, 2, lane, acc[p]); mfma_tile32<false>(CURBUF, wave, acc[p], out[p], lane)),
               32, wc + 7 * CH32)
        if (c < 3) {
            STAGE1(FOR_P(xbuf_get(xbuf + p * XBUF_FLOATS, 3, lane, acc[p]); mfma_tile32<false>(CURBUF, wave, acc[p], out[p], lane)),
                   32, wc + 8 * CH32)
        } else {
            FOR_P(xbuf_get(xbuf + p * XBUF_FLOATS, 3, lane, acc[p]); mfma_tile32<false>(CURBUF, wave, acc[p], out[p], lane))
            __syncthreads();          // every wave is done reading the hidden tiles before they are overwritten
        }
    }
    // ---- h_E = mask * LN3(x1 + ffn) ---------------------------------------------------------------------
    FOR_P(
        // residual: this wave's tile of x1 (wave is scalar: uniform branches, static register indices)
        if (wave == 0) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[p][r] += x[p][0][r]; }
        else if (wave == 1) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[p][r] += x[p][1][r]; }
        else if (wave == 2) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[p][r] += x[p][2][r]; }
        else { _Pragma("unroll") for (int r = 0; r < 16; r++) out[p][r] += x[p][3][r]; }
        xbuf_put(xbuf + p * XBUF_FLOATS, wave, lane, out[p]))
    __syncthreads();
    FOR_P(
        _Pragma("unroll") for (int t = 0; t < 4; t++) xbuf_get(xbuf + p * XBUF_FLOATS, t, lane, x[p][t]);
        float mean3;
        const float rstd = ln_center(x[p], mean3);
        _Pragma("unroll") for (int r = 0; r < 16; r++) out[p][r] -= mean3;
        ln_affine_tile(out[p], rstd, A.g3 + 32 * wave, A.be3 + 32 * wave, h);
        _Pragma("unroll") for (int r = 0; r < 16; r++) out[p][r] *= me[p];
        if (j < K) store_tile(A.hE_out + ((size_t)n[p] * K + j) * 128 + 32 * wave, h, out[p]))
}

// ---------------------------------------------------------------------------------------------
static const size_t EDGE2_SMEM = (2 * WBUF_FLOATS + XBUF2_FLOATS) * sizeof(float);

// number of dual / single workgroups for N residues on a device with `slots` resident workgroups
static void split_workgroups(int N, int slots, int *n_dual, int *n_single) {
    if (N <= slots) { *n_dual = 0; *n_single = N; }
    else if (N <= 2 * slots) { *n_dual = N - slots; *n_single = slots - *n_dual; }
    else { *n_dual = N / 2; *n_single = N & 1; }
}

static int wg_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            slots = 2 * prop.multiProcessorCount;
        else slots = 512;
    }
    return slots;
}

EdgeArgs pp_edge_args(pp_ctx *c, int layer, bool edge);

pp_status pp_launch_node_message2(pp_ctx *c, int layer, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_message2),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE2_SMEM));
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_update2),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE2_SMEM));
        attr = true;
    }
    Edge2Args B;
    B.e = pp_edge_args(c, layer, false);
    int ns;
    split_workgroups(c->N, wg_slots(), &B.n_dual, &ns);
    hipLaunchKernelGGL(k_node_message2, dim3(B.n_dual + ns), dim3(ET), EDGE2_SMEM, s, B);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_edge_update2(pp_ctx *c, int layer, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_node_message2),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE2_SMEM));
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_edge_update2),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE2_SMEM));
        attr = true;
    }
    Edge2Args B;
    B.e = pp_edge_args(c, layer, true);
    int ns;
    split_workgroups(c->N, wg_slots(), &B.n_dual, &ns);
    hipLaunchKernelGGL(k_edge_update2, dim3(B.n_dual + ns), dim3(ET), EDGE2_SMEM, s, B);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

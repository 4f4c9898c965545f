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
#include "pp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define ET 256
#define WBUF_FLOATS (128 * 36)          // one weight chunk slot, row stride 36 (32 cols) or 28 (24 cols)
#define XBUF_FLOATS (4 * 4 * 64 * 4)    // exchange buffer: [tile][quad][lane] float4

// Timing-only ablation switches (tools/debug/ablate_edge.py); never defined in a shipped build.
#ifdef PP_X_NOMFMA
#define MFMA(a, b, c) ((c) + (a) * (b))
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif
#ifdef PP_X_NOBARRIER
#define STAGE_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define STAGE_SYNC() __syncthreads()
#endif

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
    const float *wstream;      // this kernel's weight chunks, packed in consumption order
    const float *b_mid, *b_out;
    const float *g2, *be2, *g3, *be3;
    const float *ffn_in_b, *ffn_out_b;
    unsigned long long *dbg;   // PP_X_STAMP builds only: [N][4 waves][64 stages][4 stamps]
};

// one weight chunk in flight through registers: [128 rows][NC cols], NC = 32 (4 float4 per thread) or 24 (3)
struct WRegs {
    f32x4v v[4];
};

// The weight chunks are pre-packed (pp_plan_create) in the order the kernel consumes them, each [128 rows][NC cols]
// chunk contiguous: a chunk is one linear 16 KB (12 KB) read that spreads over every L2 channel.  (Reading the chunks
// in place from the [out][in] matrices put all 128 rows of a chunk of the 2 KB-stride FFN matrix on one or two
// channels and made the kernel L2-bound.)   global chunk -> registers -> LDS (row stride NC + 4)
template <int NC>
__device__ __forceinline__ void chunk_load(const float *__restrict__ g, WRegs &r, int tid) {
    constexpr int PER_ROW = NC / 4, PER_T = (128 * PER_ROW) / ET;
    static_assert((128 * PER_ROW) % ET == 0 && PER_T <= 4, "chunk shape");
#ifdef PP_X_NOLOAD
    for (int m = 0; m < PER_T; m++) r.v[m] = f32x4v{0.001f * tid, 0.f, 0.f, 0.f};
    return;
#endif
#pragma unroll
    for (int m = 0; m < PER_T; m++) r.v[m] = *reinterpret_cast<const f32x4v *>(g + 4 * (tid + ET * m));
}
template <int NC>
__device__ __forceinline__ void chunk_store(float *lds, const WRegs &r, int tid) {
    constexpr int PER_ROW = NC / 4, PER_T = (128 * PER_ROW) / ET;
#ifdef PP_X_NOSTORE
    return;
#endif
#pragma unroll
    for (int m = 0; m < PER_T; m++) {
        int idx = tid + ET * m;
        int row = idx / PER_ROW, c4 = idx - row * PER_ROW;
        *reinterpret_cast<f32x4v *>(lds + row * (NC + 4) + 4 * c4) = r.v[m];
    }
}

// acc += W[32 wave .. +32, chunk cols] * x     (SWAP: acc += x * W^T, edges on rows / features on lanes)
template <bool SWAP>
__device__ __forceinline__ void mfma_tile32(const float *lds, int wave, const f32x16 &x, f32x16 &acc, int lane) {
    const float *base = lds + (32 * wave + (lane & 31)) * 36 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(base + 8 * q);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (SWAP) acc = MFMA(x[4 * q + p], a[p], acc);
            else acc = MFMA(a[p], x[4 * q + p], acc);
        }
    }
}

// geometry chunk: 24 inputs = 12 k-steps; lane half h supplies input 12 h + m at step m
__device__ __forceinline__ void mfma_tile24(const float *lds, int wave, const float (&g)[12], f32x16 &acc, int lane) {
    const float *base = lds + (32 * wave + (lane & 31)) * 28 + 12 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(base + 4 * q);
#pragma unroll
        for (int p = 0; p < 4; p++) acc = MFMA(a[p], g[4 * q + p], acc);
    }
}

// one tile (16 registers) <-> 32 consecutive features of a row-major vector
__device__ __forceinline__ void load_tile(const float *__restrict__ row32, int h, f32x16 &d) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(row32 + 8 * q + 4 * h);
        d[4 * q] = a[0]; d[4 * q + 1] = a[1]; d[4 * q + 2] = a[2]; d[4 * q + 3] = a[3];
    }
}
__device__ __forceinline__ void add_tile(const float *__restrict__ row32, int h, f32x16 &d) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(row32 + 8 * q + 4 * h);
        d[4 * q] += a[0]; d[4 * q + 1] += a[1]; d[4 * q + 2] += a[2]; d[4 * q + 3] += a[3];
    }
}
__device__ __forceinline__ void store_tile(float *__restrict__ row32, int h, const f32x16 &d) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
        *reinterpret_cast<f32x4v *>(row32 + 8 * q + 4 * h) = a;
    }
}
__device__ __forceinline__ void relu_tile(f32x16 &d) {
#pragma unroll
    for (int r = 0; r < 16; r++) d[r] = fmaxf(d[r], 0.f);
}

// exchange buffer: tile t, quad q, lane l -> float4
__device__ __forceinline__ void xbuf_put(float *xbuf, int t, int lane, const f32x16 &d) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
        *reinterpret_cast<f32x4v *>(xbuf + ((t * 4 + q) * 64 + lane) * 4) = a;
    }
}
__device__ __forceinline__ void xbuf_get(const float *xbuf, int t, int lane, f32x16 &d) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(xbuf + ((t * 4 + q) * 64 + lane) * 4);
        d[4 * q] = a[0]; d[4 * q + 1] = a[1]; d[4 * q + 2] = a[2]; d[4 * q + 3] = a[3];
    }
}

// LayerNorm statistics over the 128 features of this lane's edge (64 here, 64 in lane ^ 32); v is centred in
// place; returns 1/std, writes the mean
__device__ __forceinline__ float ln_center(f32x16 (&v)[4], float &mean_out) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += v[t][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.f / 128.f);
    mean_out = mean;
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
    return 1.f / sqrtf(q * (1.f / 128.f) + 1e-5f);
}
// centred tile -> tile * rstd * gamma + beta
__device__ __forceinline__ void ln_affine_tile(f32x16 &v, float rstd, const float *__restrict__ gamma32,
                                               const float *__restrict__ beta32, int h) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v g = *reinterpret_cast<const f32x4v *>(gamma32 + 8 * q + 4 * h);
        f32x4v b = *reinterpret_cast<const f32x4v *>(beta32 + 8 * q + 4 * h);
#pragma unroll
        for (int p = 0; p < 4; p++) v[4 * q + p] = fmaf(v[4 * q + p] * rstd, g[p], b[p]);
    }
}

// 72 invariant point features of edge (i, j); g[c][m] = feature 24 c + 12 h + m (what this lane half feeds the MFMA)
__device__ __forceinline__ void edge_geometry(const float *__restrict__ pts_i, const float *__restrict__ fr,
                                              const float *__restrict__ pts_j, int h, float (&g)[3][12]) {
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
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int m = 0; m < 12; m++) g[c][m] = h ? geom[24 * c + 12 + m] : geom[24 * c + m];
}

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
        chunk_load<RL_NC>((RL_PTR), RL, tid);                              \
        { COMPUTE; }                                                       \
        STAMP(1)                                                           \
        chunk_store<RS_NC>(cur ? wbuf0 : wbuf1, RS, tid);                  \
        STAMP(2)                                                           \
        STAGE_SYNC();                                                      \
        STAMP(3)                                                           \
        NEXT_STAGE();                                                      \
        cur ^= 1;                                                          \
    }
// last stages of a kernel: nothing further to load
#define STAGE2_NOLOAD(COMPUTE, RS, RS_NC)                                  \
    {                                                                      \
        { COMPUTE; }                                                       \
        chunk_store<RS_NC>(cur ? wbuf0 : wbuf1, RS, tid);                  \
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
    chunk_load<32>(ws + CHUNK_OFF(0), RA, tid);                                                           \
    chunk_store<32>(wbuf0, RA, tid);                                                                      \
    chunk_load<32>(ws + CHUNK_OFF(1), RB, tid);                                                           \
    __syncthreads();                                                                                      \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[0], acc, lane), RB, 32, RA, 32, ws + CHUNK_OFF(2))          \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[1], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(3))          \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[2], acc, lane), RB, 32, RA, 24, ws + CHUNK_OFF(4))          \
    STAGE2(mfma_tile32<false>(CURBUF, wave, x[3], acc, lane), RA, 24, RB, 24, ws + CHUNK_OFF(5))          \
    /* x[] is dead from here to the exchange: build the 72 point features in its place */                \
    edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g);     \
    STAGE2(mfma_tile24(CURBUF, wave, g[0], acc, lane), RB, 24, RA, 24, ws + CHUNK_OFF(6))                 \
    STAGE2(mfma_tile24(CURBUF, wave, g[1], acc, lane), RA, 24, RB, 32, ws + CHUNK_OFF(7))                 \
    STAGE2(mfma_tile24(CURBUF, wave, g[2], acc, lane); relu_tile(acc); xbuf_put(xbuf, wave, lane, acc),   \
           RB, 32, RA, 32, ws + CHUNK_OFF(8))

// ---------------------------------------------------------------------------------------------
// node message: S[i] = (1/K) sum_j mask_ij relu(W_mid relu(W_in [..]) + b), msum[i] = (1/K) sum_j mask_ij
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ET, 3)
k_node_message(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + WBUF_FLOATS, *xbuf = smem + 2 * WBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    const int K = A.K;
    int cur = 0;
#ifdef PP_X_STAMP
    int stage_no = 0;
#endif
    if (A.rmask[n] == 0.f) {              // masked / padded residue: whole workgroup leaves
        if (tid < 128) A.S[(size_t)n * 128 + tid] = 0.f;
        if (tid == 0) A.msum[n] = 0.f;
        return;
    }

    f32x16 x[4], acc;
    float g[3][12];
    const int jj = j < K ? j : K - 1;
    const int nbr = A.eidx[(size_t)n * K + jj];
    {
        const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
#pragma unroll
        for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
    }
    const float *ws = A.wstream;          // chunks: W_B 0..3, W_G 4..6, W_mid 7..10
    WRegs RA, RB;
    FIRST_LAYER()
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        const float b = A.b_mid[32 * wave + j];           // SWAP form: feature on the lane
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = b;
    }
    STAGE2(mfma_tile32<true>(CURBUF, wave, x[0], acc, lane), RA, 32, RB, 32, ws + CHUNK_OFF(9))
    STAGE2(mfma_tile32<true>(CURBUF, wave, x[1], acc, lane), RB, 32, RA, 32, ws + CHUNK_OFF(10))
    STAGE2_NOLOAD(mfma_tile32<true>(CURBUF, wave, x[2], acc, lane), RA, 32)
    {
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
        if (tid == 0) A.msum[n] = ms * A.inv_K;
    }
}

// ---------------------------------------------------------------------------------------------
// edge update: h_E <- mask * LN3(x1 + FFN(x1)),  x1 = LN2(h_E + mask * MLP3([..]))
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ET, 2)
k_edge_update(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + WBUF_FLOATS, *xbuf = smem + 2 * WBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    const int K = A.K;
    const int jj = j < K ? j : K - 1;
    int cur = 0;
#ifdef PP_X_STAMP
    int stage_no = 0;
#endif
    if (A.rmask[n] == 0.f) {              // masked / padded residue: its edges are zero, whole workgroup leaves
        if (j < K) {
            f32x4v z = {0.f, 0.f, 0.f, 0.f};
            float *orow = A.hE_out + ((size_t)n * K + j) * 128 + 32 * wave;
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4v *>(orow + 8 * q + 4 * h) = z;
        }
        return;
    }

    f32x16 x[4], acc, out;
    float g[3][12];
    const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
    const int nbr = A.eidx[(size_t)n * K + jj];
    const float me = A.mask_att[(size_t)n * 32 + j];
    {
#pragma unroll
        for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
    }
    const float *ws = A.wstream;   // chunks: W_B 0..3, W_G 4..6, W_mid 7..10, W_out 11..14, then per c: W1 x4, W2 x4
    WRegs RA, RB;
    FIRST_LAYER()
    // ---- second layer (chunks 7..10) -------------------------------------------------------------
    {
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
    {
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
    {
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
        load_tile(A.ffn_in_b + 128 * c + 32 * wave, h, acc);
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
            xbuf_get(xbuf, 3, lane, acc);
            mfma_tile32<false>(CURBUF, wave, acc, out, lane);
            __syncthreads();          // every wave is done reading the hidden tiles before they are overwritten
        }
    }
    // ---- h_E = mask * LN3(x1 + ffn) ---------------------------------------------------------------------
    // residual: this wave's tile of x1 (wave is scalar: four uniform branches, static register indices)
    if (wave == 0) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[0][r]; }
    else if (wave == 1) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[1][r]; }
    else if (wave == 2) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[2][r]; }
    else { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[3][r]; }
    xbuf_put(xbuf, wave, lane, out);
    __syncthreads();
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

static const size_t EDGE_SMEM = (2 * WBUF_FLOATS + XBUF_FLOATS) * sizeof(float);

// resident workgroups per CU the runtime predicts for the two kernels (measurement aid)
void pp_edge_occupancy(int *node_msg, int *edge_upd) {
    hipOccupancyMaxActiveBlocksPerMultiprocessor(node_msg, reinterpret_cast<const void *>(k_node_message), ET, EDGE_SMEM);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(edge_upd, reinterpret_cast<const void *>(k_edge_update), ET, EDGE_SMEM);
}

pp_status pp_launch_node_message(pp_ctx *c, int layer, hipStream_t s) {
    EdgeArgs A = edge_args(c, layer, false);
    hipLaunchKernelGGL(k_node_message, dim3(c->N), dim3(ET), EDGE_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_edge_update(pp_ctx *c, int layer, hipStream_t s) {
    EdgeArgs A = edge_args(c, layer, true);
    hipLaunchKernelGGL(k_edge_update, dim3(c->N), dim3(ET), EDGE_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

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
// 128-vector it needs as B operands.  Since wave w needs only weight rows 32w..32w+31, each wave streams its own
// quarter of every [128 rows][32 cols] weight chunk (pre-packed in consumption order) by LDS-DMA into a private
// two-slot ring: no barrier on the weight path, only around the exchanges.  53 KB of LDS and <=168 VGPRs per
// workgroup let 3 workgroups share a CU, so one workgroup's barrier / LDS latency is covered by another's MFMAs,
// and 739 residues x 4 waves spread evenly over the 1024 SIMDs.
// Kernels: k_edge_static (once per complex: layer 0's W_B h_E0), k_node_message (layer 0), k_edge_update (edge update of
// layer l, then the node message of layer l+1 on the fresh edges, in the same workgroup).
//
// The 456-wide first layer is never materialised: W_in [h_V_i | h_E_ij | h_V_j | geom] =
// (W_A h_V_i + b) + W_C h_V_j  (node-level, precomputed per residue in pp_node.hip, gathered here)
// + W_B h_E_ij + W_G geom_ij (MFMA here; the 72 invariant-point features are built in registers).
#include "pp_internal.h"

// exact fp32 operands: no f16 range to check
unsigned int pp_edge_range_hits(int) { return 0; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define ET 256
#define CH32 (128 * 32)                 // floats per packed weight chunk (16 KB); a wave's quarter is 1024 floats
#define XBUF_FLOATS (4 * 4 * 64 * 4)    // exchange buffer: [tile][quad][lane] float4
#define PARAM_FLOATS 1152               // edge kernel: small per-layer vectors staged once

// Timing-only ablation switch (tools/debug/ablate_edge.py); never defined in a shipped build.
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
    const float *wstream;      // this kernel's weight chunks, packed in consumption order (pp_api.hip put_chunk)
    const float *params;       // edge kernel: b_mid | b_out | ffn_out_b | g2 | be2 | ffn_in_b[512]
    const float *g3, *be3;     // edge kernel: last LayerNorm (read in the epilogue)
    const float *b_mid;        // node kernel (per-lane read)
    const float *Z;            // layer 0: precomputed W_B h_E0 of this message function [N][K][128]
    const float *pts2, *PA2, *PC2, *b_mid2;   // fused edge update: node-level inputs / bias of the NEXT node message
    float *Znm, *Zem;          // k_edge_static outputs
};
enum { P_BMID = 0, P_BOUT = 128, P_FOB = 256, P_G2 = 384, P_BE2 = 512, P_FIB = 640 };

// ---- weight pipeline: LDS-DMA, private to each wave ------------------------------------------------------------
// Wave w only ever needs rows 32w..32w+31 of a weight chunk (its output tile), so its quarter of every chunk is
// packed (pp_plan_create) as [quad q][lane][4 floats] = exactly the A-operand registers of 16 MFMAs: one 4 KB piece,
// copied global -> LDS by four `global_load_lds_dwordx4` (1 KB each, no VGPRs, no ds_write) into a per-wave ring of S
// slots and read back lane-linear (conflict-free ds_read_b128).  No other wave touches the slot, so the only
// ordering needed is this wave's own counted s_waitcnt vmcnt (MI355X_MICROARCH.md, co-residence item 7): the weight
// stream needs NO workgroup barrier; barriers remain only around the activation exchange buffer.
// hipcc does not count these loads: between the prologue and the epilogue the kernels issue no ordinary global
// loads (every small vector is staged to LDS or registers up front), so no compiler-made vmcnt(0) drains the ring.
//
// HAZARD (found while moving these kernels to split-f16 MFMA, tools/debug/experiments/pp_edge_f16_split.hip, where
// stages are 4x shorter; measured with tools/debug/edge_repro.py / soak.py): an LDS-DMA instruction reads its address
// VGPRs LATE -- when the memory pipeline accepts it, which with several workgroups per CU can be hundreds of cycles after
// issue -- and nothing interlocks a later VALU write to those registers.  hipcc, for which an asm's inputs are dead at
// its end, reuses them at once; the copy then fetches from a garbage address: sporadic wrong weight tiles, only with
// co-resident workgroups, gone with an s_waitcnt vmcnt(0) after every issue.  The fp32 kernels never showed it (their
// soak is clean over millions of workgroup launches), but they are written to the safe pattern anyway: the per-lane part
// of every DMA address lives in ONE register for the whole kernel (`laneoff` = lane * 16, `rowoff` for the residual
// gather), each statement takes it read-write and every wait names it, so it is never handed to anything else; the
// wave-uniform part of the address goes in SGPRs; and no lane is masked off (divergent control flow) while a copy is in
// flight -- surplus lanes use clamped indices and rewrite the same value.
// M0 (the LDS destination) and the SGPR pairs holding the wave-uniform part of the address are written only by these
// statements, into registers hipcc never allocates here (s90..s99; it uses ~60 from s0 up and no M0), and then rest until
// the next issue a whole stage later: whatever the hardware reads late, it finds unchanged.
__device__ __forceinline__ void dma_chunk(const float *sbase, unsigned &laneoff, unsigned lds_dst) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"          // this wave's reads of the slot being refilled have returned
                 "s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 s[98:99], %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99]\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99] offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99] offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99] offset:3072"
                 : "+v"(laneoff) : "s"(sbase), "s"(lds_dst) : "memory", "s98", "s99");
}
// one tile of a row-major [.][128] row set, gathered per lane (byte offset rowoff of this lane's 32-feature row piece
// + 16 h from sbase) into the [quad][lane][4] register image at lds_dst: quad q is the float4 at + 32 q bytes.  The
// instruction offset moves the global AND the LDS address by the same amount, so quad q is issued with offset 1024 q
// (its LDS place) from a base pulled back by 992 q; M0 is the same for all four.
__device__ __forceinline__ void dma_tile(const float *sbase, unsigned &rowoff, unsigned lds_dst) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 s[90:91], %1\n\t"
                 "s_sub_u32 s92, s90, 992\n\ts_subb_u32 s93, s91, 0\n\t"
                 "s_sub_u32 s94, s90, 1984\n\ts_subb_u32 s95, s91, 0\n\t"
                 "s_sub_u32 s96, s90, 2976\n\ts_subb_u32 s97, s91, 0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, s[90:91]\n\t"
                 "global_load_lds_dwordx4 %0, s[92:93] offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, s[94:95] offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, s[96:97] offset:3072"
                 : "+v"(rowoff) : "s"(sbase), "s"(lds_dst)
                 : "memory", "scc", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97");
}
// counted wait; names the DMA address registers so that they stay allocated (see HAZARD)
template <int N>
__device__ __forceinline__ void wait_vm(unsigned laneoff, unsigned rowoff) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N), "v"(laneoff), "v"(rowoff) : "memory");
}

// acc += W[32 wave .. +32, chunk cols] * x     (SWAP: acc += x * W^T, edges on rows / features on lanes)
template <bool SWAP>
__device__ __forceinline__ void mfma_tile32(const float *wslot, const f32x16 &x, f32x16 &acc, int lane) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(wslot + (q * 64 + lane) * 4);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (SWAP) acc = MFMA(x[4 * q + p], a[p], acc);
            else acc = MFMA(a[p], x[4 * q + p], acc);
        }
    }
}

// geometry chunk: 24 inputs = 12 k-steps; lane half h supplies input 12 h + m at step m (quad 3 of the slot is padding)
__device__ __forceinline__ void mfma_tile24(const float *wslot, const float (&g)[12], f32x16 &acc, int lane) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        f32x4v a = *reinterpret_cast<const f32x4v *>(wslot + (q * 64 + lane) * 4);
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

#define DMA_PER_CHUNK 4
// Stage k of a kernel with NCH chunks: refill the slot freed by stage k-1 with chunk k+S-1, wait until chunk k has
// landed (all but the younger chunks' DMAs retired), compute on it.
#define WSTAGE(k, NCH, BODY)                                                                                   \
    {                                                                                                          \
        if constexpr ((k) + S - 1 < (NCH))                                                                     \
            dma_chunk(wsb + (size_t)((k) + S - 1) * CH32, laneoff, slot0 + (((k) + S - 1) % S) * 4096u);       \
        wait_vm<DMA_PER_CHUNK * (((NCH) - 1 - (k)) < (S - 1) ? ((NCH) - 1 - (k)) : (S - 1))>(laneoff, rowoff);             \
        const float *wslot = wl + ((k) % S) * 1024;                                                            \
        BODY;                                                                                                  \
    }

// shared first layer: acc (tile `wave`) = PA_i + PC_j + W_B h_E + W_G geom, ReLU.  Chunks W_B x4 (absent when ST0:
// layer 0's W_B h_E0 is timestep-invariant and arrives precomputed in acc), then W_G x3.  C0 = number of W_B chunks.
#define FIRST_LAYER(NCH)                                                          \
    if constexpr (!ST0) {                                                         \
        WSTAGE(0, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))                \
        WSTAGE(1, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))                \
        WSTAGE(2, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))                \
        WSTAGE(3, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))                \
    }                                                                             \
    WSTAGE(C0 + 0, NCH, mfma_tile24(wslot, g[0], acc, lane))                      \
    WSTAGE(C0 + 1, NCH, mfma_tile24(wslot, g[1], acc, lane))                      \
    WSTAGE(C0 + 2, NCH, mfma_tile24(wslot, g[2], acc, lane))                      \
    relu_tile(acc);                                                               \
    xbuf_put(xbuf, wave, lane, acc);                                              \
    __syncthreads();

#define PROLOGUE_PIPE()                                                                        \
    const float *wsb = A.wstream + wave * 1024;          /* wave-uniform part of the chunk addresses */ \
    unsigned laneoff = (unsigned)lane * 16u;              /* the one per-lane DMA address register */     \
    unsigned rowoff = 0;                                   /* the residual gather's, where used */          \
    const float *wl = smem + wave * (S * 1024);                                                \
    const unsigned slot0 = (unsigned)(size_t)wl;                                               \
    _Pragma("unroll") for (int pk = 0; pk < S - 1; pk++) dma_chunk(wsb + (size_t)pk * CH32, laneoff, slot0 + pk * 4096u);

// ---------------------------------------------------------------------------------------------
// node message: S[i] = (1/K) sum_j mask_ij relu(W_mid relu(W_in [..]) + b), msum[i] = (1/K) sum_j mask_ij
// ---------------------------------------------------------------------------------------------
#ifndef PP_NM_SLOTS
#define PP_NM_SLOTS 2
#endif
#ifndef PP_EU_SLOTS
#define PP_EU_SLOTS 2
#endif
#ifndef PP_EU_WGS
#define PP_EU_WGS 3
#endif

template <int S, bool ST0>
__global__ void __launch_bounds__(ET, 3)
k_node_message(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xbuf = smem + 4 * S * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    const int K = A.K;
    if (A.rmask[n] == 0.f) {              // masked / padded residue: whole workgroup leaves
        if (tid < 128) A.S[(size_t)n * 128 + tid] = 0.f;
        if (tid == 0) A.msum[n] = 0.f;
        return;
    }
    constexpr int C0 = ST0 ? 0 : 4;
    constexpr int NCH = C0 + 7;           // chunks: [W_B x4,] W_G x3, W_mid x4
    PROLOGUE_PIPE()

    f32x16 x[4], acc;
    float g[3][12];
    const int jj = j < K ? j : K - 1;
    const int nbr = A.eidx[(size_t)n * K + jj];
    const float bmid = A.b_mid[32 * wave + j];            // SWAP form: feature on the lane
    float m16[16];
    {
        const float *mrow = A.mask_att + (size_t)n * 32;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v mm = *reinterpret_cast<const f32x4v *>(mrow + 8 * q + 4 * h);
            m16[4 * q] = mm[0]; m16[4 * q + 1] = mm[1]; m16[4 * q + 2] = mm[2]; m16[4 * q + 3] = mm[3];
        }
    }
    edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g);
    {
        const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
        if constexpr (!ST0) {
#pragma unroll
            for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
        }
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
        if constexpr (ST0) add_tile(A.Z + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);
    }
    FIRST_LAYER(NCH)
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = bmid;
    }
    WSTAGE(C0 + 3, NCH, mfma_tile32<true>(wslot, x[0], acc, lane))
    WSTAGE(C0 + 4, NCH, mfma_tile32<true>(wslot, x[1], acc, lane))
    WSTAGE(C0 + 5, NCH, mfma_tile32<true>(wslot, x[2], acc, lane))
    WSTAGE(C0 + 6, NCH, mfma_tile32<true>(wslot, x[3], acc, lane))
    {
        // rows (registers) are edges e = 8 (r>>2) + 4 h + (r&3); mask and reduce over them
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
// FFN hidden block c (chunks 15 + 8c ..): W1 s=0..3 -> hidden tile 4c+wave -> exchange -> W2 s'=0..3 accumulate into out
#define FFN_BLOCK(c)                                                                                         \
    load_tile(prm + P_FIB + 128 * (c) + 32 * wave, h, acc);                                                  \
    WSTAGE(C0 + 11 + 8 * (c) + 0, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))                                \
    WSTAGE(C0 + 11 + 8 * (c) + 1, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))                                \
    WSTAGE(C0 + 11 + 8 * (c) + 2, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))                                \
    WSTAGE(C0 + 11 + 8 * (c) + 3, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))                                \
    relu_tile(acc);                                                                                          \
    __syncthreads();          /* every wave is done reading the previous exchange */                        \
    xbuf_put(xbuf, wave, lane, acc);                                                                         \
    __syncthreads();                                                                                         \
    WSTAGE(C0 + 11 + 8 * (c) + 4, NCH, xbuf_get(xbuf, 0, lane, acc); mfma_tile32<false>(wslot, acc, out, lane))   \
    WSTAGE(C0 + 11 + 8 * (c) + 5, NCH, xbuf_get(xbuf, 1, lane, acc); mfma_tile32<false>(wslot, acc, out, lane))   \
    WSTAGE(C0 + 11 + 8 * (c) + 6, NCH, xbuf_get(xbuf, 2, lane, acc); mfma_tile32<false>(wslot, acc, out, lane))   \
    WSTAGE(C0 + 11 + 8 * (c) + 7, NCH, xbuf_get(xbuf, 3, lane, acc); mfma_tile32<false>(wslot, acc, out, lane))

// FUSE: the workgroup goes straight on to the NEXT layer's node message of its residue (same 32 edges, whose new
// h_E it holds; the node-level inputs PA2 / PC2 / pts2 were written by the node update that ran before this kernel):
// one launch, one prologue and one read of h_E less per layer.
template <int S, bool ST0, bool FUSE>
__global__ void __launch_bounds__(ET, PP_EU_WGS)
k_edge_update(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xbuf = smem + 4 * S * 1024, *prm = xbuf + XBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    const int K = A.K;
    const int jj = j < K ? j : K - 1;
    if (A.rmask[n] == 0.f) {              // masked / padded residue: its edges are zero, whole workgroup leaves
        if (j < K) {
            f32x4v z = {0.f, 0.f, 0.f, 0.f};
            float *orow = A.hE_out + ((size_t)n * K + j) * 128 + 32 * wave;
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4v *>(orow + 8 * q + 4 * h) = z;
        }
        if constexpr (FUSE) {
            if (tid < 128) A.S[(size_t)n * 128 + tid] = 0.f;
            if (tid == 0) A.msum[n] = 0.f;
        }
        return;
    }
    // chunks: [W_B x4,] W_G x3, W_mid x4, W_out x4, then per hidden block c: W1 x4, W2 x4
    constexpr int C0 = ST0 ? 0 : 4;
    constexpr int NEU = C0 + 43;                       // chunks of the edge update itself
    constexpr int NCH = NEU + (FUSE ? 11 : 0);         // + W_B x4, W_G x3, W_mid x4 of the next node message
    PROLOGUE_PIPE()

    f32x16 x[4], acc, out;
    float g[3][12];
    const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
    const int nbr = A.eidx[(size_t)n * K + jj];
    const float me = A.mask_att[(size_t)n * 32 + jj];          // (lanes j >= K mirror edge K - 1 throughout)
    // the small per-layer vectors go to LDS once (published by the first exchange barrier)
#pragma unroll
    for (int it = 0; it < (PARAM_FLOATS / 4 + ET - 1) / ET; it++) {          // every lane active (see HAZARD): clamped index
        const int i = min(tid + it * ET, PARAM_FLOATS / 4 - 1);
        *reinterpret_cast<f32x4v *>(prm + 4 * i) = *reinterpret_cast<const f32x4v *>(A.params + 4 * i);
    }
    edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g);
    __builtin_amdgcn_sched_barrier(0);        // geometry temporaries die before the activation tiles are loaded
    {
        if constexpr (!ST0) {
#pragma unroll
            for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
        }
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
        if constexpr (ST0) add_tile(A.Z + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);
    }
    FIRST_LAYER(NCH)
    // ---- second layer (chunks 7..10) -------------------------------------------------------------
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        load_tile(prm + P_BMID + 32 * wave, h, acc);
    }
    WSTAGE(C0 + 3, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))
    WSTAGE(C0 + 4, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))
    WSTAGE(C0 + 5, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))
    WSTAGE(C0 + 6, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))
    relu_tile(acc);
    __syncthreads();
    xbuf_put(xbuf, wave, lane, acc);
    __syncthreads();
    // ---- third layer (chunks 11..14) --------------------------------------------------------------
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        load_tile(prm + P_BOUT + 32 * wave, h, acc);
    }
    // The exchange buffer is idle during this layer: once every wave has its B operands (barrier), each wave parks
    // the residual input of the first LayerNorm -- its own tile of h_E -- in its own exchange tile by LDS-DMA.  The copy
    // is older than the weight chunks issued below, so the vmcnt wait of the layer's last stage covers it.
    __syncthreads();
    rowoff = (unsigned)((((size_t)n * K + jj) * 128 + 32 * wave + 4 * h) * sizeof(float));
    dma_tile(A.hE_in, rowoff, (unsigned)(size_t)(xbuf + wave * 1024));
    WSTAGE(C0 + 7, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))
    WSTAGE(C0 + 8, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))
    WSTAGE(C0 + 9, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))
    WSTAGE(C0 + 10, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))
    // publish v = h_E + mask * m for the first LayerNorm (own tile: read, then overwritten in place)
    xbuf_get(xbuf, wave, lane, out);
#pragma unroll
    for (int r = 0; r < 16; r++) out[r] = fmaf(acc[r], me, out[r]);
    xbuf_put(xbuf, wave, lane, out);
    __syncthreads();
    {
        // x1 = LN2(v): every wave normalises the full vector (it needs all of x1 as B operands)
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        float mean;
        float rstd = ln_center(x, mean);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            // compiler fence tied to the data flow: the gamma / beta reads of tile t are issued only once rstd (and the
            // previous tile) exist, so one tile's worth of them is live at a time (168-VGPR budget)
            if (t == 0) asm volatile("" : "+v"(rstd) : : "memory");
            else asm volatile("" : "+v"(x[t > 0 ? t - 1 : 0][15]) : : "memory");
            ln_affine_tile(x[t], rstd, prm + P_G2 + 32 * t, prm + P_BE2 + 32 * t, h);
        }
        load_tile(prm + P_FOB + 32 * wave, h, out);
    }
    // ---- FFN 128 -> 512 -> 128 in four hidden blocks of 128 ------------------------------------------
    FFN_BLOCK(0)
    FFN_BLOCK(1)
    FFN_BLOCK(2)
    FFN_BLOCK(3)
    // ---- h_E = mask * LN3(x1 + ffn) ---------------------------------------------------------------------
    if (wave == 0) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[0][r]; }
    else if (wave == 1) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[1][r]; }
    else if (wave == 2) { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[2][r]; }
    else { _Pragma("unroll") for (int r = 0; r < 16; r++) out[r] += x[3][r]; }
    __syncthreads();
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
    // lanes j >= K mirror edge K - 1 (same inputs, same value): they store it again rather than being masked off
    store_tile(A.hE_out + ((size_t)n * K + jj) * 128 + 32 * wave, h, out);
    if constexpr (FUSE) {
        // ---- next layer's node message on the fresh edges ------------------------------------------------
        // its inputs are fetched here and not earlier: offsets made opaque behind `out` (scalar ones stay scalar)
        int o_pts = n * 48, o_fr = n * 12, o_pa = n * 128;
        int o_ptsj = nbr * 48, o_pc = nbr * 128;
        __syncthreads();                                      // every wave has read the LayerNorm exchange
        xbuf_put(xbuf, wave, lane, out);
        // `out` is dead from here; the fence keeps the input fetches below it (they would otherwise be hoisted to the
        // top of the kernel), and the geometry arithmetic fills the wait for the other waves' tiles
        asm volatile("" : "+s"(o_pts), "+s"(o_fr), "+s"(o_pa), "+v"(o_ptsj), "+v"(o_pc) : : "memory");
        edge_geometry(A.pts2 + o_pts, A.frames + o_fr, A.pts2 + o_ptsj, h, g);
        load_tile(A.PA2 + o_pa + 32 * wave, h, acc);
        add_tile(A.PC2 + o_pc + 32 * wave, h, acc);
        const float bmid = A.b_mid2[32 * wave + j];           // SWAP form: feature on the lane
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
        WSTAGE(NEU + 0, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))
        WSTAGE(NEU + 1, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))
        WSTAGE(NEU + 2, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))
        WSTAGE(NEU + 3, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))
        WSTAGE(NEU + 4, NCH, mfma_tile24(wslot, g[0], acc, lane))
        WSTAGE(NEU + 5, NCH, mfma_tile24(wslot, g[1], acc, lane))
        WSTAGE(NEU + 6, NCH, mfma_tile24(wslot, g[2], acc, lane))
        relu_tile(acc);
        __syncthreads();
        xbuf_put(xbuf, wave, lane, acc);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, x[t]);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = bmid;
        WSTAGE(NEU + 7, NCH, mfma_tile32<true>(wslot, x[0], acc, lane))
        WSTAGE(NEU + 8, NCH, mfma_tile32<true>(wslot, x[1], acc, lane))
        WSTAGE(NEU + 9, NCH, mfma_tile32<true>(wslot, x[2], acc, lane))
        WSTAGE(NEU + 10, NCH, mfma_tile32<true>(wslot, x[3], acc, lane))
        // rows (registers) are edges e = 8 (r>>2) + 4 h + (r&3); mask and reduce over them
        int o_m = n * 32 + 4 * h;
        asm volatile("" : "+v"(o_m) : "v"(acc[0]));
        float sacc = 0.f, ms = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4v mm = *reinterpret_cast<const f32x4v *>(A.mask_att + o_m + 8 * q);
#pragma unroll
            for (int pq = 0; pq < 4; pq++) {
                sacc = fmaf(fmaxf(acc[4 * q + pq], 0.f), mm[pq], sacc);
                ms += mm[pq];
            }
        }
        sacc += __shfl_xor(sacc, 32);
        ms += __shfl_xor(ms, 32);
        if (h == 0) A.S[(size_t)n * 128 + 32 * wave + j] = sacc * A.inv_K;
        if (tid == 0) A.msum[n] = ms * A.inv_K;
    }
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// once per complex: Z_nm = W_B(node message, layer 0) h_E0 and Z_em = W_B(edge message, layer 0) h_E0.  h_E0 never
// changes during sampling, so the layer-0 kernels skip four of their stages and start from these tiles.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(ET, 3)
k_edge_static(EdgeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    const int K = A.K;
    if (A.rmask[n] == 0.f) return;        // never read: the layer kernels leave masked residues early as well
    constexpr int NCH = 8;                // chunks: W_B(node message) x4, W_B(edge message) x4
    PROLOGUE_PIPE()
    f32x16 x[4], acc;
    const int jj = j < K ? j : K - 1;
    const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
#pragma unroll
    for (int t = 0; t < 4; t++) load_tile(hrow + 32 * t, h, x[t]);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    WSTAGE(0, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))
    WSTAGE(1, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))
    WSTAGE(2, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))
    WSTAGE(3, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))
    store_tile(A.Znm + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);      // lanes j >= K mirror edge K - 1
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    WSTAGE(4, NCH, mfma_tile32<false>(wslot, x[0], acc, lane))
    WSTAGE(5, NCH, mfma_tile32<false>(wslot, x[1], acc, lane))
    WSTAGE(6, NCH, mfma_tile32<false>(wslot, x[2], acc, lane))
    WSTAGE(7, NCH, mfma_tile32<false>(wslot, x[3], acc, lane))
    store_tile(A.Zem + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);
}

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
    A.wstream = edge ? p->lt[layer].em_stream : p->lt[layer].nm_stream;
    A.params = p->lt[layer].em_params;
    A.g3 = p->w + o.norm_g[3]; A.be3 = p->w + o.norm_b[3];
    A.b_mid = p->w + o.nm_mid_b;
    A.Z = edge ? c->Zem : c->Znm;
    A.Znm = c->Znm; A.Zem = c->Zem;
    A.pts2 = c->ptsN; A.PA2 = c->PAn; A.PC2 = c->PCn;
    A.b_mid2 = p->w + p->off.layer[layer < 2 ? layer + 1 : 2].nm_mid_b;
    return A;
}

static const size_t NM_SMEM = (4 * PP_NM_SLOTS * 1024 + XBUF_FLOATS) * sizeof(float);
static const size_t EU_SMEM = (4 * PP_EU_SLOTS * 1024 + XBUF_FLOATS + PARAM_FLOATS) * sizeof(float);
static const size_t ST_SMEM = (4 * 2 * 1024) * sizeof(float);

static bool edge_attrs() {
    static bool done = false, ok = false;
    if (!done) {
        done = true;
        auto set = [](const void *f, size_t bytes) {
            return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
        };
        ok = set(reinterpret_cast<const void *>(k_node_message<PP_NM_SLOTS, false>), NM_SMEM) &&
             set(reinterpret_cast<const void *>(k_node_message<PP_NM_SLOTS, true>), NM_SMEM) &&
             set(reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, false, true>), EU_SMEM) &&
             set(reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, true, true>), EU_SMEM);
    }
    return ok;
}

// resident workgroups per CU the runtime predicts for the two kernels (measurement aid)
void pp_edge_occupancy(int *node_msg, int *edge_upd) {
    edge_attrs();
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(node_msg, reinterpret_cast<const void *>(k_node_message<PP_NM_SLOTS, false>), ET, NM_SMEM);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(edge_upd, reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, false, true>), ET, EU_SMEM);
}

#define EDGE_ATTR_CHECK()                                                                                   \
    if (!edge_attrs()) {                                                                                    \
        pp_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the edge kernels");        \
        return PP_ERR_HIP;                                                                                  \
    }

extern "C" int pp_edge_variant(void) { return 0; }
bool pp_edge_fused() { return true; }

pp_status pp_launch_edge_static(pp_ctx *c, hipStream_t s) {
    EDGE_ATTR_CHECK()
    EdgeArgs A = edge_args(c, 0, false);
    A.wstream = c->plan->static_stream;
    hipLaunchKernelGGL(k_edge_static<2>, dim3(c->N), dim3(ET), ST_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_node_message(pp_ctx *c, int layer, hipStream_t s) {
    EDGE_ATTR_CHECK()
    EdgeArgs A = edge_args(c, layer, false);
    if (layer == 0) PP_LAUNCH(c, (k_node_message<PP_NM_SLOTS, true>), dim3(c->N), dim3(ET), NM_SMEM, s, A);
    else PP_LAUNCH(c, (k_node_message<PP_NM_SLOTS, false>), dim3(c->N), dim3(ET), NM_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// layers 0 and 1 only (the reference's layer-2 edge update is dead code); also produces S / msum of layer + 1
pp_status pp_launch_edge_update(pp_ctx *c, int layer, hipStream_t s) {
    EDGE_ATTR_CHECK()
    if (layer < 0 || layer > 1) { pp_set_error("pp_launch_edge_update: layer must be 0 or 1"); return PP_ERR_INVALID; }
    EdgeArgs A = edge_args(c, layer, true);
    if (layer == 0) PP_LAUNCH(c, (k_edge_update<PP_EU_SLOTS, true, true>), dim3(c->N), dim3(ET), EU_SMEM, s, A);
    else PP_LAUNCH(c, (k_edge_update<PP_EU_SLOTS, false, true>), dim3(c->N), dim3(ET), EU_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

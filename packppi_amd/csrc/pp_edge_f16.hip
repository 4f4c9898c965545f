// EXPERIMENTAL, opt-in (PACKPPI_EDGE=f16 python -m packppi_amd.build --force): the edge-level stages of
// InvariantPointMessagePassing (layers.py:65-148) on the F16 matrix pipe with fp32-level accuracy.  The shipped kernels
// are the exact-fp32 ones in pp_edge.hip; this file has the same structure and replaces the same launchers.
//
// Same decomposition as pp_edge.hip: a workgroup of 4 waves owns ONE residue and its K<=32 edges, activations live in the
// 32x32 MFMA accumulator layout (lane = (edge, half h); register r of tile t <-> feature 32 t + 8 (r >> 2) + 4 h + (r & 3))
// so that a layer's outputs are the next layer's B operands, wave w computes output tile w of every layer (N-split)
// and publishes it through a 16 KB LDS exchange buffer, weights arrive by wave-private LDS-DMA, the 456-wide first layer
// is split by linearity.  What differs:
//   * arithmetic: every fp32 product is three v_mfma_f32_32x32x16_f16 on two-way f16 splits (below);
//   * occupancy: ONE workgroup per CU.  With several waves interleaving on a SIMD these kernels produced rare wrong
//     tiles that could not be fully explained ("Split-f16"); with one wave per SIMD they are bit-reproducible
//     over millions of workgroup launches (tools/debug/soak.py).  The LDS request (96 / 84.5 KB) enforces it;
//   * pipeline: since no other workgroup hides latency, the DMA ring is 4-5 slots deep (counted waits) and the A operands
//     of stage k+1 are read into the next of three rotating register sets while stage k computes;
//   * no fusion of the next node message into the edge update (the fused tail was the least reliable part).
// Measured (MI355X, T1124, 100 steps): 30.0 k residues/s against 26.6 k for pp_edge.hip; single workgroups take 18 us
// (edge update) / 7 us (node message), so small complexes gain most (L = 256: 19.8 us vs 43 us per edge update).
#include "pp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---- split-f16 arithmetic -------------------------------------------------------------------------------------
// Every fp32 operand x of the dense layers is carried as two f16 numbers, hi = f16(x) and lo = f16(x - hi)
// (x = hi + lo to 2^-22 relative; f16 subnormals are honoured by the MFMA, measured), and a product W x is three
// v_mfma_f32_32x32x16_f16 with fp32 accumulation: Wh xh + Wh xl + Wl xh (the dropped Wl xl term is 2^-22 relative).
// Measured (tools/debug/ubench/mfma_f16_probe.hip): K = 128 dot products come out as close to fp64 as a sequential fp32
// FMA chain does (2.7e-6 vs 4.1e-6 max abs error), at 3 x 32 cycles per 16-deep step against 8 x 64 cycles for
// v_mfma_f32_32x32x2_f32: 5.3x the FP32-matrix rate, with the same bytes per weight (2 x 2) and per activation.
// Range: |x| must stay below 65504 (LayerNorm-bounded activations and O(1) weights do by orders of magnitude).
//
// One tile (32 features of this lane's edge) as MFMA operands: k-step s in {0,1} carries accumulator registers
// 8s..8s+7 of the tile, i.e. features 32 t + 8 (2 s + (i >> 2)) + 4 h + (i & 3), i = 0..7, for lane half h.
struct HT {
    h8 hi[2], lo[2];
};
__device__ __forceinline__ void split_tile(const f32x16 &v, HT &o) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float x = __builtin_amdgcn_fmed3f(v[8 * s + i], -65504.f, 65504.f);    // saturate, never inf
            const _Float16 hh = (_Float16)x;
            o.hi[s][i] = hh;
            o.lo[s][i] = (_Float16)(x - (float)hh);
        }
}
__device__ __forceinline__ void join_tile(const HT &t, f32x16 &v) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 8; i++) v[8 * s + i] = (float)t.hi[s][i] + (float)t.lo[s][i];
}

#define ET 256
#define CH32 (128 * 32)                 // floats per packed weight chunk (16 KB); a wave's quarter is 1024 floats
#define XBUF_FLOATS (4 * 4 * 64 * 4)    // exchange buffer: [tile][quad][lane] float4
#define PARAM_FLOATS 1152               // edge kernel: small per-layer vectors staged once

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
// HAZARD (measured, tools/debug/edge_repro.py): an LDS-DMA instruction reads its address VGPRs LATE -- when the memory
// pipeline accepts it, which under load (several workgroups per CU) can be hundreds of cycles after issue -- and
// nothing interlocks a later VALU write to those registers.  hipcc, for which the asm's inputs are dead at its end,
// reuses them at once; the copy then fetches from a garbage address (sporadic wrong weight tiles, only with co-resident
// workgroups, gone with an s_waitcnt vmcnt(0) after every issue).  So the per-lane part of every DMA address lives in
// ONE register per wave for the whole kernel (`laneoff` = lane * 16, or the gather offset of dma_tile): each statement
// takes it read-write and the next wait takes it as input, so the compiler keeps it intact across the window; the
// wave-uniform part of the address goes in fixed SGPRs written only by these statements.  The same holds for M0 (LDS
// destination) and for EXEC: a lane masked off by a LATER divergent branch is dropped from a copy still in flight, so
// between an issue and its wait the kernels keep every lane active (clamped indices instead of predication).
// M0 (the LDS destination) is likewise left alone after the issue: it is written only by the next DMA statement
// (hipcc emits no M0 use of its own in these kernels; checked in the ISA).
__device__ __forceinline__ void dma_chunk(const float *&sbase, unsigned &laneoff, unsigned lds_dst) {
    // M0 (LDS destination) and s[98:99] (wave-uniform chunk address) are written ONLY here, after the previous copy has
    // completed (the caller's s_waitcnt vmcnt(0)), and then rest until the next issue; hipcc emits no M0 use of its own in
    // these kernels and allocates SGPRs from s0 up (~65 used).
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"          // this wave's reads of the slot being refilled have returned
                 "s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 s[98:99], %1\n\ts_nop 4\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99]\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99] offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99] offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, s[98:99] offset:3072"
                 : "+v"(laneoff) : "s"(sbase), "s"(lds_dst) : "memory", "s98", "s99");
    sbase += CH32;          // the running chunk pointer
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// A operands of one stage in registers: [s0 hi, s0 lo, s1 hi, s1 lo].  Three such sets rotate (stage k computes from
// set k % 3 while set (k+1) % 3 is being loaded): the set a load lands in was last read by the MFMAs of stage k - 2, which
// have retired by then, so a returning load can never overwrite operands of an MFMA that is still queued.
struct AOp {
    h8 r[4];
};
__device__ __forceinline__ void load_A(const float *wslot, int lane, AOp &a) {
    const h8 *w = reinterpret_cast<const h8 *>(wslot) + lane;
    a.r[0] = w[0]; a.r[1] = w[64]; a.r[2] = w[128]; a.r[3] = w[192];
}
template <bool SWAP>
__device__ __forceinline__ void mfma_chunk_r(const AOp &a, const HT &x, f32x16 &acc) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (SWAP) {
            acc = MFMA16(x.hi[s], a.r[2 * s], acc);
            acc = MFMA16(x.lo[s], a.r[2 * s], acc);
            acc = MFMA16(x.hi[s], a.r[2 * s + 1], acc);
        } else {
            acc = MFMA16(a.r[2 * s], x.hi[s], acc);
            acc = MFMA16(a.r[2 * s], x.lo[s], acc);
            acc = MFMA16(a.r[2 * s + 1], x.hi[s], acc);
        }
    }
}
// the 72 invariant-point features as operands: global k-step S = 0..4 carries features 16 S + 8 h + i (zero beyond 71);
// geometry chunk C holds k-steps 2C and 2C+1 (the last one is all padding and skipped)
struct HG {
    h8 hi[5], lo[5];
};
template <int C>
__device__ __forceinline__ void mfma_geo_r(const AOp &a, const HG &g, f32x16 &acc) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (2 * C + s < 5) {
            acc = MFMA16(a.r[2 * s], g.hi[2 * C + s], acc);
            acc = MFMA16(a.r[2 * s], g.lo[2 * C + s], acc);
            acc = MFMA16(a.r[2 * s + 1], g.hi[2 * C + s], acc);
        }
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

// split tiles through the same 4 KB per tile: [hi s0 | hi s1 | lo s0 | lo s1][lane] h8
__device__ __forceinline__ void xbuf_put_h(float *xbuf, int t, int lane, const HT &d) {
    h8 *xb = reinterpret_cast<h8 *>(xbuf) + (t * 4) * 64 + lane;
    xb[0] = d.hi[0]; xb[64] = d.hi[1]; xb[128] = d.lo[0]; xb[192] = d.lo[1];
}
__device__ __forceinline__ void xbuf_get_h(const float *xbuf, int t, int lane, HT &d) {
    const h8 *xb = reinterpret_cast<const h8 *>(xbuf) + (t * 4) * 64 + lane;
    d.hi[0] = xb[0]; d.hi[1] = xb[64]; d.lo[0] = xb[128]; d.lo[1] = xb[192];
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

// 72 invariant point features of edge (i, j), split and laid out as MFMA operands (HG)
__device__ __forceinline__ void edge_geometry(const float *__restrict__ pts_i, const float *__restrict__ fr,
                                              const float *__restrict__ pts_j, int h, HG &g) {
    float geom[80];
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
    for (int k = 72; k < 80; k++) geom[k] = 0.f;
#pragma unroll
    for (int S5 = 0; S5 < 5; S5++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float x = h ? geom[16 * S5 + 8 + i] : geom[16 * S5 + i];
            const _Float16 hh = (_Float16)x;
            g.hi[S5][i] = hh;
            g.lo[S5][i] = (_Float16)(x - (float)hh);
        }
}

// one workgroup per CU: nothing else hides a copy's latency, so the ring is S slots deep -- stage k starts chunk k+S-1
// into the slot stage k-1 read and waits, counted, until chunk k has landed (copies complete in issue order)
template <int N>
__device__ __forceinline__ void wait_vmN(unsigned laneoff) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N), "v"(laneoff) : "memory");
}
#define WSTAGE(k, NCH, BODY)                                                                                   \
    {                                                                                                          \
        if constexpr ((k) + S - 1 < (NCH)) dma_chunk(wsb, laneoff, slot0 + (((k) + S - 1) % S) * 4096u);       \
        if constexpr ((k) + 1 < (NCH)) {      /* chunk k+1 has landed -> its A operands go to the next register set */ \
            wait_vmN<4 * (((NCH) - 2 - (k)) < (S - 2) ? ((NCH) - 2 - (k)) : (S - 2))>(laneoff);                \
            load_A(wl + (((k) + 1) % S) * 1024, lane, AR[((k) + 1) % 3]);                                      \
        }                                                                                                      \
        const AOp &AK = AR[(k) % 3];                                                                           \
        HT &HK = HR[(k) % 3];                                                                                  \
        BODY;                                                                                                  \
    }

// shared first layer: acc (tile `wave`) = PA_i + PC_j + W_B h_E + W_G geom, ReLU.  Chunks W_B x4 (absent when ST0:
// layer 0's W_B h_E0 is timestep-invariant and arrives precomputed in acc), then W_G x3.  C0 = number of W_B chunks.
#define FIRST_LAYER(NCH)                                                          \
    if constexpr (!ST0) {                                                         \
        WSTAGE(0, NCH, mfma_chunk_r<false>(AK, x[0], acc))                 \
        WSTAGE(1, NCH, mfma_chunk_r<false>(AK, x[1], acc))                 \
        WSTAGE(2, NCH, mfma_chunk_r<false>(AK, x[2], acc))                 \
        WSTAGE(3, NCH, mfma_chunk_r<false>(AK, x[3], acc))                 \
    }                                                                             \
    WSTAGE(C0 + 0, NCH, mfma_geo_r<0>(AK, g, acc))                         \
    WSTAGE(C0 + 1, NCH, mfma_geo_r<1>(AK, g, acc))                         \
    WSTAGE(C0 + 2, NCH, mfma_geo_r<2>(AK, g, acc))                         \
    relu_tile(acc);                                                               \
    split_tile(acc, ht);                                                          \
    xbuf_put_h(xbuf, wave, lane, ht);                                             \
    __syncthreads();

#define PROLOGUE_PIPE()                                                                        \
    static_assert(S >= 3, "the operand prefetch reads slot k+1 while slot k+S-1 is refilled");  \
    const float *wsb = A.wstream + wave * 1024;          /* running chunk pointer, wave-uniform: SGPRs */ \
    unsigned laneoff = (unsigned)lane * 16u;              /* the one per-lane address register */ \
    const float *wl = smem + wave * (S * 1024);                                                \
    const unsigned slot0 = (unsigned)(size_t)wl;                                               \
    AOp AR[3];                                                                                 \
    HT HR[3];                                                                                  \
    _Pragma("unroll") for (int pk = 0; pk < S - 1; pk++) dma_chunk(wsb, laneoff, slot0 + pk * 4096u);
// after the prologue's own loads have been issued: chunk 0's operands into the first register set
#define PROLOGUE_OPERANDS()                                                                    \
    wait_vmN<4 * (S - 2)>(laneoff);                                                            \
    load_A(wl, lane, AR[0]);

// ---------------------------------------------------------------------------------------------
// node message: S[i] = (1/K) sum_j mask_ij relu(W_mid relu(W_in [..]) + b), msum[i] = (1/K) sum_j mask_ij
// ---------------------------------------------------------------------------------------------
// Shipped configuration: ONE workgroup per CU (one wave per SIMD), LDS-DMA ring of 5 / 4 slots per wave.  With several
// waves interleaving on a SIMD the split-f16 kernels produced rare wrong tiles (see HAZARD notes); with one wave per
// SIMD they have been bit-reproducible over millions of workgroup launches (tools/debug/soak.py).  The LDS requests
// (96 KB / 84.5 KB) are what enforces the exclusivity.
#ifndef PP_NM_SLOTS
#define PP_NM_SLOTS 5
#endif
#ifndef PP_EU_SLOTS
#define PP_EU_SLOTS 4
#endif
#ifndef PP_EU_WGS
#define PP_EU_WGS 1
#endif
#ifndef PP_NM_WGS
#define PP_NM_WGS 1
#endif

template <int S, bool ST0>
__global__ void __launch_bounds__(ET, PP_NM_WGS)
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

    HT x[4], ht;
    f32x16 acc;
    HG g;
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
            for (int t = 0; t < 4; t++) { load_tile(hrow + 32 * t, h, acc); split_tile(acc, x[t]); }
        }
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
        if constexpr (ST0) add_tile(A.Z + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);
    }
    PROLOGUE_OPERANDS()
    FIRST_LAYER(NCH)
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get_h(xbuf, t, lane, x[t]);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = bmid;
    }
    WSTAGE(C0 + 3, NCH, mfma_chunk_r<true>(AK, x[0], acc))
    WSTAGE(C0 + 4, NCH, mfma_chunk_r<true>(AK, x[1], acc))
    WSTAGE(C0 + 5, NCH, mfma_chunk_r<true>(AK, x[2], acc))
    WSTAGE(C0 + 6, NCH, mfma_chunk_r<true>(AK, x[3], acc))
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
    WSTAGE(C0 + 11 + 8 * (c) + 0, NCH, mfma_chunk_r<false>(AK, x[0], acc))                                \
    WSTAGE(C0 + 11 + 8 * (c) + 1, NCH, mfma_chunk_r<false>(AK, x[1], acc))                                \
    WSTAGE(C0 + 11 + 8 * (c) + 2, NCH, mfma_chunk_r<false>(AK, x[2], acc))                                \
    WSTAGE(C0 + 11 + 8 * (c) + 3, NCH, mfma_chunk_r<false>(AK, x[3], acc))                                \
    relu_tile(acc);                                                                                          \
    split_tile(acc, ht);                                                                                     \
    __syncthreads();          /* every wave is done reading the previous exchange */                        \
    xbuf_put_h(xbuf, wave, lane, ht);                                                                        \
    __syncthreads();                                                                                         \
    WSTAGE(C0 + 11 + 8 * (c) + 4, NCH, xbuf_get_h(xbuf, 0, lane, HK); mfma_chunk_r<false>(AK, HK, out))   \
    WSTAGE(C0 + 11 + 8 * (c) + 5, NCH, xbuf_get_h(xbuf, 1, lane, HK); mfma_chunk_r<false>(AK, HK, out))   \
    WSTAGE(C0 + 11 + 8 * (c) + 6, NCH, xbuf_get_h(xbuf, 2, lane, HK); mfma_chunk_r<false>(AK, HK, out))   \
    WSTAGE(C0 + 11 + 8 * (c) + 7, NCH, xbuf_get_h(xbuf, 3, lane, HK); mfma_chunk_r<false>(AK, HK, out))

// FUSE (-DPP_FUSE_NM, OFF by default): the workgroup goes straight on to the NEXT layer's node message of its residue
// (same 32 edges, whose new h_E it holds; the node-level inputs PA2 / PC2 / pts2 were written by the node update that
// ran before this kernel): one launch, one prologue and one read of h_E less per layer (~5 % of a step).  Disabled in
// the split-f16 build: with several workgroups per CU the fused tail produced rare wrong S rows (about one workgroup
// in a thousand, never with one workgroup per CU, h_E itself always bit-exact) that survived every fix of the LDS-DMA
// hazards below and also show without LDS-DMA; the stand-alone node-message kernel, same code, is bit-reproducible
// (tools/debug/edge_repro.py, score_check3.py).  Unresolved -> not shipped.
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

    HT x[4], ht;
    f32x16 acc, out;
    HG g;
    const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
    const int nbr = A.eidx[(size_t)n * K + jj];
    const float me = A.mask_att[(size_t)n * 32 + jj];          // (lanes j >= K mirror edge K - 1 throughout)
    // the small per-layer vectors go to LDS once (published by the first exchange barrier)
    // (no divergent control flow while a weight copy is in flight -- see HAZARD: the surplus threads of the second trip
    //  rewrite the last float4 with the same value instead of being masked off)
#pragma unroll
    for (int it = 0; it < (PARAM_FLOATS / 4 + ET - 1) / ET; it++) {
        const int i = min(tid + it * ET, PARAM_FLOATS / 4 - 1);
        *reinterpret_cast<f32x4v *>(prm + 4 * i) = *reinterpret_cast<const f32x4v *>(A.params + 4 * i);
    }
    edge_geometry(A.pts + (size_t)n * 48, A.frames + (size_t)n * 12, A.pts + (size_t)nbr * 48, h, g);
    __builtin_amdgcn_sched_barrier(0);        // geometry temporaries die before the activation tiles are loaded
    {
        if constexpr (!ST0) {
#pragma unroll
            for (int t = 0; t < 4; t++) { load_tile(hrow + 32 * t, h, acc); split_tile(acc, x[t]); }
        }
        load_tile(A.PA + (size_t)n * 128 + 32 * wave, h, acc);
        add_tile(A.PC + (size_t)nbr * 128 + 32 * wave, h, acc);
        if constexpr (ST0) add_tile(A.Z + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);
    }
    PROLOGUE_OPERANDS()
    FIRST_LAYER(NCH)
    // ---- second layer (chunks 7..10) -------------------------------------------------------------
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get_h(xbuf, t, lane, x[t]);
        load_tile(prm + P_BMID + 32 * wave, h, acc);
    }
    WSTAGE(C0 + 3, NCH, mfma_chunk_r<false>(AK, x[0], acc))
    WSTAGE(C0 + 4, NCH, mfma_chunk_r<false>(AK, x[1], acc))
    WSTAGE(C0 + 5, NCH, mfma_chunk_r<false>(AK, x[2], acc))
    WSTAGE(C0 + 6, NCH, mfma_chunk_r<false>(AK, x[3], acc))
    relu_tile(acc);
    split_tile(acc, ht);
    __syncthreads();
    xbuf_put_h(xbuf, wave, lane, ht);
    __syncthreads();
    // ---- third layer (chunks 11..14) --------------------------------------------------------------
    {
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get_h(xbuf, t, lane, x[t]);
        load_tile(prm + P_BOUT + 32 * wave, h, acc);
    }
    WSTAGE(C0 + 7, NCH, mfma_chunk_r<false>(AK, x[0], acc))
    WSTAGE(C0 + 8, NCH, mfma_chunk_r<false>(AK, x[1], acc))
    WSTAGE(C0 + 9, NCH, mfma_chunk_r<false>(AK, x[2], acc))
    WSTAGE(C0 + 10, NCH, mfma_chunk_r<false>(AK, x[3], acc))
    // publish v = h_E + mask * m for the first LayerNorm (own tile: read, then overwritten in place)
    load_tile(hrow + 32 * wave, h, out);        // residual input: this wave's tile of h_E (L2-resident re-read)
#pragma unroll
    for (int r = 0; r < 16; r++) out[r] = fmaf(acc[r], me, out[r]);
    __syncthreads();                            // every wave has its B operands of this layer
    xbuf_put(xbuf, wave, lane, out);
    __syncthreads();
    {
        // x1 = LN2(v): every wave normalises the full vector in fp32 (it needs all of x1 as B operands), then splits it
        f32x16 v4[4];
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, v4[t]);
        float mean;
        float rstd = ln_center(v4, mean);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            // compiler fence tied to the data flow: the gamma / beta reads of tile t are issued only once rstd (and the
            // previous tile) exist, so one tile's worth of them is live at a time (168-VGPR budget)
            if (t == 0) asm volatile("" : "+v"(rstd) : : "memory");
            else asm volatile("" : "+v"(v4[t > 0 ? t - 1 : 0][15]) : : "memory");
            ln_affine_tile(v4[t], rstd, prm + P_G2 + 32 * t, prm + P_BE2 + 32 * t, h);
            split_tile(v4[t], x[t]);
        }
        load_tile(prm + P_FOB + 32 * wave, h, out);
    }
    // ---- FFN 128 -> 512 -> 128 in four hidden blocks of 128 ------------------------------------------
    FFN_BLOCK(0)
    FFN_BLOCK(1)
    FFN_BLOCK(2)
    FFN_BLOCK(3)
    // ---- h_E = mask * LN3(x1 + ffn) ---------------------------------------------------------------------
    // residual: this wave's tile of x1, rebuilt from its split form (wave is scalar: uniform branches, static indices)
    if (wave == 0) join_tile(x[0], acc);
    else if (wave == 1) join_tile(x[1], acc);
    else if (wave == 2) join_tile(x[2], acc);
    else join_tile(x[3], acc);
#pragma unroll
    for (int r = 0; r < 16; r++) out[r] += acc[r];
    __syncthreads();
    xbuf_put(xbuf, wave, lane, out);
    __syncthreads();
    float mean3, rstd;
    {
        f32x16 v4[4];
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get(xbuf, t, lane, v4[t]);
        rstd = ln_center(v4, mean3);
    }
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
        split_tile(out, ht);
        __syncthreads();                                      // every wave has read the LayerNorm exchange
        xbuf_put_h(xbuf, wave, lane, ht);
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
        for (int t = 0; t < 4; t++) xbuf_get_h(xbuf, t, lane, x[t]);
        WSTAGE(NEU + 0, NCH, mfma_chunk_r<false>(AK, x[0], acc))
        WSTAGE(NEU + 1, NCH, mfma_chunk_r<false>(AK, x[1], acc))
        WSTAGE(NEU + 2, NCH, mfma_chunk_r<false>(AK, x[2], acc))
        WSTAGE(NEU + 3, NCH, mfma_chunk_r<false>(AK, x[3], acc))
        WSTAGE(NEU + 4, NCH, mfma_geo_r<0>(AK, g, acc))
        WSTAGE(NEU + 5, NCH, mfma_geo_r<1>(AK, g, acc))
        WSTAGE(NEU + 6, NCH, mfma_geo_r<2>(AK, g, acc))
        relu_tile(acc);
        split_tile(acc, ht);
        __syncthreads();
        xbuf_put_h(xbuf, wave, lane, ht);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) xbuf_get_h(xbuf, t, lane, x[t]);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = bmid;
        WSTAGE(NEU + 7, NCH, mfma_chunk_r<true>(AK, x[0], acc))
        WSTAGE(NEU + 8, NCH, mfma_chunk_r<true>(AK, x[1], acc))
        WSTAGE(NEU + 9, NCH, mfma_chunk_r<true>(AK, x[2], acc))
        WSTAGE(NEU + 10, NCH, mfma_chunk_r<true>(AK, x[3], acc))
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
__global__ void __launch_bounds__(ET, PP_NM_WGS)
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
    HT x[4];
    f32x16 acc;
    const int jj = j < K ? j : K - 1;
    const float *hrow = A.hE_in + ((size_t)n * K + jj) * 128;
#pragma unroll
    for (int t = 0; t < 4; t++) { load_tile(hrow + 32 * t, h, acc); split_tile(acc, x[t]); }
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    PROLOGUE_OPERANDS()
    WSTAGE(0, NCH, mfma_chunk_r<false>(AK, x[0], acc))
    WSTAGE(1, NCH, mfma_chunk_r<false>(AK, x[1], acc))
    WSTAGE(2, NCH, mfma_chunk_r<false>(AK, x[2], acc))
    WSTAGE(3, NCH, mfma_chunk_r<false>(AK, x[3], acc))
    store_tile(A.Znm + ((size_t)n * K + jj) * 128 + 32 * wave, h, acc);      // lanes j >= K mirror edge K - 1
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    WSTAGE(4, NCH, mfma_chunk_r<false>(AK, x[0], acc))
    WSTAGE(5, NCH, mfma_chunk_r<false>(AK, x[1], acc))
    WSTAGE(6, NCH, mfma_chunk_r<false>(AK, x[2], acc))
    WSTAGE(7, NCH, mfma_chunk_r<false>(AK, x[3], acc))
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
static const size_t ST_SMEM = (4 * PP_NM_SLOTS * 1024) * sizeof(float);

static bool edge_attrs() {
    static bool done = false, ok = false;
    if (!done) {
        done = true;
        auto set = [](const void *f, size_t bytes) {
            return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
        };
        ok = set(reinterpret_cast<const void *>(k_edge_static<PP_NM_SLOTS>), ST_SMEM) &&
             set(reinterpret_cast<const void *>(k_node_message<PP_NM_SLOTS, false>), NM_SMEM) &&
             set(reinterpret_cast<const void *>(k_node_message<PP_NM_SLOTS, true>), NM_SMEM) &&
#ifndef PP_FUSE_NM
             set(reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, false, false>), EU_SMEM) &&
             set(reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, true, false>), EU_SMEM);
#else
             set(reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, false, true>), EU_SMEM) &&
             set(reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, true, true>), EU_SMEM);
#endif
    }
    return ok;
}

// resident workgroups per CU the runtime predicts for the two kernels (measurement aid)
void pp_edge_occupancy(int *node_msg, int *edge_upd) {
    edge_attrs();
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(node_msg, reinterpret_cast<const void *>(k_node_message<PP_NM_SLOTS, false>), ET, NM_SMEM);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(edge_upd, reinterpret_cast<const void *>(k_edge_update<PP_EU_SLOTS, false, false>), ET, EU_SMEM);
}

#define EDGE_ATTR_CHECK()                                                                                   \
    if (!edge_attrs()) {                                                                                    \
        pp_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the edge kernels");        \
        return PP_ERR_HIP;                                                                                  \
    }

pp_status pp_launch_edge_static(pp_ctx *c, hipStream_t s) {
    EDGE_ATTR_CHECK()
    EdgeArgs A = edge_args(c, 0, false);
    A.wstream = c->plan->static_stream;
    hipLaunchKernelGGL(k_edge_static<PP_NM_SLOTS>, dim3(c->N), dim3(ET), ST_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_node_message(pp_ctx *c, int layer, hipStream_t s) {
    EDGE_ATTR_CHECK()
    EdgeArgs A = edge_args(c, layer, false);
    if (layer == 0) hipLaunchKernelGGL((k_node_message<PP_NM_SLOTS, true>), dim3(c->N), dim3(ET), NM_SMEM, s, A);
    else hipLaunchKernelGGL((k_node_message<PP_NM_SLOTS, false>), dim3(c->N), dim3(ET), NM_SMEM, s, A);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// true when k_edge_update also computes the next layer's node message (build with -DPP_FUSE_NM)
bool pp_edge_fused() {
#ifdef PP_FUSE_NM
    return true;
#else
    return false;
#endif
}

// layers 0 and 1 only (the reference's layer-2 edge update is dead code)
pp_status pp_launch_edge_update(pp_ctx *c, int layer, hipStream_t s) {
    EDGE_ATTR_CHECK()
    if (layer < 0 || layer > 1) { pp_set_error("pp_launch_edge_update: layer must be 0 or 1"); return PP_ERR_INVALID; }
    EdgeArgs A = edge_args(c, layer, true);
#ifdef PP_FUSE_NM
    if (layer == 0) hipLaunchKernelGGL((k_edge_update<PP_EU_SLOTS, true, true>), dim3(c->N), dim3(ET), EU_SMEM, s, A);
    else hipLaunchKernelGGL((k_edge_update<PP_EU_SLOTS, false, true>), dim3(c->N), dim3(ET), EU_SMEM, s, A);
#else
    if (layer == 0) hipLaunchKernelGGL((k_edge_update<PP_EU_SLOTS, true, false>), dim3(c->N), dim3(ET), EU_SMEM, s, A);
    else hipLaunchKernelGGL((k_edge_update<PP_EU_SLOTS, false, false>), dim3(c->N), dim3(ET), EU_SMEM, s, A);
#endif
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

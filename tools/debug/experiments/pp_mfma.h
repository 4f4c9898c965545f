// Device helpers shared by the MFMA edge kernels (pp_edge.hip, pp_edge2.hip): accumulator-layout tiles,
// weight-chunk staging, the LDS exchange buffer, LayerNorm over an edge's 128 features, invariant-point geometry.
// See the header comment of pp_edge.hip for the register / lane conventions.
#pragma once
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
template <int NC, int NT = ET>
__device__ __forceinline__ void chunk_load(const float *__restrict__ g, WRegs &r, int tid) {
    constexpr int TOTAL = 128 * NC / 4, PER_T = (TOTAL + NT - 1) / NT;
    static_assert(PER_T <= 4, "chunk shape");
#ifdef PP_X_NOLOAD
    for (int m = 0; m < PER_T; m++) r.v[m] = f32x4v{0.001f * tid, 0.f, 0.f, 0.f};
    return;
#endif
#pragma unroll
    for (int m = 0; m < PER_T; m++)
        if (TOTAL % NT == 0 || tid + NT * m < TOTAL) r.v[m] = *reinterpret_cast<const f32x4v *>(g + 4 * (tid + NT * m));
}
template <int NC, int NT = ET>
__device__ __forceinline__ void chunk_store(float *lds, const WRegs &r, int tid) {
    constexpr int PER_ROW = NC / 4, TOTAL = 128 * PER_ROW, PER_T = (TOTAL + NT - 1) / NT;
#ifdef PP_X_NOSTORE
    return;
#endif
#pragma unroll
    for (int m = 0; m < PER_T; m++) {
        int idx = tid + NT * m;
        if (TOTAL % NT == 0 || idx < TOTAL) {
            int row = idx / PER_ROW, c4 = idx - row * PER_ROW;
            *reinterpret_cast<f32x4v *>(lds + row * (NC + 4) + 4 * c4) = r.v[m];
        }
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


// Timestep-invariant graph construction (once per complex): kNN over CA atoms, residue frames,
// 468-d edge features -> Linear(468,128) -> LayerNorm.   Reference: encoder.py:105-118 (kNN),
// :34-47 (relpos), :120-153 (RBF), :164-196 (inter-residue dihedrals), :243-244 (embedding),
// rigid_utils.py:1127-1179 (frames).
#include "pp_internal.h"
#include "pp_topk_aten.h"

#define KNN_THREADS 256
#define KNN_PAR_MAX 2047      // rows up to this length take ATen's nth_element branch (K * 64 > L for K = 32): done by the whole block

// The distance arithmetic that decides neighbour membership must round like the reference's
// separate mul/add/sqrt ops, so no FMA contraction in this file's geometric helpers.
#pragma clang fp contract(off)

__device__ __forceinline__ float dist_eps(const float *a, const float *b, float eps) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return sqrtf(s + eps);
}

// ---------------------------------------------------------------------------------------------
// frames, backbone atom table, virtual CB
// ---------------------------------------------------------------------------------------------
__global__ void k_frames(const float *__restrict__ X, int N, float *__restrict__ frames,
                         float *__restrict__ bbpos) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *x = X + (size_t)n * 42;
    float Nn[3], CA[3], C[3], O[3];
    for (int k = 0; k < 3; k++) { Nn[k] = x[k]; CA[k] = x[3 + k]; C[k] = x[6 + k]; O[k] = x[9 + k]; }
    // Gram-Schmidt: e0 along C-CA, e1 from N-CA (from_3_points(N, CA, C, fixed=True))
    float a[3], b[3];
    for (int k = 0; k < 3; k++) { a[k] = C[k] - CA[k]; b[k] = Nn[k] - CA[k]; }
    float na = sqrtf(((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) + 1e-8f);
    for (int k = 0; k < 3; k++) a[k] = a[k] / na;
    float dot = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    for (int k = 0; k < 3; k++) b[k] = b[k] - a[k] * dot;
    float nb = sqrtf(((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]) + 1e-8f);
    for (int k = 0; k < 3; k++) b[k] = b[k] / nb;
    float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    float *f = frames + (size_t)n * 12;
    for (int r = 0; r < 3; r++) { f[3 * r] = a[r]; f[3 * r + 1] = b[r]; f[3 * r + 2] = c[r]; }
    for (int k = 0; k < 3; k++) f[9 + k] = CA[k];
    // virtual CB (encoder.py:137-142)
    float bb[3], cc[3];
    for (int k = 0; k < 3; k++) { bb[k] = CA[k] - Nn[k]; cc[k] = C[k] - CA[k]; }
    float aa[3] = {bb[1] * cc[2] - bb[2] * cc[1], bb[2] * cc[0] - bb[0] * cc[2], bb[0] * cc[1] - bb[1] * cc[0]};
    float *p = bbpos + (size_t)n * 15;
    for (int k = 0; k < 3; k++) {
        p[k] = Nn[k]; p[3 + k] = CA[k]; p[6 + k] = C[k]; p[9 + k] = O[k];
        p[12 + k] = ((-0.58273431f * aa[k] + 0.56802827f * bb[k]) - 0.54067466f * cc[k]) + CA[k];
    }
}

// ---------------------------------------------------------------------------------------------
// kNN: one block per residue row; K rounds of (value, index)-lexicographic arg-min over the row's adjusted distances
// (encoder.py:105-118), then -- `ties` != PP_KNN_TIES_LOWER_INDEX -- the reference CPU path's own choice where values are
// exactly equal: torch.topk's selection depends on libstdc++'s nth_element / sort running over the whole row
// (pp_topk_aten.h), so a row that has two equal values among its K + 1 smallest (PP_KNN_TIES_ATEN_CPU; only rank K = rank
// K + 1, a MEMBERSHIP tie, in PP_KNN_TIES_ATEN_MEMBER) is redone by one lane running exactly that code on the row in LDS.
// Rows without such a tie have one possible answer and never pay for it.
// ---------------------------------------------------------------------------------------------
// minimum of a 64-bit key over the wave, in every lane: four DPP butterfly steps inside each row of 16 lanes (VALU speed; the
// ds_bpermute form of __shfl_xor costs an LDS round trip per step, twelve dependent ones per arg-min), then the four row results
// through SGPRs.  Keys are (float bits of a non-negative distance) << 32 | index: unsigned order = (value, index) order, +inf and
// NaN patterns last.
__device__ __forceinline__ unsigned long long wave_min_key(unsigned hi, unsigned lo) {
#define KNN_DPP_STEP(ctrl)                                                                                              \
    {                                                                                                                   \
        const unsigned oh = (unsigned)__builtin_amdgcn_mov_dpp((int)hi, ctrl, 0xF, 0xF, true);                          \
        const unsigned ol = (unsigned)__builtin_amdgcn_mov_dpp((int)lo, ctrl, 0xF, 0xF, true);                          \
        const bool less = oh < hi || (oh == hi && ol < lo);                                                             \
        hi = less ? oh : hi;                                                                                            \
        lo = less ? ol : lo;                                                                                            \
    }
    KNN_DPP_STEP(0xB1)      // quad_perm [1,0,3,2]
    KNN_DPP_STEP(0x4E)      // quad_perm [2,3,0,1]
    KNN_DPP_STEP(0x141)     // row_half_mirror
    KNN_DPP_STEP(0x140)     // row_mirror
#undef KNN_DPP_STEP
    unsigned long long best = ~0ull;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        const unsigned long long k = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 16 * row) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)lo, 16 * row);
        best = k < best ? k : best;
    }
    return best;
}

__global__ void __launch_bounds__(KNN_THREADS)
k_knn(const float *__restrict__ X, const float *__restrict__ rmask, const int2 *__restrict__ seg, int K, int ties, int Lmax,
      int32_t *__restrict__ eidx, float *__restrict__ mask_att) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    pp_tk_pair *q = reinterpret_cast<pp_tk_pair *>(smem_raw);   // [L] (adjusted distance, index in the complex)
    __shared__ float red_v[KNN_THREADS / 64];
    __shared__ float s_max;
    __shared__ float pick_v[33];
    __shared__ int pick_i[33];
    const int n = blockIdx.x;
    const int row0 = seg[n].x, L = seg[n].y;          // this row's complex: rows row0 .. row0 + L - 1
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float mi = rmask[n];
    const float *ca_i = X + (size_t)n * 42 + 3;
    float ci[3] = {ca_i[0], ca_i[1], ca_i[2]};
    float lmax = 0.f;
    for (int j = tid; j < L; j += KNN_THREADS) {
        const float *ca_j = X + (size_t)(row0 + j) * 42 + 3;
        float m2 = mi * rmask[row0 + j];
        float v = m2 * dist_eps(ca_j, ci, 1e-6f);
        q[j].v = v;
        q[j].i = j;
        lmax = fmaxf(lmax, v);
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0) red_v[wid] = lmax;
    __syncthreads();
    if (tid == 0) {
        float m = red_v[0];
        for (int w = 1; w < KNN_THREADS / 64; w++) m = fmaxf(m, red_v[w]);
        s_max = m;
    }
    __syncthreads();
    const float dmax = s_max;
    for (int j = tid; j < L; j += KNN_THREADS) {
        float m2 = mi * rmask[row0 + j];
        q[j].v = q[j].v + (2.f * (1.f - m2)) * dmax;
    }
    __syncthreads();
    // round K (when the complex has more than K residues) only looks at the next value: is rank K + 1 equal to rank K?
    // One barrier per round: the four wave candidates go through one of two alternating LDS slots, every thread merges them
    // (same answer everywhere), the thread that owns the winning element in the strided scan retires it, and the global
    // stores wait until the rounds are over.
    __shared__ unsigned long long red_k[2][KNN_THREADS / 64];
    const int rounds = K + ((ties != PP_KNN_TIES_LOWER_INDEX && L > K) ? 1 : 0);
    int tie = 0;
    unsigned prev = 0;
    for (int k = 0; k < rounds; k++) {
        // the scan starts at +inf: an element retired to +inf in an earlier round (or a non-finite distance) is never
        // picked, so a segment with fewer than K finite entries falls through to the masked-out fallback below
        unsigned bv = 0x7f800000u, bi = 0x7fffffffu;
        for (int j = tid; j < L; j += KNN_THREADS) {
            const unsigned v = __float_as_uint(q[j].v);
            if (v < bv) { bv = v; bi = (unsigned)j; }      // strided scan keeps the lowest j among equal values
        }
        const unsigned long long wk = wave_min_key(bv, bi);
        if (lane == 0) red_k[k & 1][wid] = wk;
        __syncthreads();
        unsigned long long best = red_k[k & 1][0];
#pragma unroll
        for (int w = 1; w < KNN_THREADS / 64; w++) {
            const unsigned long long o = red_k[k & 1][w];
            best = o < best ? o : best;
        }
        const unsigned vb = (unsigned)(best >> 32);
        const int ix = (int)(unsigned)best;
        if (k > 0 && vb == prev && vb < 0x7f800000u) tie |= (k == K) ? 2 : 1;      // (equal finite values; NaN never equals)
        prev = vb;
        if (tid == 0) { pick_v[k] = __uint_as_float(vb); pick_i[k] = ix; }
        if (k < K && ix < L && (ix % KNN_THREADS) == tid) q[ix].v = INFINITY;      // only this thread ever scans element ix
    }
    __syncthreads();
    if (tid < K) {
        const int ix = pick_i[tid];
        // nothing finite left (a segment shorter than K, or NaN coordinates): the row itself, masked out
        eidx[(size_t)n * K + tid] = ix < L ? row0 + ix : n;
        mask_att[(size_t)n * 32 + tid] = ix < L ? mi * rmask[row0 + ix] : 0.f;
    }
    if (tid < 32 && tid >= K) mask_att[(size_t)n * 32 + tid] = 0.f;
    const int want = ties == PP_KNN_TIES_ATEN_CPU ? 3 : (ties == PP_KNN_TIES_ATEN_MEMBER ? 2 : 0);
    if (!(tie & want) || mi == 0.f) return;         // (a masked row's list is never used: mask_att is 0 throughout)
    if (tid < K && pick_i[tid] < L) q[pick_i[tid]].v = pick_v[tid];      // the row as it was before the rounds
    __syncthreads();
    if ((long)K * 64 <= (long)L || L > KNN_PAR_MAX) {
        // ATen's partial_sort branch (rows of 2048 residues and more): one lane, heap code as it stands
        if (tid == 0) pp_tk_topk_smallest(q, L, K);
    } else {
        // std::nth_element(q, q + K - 1, q + L) with the whole block: every partition pass is the closed form of
        // pp_topk_aten.h (pp_tk_partition_pivot_lists) -- stop positions by prefix sums, the number of exchanges by a
        // reduction, the exchanges themselves in parallel; then std::sort of the first K - 1 by one lane (31 elements).
        // the two stop-position lists sit behind the LONGEST row of the context (Lmax, what the launch sized the allocation
        // for), not behind this row: a packed context mixes rows of different lengths
        short *Apos = reinterpret_cast<short *>(q + Lmax), *Bpos = Apos + (KNN_PAR_MAX + 1);
        __shared__ int wtL[2][KNN_THREADS / 64], wtR[2][KNN_THREADS / 64], wtT[KNN_THREADS / 64];
        const unsigned long long lt = (1ull << lane) - 1ull;
        int first = 0, last = L, depth = 2 * pp_tk_lg(L);
        const int nth = K - 1;
        bool heap_done = false;
        while (last - first > 3) {
            if (depth == 0) {                      // introselect's fallback (never seen outside adversarial inputs)
                if (tid == 0) {
                    pp_tk_heap_select(q + first, q + nth + 1, q + last);
                    pp_tk_swap(q + first, q + nth);
                }
                heap_done = true;
                break;
            }
            --depth;
            if (tid == 0) pp_tk_median_to_first(q + first, q + first + 1, q + first + (last - first) / 2, q + last - 1);
            __syncthreads();
            const pp_tk_pair pivot = q[first];
            const int m = last - first;
            int baseL = 0, baseR = 0;
            for (int c0 = 0, it = 0; c0 < m - 1; c0 += KNN_THREADS, it++) {
                const int o = c0 + tid;
                const bool in = o < m - 1;
                const int iA = first + 1 + o, iB = last - 1 - o;           // ascending / descending position of this thread
                const bool fl = in && !pp_tk_less(q[in ? iA : first], pivot);
                const bool fr = in && !pp_tk_less(pivot, q[in ? iB : first]);
                const unsigned long long bl = __ballot(fl), br = __ballot(fr);
                if (lane == 0) { wtL[it & 1][wid] = __popcll(bl); wtR[it & 1][wid] = __popcll(br); }
                __syncthreads();
                int offL = baseL, offR = baseR;
#pragma unroll
                for (int w = 0; w < KNN_THREADS / 64; w++) {
                    const int a = wtL[it & 1][w], b = wtR[it & 1][w];
                    if (w < wid) { offL += a; offR += b; }
                    baseL += a; baseR += b;
                }
                if (fl) Apos[offL + __popcll(bl & lt)] = (short)iA;
                if (fr) Bpos[offR + __popcll(br & lt)] = (short)iB;
            }
            if (tid == 0) Bpos[baseR] = (short)first;          // the pivot: where the downward scan stops at the latest
            const int nA = baseL, nB = baseR + 1;
            __syncthreads();
            int cnt = 0;
            for (int t = tid; t < nA && t < nB; t += KNN_THREADS) cnt += Apos[t] < Bpos[t] ? 1 : 0;
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0) wtT[wid] = cnt;
            __syncthreads();
            int T = 0;
#pragma unroll
            for (int w = 0; w < KNN_THREADS / 64; w++) T += wtT[w];
            const int a_next = T < nA ? (int)Apos[T] : last, b_prev = T > 0 ? (int)Bpos[T - 1] : last;
            for (int t = tid; t < T; t += KNN_THREADS) pp_tk_swap(q + Apos[t], q + Bpos[t]);
            const int cut = a_next < b_prev ? a_next : b_prev;
            __syncthreads();
            if (cut <= nth) first = cut; else last = cut;
        }
        if (tid == 0) {
            if (!heap_done) pp_tk_insertion_sort(q + first, q + last);
            pp_tk_sort(q, q + K - 1);
        }
    }
    if (tid == 0) {
        for (int k = 0; k < K; k++) {
            const int ix = q[k].i;
            eidx[(size_t)n * K + k] = row0 + ix;
            mask_att[(size_t)n * 32 + k] = mi * rmask[row0 + ix];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// edge features + embedding: one block (128 threads) per residue, its K edges.
// ---------------------------------------------------------------------------------------------
#define EF_THREADS 128
#define EF_RBF 400
#define EF_STRIDE 404      // padded row of the per-edge RBF vector in LDS

__global__ void __launch_bounds__(EF_THREADS)
k_edge_embed(const float *__restrict__ bbpos, const int32_t *__restrict__ eidx,
             const int64_t *__restrict__ res_index, const int64_t *__restrict__ chain, int K,
             const float *__restrict__ WT /*[468][128]*/, const float *__restrict__ bias,
             const float *__restrict__ ln_g, const float *__restrict__ ln_b, float *__restrict__ hE0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *rbf = reinterpret_cast<float *>(smem_raw);           // [32][EF_STRIDE]
    __shared__ float s_extra[32][4];                            // relpos idx, etype, phi, psi
    __shared__ float s_pos[33][15];                             // slot 32 = centre residue
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < 15; t += EF_THREADS) s_pos[32][t] = bbpos[(size_t)n * 15 + t];
    for (int t = tid; t < K * 15; t += EF_THREADS) {
        int e = t / 15, c = t - e * 15;
        s_pos[e][c] = bbpos[(size_t)eidx[(size_t)n * K + e] * 15 + c];
    }
    __syncthreads();
    // 25 atom-pair distances -> 16 RBFs each (centre atom a major, neighbour atom b minor)
    for (int item = tid; item < K * 25; item += EF_THREADS) {
        int e = item / 25, pr = item - e * 25;
        int a = pr / 5, bq = pr - a * 5;
        float dd = dist_eps(&s_pos[32][3 * a], &s_pos[e][3 * bq], 1e-6f);
        float *o = rbf + e * EF_STRIDE + pr * 16;
        for (int r = 0; r < 16; r++) {
            // torch.linspace(0, 20, 16): ascending from the start below the midpoint, descending from
            // the end above it, fp32 step
            const float step = 20.0f / 15.0f;
            float mu = (r < 8) ? step * (float)r : 20.0f - step * (float)(15 - r);
            float z = (dd - mu) / 1.25f;
            o[r] = expf(-(z * z));
        }
    }
    if (tid < K) {
        int e = tid;
        int j = eidx[(size_t)n * K + e];
        long off = (long)(res_index[n] - res_index[j]) + 32;
        off = off < 0 ? 0 : (off > 64 ? 64 : off);
        s_extra[e][0] = (float)off;
        s_extra[e][1] = (chain[n] == chain[j]) ? 2.f : 1.f;
        // phi_ij = dih(C_i, N_j, CA_j, C_j), psi_ij = dih(N_i, CA_i, C_i, N_j), rounded as the reference rounds them
        // (pp_internal.h); the j == i edge included
        const float phi = pp_pair_dihedral_t(&s_pos[32][6], &s_pos[e][0], &s_pos[e][3], &s_pos[e][6]);
        const float psi = pp_pair_dihedral_t(&s_pos[32][0], &s_pos[32][3], &s_pos[32][6], &s_pos[e][0]);
        s_extra[e][2] = phi;
        s_extra[e][3] = psi;
    }
    __syncthreads();
    const int f = tid;
    float acc[32];
#pragma unroll
    for (int e = 0; e < 32; e++) acc[e] = 0.f;
    for (int m = 0; m < EF_RBF; m += 4) {
        float w0 = WT[(size_t)(65 + m) * 128 + f], w1 = WT[(size_t)(66 + m) * 128 + f];
        float w2 = WT[(size_t)(67 + m) * 128 + f], w3 = WT[(size_t)(68 + m) * 128 + f];
#pragma unroll
        for (int e = 0; e < 32; e++) {
            float4 r = *reinterpret_cast<const float4 *>(rbf + e * EF_STRIDE + m);
            acc[e] = fmaf(w0, r.x, acc[e]);
            acc[e] = fmaf(w1, r.y, acc[e]);
            acc[e] = fmaf(w2, r.z, acc[e]);
            acc[e] = fmaf(w3, r.w, acc[e]);
        }
    }
    __syncthreads();
    float *pre = rbf;                                            // reuse as [32][128]
    const float bf = bias[f], w465 = WT[(size_t)465 * 128 + f], w466 = WT[(size_t)466 * 128 + f],
                w467 = WT[(size_t)467 * 128 + f];
#pragma unroll
    for (int e = 0; e < 32; e++) {
        if (e < K) {
            int rp = (int)s_extra[e][0];
            float v = acc[e] + bf + WT[(size_t)rp * 128 + f];
            v = fmaf(w465, s_extra[e][1], v);
            v = fmaf(w466, s_extra[e][2], v);
            v = fmaf(w467, s_extra[e][3], v);
            pre[e * 128 + f] = v;
        }
    }
    __syncthreads();
    // LayerNorm per edge: each 64-lane wave takes edges wave, wave+2, ...; 2 features per lane
    const int lane = tid & 63, wid = tid >> 6;
    for (int e = wid; e < K; e += EF_THREADS / 64) {
        float v0 = pre[e * 128 + lane], v1 = pre[e * 128 + 64 + lane];
        float s = v0 + v1;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        float mean = s * (1.f / 128.f);
        float d0 = v0 - mean, d1 = v1 - mean;
        float q = d0 * d0 + d1 * d1;
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        float rstd = 1.f / sqrtf(q * (1.f / 128.f) + 1e-5f);
        float *o = hE0 + ((size_t)n * K + e) * 128;
        o[lane] = d0 * rstd * ln_g[lane] + ln_b[lane];
        o[64 + lane] = d1 * rstd * ln_g[64 + lane] + ln_b[64 + lane];
    }
}

// caller-provided neighbour lists (pp_ctx_set_graph): per-complex numbering [N][K] int64 -> global rows + pair mask
__global__ void k_import_graph(const int64_t *__restrict__ E_idx, const int2 *__restrict__ seg,
                               const float *__restrict__ rmask, int N, int K, int32_t *__restrict__ eidx,
                               float *__restrict__ mask_att, int *__restrict__ bad) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * 32) return;
    const int n = t >> 5, k = t & 31;
    if (k >= K) { mask_att[(size_t)n * 32 + k] = 0.f; return; }
    const int64_t j = E_idx[(size_t)n * K + k];
    const int2 sg = seg[n];
    if (j < 0 || j >= sg.y) { atomicAdd(bad, 1); eidx[(size_t)n * K + k] = n; mask_att[(size_t)n * 32 + k] = 0.f; return; }
    eidx[(size_t)n * K + k] = sg.x + (int)j;
    mask_att[(size_t)n * 32 + k] = rmask[n] * rmask[sg.x + (int)j];
}

pp_status pp_launch_prepare(pp_ctx *c, hipStream_t s, const int64_t *E_idx) {
    const int N = c->N;
    if (E_idx) {
        // neighbour lists given: frames and backbone atoms are in place already, only the edge embedding follows
        PP_HIP_CHECK(hipMemsetAsync(c->scal, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_import_graph, dim3((N * 32 + 255) / 256), dim3(256), 0, s, E_idx, c->seg, c->b.residue_mask, N, c->K,
                           c->eidx, c->mask_att, reinterpret_cast<int *>(c->scal));
    } else {
    hipLaunchKernelGGL(k_frames, dim3((N + 127) / 128), dim3(128), 0, s, c->b.X, N, c->frames, c->bbpos);
    // L = the longest complex of the context; the stop-position lists of the parallel partition (8 KB) always: in a packed
    // context a shorter complex can take that branch next to one that is too long for it
    const size_t smem = (size_t)c->L * sizeof(pp_tk_pair) + 2 * (KNN_PAR_MAX + 1) * sizeof(short);
    if (smem > 64 * 1024) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_knn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    hipLaunchKernelGGL(k_knn, dim3(N), dim3(KNN_THREADS), smem, s, c->b.X, c->b.residue_mask, c->seg, c->K,
                       c->plan->knn_ties, c->L, c->eidx, c->mask_att);
    }
#ifdef PP_EDGE_F16
    return pp_launch_edge_embed_f16(c, s);      // MFMA form (pp_edge_f16.hip); k_edge_embed below is the fp32 build's
#else
    const pp_plan *p = c->plan;
    size_t smem2 = (size_t)32 * EF_STRIDE * sizeof(float);
    hipLaunchKernelGGL(k_edge_embed, dim3(N), dim3(EF_THREADS), smem2, s, c->bbpos, c->eidx, c->b.residue_index,
                       c->b.chain_indices, c->K, p->edge_emb_T, p->w + p->off.edge_emb_b,
                       p->w + p->off.norm_edges_g, p->w + p->off.norm_edges_b, c->hE0);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
#endif
}

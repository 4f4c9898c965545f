// C ABI of libpackppi_hip.so: plan / ctx lifetime and the per-call kernel schedules.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "pp_internal.h"
#include "pp_topk_aten.h"
#include <algorithm>
#include <cstring>
#include <cmath>
#include <chrono>

static thread_local std::string g_err;
void pp_set_error(const std::string &msg) { g_err = msg; }

extern "C" const char *pp_last_error(void) { return g_err.c_str(); }
extern "C" int pp_version(void) { return 101; }
// build stamp (packppi_amd/build.py passes -DPP_BUILD_ID="<sources>-<flags>"); the marker prefix lets build.py read it from
// the file without loading the library
#ifndef PP_BUILD_ID
#define PP_BUILD_ID "unstamped-unstamped"
#endif
static const char g_build_id[] = "PP_BUILD_ID=" PP_BUILD_ID;
extern "C" const char *pp_build_id(void) { return g_build_id + 12; }

#define FAIL(code, msg)      \
    do {                     \
        pp_set_error(msg);   \
        return code;         \
    } while (0)

// ---------------------------------------------------------------------------------------------
// host-side transpose of W[rows][ld] columns [c0, c0+cols) into dst[cols][rows]
static size_t put_T(std::vector<float> &arena, const float *W, int rows, int ld, int c0, int cols) {
    size_t at = arena.size();
    at = (at + 3) & ~size_t(3);
    arena.resize(at + (size_t)rows * cols);
    float *d = arena.data() + at;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) d[(size_t)c * rows + r] = W[(size_t)r * ld + c0 + c];
    return at;
}

// k-quad interleaved transpose for the node kernels (pp_node.hip): dst[in / 4][out][in % 4], so that a thread owning
// output column `out` reads four consecutive reduction inputs with one 16-byte load (cols % 4 == 0)
static size_t put_T4(std::vector<float> &arena, const float *W, int rows, int ld, int c0, int cols) {
    size_t at = arena.size();
    at = (at + 3) & ~size_t(3);
    arena.resize(at + (size_t)rows * cols);
    float *d = arena.data() + at;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) d[((size_t)(c >> 2) * rows + r) * 4 + (c & 3)] = W[(size_t)r * ld + c0 + c];
    return at;
}

// fp32 -> IEEE binary16 bits, round to nearest even (subnormals kept: the MFMA honours them)
static uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                   // rounds to >= 65520: overflow
    if (x < 0x33000001u) return (uint16_t)sign;                                 // < 2^-25: rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;                                  // subnormal: shift further
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    uint32_t out = e < -14 ? r : (((uint32_t)(e + 15) << 10) + (r - 0x400u));   // a carry out of the mantissa bumps e
    return (uint16_t)(sign | out);
}
static float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400u), (int)e - 25);
    uint32_t b;
    memcpy(&b, &v, 4);
    b |= sign;
    memcpy(&v, &b, 4);
    return v;
}

#ifdef PP_EDGE_F16      // experimental split-f16 edge kernels (pp_edge_f16.hip): PACKPPI_EDGE=f16 python -m packppi_amd.build
// Append one weight chunk (a K = 32 slice of a 128-row layer) packed for the edge kernels' wave-private LDS-DMA pipeline
// and split-f16 arithmetic (pp_edge.hip): [wave 4][k-step s 2][part hi|lo 2][lane 64][i 8] halves = 16 KB, where lane =
// (row & 31, half h) of wave row >> 5 holds the A-operand of v_mfma_f32_32x32x16_f16 for k-step s: input column
//   col(s, h, i)   (`colmap`; < 0 = zero padding)  --  32-wide slices: col0 + 8 (2 s + (i >> 2)) + 4 h + (i & 3), the order
//   in which the accumulator registers of the producing layer become B operands.
template <typename ColMap>
static void put_chunk_f16(std::vector<float> &arena, const float *W, int ld, int row0, ColMap colmap) {
    size_t at = arena.size();
    arena.resize(at + (size_t)128 * 32, 0.f);
    uint16_t *d = reinterpret_cast<uint16_t *>(arena.data() + at);
    for (int wave = 0; wave < 4; wave++)
        for (int s = 0; s < 2; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int i = 0; i < 8; i++) {
                    const int row = row0 + 32 * wave + (lane & 31), h = lane >> 5;
                    const int col = colmap(wave, s, h, i);
                    const float w = col >= 0 ? W[(size_t)row * ld + col] : 0.f;
                    const uint16_t hi = f2h(w);
                    const uint16_t lo = f2h(w - h2f(hi));
                    uint16_t *base = d + (size_t)wave * 2048;      // 4 KB per wave = 2048 halves
                    base[((2 * s + 0) * 64 + lane) * 8 + i] = hi;
                    base[((2 * s + 1) * 64 + lane) * 8 + i] = lo;
                }
}
static void put_chunk(std::vector<float> &arena, const float *W, int ld, int row0, int col0, int ncols) {
    (void)ncols;
    put_chunk_f16(arena, W, ld, row0, [col0](int, int s, int h, int i) { return col0 + 8 * (2 * s + (i >> 2)) + 4 * h + (i & 3); });
}
// chunk at position p of a ROTATED layer (pp_edge_f16.hip, "ROTATED TILE ORDER"): wave w's quarter holds the columns of input
// tile (w + p) & 3 of the 128-wide block at col_base -- every wave starts a layer with the tile it produced itself
static void put_chunk_rot(std::vector<float> &arena, const float *W, int ld, int row0, int col_base, int p) {
    put_chunk_f16(arena, W, ld, row0, [col_base, p](int wave, int s, int h, int i) {
        return col_base + 32 * ((wave + p) & 3) + 8 * (2 * s + (i >> 2)) + 4 * h + (i & 3);
    });
}
// geometry chunk C of a message MLP's first layer: features f = 16 (2 C + s) + 8 h + i of the 72 (columns 384 + f)
// Lane half h of the geometry operand carries the features of points 4h .. 4h+3 only (so the four waves of a workgroup
// compute one point each), point-major: k-step q = 0..3 holds point 4h + q as
// p_loc xyz | |p_loc| | local neighbour xyz | its norm; k-step 4 holds the four distances | 0 x4.  k-step S = 2 C + s.
static void put_geo_chunk(std::vector<float> &arena, const float *W, int C) {
    put_chunk_f16(arena, W, 456, 0, [C](int, int s, int h, int i) {
        const int S5 = 2 * C + s;
        int f;
        if (S5 < 4) {
            const int pt = 4 * h + S5;
            if (i < 3) f = 3 * pt + i;
            else if (i == 3) f = 24 + pt;
            else if (i < 7) f = 32 + 3 * pt + (i - 4);
            else f = 56 + pt;
        } else if (S5 == 4 && i < 4) {
            f = 64 + 4 * h + i;
        } else {
            return -1;
        }
        return 384 + f;
    });
}
#else
// Append one weight chunk = the [128 rows][ncols] block of W (row stride ld) at (row0, col0), packed for the edge
// kernels' wave-private LDS-DMA pipeline (pp_edge.hip): [wave 4][quad 4][lane 64][4 floats], where lane = (row & 31,
// half h) of wave row >> 5 holds the A-operand registers of MFMA steps 4q..4q+3:
//   ncols == 32:  W[row][col0 + 8 q + 4 h + p]                 (k-order F of the accumulator layout)
//   ncols == 24:  W[row][col0 + 12 h + 4 q + p], quad 3 = 0    (geometry chunks: half h feeds inputs 12 h .. 12 h + 11)
static void put_chunk(std::vector<float> &arena, const float *W, int ld, int row0, int col0, int ncols) {
    size_t at = arena.size();
    arena.resize(at + (size_t)128 * 32, 0.f);
    float *d = arena.data() + at;
    for (int wave = 0; wave < 4; wave++)
        for (int q = 0; q < (ncols == 32 ? 4 : 3); q++)
            for (int lane = 0; lane < 64; lane++)
                for (int pp = 0; pp < 4; pp++) {
                    int row = 32 * wave + (lane & 31), h = lane >> 5;
                    int col = ncols == 32 ? 8 * q + 4 * h + pp : 12 * h + 4 * q + pp;
                    d[((wave * 4 + q) * 64 + lane) * 4 + pp] = W[(size_t)(row0 + row) * ld + col0 + col];
                }
}
static void put_geo_chunk(std::vector<float> &arena, const float *W, int C) { put_chunk(arena, W, 456, 0, 384 + 24 * C, 24); }
// (the exact-fp32 edge kernels of pp_edge.hip read their input tiles in natural order)
static void put_chunk_rot(std::vector<float> &arena, const float *W, int ld, int row0, int col_base, int p) {
    put_chunk(arena, W, ld, row0, col_base + 32 * p, 32);
}
#endif
// chunk stream of one message MLP: [W_in[:,128:256] x4 unless `skip_wb`,] W_in[:,384:456] x3 (24 cols), W_mid x4
// [, W_out x4, FFN blocks].  Layer 0 skips the W_B chunks: its W_B h_E0 is precomputed once per complex (k_edge_static).
static size_t put_stream(std::vector<float> &arena, const float *w, const LayerOff &L, bool edge, bool skip_wb) {
    size_t at = (arena.size() + 3) & ~size_t(3);
    arena.resize(at);
    const float *win = w + (edge ? L.em_in_w : L.nm_in_w), *wmid = w + (edge ? L.em_mid_w : L.nm_mid_w);
    // the four chunks of a 128-wide input are consumed in rotated tile order by the split-f16 kernels (put_chunk_rot)
    if (!skip_wb)
        for (int p = 0; p < 4; p++) put_chunk_rot(arena, win, 456, 0, 128, p);
    for (int g = 0; g < 3; g++) put_geo_chunk(arena, win, g);
    for (int p = 0; p < 4; p++) put_chunk_rot(arena, wmid, 128, 0, 0, p);
    if (edge) {
        for (int p = 0; p < 4; p++) put_chunk_rot(arena, w + L.em_out_w, 128, 0, 0, p);
        for (int c = 0; c < 4; c++) {
            for (int p = 0; p < 4; p++) put_chunk_rot(arena, w + L.ed_in_w, 128, 128 * c, 0, p);
            for (int p = 0; p < 4; p++) put_chunk_rot(arena, w + L.ed_out_w, 512, 0, 128 * c, p);
        }
    }
    return at;
}
// ---- k_node_update (pp_node.hip) ------------------------------------------------------------------------------
// One slot: rows row0 .. row0+15 (those < nrows are real) of W (row stride ld), columns col0 .. col0+31 (those < col0 + ncols
// real), as the A operand of v_mfma_f32_16x16x32_f16: lane l holds W[row0 + (l & 15)][col0 + 8 (l >> 4) + j], j = 0..7;
// hi = f16(w), lo = f16((w - hi) * 2^11) (the scaling keeps lo out of the f16 subnormal range; the kernel accumulates the
// lo products separately and folds them in with 2^-11).
static void put_node_slot(uint16_t *d, const float *W, int ld, int row0, int nrows, int col0, int ncols) {
    for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
            const int r = lane & 15, k = 8 * (lane >> 4) + j;
            const float w = (W && r < nrows && k < ncols) ? W[(size_t)(row0 + r) * ld + col0 + k] : 0.f;
            const uint16_t hi = f2h(w);
            d[lane * 8 + j] = hi;
            d[512 + lane * 8 + j] = f2h((w - h2f(hi)) * PP_NU_LO_SCALE);
        }
}
// the slot list of layer l (see pp_internal.h): [wave][slot]
static size_t put_node_stream(std::vector<float> &arena, const float *w, const WeightOff &off, int l) {
    const LayerOff &L = off.layer[l];
    const bool last = l == 2;
    const int nslots = last ? PP_NU_SLOTS_LAST : PP_NU_SLOTS_MID;
    size_t at = (arena.size() + 3) & ~size_t(3);
    arena.resize(at + (size_t)PP_NU_WAVES * nslots * PP_NU_SLOT_FLOATS, 0.f);
    for (int wv = 0; wv < PP_NU_WAVES; wv++) {
        uint16_t *base = reinterpret_cast<uint16_t *>(arena.data() + at + (size_t)wv * nslots * PP_NU_SLOT_FLOATS);
        int s = 0;
        auto put = [&](const float *W, int ld, int row0, int nrows, int col0, int ncols) {
            put_node_slot(base + (size_t)s * 1024, W, ld, row0, nrows, col0, ncols);
            s++;
        };
        for (int ks = 0; ks < 4; ks++) put(w + L.nm_out_w, 128, 16 * wv, 16, 32 * ks, 32);
        for (int c = 0; c < 4; c++)
            for (int ks = 0; ks < 4; ks++) put(w + L.nd_in_w, 128, 16 * (4 * wv + c), 16, 32 * ks, 32);
        for (int ks = 0; ks < 16; ks++) put(w + L.nd_out_w, 512, 16 * wv, 16, 32 * ks, 32);
        if (!last) {
            const LayerOff &Nx = off.layer[l + 1];
            for (int ks = 0; ks < 4; ks++) put(w + L.em_in_w, 456, 16 * wv, 16, 32 * ks, 32);
            for (int ks = 0; ks < 4; ks++) put(w + L.em_in_w, 456, 16 * wv, 16, 256 + 32 * ks, 32);
            for (int ks = 0; ks < 4; ks++) put(w + Nx.nm_in_w, 456, 16 * wv, 16, 32 * ks, 32);
            for (int ks = 0; ks < 4; ks++) put(w + Nx.nm_in_w, 456, 16 * wv, 16, 256 + 32 * ks, 32);
            // the 48 point features: rows 0..23 = this layer's points_fn_edge, 24..47 = the next layer's points_fn_node
            std::vector<float> pw(48 * 128);
            memcpy(pw.data(), w + L.pts_edge_w, 24 * 128 * sizeof(float));
            memcpy(pw.data() + 24 * 128, w + Nx.pts_node_w, 24 * 128 * sizeof(float));
            for (int ks = 0; ks < 4; ks++) put(wv < 3 ? pw.data() : nullptr, 128, 16 * wv, 16, 32 * ks, 32);
        } else {
            const LayerOff &L0 = off.layer[0];
            for (int ks = 0; ks < 4; ks++) put(wv < 4 ? w + off.d0_in_w : nullptr, 128, 16 * wv, 16, 32 * ks, 32);
            for (int t = 0; t < 2; t++)
                for (int ks = 0; ks < 2; ks++) put(wv == 0 ? w + off.d0_out_w : nullptr, 64, 16 * t, 16, 32 * ks, 32);
            put(wv == 0 ? w + off.d2_in_w : nullptr, 32, 0, 16, 0, 32);
            put(wv == 0 ? w + off.d2_out_w : nullptr, 16, 0, 4, 0, 16);
            // node_embedding.weight [128][51], input columns 21..50 (6 backbone sin/cos, 8 chi sin/cos, 16 time) as one k-step
            put(w + off.node_emb_w, 51, 16 * wv, 16, 21, 30);
            for (int ks = 0; ks < 4; ks++) put(w + L0.nm_in_w, 456, 16 * wv, 16, 32 * ks, 32);
            for (int ks = 0; ks < 4; ks++) put(w + L0.nm_in_w, 456, 16 * wv, 16, 256 + 32 * ks, 32);
            for (int ks = 0; ks < 4; ks++) put(wv < 2 ? w + L0.pts_node_w : nullptr, 128, 16 * wv, wv == 0 ? 16 : 8, 32 * ks, 32);
        }
        if (s != nslots) abort();
    }
    return at;
}
static size_t put_node_params(std::vector<float> &arena, const float *w, const WeightOff &off, int l) {
    const LayerOff &L = off.layer[l];
    const bool last = l == 2;
    size_t at = (arena.size() + 3) & ~size_t(3);
    arena.resize(at + (last ? NU_P_LAST_TOTAL : NU_P_MID_TOTAL), 0.f);
    float *d = arena.data() + at;
    auto cp = [&](int dst, size_t src, int n) { memcpy(d + dst, w + src, n * sizeof(float)); };
    cp(NU_P_OUTB, L.nm_out_b, 128); cp(NU_P_G0, L.norm_g[0], 128); cp(NU_P_B0, L.norm_b[0], 128);
    cp(NU_P_FIB, L.nd_in_b, 512); cp(NU_P_FOB, L.nd_out_b, 128); cp(NU_P_G1, L.norm_g[1], 128); cp(NU_P_B1, L.norm_b[1], 128);
    if (!last) {
        const LayerOff &Nx = off.layer[l + 1];
        cp(NU_P_PAE_B, L.em_in_b, 128); cp(NU_P_PAN_B, Nx.nm_in_b, 128);
        cp(NU_P_PTS_B, L.pts_edge_b, 24); cp(NU_P_PTS_B + 24, Nx.pts_node_b, 24);
    } else {
        const LayerOff &L0 = off.layer[0];
        cp(NU_P_DB0, off.d0_in_b, 64); cp(NU_P_DB1, off.d0_out_b, 32); cp(NU_P_DB2, off.d2_in_b, 16); cp(NU_P_DB3, off.d2_out_b, 4);
        cp(NU_P_PAN0_B, L0.nm_in_b, 128); cp(NU_P_PTS0_B, L0.pts_node_b, 24);
        cp(NU_P_EMB_B, off.node_emb_b, 128); cp(NU_P_EMB_G, off.norm_nodes_g, 128); cp(NU_P_EMB_BETA, off.norm_nodes_b, 128);
    }
    return at;
}

// k_edge_static's stream: the W_B chunks of layer 0's node message, then of its edge message
static size_t put_static_stream(std::vector<float> &arena, const float *w, const LayerOff &L) {
    size_t at = (arena.size() + 3) & ~size_t(3);
    arena.resize(at);
    for (int s = 0; s < 4; s++) put_chunk(arena, w + L.nm_in_w, 456, 0, 128 + 32 * s, 32);
    for (int s = 0; s < 4; s++) put_chunk(arena, w + L.em_in_w, 456, 0, 128 + 32 * s, 32);
    return at;
}
// the edge kernel's small per-layer vectors in one block (staged to LDS once per workgroup):
// b_mid | b_out | ffn_out_b | g2 | be2 | ffn_in_b[512]   (1152 floats)
static size_t put_edge_params(std::vector<float> &arena, const float *w, const LayerOff &L) {
    size_t at = (arena.size() + 3) & ~size_t(3);
    arena.resize(at);
    auto app = [&](size_t off, int n) { arena.insert(arena.end(), w + off, w + off + n); };
    app(L.em_mid_b, 128); app(L.em_out_b, 128); app(L.ed_out_b, 128);
    app(L.norm_g[2], 128); app(L.norm_b[2], 128);
    app(L.ed_in_b, 512);
    return at;
}

#ifdef PP_EDGE_F16
#include "pp_rebalance.h"      // rebalance_relu_chains(): power-of-two rebalancing of the ReLU chains (host-only header)
#endif

template <typename T>
static pp_status upload(T **dst, const T *src, size_t n) {
    PP_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
    PP_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PP_OK;
}

// f16 operand range check (pp_internal.h): events counted since the last reset by the kernels of a -DPP_CHECK_RANGE build
extern "C" int pp_has_range_check(void) {
#ifdef PP_CHECK_RANGE
    return 1;
#else
    return 0;
#endif
}
extern "C" pp_status pp_range_check_parts(unsigned long long *edge_events, unsigned long long *node_events, int reset) {
    if (!edge_events || !node_events) FAIL(PP_ERR_INVALID, "pp_range_check_parts: null argument");
#ifdef PP_CHECK_RANGE
    PP_HIP_CHECK(hipDeviceSynchronize());
    *edge_events = (unsigned long long)pp_edge_range_hits(reset);
    *node_events = (unsigned long long)pp_node_range_hits(reset);
    return PP_OK;
#else
    *edge_events = *node_events = 0;
    FAIL(PP_ERR_UNSUPPORTED, "pp_range_check_parts: this library was built without -DPP_CHECK_RANGE (use libpackppi_hip.chk.so: "
                             "python -m packppi_amd.rangecheck)");
#endif
}
// Non-finite INPUTS.  The kernels clamp hidden activations with v_med3 / v_max, which turn a NaN into a finite number: a NaN that
// enters with the caller's tensors would come out as finite angles that mean nothing, where the reference returns NaN
// (layers.py:22-33 has no such clamp).  From finite inputs no NaN can arise inside (every weight is checked finite at plan
// creation, every LayerNorm has its eps, every hidden activation is bounded), so the inputs are what is checked: the backbone
// coordinates of unmasked rows when a context is prepared, the angles at every pp_score / pp_sample.  Bit 2 of the sticky word.
__global__ void k_flag_nonfinite(const float *__restrict__ v, int rows, int row_stride, int per_row, const float *__restrict__ rmask,
                                 unsigned *__restrict__ sat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * per_row) return;
    const int r = i / per_row, k = i - r * per_row;
    if (rmask && rmask[r] == 0.f) return;
    const float x = v[(size_t)r * row_stride + k];
    if (!(__builtin_fabsf(x) <= 3.402823466e38f)) atomicOr(sat, 4u);
}
static void flag_nonfinite(pp_ctx *c, const float *v, int row_stride, int per_row, hipStream_t s) {
    const int n = c->N * per_row;
    hipLaunchKernelGGL(k_flag_nonfinite, dim3((n + 255) / 256), dim3(256), 0, s, v, c->N, row_stride, per_row, c->b.residue_mask, c->sat);
}

// sticky saturation word of the context (every build): waits for `stream`
extern "C" pp_status pp_ctx_saturated(pp_ctx *c, int *flags, void *stream) {
    if (!c || !flags) FAIL(PP_ERR_INVALID, "pp_ctx_saturated: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned v = 0;
    PP_HIP_CHECK(hipMemcpyAsync(&v, c->sat, sizeof(v), hipMemcpyDeviceToHost, s));
    PP_HIP_CHECK(hipStreamSynchronize(s));
    *flags = (int)v;
    return PP_OK;
}
extern "C" pp_status pp_range_check(unsigned long long *events, int reset) {
    if (!events) FAIL(PP_ERR_INVALID, "pp_range_check: null argument");
#ifdef PP_CHECK_RANGE
    PP_HIP_CHECK(hipDeviceSynchronize());
    *events = (unsigned long long)pp_edge_range_hits(reset) + (unsigned long long)pp_node_range_hits(reset);
    return PP_OK;
#else
    *events = 0;
    FAIL(PP_ERR_UNSUPPORTED, "pp_range_check: this library was built without -DPP_CHECK_RANGE (use libpackppi_hip.chk.so: "
                             "python -m packppi_amd.rangecheck)");
#endif
}

extern "C" pp_status pp_plan_create(const float *weights, size_t n_weights, const pp_tables *tables, int device,
                                    pp_plan **out) {
    if (!tables || !out) FAIL(PP_ERR_INVALID, "pp_plan_create: null argument");
    WeightOff off = pp_weight_offsets();
    const bool has_net = weights != nullptr;
    if (has_net && (n_weights != off.total || off.total != PP_N_WEIGHTS))
        FAIL(PP_ERR_INVALID, "pp_plan_create: expected " + std::to_string(off.total) + " weights, got " +
                                 std::to_string(n_weights));
    if (has_net)      // the dense layers run on two-way f16 splits of the weights: every weight must be finite and inside the f16 range
        for (size_t i = 0; i < n_weights; i++)
            if (!(std::fabs(weights[i]) < 65504.f))
                FAIL(PP_ERR_INVALID, "pp_plan_create: weight " + std::to_string(i) + " is not finite or outside the f16 range (" +
                                         std::to_string(weights[i]) + ")");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FAIL(PP_ERR_NO_DEVICE, "pp_plan_create: no HIP device visible");
    if (device < 0 || device >= ndev) FAIL(PP_ERR_INVALID, "pp_plan_create: bad device index");
    PP_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    PP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        FAIL(PP_ERR_NO_DEVICE, std::string("pp_plan_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);

    pp_plan *p = new (std::nothrow) pp_plan();
    if (!p) FAIL(PP_ERR_INVALID, "out of host memory");
    p->device = device;
    p->off = off;
    p->has_network = has_net;
    p->knn_ties = PP_KNN_TIES_ATEN_CPU;
    p->annealed_temp = 3.0f;      // configs/model/sample_cfg/Sampling.yaml:4
    p->rebalanced_chains = 0;
    pp_status st;
    const float *wpack = weights;          // what the edge-level MFMA streams are packed from
#ifdef PP_EDGE_F16
    std::vector<float> rebalanced, ln_packed;
    p->ln_scale = nullptr;
    p->ln_scaled_features = 0;
    if (has_net) {
        LnScales sc;
        p->rebalanced_chains = rewrite_checkpoint(weights, off, rebalanced, ln_packed, sc);
        p->ln_scaled_features = sc.n_scaled;
        for (size_t i = 0; i < off.total; i++)
            if (!(std::fabs(rebalanced[i]) < 65504.f) || !(std::fabs(ln_packed[i]) < 65504.f)) {
                delete p;
                FAIL(PP_ERR_INVALID, "pp_plan_create: weight " + std::to_string(i) + " leaves the f16 range when its ReLU chain is "
                                     "rebalanced (run this checkpoint on libpackppi_hip.f32.so)");
            }
        weights = rebalanced.data();       // everything below -- device copy, node-level streams, transposed copies -- is made from it
        wpack = ln_packed.data();          // ... and the edge-level streams from the copy with the operand scales in its columns
        if (sc.n_scaled > 0 && (st = upload(&p->ln_scale, &sc.v[0][0], (size_t)5 * 128)) != PP_OK) return st;
    }
#endif
    if (has_net) {
    if ((st = upload(&p->w, weights, off.total)) != PP_OK) return st;

    std::vector<float> arena;
    arena.reserve(3u << 20);
    size_t o_node_emb = put_T(arena, weights + off.node_emb_w, 128, 51, 0, 51);
    size_t o_edge_emb = put_T(arena, weights + off.edge_emb_w, 128, 468, 0, 468);
    size_t o_l[3][14];
    for (int l = 0; l < 3; l++) {
        const LayerOff &L = off.layer[l];
        o_l[l][0] = put_T4(arena, weights + L.pts_node_w, 24, 128, 0, 128);
        o_l[l][1] = put_T4(arena, weights + L.pts_edge_w, 24, 128, 0, 128);
        o_l[l][2] = put_T4(arena, weights + L.nm_in_w, 128, 456, 0, 128);
        o_l[l][3] = put_T4(arena, weights + L.nm_in_w, 128, 456, 256, 128);
        o_l[l][4] = put_T4(arena, weights + L.em_in_w, 128, 456, 0, 128);
        o_l[l][5] = put_T4(arena, weights + L.em_in_w, 128, 456, 256, 128);
        o_l[l][6] = put_T4(arena, weights + L.nm_out_w, 128, 128, 0, 128);
        o_l[l][7] = put_T4(arena, weights + L.nd_in_w, 512, 128, 0, 128);
        o_l[l][8] = put_T4(arena, weights + L.nd_out_w, 128, 512, 0, 512);
        o_l[l][9] = put_stream(arena, wpack, L, false, l == 0);
        o_l[l][10] = put_stream(arena, wpack, L, true, l == 0);
        if (l < 2) put_stream(arena, wpack, off.layer[l + 1], false, false);   // fused kernel: next layer's node message follows
        o_l[l][11] = put_edge_params(arena, weights, L);
        o_l[l][12] = put_node_stream(arena, weights, off, l);
        o_l[l][13] = put_node_params(arena, weights, off, l);
    }
    size_t o_static = put_static_stream(arena, wpack, off.layer[0]);


#ifdef PP_EDGE_F16
    // edge embedding, RBF block (input columns 65..464 of encoder.edge_embedding.weight): 13 chunks of two 16-deep k-steps,
    // k-step S = atom pair S, lane half h = RBFs 8h .. 8h+7 of that pair
    size_t o_embed = (arena.size() + 3) & ~size_t(3);
    arena.resize(o_embed);
    for (int cch = 0; cch < 13; cch++)
        put_chunk_f16(arena, weights + off.edge_emb_w, 468, 0, [cch](int, int s2, int h, int i) {
            const int k = 32 * cch + 16 * s2 + 8 * h + i;
            return k < 400 ? 65 + k : -1;
        });
#endif
    size_t o_d0i = put_T4(arena, weights + off.d0_in_w, 64, 128, 0, 128);
    size_t o_d0o = put_T4(arena, weights + off.d0_out_w, 32, 64, 0, 64);
    size_t o_d2i = put_T4(arena, weights + off.d2_in_w, 16, 32, 0, 32);
    size_t o_d2o = put_T4(arena, weights + off.d2_out_w, 4, 16, 0, 16);
    if ((st = upload(&p->wT, arena.data(), arena.size())) != PP_OK) return st;
    p->node_emb_T = p->wT + o_node_emb;
    p->edge_emb_T = p->wT + o_edge_emb;
    for (int l = 0; l < 3; l++) {
        LayerT &t = p->lt[l];
        t.pts_node_wT = p->wT + o_l[l][0]; t.pts_edge_wT = p->wT + o_l[l][1];
        t.nm_A_T = p->wT + o_l[l][2]; t.nm_C_T = p->wT + o_l[l][3];
        t.em_A_T = p->wT + o_l[l][4]; t.em_C_T = p->wT + o_l[l][5];
        t.nm_out_T = p->wT + o_l[l][6];
        t.nd_in_T = p->wT + o_l[l][7]; t.nd_out_T = p->wT + o_l[l][8];
        t.nm_stream = p->wT + o_l[l][9]; t.em_stream = p->wT + o_l[l][10];
        t.em_params = p->wT + o_l[l][11];
        t.nu_stream = p->wT + o_l[l][12]; t.nu_params = p->wT + o_l[l][13];
    }
    p->static_stream = p->wT + o_static;


#ifdef PP_EDGE_F16
    p->embed_stream = p->wT + o_embed;
#endif
    p->d0_in_T = p->wT + o_d0i; p->d0_out_T = p->wT + o_d0o;
    p->d2_in_T = p->wT + o_d2i; p->d2_out_T = p->wT + o_d2o;
    }

    if ((st = upload(&p->default_frames, tables->default_frames, 21 * 8 * 16)) != PP_OK) return st;
    if ((st = upload(&p->atom14_to_group, tables->atom14_to_group, 21 * 14)) != PP_OK) return st;
    if ((st = upload(&p->atom14_mask, tables->atom14_mask, 21 * 14)) != PP_OK) return st;
    if ((st = upload(&p->lit_positions, tables->lit_positions, 21 * 14 * 3)) != PP_OK) return st;
    if ((st = upload(&p->between_radius, tables->between_radius, 21 * 14)) != PP_OK) return st;
    {
        // How far a side-chain atom can get from its CA whatever the chi angles are: the atom sits at chain(lit) in the backbone
        // frame (origin CA), the chain composes default frames and rotations about x, a rotation keeps the norm and a frame adds at
        // most the length of its translation: |atom - CA| <= |lit| + sum of |t_k| over the frames of its chain (features.py:95-194).
        // Rigorous, about 15 % above the true maximum; the proximal loop's static partner lists are built from it (pp_clash.hip).
        float ext[21];
        for (int S = 0; S < 21; S++) {
            float m = 0.f;
            for (int a = 4; a < 14; a++) {
                if (tables->atom14_mask[S * 14 + a] == 0.f) continue;
                const float *lp = tables->lit_positions + (S * 14 + a) * 3;
                float b = std::sqrt(lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]);
                const int g = tables->atom14_to_group[S * 14 + a];
                auto tlen = [&](int k) { const float *f = tables->default_frames + ((size_t)S * 8 + k) * 16; return std::sqrt(f[3] * f[3] + f[7] * f[7] + f[11] * f[11]); };
                if (g >= 4) for (int k = 4; k <= g; k++) b += tlen(k);
                else b += tlen(g);
                m = std::max(m, b);
            }
            ext[S] = m * 1.0001f + 1e-3f;
        }
        if ((st = upload(&p->side_extent, ext, 21)) != PP_OK) return st;
    }
    PP_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p->bounds_lower), 21 * 14 * 14 * sizeof(float)));
    PP_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p->bounds_upper), 21 * 14 * 14 * sizeof(float)));
    p->clash_params_set = false;
    *out = p;
    return PP_OK;
}

extern "C" void pp_plan_destroy(pp_plan *p) {
    if (!p) return;
    void *ptrs[] = {p->w, p->wT, p->default_frames, p->atom14_to_group, p->atom14_mask, p->lit_positions,
                    p->between_radius, p->bounds_lower, p->bounds_upper, p->ln_scale, p->side_extent};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    for (const ArenaSlot &sl : p->arena_pool) {
        (void)hipFree(sl.p);
    }
    delete p;
}

extern "C" pp_status pp_plan_set_knn_ties(pp_plan *p, int mode) {
    if (!p) FAIL(PP_ERR_INVALID, "pp_plan_set_knn_ties: null plan");
    if (mode != PP_KNN_TIES_LOWER_INDEX && mode != PP_KNN_TIES_ATEN_CPU && mode != PP_KNN_TIES_ATEN_MEMBER)
        FAIL(PP_ERR_INVALID, "pp_plan_set_knn_ties: unknown mode " + std::to_string(mode));
    p->knn_ties = mode;
    return PP_OK;
}

extern "C" int pp_plan_rebalanced_chains(const pp_plan *p) { return p ? p->rebalanced_chains : -1; }
extern "C" int pp_plan_ln_scaled_features(const pp_plan *p) { return p ? p->ln_scaled_features : -1; }
// HOST helper, no device call: the five operand-scale vectors pp_plan_create would choose for these weights (after the ReLU-chain
// rebalancing), [h_E0 | h_E after layer 0 | after layer 1 | x1 of layer 0 | of layer 1] x 128; the exact-fp32 build returns ones
extern "C" pp_status pp_ln_operand_scales_host(const float *weights, size_t n_weights, float *out, int *n_scaled) {
    const WeightOff off = pp_weight_offsets();
    if (!weights || !out || n_weights != off.total) FAIL(PP_ERR_INVALID, "pp_ln_operand_scales_host: bad argument");
    int n = 0;
#ifdef PP_EDGE_F16
    const LnScales sc = ln_operand_scales(weights, off);
    std::memcpy(out, &sc.v[0][0], sizeof(sc.v));
    n = sc.n_scaled;
#else
    for (int i = 0; i < 5 * 128; i++) out[i] = 1.f;
#endif
    if (n_scaled) *n_scaled = n;
    return PP_OK;
}

// HOST helper, no device call: the weight vector as pp_plan_create packs it (split-f16 build: ReLU chains rebalanced by powers of
// two; exact-fp32 build: a copy).  Lets a CPU test hold the rebalanced network to the original one through the oracle.
extern "C" pp_status pp_rebalance_weights_host(const float *weights, size_t n_weights, float *out, int *chains) {
    const WeightOff off = pp_weight_offsets();
    if (!weights || !out || n_weights != off.total) FAIL(PP_ERR_INVALID, "pp_rebalance_weights_host: bad argument");
    memcpy(out, weights, n_weights * sizeof(float));
    int c = 0;
#ifdef PP_EDGE_F16
    {
        std::vector<float> plain, packed;
        LnScales sc;
        c = rewrite_checkpoint(weights, off, plain, packed, sc);      // (`plain`: the rebalanced network with the operand scales multiplied back)
        memcpy(out, plain.data(), n_weights * sizeof(float));
    }
#endif
    if (chains) *chains = c;
    return PP_OK;
}

// sample_cfg.annealed_temp (TorsionalDiffusion.py:70-75 -> SO2VESchedule(annealed_temp=...), schedule.py:205-208)
extern "C" pp_status pp_plan_set_annealed_temp(pp_plan *p, float T) {
    if (!p) FAIL(PP_ERR_INVALID, "pp_plan_set_annealed_temp: null plan");
    // schedule.py:216-217 tests `if self.annealed_temp`: 0 (and None, which the host side maps to 0) switch the annealing off
    if (!std::isfinite(T)) FAIL(PP_ERR_INVALID, "pp_plan_set_annealed_temp: annealed_temp must be finite (0 = no annealing)");
    p->annealed_temp = T;
    return PP_OK;
}

// torch.topk(values, k, largest=False) indices of ATen's CPU kernel (pp_topk_aten.h), on the host
extern "C" pp_status pp_topk_aten_host(const float *values, int n, int k, int32_t *idx_out) {
    if (!values || !idx_out || n < 1 || k < 1 || k > n) FAIL(PP_ERR_INVALID, "pp_topk_aten_host: bad argument");
    std::vector<pp_tk_pair> q((size_t)n);
    for (int j = 0; j < n; j++) { q[j].v = values[j]; q[j].i = j; }
    pp_tk_topk_smallest(q.data(), n, k);
    for (int j = 0; j < k; j++) idx_out[j] = q[j].i;
    return PP_OK;
}

extern "C" pp_status pp_plan_set_clash_params(pp_plan *p, float tol, const float *lower, const float *upper, void *stream) {
    if (!p || !lower || !upper) FAIL(PP_ERR_INVALID, "pp_plan_set_clash_params: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the host tables are caller-owned and may be freed on return: blocking copies
    PP_HIP_CHECK(hipStreamSynchronize(s));
    PP_HIP_CHECK(hipMemcpy(p->bounds_lower, lower, 21 * 14 * 14 * sizeof(float), hipMemcpyHostToDevice));
    PP_HIP_CHECK(hipMemcpy(p->bounds_upper, upper, 21 * 14 * 14 * sizeof(float), hipMemcpyHostToDevice));
    p->clash_tol = tol;
    p->clash_params_set = true;
    return PP_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" void pp_ctx_destroy(pp_ctx *c) {
    if (!c) return;
    if (c->arena) {                                   // every workspace pointer lives in this one allocation
        // hand it to the plan's pool instead of hipFree (which synchronises the device): a context per batch is created
        // and destroyed on the sampling path.  Work already enqueued on last_stream may still be using the memory; the
        // next owner either runs on the same stream (ordered after it) or waits for that stream first.
        pp_plan *p = c->plan;
        std::lock_guard<std::mutex> g(p->pool_mutex);
        if (p->arena_pool.size() < 4) {
            p->arena_pool.push_back({c->arena, c->arena_bytes, c->last_stream});
        } else {
            (void)hipFree(c->arena);
        }
    }
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    delete c;
}

// (first row, length) of the complex every row belongs to: a padded batch [B][L] ...
__global__ void k_fill_seg(int2 *__restrict__ seg, int N, int L) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) seg[n] = make_int2((n / L) * L, L);
}
// ... or complexes packed back to back, rows off[s] .. off[s + 1] - 1 (pp_complex_prepare_packed)
__global__ void k_fill_seg_packed(int2 *__restrict__ seg, int N, const int32_t *__restrict__ off, int n_seg, int max_len) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int lo = 0, hi = n_seg - 1;                   // last s with off[s] <= n
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= n) lo = mid; else hi = mid - 1;
    }
    // the launches are sized by the caller's max_len (LDS of the neighbour search): a segment table that disagrees with it
    // is clamped to stay inside the batch and the LDS, its results are meaningless
    int start = off[lo], len = off[lo + 1] - off[lo];
    start = start < 0 ? 0 : (start > n ? n : start);
    len = len > max_len ? max_len : len;
    if (start + len > N) len = N - start;
    if (n >= start + len) len = n - start + 1 <= max_len ? n - start + 1 : max_len;
    seg[n] = make_int2(start, len);
}

static pp_status prepare_impl(pp_plan *plan, const pp_batch *b, const int32_t *seg_offsets, int n_seg, int min_len,
                              int max_len, void *stream, pp_ctx **out) {
    const bool packed = seg_offsets != nullptr;
    if (!plan || !b || !out) FAIL(PP_ERR_INVALID, "pp_complex_prepare: null argument");
    if (b->B <= 0 || b->L <= 0) FAIL(PP_ERR_INVALID, "pp_complex_prepare: B and L must be positive");
    if ((packed ? max_len : b->L) > 16384) FAIL(PP_ERR_UNSUPPORTED, "pp_complex_prepare: L > 16384 residues per complex is not supported");
    if (packed) {
        if (b->B != 1) FAIL(PP_ERR_INVALID, "pp_complex_prepare_packed: the batch tensors are [1, sum of lengths, ...]");
        if (n_seg < 1 || min_len < 1 || max_len < min_len || (long long)n_seg * min_len > b->L || (long long)n_seg * max_len < b->L)
            FAIL(PP_ERR_INVALID, "pp_complex_prepare_packed: segment count / lengths do not fit the batch");
        // K = min(32, L) is a per-batch constant of the reference (encoder.py:115): complexes shorter than 32 residues
        // cannot share a context with longer ones
        if (min_len < PP_TOP_K && min_len != max_len)
            FAIL(PP_ERR_UNSUPPORTED, "pp_complex_prepare_packed: complexes shorter than 32 residues must be prepared on their own");
    }
    if (!b->X || !b->residue_type || !b->BB_D)
        FAIL(PP_ERR_INVALID, "pp_complex_prepare: batch needs at least X, residue_type and BB_D");
    const bool net = plan->has_network;
    if (net && (!b->atom_mask || !b->residue_mask || !b->residue_index || !b->chain_indices || !b->BB_D_sincos ||
                !b->SC_D_mask || !b->chi_1pi_periodic_mask || !b->chi_2pi_periodic_mask))
        FAIL(PP_ERR_INVALID, "pp_complex_prepare: batch has a null tensor pointer");
    PP_HIP_CHECK(hipSetDevice(plan->device));
    pp_ctx *c = new (std::nothrow) pp_ctx();
    if (!c) FAIL(PP_ERR_INVALID, "out of host memory");      // value-initialised: every pointer null, prof_which = -1
    c->plan = plan;
    c->b = *b;
    c->packed = packed;
    c->B = packed ? n_seg : b->B;
    c->L = packed ? max_len : b->L;
    c->N = b->B * b->L;
    const int shortest = packed ? min_len : b->L;
    c->K = shortest < PP_TOP_K ? shortest : PP_TOP_K;
    const size_t N = c->N, K = c->K;
    pp_status st = PP_OK;
    // one arena for all workspaces (a context per batch is created and destroyed on the sampling path: ~35 hipMalloc /
    // hipFree pairs cost 1.5 ms per context, one pair 0.1 ms): sizes first, then one hipMalloc, then the pointers
    struct Slot { void **p; size_t bytes; };
    std::vector<Slot> slots;
    size_t total = 0;
#define ALLOC(field, n) { const size_t bytes_ = (((n) ? (n) : 1) * sizeof(*c->field) + 255) & ~size_t(255);          \
                          slots.push_back({reinterpret_cast<void **>(&c->field), bytes_}); total += bytes_; }
    if (net) {
    ALLOC(eidx, N * K); ALLOC(mask_att, N * 32); ALLOC(frames, N * 12); ALLOC(bbpos, N * 15);
    ALLOC(hE0, N * K * 128); ALLOC(hE, N * K * 128); ALLOC(Znm, N * K * 128); ALLOC(Zem, N * K * 128); ALLOC(hV, N * 128); ALLOC(hV_alt, N * 128); ALLOC(S, N * 128); ALLOC(msum, N);
    ALLOC(ptsN, N * 48); ALLOC(PAn, N * 128); ALLOC(PCn, N * 128);
    ALLOC(ptsE, N * 48); ALLOC(PAe, N * 128); ALLOC(PCe, N * 128);
    ALLOC(score, N * 4); ALLOC(chi_tmp, N * 4);
    }
    ALLOC(xyz, N * 42); ALLOC(rec, N * 64); ALLOC(axes, N * 24); ALLOC(rec2, N * 64); ALLOC(axes2, N * 24); ALLOC(brad, N); ALLOC(per_res, N); ALLOC(dchi, N * 4);
    ALLOC(px, N * 4); ALLOC(pm, N * 4); ALLOC(pv, N * 4); ALLOC(pz, N * 4); ALLOC(pxeff, N * 4); ALLOC(pmask, N);
    if (c->B == 1) { ALLOC(cand, (size_t)N * 4 * PP_CL_CAP); ALLOC(cand_cnt, N * 4); }      // the proximal loop is defined for one complex (optimize.py:27)
    ALLOC(scal, 64);
    ALLOC(sat, 4);
    ALLOC(seg, N);
    ALLOC(prox_part, (size_t)PP_PROX_CHUNK * N);
    c->max_steps = 1 << 20;
#undef ALLOC
    c->last_stream = static_cast<hipStream_t>(stream);
    {
        std::lock_guard<std::mutex> g(plan->pool_mutex);
        int best = -1;
        for (int i = 0; i < (int)plan->arena_pool.size(); i++)
            if (plan->arena_pool[i].bytes >= total && (best < 0 || plan->arena_pool[i].bytes < plan->arena_pool[best].bytes)) best = i;
        if (best >= 0) {
            const ArenaSlot sl = plan->arena_pool[best];
            plan->arena_pool.erase(plan->arena_pool.begin() + best);
            if (sl.stream != c->last_stream) (void)hipStreamSynchronize(sl.stream);
            c->arena = sl.p;
            c->arena_bytes = sl.bytes;
        }
    }
    if (!c->arena && hipMalloc(&c->arena, total) != hipSuccess) {
        pp_set_error("hipMalloc of the context workspace failed");
        c->arena = nullptr;
        st = PP_ERR_HIP;
    } else {
        if (!c->arena_bytes) c->arena_bytes = total;
        char *base = static_cast<char *>(c->arena);
        for (const Slot &sl : slots) { *sl.p = base; base += sl.bytes; }
    }
    if (st == PP_OK) {
        hipStream_t s_ = static_cast<hipStream_t>(stream);
        if (packed) hipLaunchKernelGGL(k_fill_seg_packed, dim3((c->N + 255) / 256), dim3(256), 0, s_, c->seg, c->N, seg_offsets, n_seg, c->L);
        else hipLaunchKernelGGL(k_fill_seg, dim3((c->N + 255) / 256), dim3(256), 0, s_, c->seg, c->N, c->L);
        if (hipGetLastError() != hipSuccess) { pp_set_error("segment table launch failed"); st = PP_ERR_HIP; }
    }
    if (st == PP_OK && hipMemsetAsync(c->sat, 0, 4 * sizeof(unsigned), static_cast<hipStream_t>(stream)) != hipSuccess) {
        pp_set_error("clearing the saturation word failed");
        st = PP_ERR_HIP;
    }
    if (net && st == PP_OK) flag_nonfinite(c, c->b.X, 42, 12, static_cast<hipStream_t>(stream));      // N, CA, C, O of the unmasked rows
    if (net && st == PP_OK) st = pp_launch_prepare(c, static_cast<hipStream_t>(stream));
    if (net && st == PP_OK) st = pp_launch_edge_static(c, static_cast<hipStream_t>(stream));
    if (st != PP_OK) { pp_ctx_destroy(c); return st; }
    *out = c;
    return PP_OK;
}

extern "C" pp_status pp_complex_prepare(pp_plan *plan, const pp_batch *b, void *stream, pp_ctx **out) {
    return prepare_impl(plan, b, nullptr, 0, 0, 0, stream, out);
}

extern "C" pp_status pp_complex_prepare_packed(pp_plan *plan, const pp_batch *b, const int32_t *seg_offsets, int n_seg,
                                               int min_len, int max_len, void *stream, pp_ctx **out) {
    if (!seg_offsets) FAIL(PP_ERR_INVALID, "pp_complex_prepare_packed: null segment offsets");
    return prepare_impl(plan, b, seg_offsets, n_seg, min_len, max_len, stream, out);
}

__global__ void k_widen_idx(const int32_t *__restrict__ src, int64_t *__restrict__ dst, size_t n, const int2 *__restrict__ seg, int K) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        size_t node = i / K;
        dst[i] = (int64_t)src[i] - (int64_t)seg[node].x;      // back to the per-complex residue numbering of the reference
    }
}

extern "C" pp_status pp_ctx_get_graph(pp_ctx *c, int64_t *E_idx, float *hE0, void *stream) {
    if (!c) FAIL(PP_ERR_INVALID, "pp_ctx_get_graph: null ctx");
    if (!c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_ctx_get_graph: plan was created without network weights");
    hipStream_t s = static_cast<hipStream_t>(stream);
    size_t n = (size_t)c->N * c->K;
    if (E_idx) hipLaunchKernelGGL(k_widen_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->eidx, E_idx, n, c->seg, c->K);
    if (hE0) PP_HIP_CHECK(hipMemcpyAsync(hE0, c->hE0, n * 128 * sizeof(float), hipMemcpyDeviceToDevice, s));
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

extern "C" pp_status pp_ctx_set_graph(pp_ctx *c, const int64_t *E_idx, void *stream) {
    if (c) c->last_stream = static_cast<hipStream_t>(stream);
    if (!c || !E_idx) FAIL(PP_ERR_INVALID, "pp_ctx_set_graph: null argument");
    if (!c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_ctx_set_graph: plan was created without network weights");
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    pp_status st = pp_launch_prepare(c, s, E_idx);
    if (st == PP_OK) st = pp_launch_edge_static(c, s);
    if (st != PP_OK) return st;
    int bad = 0;          // inspection-grade call: it validates the indices, which needs one read-back
    PP_HIP_CHECK(hipMemcpyAsync(&bad, c->scal, sizeof(int), hipMemcpyDeviceToHost, s));
    PP_HIP_CHECK(hipStreamSynchronize(s));
    if (bad) FAIL(PP_ERR_INVALID, "pp_ctx_set_graph: " + std::to_string(bad) + " neighbour indices outside their complex");
    return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// per-step scalars (schedule.py:165-174,198-235; layers.py:257-268), fp32 like the reference's tensors
static void fill_step(StepParams *sp, float t, float dt, float T) {
    const double PI_D = 3.14159265358979323846;
    const double lo = log(0.01 * PI_D), hi = log(PI_D);
    memset(sp, 0, sizeof(*sp));
    // sinusoidal embedding of t * 10000
    const float ts = t * 10000.0f;
    const float nemb = (float)(-(log(10000.0) / 7.0));
    for (int i = 0; i < 8; i++) {
        float freq = expf((float)i * nemb);
        float arg = ts * freq;
        sp->temb[i] = (float)sin((double)arg);
        sp->temb[8 + i] = (float)cos((double)arg);
    }
    float sigma = expf((float)lo + (float)(hi - lo) * t);
    float g = sigma * (float)sqrt(2.0 * log(PI_D / (0.01 * PI_D)));
    float ratio = sigma / (float)exp(hi);
    float alpha = 1.0f - ratio * ratio;
    sp->w = T != 0.f ? T / (alpha + (1.0f - alpha) * T) : 1.0f;      // schedule.py:216-217: a falsy annealed_temp means weight 1
    sp->c_ode = (0.5f * (g * g)) * dt;
    sp->c_drift = (g * g) * dt;
    sp->c_diff = g * sqrtf(dt);
}

// pp_profile_kernel: arm the launch of kernel class `which` that follows (the launcher's PP_LAUNCH takes the event pair)
static inline void prof_arm(pp_ctx *c, int which) { c->prof_armed = c->prof_which == which; }
static inline void prof_disarm(pp_ctx *c) { c->prof_armed = false; }
bool pp_prof_take(pp_ctx *c, hipEvent_t *e0, hipEvent_t *e1) {
    c->prof_armed = false;
    while (c->prof_ev.size() < c->prof_n + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return false;
        c->prof_ev.push_back(e);
    }
    *e0 = c->prof_ev[c->prof_n++];
    *e1 = c->prof_ev[c->prof_n++];
    return true;
}

static pp_status run_network(pp_ctx *c, hipStream_t s, int step, int last_mode, float *chi, int mode, const float *noise,
                             const StepParams *cur, const StepParams *next) {
    pp_status st;
    for (int l = 0; l < 3; l++) {
        if (l == 0 || !pp_edge_fused()) {   // fused build: layers 1 and 2 come from the tail of the previous edge update
            prof_arm(c, 0);
            st = pp_launch_node_message(c, l, s);
            prof_disarm(c);
            if (st != PP_OK) return st;
        }
        if (l < 2) {
            prof_arm(c, 2);
            st = pp_launch_node_update(c, l, PP_NU_MID, chi, step, mode, noise, nullptr, nullptr, s);
            prof_disarm(c);
            if (st != PP_OK) return st;
            prof_arm(c, 1);
            st = pp_launch_edge_update(c, l, s);
            prof_disarm(c);
            if (st != PP_OK) return st;
        } else {
            prof_arm(c, 2);
            st = pp_launch_node_update(c, l, last_mode, chi, step, mode, noise, cur, last_mode == PP_NU_STEP ? next : nullptr, s);
            prof_disarm(c);
            if (st != PP_OK) return st;
        }
    }
    return PP_OK;
}

extern "C" pp_status pp_score(pp_ctx *c, const float *chi, float t, float *score, float *hV, void *stream) {
    if (c) c->last_stream = static_cast<hipStream_t>(stream);
    if (!c || !chi || !score) FAIL(PP_ERR_INVALID, "pp_score: null argument");
    if (!c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_score: plan was created without network weights");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    StepParams sp;
    fill_step(&sp, t, 0.f, c->plan->annealed_temp);
    pp_status st;
    flag_nonfinite(c, chi, 4, 4, s);
    if ((st = pp_launch_node_embed(c, chi, sp, s)) != PP_OK) return st;
    if ((st = run_network(c, s, 0, PP_NU_SCORE, nullptr, PP_MODE_ODE, nullptr, &sp, nullptr)) != PP_OK) return st;
    PP_HIP_CHECK(hipMemcpyAsync(score, c->score, (size_t)c->N * 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (hV) PP_HIP_CHECK(hipMemcpyAsync(hV, c->hV, (size_t)c->N * 128 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return PP_OK;
}

extern "C" pp_status pp_sample(pp_ctx *c, float *chi, const float *schedule, int n_schedule, int mode,
                               const float *sde_noise, void *stream) {
    if (c) c->last_stream = static_cast<hipStream_t>(stream);
    if (!c || !chi || !schedule) FAIL(PP_ERR_INVALID, "pp_sample: null argument");
    if (!c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_sample: plan was created without network weights");
    if (n_schedule < 2) FAIL(PP_ERR_INVALID, "pp_sample: schedule needs at least 2 times");
    if (n_schedule - 1 > c->max_steps) FAIL(PP_ERR_UNSUPPORTED, "pp_sample: more than 2^20 steps");
    if (mode != PP_MODE_ODE && mode != PP_MODE_SDE) FAIL(PP_ERR_INVALID, "pp_sample: unknown mode");
    if (mode == PP_MODE_SDE && !sde_noise) FAIL(PP_ERR_INVALID, "pp_sample: sde mode needs the per-step noise tensor");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    const int nsteps = n_schedule - 1;
    // per-step scalars are kernel arguments: nothing is staged, nothing waits for the stream
    std::vector<StepParams> steps((size_t)nsteps);
    for (int j = 0; j < nsteps; j++) fill_step(&steps[j], schedule[j], schedule[j] - schedule[j + 1], c->plan->annealed_temp);
    pp_status st;
    // (A hipGraph replay of the loop was measured and dropped: with no stray event records in the stream the kernel
    // trace shows back-to-back dispatches, and capture + replay was 2 % slower than plain launches.)
    flag_nonfinite(c, chi, 4, 4, s);
    if ((st = pp_launch_node_embed(c, chi, steps[0], s)) != PP_OK) return st;
    static const bool dbg = PP_GETENV("PP_DEBUG") != nullptr;
    const auto h0 = std::chrono::steady_clock::now();
    for (int j = 0; j < nsteps; j++) {
        if ((st = run_network(c, s, j, PP_NU_STEP, chi, mode, sde_noise, &steps[j], j + 1 < nsteps ? &steps[j + 1] : nullptr)) != PP_OK) return st;
    }
    if (dbg) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        fprintf(stderr, "[pp] pp_sample: host enqueue of %d steps took %.0f us (%.1f us per step)\n", nsteps, us, us / nsteps);
    }
    return PP_OK;
}

extern "C" pp_status pp_atom14(pp_ctx *c, const float *chi, float *xyz, void *stream) {
    if (c) c->last_stream = static_cast<hipStream_t>(stream);
    if (!c || !chi || !xyz) FAIL(PP_ERR_INVALID, "pp_atom14: null argument");
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    return pp_launch_atom14(c, chi, xyz, static_cast<hipStream_t>(stream));
}

extern "C" pp_status pp_clash(pp_ctx *c, const float *chi, float *per_res, float *dchi, void *stream) {
    if (c) c->last_stream = static_cast<hipStream_t>(stream);
    if (!c || !chi || !per_res) FAIL(PP_ERR_INVALID, "pp_clash: null argument");
    if (!c->plan->clash_params_set) FAIL(PP_ERR_INVALID, "pp_clash: call pp_plan_set_clash_params first");
    if (!c->b.atom_mask || !c->b.residue_index) FAIL(PP_ERR_INVALID, "pp_clash: batch lacks atom_mask / residue_index");
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    pp_status st;
    if ((st = pp_launch_atom14(c, chi, c->xyz, s)) != PP_OK) return st;
    return pp_launch_clash(c, c->xyz, per_res, dchi, s);
}

extern "C" pp_status pp_proximal(pp_ctx *c, const float *chi, float lamda, int num_steps, float *chi_traj,
                                 float *chi_last, float *losses, void *stream) {
    if (c) c->last_stream = static_cast<hipStream_t>(stream);
    if (!c || !chi || !losses) FAIL(PP_ERR_INVALID, "pp_proximal: null argument");
    if (c->B != 1) FAIL(PP_ERR_INVALID, "pp_proximal: batch.num_proteins must be 1 (optimize.py:27); optimise the complexes of a packed or padded batch one by one");
    if (num_steps < 1) FAIL(PP_ERR_INVALID, "pp_proximal: num_steps must be >= 1");
    if (!c->plan->clash_params_set) FAIL(PP_ERR_INVALID, "pp_proximal: call pp_plan_set_clash_params first");
    if (!c->b.atom_mask || !c->b.residue_index) FAIL(PP_ERR_INVALID, "pp_proximal: batch lacks atom_mask / residue_index");
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    return pp_launch_proximal(c, chi, lamda, num_steps, chi_traj, chi_last, losses, static_cast<hipStream_t>(stream));
}

// Measurement aid (bench.py): average duration of one launch of a hot kernel, timed with HIP events on
// `stream` around `iters` back-to-back launches.  which: 0 = node message, 1 = edge update (layer 1 weights).
// The ctx must have been through pp_score / pp_sample so that its state buffers hold real activations.
extern "C" pp_status pp_time_kernel(pp_ctx *c, int which, int iters, float *avg_ms, void *stream) {
    if (!c || !avg_ms || iters < 1) FAIL(PP_ERR_INVALID, "pp_time_kernel: bad argument");
    if (!c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_time_kernel: plan was created without network weights");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    if (PP_GETENV("PP_DEBUG")) {
        int a = 0, b = 0;
        pp_edge_occupancy(&a, &b);
        fprintf(stderr, "[pp] resident workgroups/CU: k_node_message %d, k_edge_update %d\n", a, b);
    }
    hipEvent_t e0, e1;
    PP_HIP_CHECK(hipEventCreate(&e0));
    PP_HIP_CHECK(hipEventCreate(&e1));
    pp_status st = PP_OK;
    for (int w = 0; w < 2 && st == PP_OK; w++) st = which == 0 ? pp_launch_node_message(c, 0, s) : pp_launch_edge_update(c, 1, s);
    PP_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters && st == PP_OK; i++) st = which == 0 ? pp_launch_node_message(c, 0, s) : pp_launch_edge_update(c, 1, s);
    PP_HIP_CHECK(hipEventRecord(e1, s));
    PP_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    PP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)iters;
    return st;
}

// Diagnostics -- libpackppi_hip.dbg.so only (-DPP_DIAG; tools/debug and the per-layer parity test): run single launches or a
// prefix of one network evaluation, read or restore an internal buffer.  Declared in include/packppi_hip.h under PP_DIAG.
#ifdef PP_DIAG
extern "C" pp_status pp_debug_edge(pp_ctx *c, int layer, void *stream) {
    if (!c || !c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_debug_edge: bad ctx");
    return pp_launch_edge_update(c, layer, static_cast<hipStream_t>(stream));
}
extern "C" pp_status pp_debug_nm(pp_ctx *c, int layer, void *stream) {
    if (!c || !c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_debug_nm: bad ctx");
    return pp_launch_node_message(c, layer, static_cast<hipStream_t>(stream));
}
extern "C" pp_status pp_debug_set_hE(pp_ctx *c, const float *src, size_t n) {
    if (!c || !src) FAIL(PP_ERR_INVALID, "pp_debug_set_hE: null");
    PP_HIP_CHECK(hipMemcpy(c->hE, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return PP_OK;
}
// which: 0 h_E, 1 S, 2 msum, 3 h_E0, 4 Z_em, 5 h_V
extern "C" pp_status pp_debug_buffer(pp_ctx *c, int which, float *dst, size_t n) {
    if (!c || !dst) FAIL(PP_ERR_INVALID, "pp_debug_buffer: null");
    const float *src = which == 0 ? c->hE : which == 1 ? c->S : which == 2 ? c->msum : which == 3 ? c->hE0 : which == 4 ? c->Zem : c->hV;
    PP_HIP_CHECK(hipDeviceSynchronize());
    PP_HIP_CHECK(hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return PP_OK;
}
// The first `n_launches` kernel launches of one network evaluation at time t (pp_score's schedule: node embedding, then
// NM0, NU0, EU0(+NM1), NU1, EU1(+NM2), NU2): after 1 + 2 the layer-0 h_V is in place, after 1 + 3 the layer-0 h_E, after
// 1 + 4 / 1 + 5 the same of layer 1, after 1 + 6 the final h_V (mpnn.py:47-62, layers.py:119-148).
extern "C" pp_status pp_debug_score_prefix(pp_ctx *c, const float *chi, float t, int n_launches, void *stream) {
    if (!c || !chi || !c->plan->has_network) FAIL(PP_ERR_INVALID, "pp_debug_score_prefix: bad argument");
    if (!pp_edge_fused()) FAIL(PP_ERR_UNSUPPORTED, "pp_debug_score_prefix: needs the fused edge update");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    StepParams sp;
    fill_step(&sp, t, 0.f, c->plan->annealed_temp);
    pp_status st = PP_OK;
    int k = 0;
    auto more = [&]() { return st == PP_OK && k++ < n_launches; };
    if (more()) st = pp_launch_node_embed(c, chi, sp, s);
    if (more()) st = pp_launch_node_message(c, 0, s);
    for (int l = 0; l < 3; l++) {
        if (more()) st = pp_launch_node_update(c, l, l < 2 ? PP_NU_MID : PP_NU_SCORE, nullptr, 0, PP_MODE_ODE, nullptr, l < 2 ? nullptr : &sp, nullptr, s);
        if (l < 2 && more()) st = pp_launch_edge_update(c, l, s);
    }
    return st;
}
#endif

// Measurement aid (bench.py): in-situ duration of one hot kernel.  After pp_profile_kernel(ctx, which) every launch of
// that kernel inside pp_score / pp_sample carries a start / stop HIP event pair (hipExtLaunchKernelGGL: the
// dispatch's own begin and end timestamps);
// pp_profile_read synchronises, sums the pair intervals, reports (total ms, launches) and switches profiling off.
extern "C" pp_status pp_profile_kernel(pp_ctx *c, int which) {
    if (!c || which < 0 || which > 3)
        FAIL(PP_ERR_INVALID, "pp_profile_kernel: which must be 0 (node message), 1 (edge update), 2 (node update) or 3 (the Adam-step launch of pp_proximal: clash + gradient + step + reconstruction)");
    c->prof_which = which;
    c->prof_n = 0;
    return PP_OK;
}

extern "C" pp_status pp_profile_read(pp_ctx *c, float *total_ms, int *launches) {
    if (!c || !total_ms || !launches) FAIL(PP_ERR_INVALID, "pp_profile_read: null argument");
    PP_HIP_CHECK(hipSetDevice(c->plan->device));
    double tot = 0.0;
    size_t pairs = c->prof_n / 2;
    if (pairs) PP_HIP_CHECK(hipEventSynchronize(c->prof_ev[2 * pairs - 1]));
    for (size_t i = 0; i < pairs; i++) {
        float ms = 0.f;
        PP_HIP_CHECK(hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms = (float)tot;
    *launches = (int)pairs;
    c->prof_which = -1;
    c->prof_n = 0;
    return PP_OK;
}


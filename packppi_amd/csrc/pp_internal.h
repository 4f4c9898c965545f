// Internal declarations shared by the HIP translation units of libpackppi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <vector>
#include <mutex>
#include <stdint.h>
#include <string>

#include "../../include/packppi_hip.h"

// ---- product vs laboratory ------------------------------------------------------------------------------------------
// Five measurement switches are left in the sources, all in pp_edge_f16.hip, all behind -DPP_LAB and a TAGGED library:
//   -DPP_X_TS [-DPP_X_TS_FINE=k0]  phase / stage clocks of wave 0 into the dbg buffer (tools/debug/phase_*.py, stage_times.py);
//   -DPP_X_NOWLOAD, -DPP_X_E_NOMFMA  ablations (WRONG RESULTS): no weight fetches / no matrix instructions -- what the launch time
//                                    is made of (profiles/r02_edge_ablation.txt);
//   -DPP_X_NOSAT                   no sticky-flag bookkeeping (A/B of its price, tools/debug/ab_sat.sh).
// The switches of experiments that are settled (node-update stop points and ablations, wave priorities, conversion variants,
// the clash ablations ...) were removed from the sources in round 5 -- the device code of all four libraries is the same to the
// instruction (profiles/r05_switch_cleanup.txt); the builds behind the older records are in the history (up to 5cba218).
//   -DPP_LAB   is required by every PP_X_* switch and by the tuning parameters below;
//   -DPP_DIAG  (implied by PP_LAB; the libpackppi_hip.dbg.so build) compiles the pp_debug_* exports and the getenv switches
//              of the launchers (PP_NU_SPLIT, PP_EDGE_MIX, PP_EDGE_R, PP_NM_R, PP_NM_MIX, PP_NODE_F16, PP_DEBUG): same kernels, same
//              results, forced launch shapes -- the default / .f32 / .chk libraries have neither.
// packppi_amd/build.py refuses these flags for the three product libraries and lib.load() refuses a library whose flag
// stamp is not one of the known sets.
#if defined(PP_LAB) && !defined(PP_DIAG)
#define PP_DIAG
#endif
#if !defined(PP_LAB) &&                                                                                                       \
    (defined(PP_X_TS) || defined(PP_X_TS_FINE) || defined(PP_X_NOWLOAD) || defined(PP_X_E_NOMFMA) || defined(PP_X_NOSAT) ||      \
     defined(PP_WDEPTH) || defined(PP_WDEPTH_R1) || defined(PP_NXB_R1) || defined(PP_WGS2) || defined(PP_WGS) ||                 \
     defined(PP_NM_WGS2) ||                                                                                                  \
     defined(PP_LDS_PAD) || defined(PP_NO_FUSE_NM) || defined(PP_NU_DEPTH))
#error "PP_X_* / tuning switches compile laboratory variants (most give wrong results): add -DPP_LAB and build a TAGGED library (python -m packppi_amd.build --tag NAME -DPP_LAB -DPP_X_...)"
#endif
#ifdef PP_DIAG
#define PP_GETENV(name) getenv(name)
#else
#define PP_GETENV(name) ((const char *)nullptr)      /* product builds read no environment switches */
#endif

#define PP_H 128          // hidden width
#define PP_MSG_IN 456     // message MLP input width
#define PP_NPTS 8         // invariant points per node
#define PP_PROX_CHUNK 64   // proximal steps whose loss terms are parked before one reduction
#define PP_CL_CAP 96       // static clash-partner candidates kept per (residue, wave) -- k_clash_cand / k_clash<true>

#include "pp_weights.h"      // LayerOff / WeightOff / pp_weight_offsets(): offsets into the concatenated weight buffer

// Transposed ([in][out]) copies used by the node-level (VALU) kernels, per layer.
struct LayerT {
    const float *pts_node_wT, *pts_edge_wT;          // [128][24]
    const float *nm_A_T, *nm_C_T, *em_A_T, *em_C_T;  // W_in[:, 0:128]^T, W_in[:, 256:384]^T  -> [128][128]
    const float *nm_out_T;                           // [128][128]
    const float *nd_in_T;                            // [128][512]
    const float *nd_out_T;                           // [512][128]
    const float *nm_stream, *em_stream;              // MFMA weight chunks packed in consumption order (pp_edge.hip)
    const float *em_params;                          // edge kernel small vectors, one block
    const float *nu_stream;                          // k_node_update: split-f16 weight slots, [wave][slot] (pp_api.hip put_node_stream)
    const float *nu_params;                          // k_node_update: small per-layer vectors, one block (NU_P_* offsets)
};

// ---- k_node_update (pp_node.hip): packed weight stream and parameter block ---------------------------------------
// One slot = the A operand of one (16-feature tile, 32-deep k-step) of v_mfma_f32_16x16x32_f16 as split f16:
// [hi | lo * 2^11][lane 64][8 halves] = 2 KB; a wave's slots are contiguous in the order it consumes them.
// Layers 0 and 1 use the MID list (next = this layer's edge message + the next layer's node message), layer 2 the LAST
// list (decoder + the layer-0 node-message inputs of the next step).
#define PP_NU_WAVES 8
#define PP_NU_SLOT_FLOATS 512
#define PP_NU_SLOTS_MID 56     // W_out 4 | FFN-in 16 | FFN-out 16 | PAe 4 | PCe 4 | PAn 4 | PCn 4 | pts 4 (waves 0-2)
#define PP_NU_SLOTS_LAST 59    // W_out 4 | FFN-in 16 | FFN-out 16 | D1 4 (waves 0-3) | D2 4, D3 1, D4 1 (wave 0) | embed 1 | PAn 4 | PCn 4 | pts 4 (waves 0-1)
#define PP_NU_LO_SCALE 2048.0f
enum {
    NU_P_OUTB = 0, NU_P_G0 = 128, NU_P_B0 = 256, NU_P_FIB = 384, NU_P_FOB = 896, NU_P_G1 = 1024, NU_P_B1 = 1152,
    // MID
    NU_P_PAE_B = 1280, NU_P_PAN_B = 1408, NU_P_PTS_B = 1536, NU_P_MID_TOTAL = 1600,
    // LAST
    NU_P_DB0 = 1280, NU_P_DB1 = 1344, NU_P_DB2 = 1376, NU_P_DB3 = 1392, NU_P_PAN0_B = 1408, NU_P_PTS0_B = 1536,
    NU_P_EMB_B = 1568, NU_P_EMB_G = 1696, NU_P_EMB_BETA = 1824, NU_P_LAST_TOTAL = 1952
};

struct ArenaSlot {
    void *p;
    size_t bytes;
    hipStream_t stream;      // work on this stream may still be using the memory
};
struct pp_plan {
    int device;
    std::vector<ArenaSlot> arena_pool;   // workspaces of destroyed contexts, reused by the next pp_complex_prepare
    std::mutex pool_mutex;
    bool has_network;         // false: geometry-only plan (atom14 / clash / proximal)
    int knn_ties;             // PP_KNN_TIES_*: what the neighbour search does on exactly equal distances
    float annealed_temp;      // sample_cfg.annealed_temp (the T of schedule.py:205-208), default 3
    int rebalanced_chains;    // split-f16 build: ReLU chains whose layers were rescaled by a power of two (pp_api.hip rebalance_relu_chains)
    float *ln_scale = nullptr;       // split-f16 build: [5][128] power-of-two operand scales behind small LayerNorm gains (pp_rebalance.h), or null
    int ln_scaled_features = 0;      // ... how many of them differ from 1
    float *w;                 // device copy of all weights, original layouts
    WeightOff off;
    float *wT;                // device arena of transposed copies
    LayerT lt[3];
    const float *node_emb_T;  // [51][128]
    const float *edge_emb_T;  // [468][128]
    const float *d0_in_T, *d0_out_T, *d2_in_T, *d2_out_T;   // [128][64] [64][32] [32][16] [16][4]
    const float *static_stream;                             // k_edge_static: W_B chunks of layer 0 (node, edge message)
    const float *embed_stream;                              // split-f16 build: the 400 RBF columns of the edge embedding, 13 chunks
    // chemistry tables (device)
    float *default_frames;    // [21][8][16]
    int32_t *atom14_to_group; // [21][14]
    float *atom14_mask;       // [21][14]
    float *lit_positions;     // [21][14][3]
    float *between_radius;    // [21][14]
    float *side_extent;       // [21] upper bound of |side-chain atom - CA| over all chi, per residue type (pp_api.hip clash_extents)
    float *bounds_lower, *bounds_upper;   // [21][14][14]
    float clash_tol;
    bool clash_params_set;
};

// Per-step scalars of the reverse process, computed on the host (schedule.py:198-235) and handed to the kernels as arguments.
struct StepParams {
    float temb[16];     // sinusoidal embedding of t
    float c_ode;        // 0.5 * g^2 * dt
    float w;            // annealed weight
    float c_drift;      // g^2 * dt        (sde)
    float c_diff;       // g * sqrt(dt)    (sde)
    float pad[12];
};

struct pp_ctx {
    pp_plan *plan;
    pp_batch b;               // caller-owned device pointers
    int B, L, K, N;           // N = B*L nodes (packed context: B = number of complexes, L = longest, N = sum of lengths)
    bool packed = false;      // rows of the complexes back to back, no padding rows (pp_complex_prepare_packed)
    int2 *seg;                // [N] (first row, length) of the complex each row belongs to
    // static per complex
    int32_t *eidx;            // [N][K] global node index of each neighbour
    float *mask_att;          // [N][32] mask_i*mask_j (0 for slots >= K)
    float *frames;            // [N][12]  R (row-major 9) | t (3)
    float *bbpos;             // [N][5][3] N CA C O CB*
    float *hE0;               // [N][K][128]
    float *Znm, *Zem;         // [N][K][128] W_B h_E0 of layer 0's node / edge message (k_edge_static)
    // per-evaluation state
    float *hE;                // [N][K][128]
    float *hV;                // [N][128]
    float *hV_alt;            // [N][128] the split node-update launches write the new h_V here, then the two pointers swap
    float *S;                 // [N][128]  masked mean of the node-message hidden layer
    float *msum;              // [N]
    float *ptsN, *PAn, *PCn;  // node-message inputs [N][48] [N][128] [N][128]
    float *ptsE, *PAe, *PCe;  // edge-message inputs
    float *score;             // [N][4]
    float *chi_tmp;           // [N][4]
    void *arena = nullptr;          // ONE device allocation behind every workspace pointer above (one hipMalloc / hipFree per ctx)
    size_t arena_bytes = 0;
    hipStream_t last_stream = nullptr;   // the stream of the most recent call on this context (arena hand-over)
    int max_steps;
    // clash / proximal workspaces
    float *xyz;               // [N][14][3]
    float *rec;               // [N][16][4] packed per-residue records for k_clash (positions + radii, CA + reach, ids)
    float *axes;              // [N][4][6]  chi-frame x-axis (3) | origin (3)
    float *rec2, *axes2;      // the other buffers of the proximal loop: step t reads rec / axes of step t and writes those of step t + 1
    float *brad;              // [N] bounding radius around CA
    float *per_res;           // [N]
    float *dchi;              // [N][4]
    float *px, *pm, *pv, *pz, *pxeff;   // proximal: param, Adam moments, anchor, effective chi  [N][4]
    uint8_t *pmask;           // [N]
    int32_t *cand;            // [N][4][PP_CL_CAP] proximal: static clash-partner candidates of every (residue, wave of its workgroup)
    int32_t *cand_cnt;        // [N][4] their number, -1 = more than PP_CL_CAP (that wave scans all partners as before)
    float *prox_part;         // [PP_PROX_CHUNK][ceil(N / 16)] per-block loss terms of the proximal steps
    float *scal;              // small scalar scratch
    unsigned *sat;            // sticky word: bit 0 = an edge kernel, bit 1 = a node kernel clamped a hidden activation at 65504
    // in-situ kernel timing (pp_profile_kernel): every launch of one hot kernel carries a start / stop event pair
    // (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, what rocprofv3's kernel trace reports)
    int prof_which = -1;       // -1 off, 0 node message, 1 edge update, 2 node update
    bool prof_armed = false;   // the next PP_LAUNCH on this ctx is one of the profiled kernel
    std::vector<hipEvent_t> prof_ev;
    size_t prof_n = 0;         // events used (2 per launch)
};

void pp_set_error(const std::string &msg);
#define PP_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            pp_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return PP_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ---- f16 operand range check (-DPP_CHECK_RANGE: the libpackppi_hip.chk.so build; packppi_amd/rangecheck.py) -----------------
// The split-f16 kernels saturate hidden activations at the f16 maximum (65504) and assume every other operand is far below it.
// That holds by orders of magnitude for the seeded fixtures; a trained checkpoint is checked, not trusted: in this build every
// fp32 value that is about to be split into f16 operands is compared with the limit and counted (per translation unit; atomics:
// this build is for checking, not for timing).  The default build compiles none of it.
#ifdef PP_CHECK_RANGE
#define PP_RANGE_COUNTER static __device__ unsigned int g_range_hits;
#define PP_RANGE(x)                                                                                    \
    {                                                                                                  \
        const float rx_ = (x);                                                                         \
        if (!(__builtin_fabsf(rx_) < 65504.f)) atomicAdd(&g_range_hits, 1u);                           \
    }
#define PP_RANGE_READER(name)                                                                          \
    unsigned int name(int reset) {                                                                     \
        unsigned int v = 0, z = 0;                                                                     \
        (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_range_hits), sizeof(v));                            \
        if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_range_hits), &z, sizeof(z));                   \
        return v;                                                                                      \
    }
#else
#define PP_RANGE_COUNTER
#define PP_RANGE(x)
#define PP_RANGE_READER(name) \
    unsigned int name(int) { return 0; }
#endif
// ---- sticky f16 saturation flag of the DEFAULT kernels ------------------------------------------------------------------
// Hidden activations are clamped at the f16 maximum before they are split (one v_med3).  Every build remembers when that
// happened: the clamped halves are folded into a per-lane running maximum (v_pk_max_u16 on the packed pair: non-negative
// f16 bit patterns order like integers), and a lane that saw 0x7BFF ORs a bit into the
// context's sticky word when the kernel ends.  pp_ctx_saturated reads it.  NaN activations are not flagged (v_med3 turns
// them into a finite value first): they cannot arise from finite inputs here, see relu_sat in pp_node.hip.
__device__ __forceinline__ bool pp_sat_hit(unsigned packed_max) {
    return (packed_max & 0xffffu) >= 0x7bffu || (packed_max >> 16) >= 0x7bffu;
}
unsigned int pp_edge_range_hits(int reset);
unsigned int pp_node_range_hits(int reset);

// next (start, stop) event pair of an armed profiling run; false when profiling is off for this launch
bool pp_prof_take(pp_ctx *c, hipEvent_t *e0, hipEvent_t *e1);
#define PP_LAUNCH(c, kernel, grid, block, shmem, s, ...)                                              \
    do {                                                                                               \
        hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                       \
        if ((c)->prof_armed && pp_prof_take((c), &e0_, &e1_))                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, s, e0_, e1_, 0, __VA_ARGS__);            \
        else                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, shmem, s, __VA_ARGS__);                            \
    } while (0)

// ---- inter-residue dihedral, rounded like the reference's ATen CPU ops (encoder.py:164-174) ------------------------------
// sgn * arccos(n1 . n2) has no clamp: where four atoms are coplanar (every j == i edge; on ideal-geometry backbones also a few
// trans pairs with |psi| within 3e-4 of pi) the argument is 1 or -1 up to rounding, and one ulp decides between NaN -> 0,
// exactly pi and pi - 5e-4: the feature is rounding noise of size pi.  The only way to agree with the reference there is to
// round as it does.  Measured against torch 2.10 CPU (tools/debug/aten_rounding_probe.py, 76 864 dihedrals of four C5
// complexes, argument and sign bit-equal):  torch.cross  c = fma(a1, b2, -(a2 * b1))  (second product rounded first);
// torch.norm  sqrt(fma(z, z, fma(y, y, x * x)));  (a * b).sum(-1)  ((p0 + p1) + p2) on separately rounded products; IEEE
// division.  arccos itself differs from ATen's by at most an ulp, which is smooth.
#ifdef __HIPCC__
__device__ __forceinline__ void pp_cross_t(const float *a, const float *b, float *o) {
#pragma clang fp contract(off)
    const float q0 = a[2] * b[1], q1 = a[0] * b[2], q2 = a[1] * b[0];
    o[0] = __builtin_fmaf(a[1], b[2], -q0);
    o[1] = __builtin_fmaf(a[2], b[0], -q1);
    o[2] = __builtin_fmaf(a[0], b[1], -q2);
}
__device__ __forceinline__ float pp_dot_t(const float *a, const float *b) {
#pragma clang fp contract(off)
    const float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
    return (p0 + p1) + p2;
}
__device__ __forceinline__ void pp_unit_t(float *v) {
#pragma clang fp contract(off)
    const float x2 = v[0] * v[0];
    const float n = sqrtf(__builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], x2)));
    for (int k = 0; k < 3; k++) {
        const float q = v[k] / n;
        v[k] = (q != q) ? 0.f : q;           // nan_to_num of 0 / 0; +-inf cannot occur for finite input
    }
}
__device__ __forceinline__ float pp_pair_dihedral_t(const float *p0, const float *p1, const float *p2, const float *p3) {
#pragma clang fp contract(off)
    float u0[3], u1[3], u2[3], n1[3], n2[3], c12[3];
    for (int k = 0; k < 3; k++) { u0[k] = p2[k] - p1[k]; u1[k] = p0[k] - p1[k]; u2[k] = p3[k] - p2[k]; }
    pp_cross_t(u0, u1, n1); pp_unit_t(n1);
    pp_cross_t(u0, u2, n2); pp_unit_t(n2);
    pp_cross_t(u1, u2, c12);
    const float sg = pp_dot_t(c12, u0);
    const float sgn = (sg > 0.f) ? 1.f : ((sg < 0.f) ? -1.f : 0.f);
    const float ang = sgn * acosf(pp_dot_t(n1, n2));
    return (ang != ang) ? 0.f : ang;
}
#endif

// ---- launchers implemented in the kernel translation units ----------------------------------
pp_status pp_launch_prepare(pp_ctx *c, hipStream_t s, const int64_t *E_idx = nullptr);   // E_idx: given neighbour lists instead of the kNN search
pp_status pp_launch_node_embed(pp_ctx *c, const float *chi, const StepParams &sp, hipStream_t s);
// cur: this step's scalars (layer 2 inside sampling); next: the next step, if its node embedding is to follow (else null)
pp_status pp_launch_node_update(pp_ctx *c, int layer, int last_mode, float *chi, int step, int mode,
                                const float *noise, const StepParams *cur, const StepParams *next, hipStream_t s);
pp_status pp_launch_edge_static(pp_ctx *c, hipStream_t s);
#ifdef PP_EDGE_F16
pp_status pp_launch_edge_embed_f16(pp_ctx *c, hipStream_t s);   // pp_edge_f16.hip: MFMA form of k_edge_embed
#endif
pp_status pp_launch_node_message(pp_ctx *c, int layer, hipStream_t s);
bool pp_edge_fused();            // does pp_launch_edge_update also compute the next layer's node message?
pp_status pp_launch_edge_update(pp_ctx *c, int layer, hipStream_t s);   // + node message of layer + 1
pp_status pp_launch_atom14(pp_ctx *c, const float *chi, float *xyz, hipStream_t s);
pp_status pp_launch_clash(pp_ctx *c, const float *xyz, float *per_res, float *dchi, hipStream_t s, bool use_candidates = false);
pp_status pp_launch_proximal(pp_ctx *c, const float *chi, float lamda, int nsteps, float *traj,
                             float *chi_last, float *losses, hipStream_t s);

void pp_edge_occupancy(int *node_msg, int *edge_upd);

// last_mode values for pp_launch_node_update
#define PP_NU_MID 0        // layers 0,1: update + edge-message inputs + next layer's node-message inputs
#define PP_NU_SCORE 1      // layer 2, single evaluation: update + decoder
#define PP_NU_STEP 2       // layer 2 inside sampling: update + decoder + reverse step + next embed

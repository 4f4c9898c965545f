/*
 * packppi_hip.h -- C ABI of the MI355X (gfx950) side-chain sampling path.
 *
 * The reference (Jackz915/PackPPI) is pure Python/PyTorch and has no FFI; the entry points
 * below are what a binding for its hot path binds.  Each one names the
 * reference call it replaces (paths relative to the upstream repo root).
 *
 * Conventions
 *   - every `const float*` / `int64_t*` argument that is not marked HOST is a caller-owned,
 *     contiguous DEVICE pointer (e.g. torch.Tensor.data_ptr() of a ROCm tensor);
 *   - all floating data is fp32, indices are int64 (as in the reference batch object);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it and the
 *     library never synchronises the device, except where stated;
 *   - no exceptions cross the ABI: calls return pp_status, pp_last_error() gives the message
 *     of the calling thread's most recent failure;
 *   - a pp_ctx is not thread-safe; different ctxs may be driven from different threads.
 */
#ifndef PACKPPI_HIP_H
#define PACKPPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pp_status {
    PP_OK = 0,
    PP_ERR_INVALID = 1,     /* bad argument / shape */
    PP_ERR_HIP = 2,         /* a HIP runtime call failed */
    PP_ERR_UNSUPPORTED = 3, /* valid request this build does not implement */
    PP_ERR_NO_DEVICE = 4    /* no gfx950 device visible */
} pp_status;

typedef struct pp_plan pp_plan; /* weights + chemistry tables resident on one GPU */
typedef struct pp_ctx pp_ctx;   /* one batch of complexes: cached graph, frames, workspaces */

#define PP_N_WEIGHTS 1439172u /* fp32 parameters of the score network (112 tensors) */
#define PP_HIDDEN 128
#define PP_TOP_K 32
#define PP_MODE_ODE 0
#define PP_MODE_SDE 1
/* What the neighbour search does where two CA distances of a row are exactly equal (pp_plan_set_knn_ties). */
#define PP_KNN_TIES_LOWER_INDEX 0 /* the residue with the lower index first                                          */
#define PP_KNN_TIES_ATEN_CPU 1    /* default: the choice AND order of torch.topk on CPU (the reference CPU path)       */
#define PP_KNN_TIES_ATEN_MEMBER 2 /* that choice only where it decides membership (rank K == rank K+1), else lower index */

/* Residue-chemistry tables (HOST pointers; values of src/utils/residue_constants.py:595-677). */
typedef struct pp_tables {
    const float *default_frames;   /* [21,8,4,4] restype_rigid_group_default_frame           */
    const int32_t *atom14_to_group;/* [21,14]    restype_atom14_to_rigid_group               */
    const float *atom14_mask;      /* [21,14]    restype_atom14_mask                         */
    const float *lit_positions;    /* [21,14,3]  restype_atom14_rigid_group_positions        */
    const float *between_radius;   /* [21,14]    vdW radius table of clash.py:263-287        */
} pp_tables;

/* The reference `batch` object (complex_dataset.py:123-139, complex_datamodule.py:205-224). */
typedef struct pp_batch {
    int32_t B, L;
    const float *X;                /* [B,L,14,3] */
    const float *atom_mask;        /* [B,L,14]   */
    const int64_t *residue_type;   /* [B,L]      */
    const float *residue_mask;     /* [B,L]      */
    const int64_t *residue_index;  /* [B,L]      */
    const int64_t *chain_indices;  /* [B,L]      */
    const float *BB_D;             /* [B,L,3]    */
    const float *BB_D_sincos;      /* [B,L,3,2]  */
    const float *SC_D;             /* [B,L,4]    */
    const float *SC_D_mask;        /* [B,L,4]    */
    const uint8_t *chi_1pi_periodic_mask; /* [B,L,4] bool */
    const uint8_t *chi_2pi_periodic_mask; /* [B,L,4] bool */
} pp_batch;

int pp_version(void);
const char *pp_last_error(void);

/* Build stamp: "<sources>-<flags>", two 16-digit hex prefixes of the SHA-256 of (a) every .hip / .h file of packppi_amd/csrc
 * plus this header, (b) the compiler flags, as computed by packppi_amd/build.py when the library was compiled.  The Python
 * binding refuses a library whose <sources> part differs from the sources on disk (a stale prebuilt .so). */
const char *pp_build_id(void);

/* Replaces TDiffusionModule.__init__ + load_from_checkpoint (TorsionalDiffusion.py:22-82,
 * eval_diffusion.py:29-41).  `weights` is a HOST buffer holding the 112 state_dict tensors
 * concatenated in the order of packppi_amd/weights.py::weight_spec(), nn.Linear [out,in].
 * weights == NULL gives a geometry-only plan (pp_atom14 / pp_clash / pp_proximal, the
 * checkpoint-free path of src/proximal_optimize.py); its batches need only X, residue_type,
 * BB_D and, for the clash calls, atom_mask and residue_index. */
pp_status pp_plan_create(const float *weights, size_t n_weights, const pp_tables *tables, int device,
                         pp_plan **plan);
void pp_plan_destroy(pp_plan *plan);

/* Replaces rc.make_atom14_dists_bounds(tol, vtf) as consumed by find_sc_violations
 * (clash.py:299-308): `lower`/`upper` are HOST [21,14,14] tables for these parameters.  The tables are
 * caller-owned and may be freed on return, so this call waits for `stream` and copies them synchronously
 * (the one blocking call besides the measurement aids; it runs once per parameter set, not per batch). */
pp_status pp_plan_set_clash_params(pp_plan *plan, float overlap_tolerance, const float *lower,
                                   const float *upper, void *stream);

/* ProteinEncoder._dist takes torch.topk(D_adjust, K, largest=False) (encoder.py:105-118), which leaves the choice between
 * exactly equal distances to the implementation -- and on ideal-geometry or 0.001-Angstrom-grid coordinates equal distances
 * do occur (about one row in five of the synthetic complexes has an order tie, one complex in thirty a membership tie at rank
 * K / K+1, which changes the graph).  PP_KNN_TIES_ATEN_CPU (the default) reproduces the reference CPU path: rows that hold a
 * tie among their K+1 smallest values are redone on the device with the selection ATen's CPU topk makes (std::nth_element +
 * std::sort, or std::partial_sort when 64 K <= L, of libstdc++ over (value, index) pairs with a value-only comparator;
 * csrc/pp_topk_aten.h).  Applies to contexts prepared afterwards. */
pp_status pp_plan_set_knn_ties(pp_plan *plan, int mode);

/* Replaces sample_cfg.annealed_temp as TDiffusionModule.__init__ hands it to both SO2VESchedule instances
 * (TorsionalDiffusion.py:70-75; configs/model/sample_cfg/Sampling.yaml:4, default 3): the T of the annealed weight
 * w = T / (alpha + (1 - alpha) T) in SO2VESchedule.step (schedule.py:216-217); 0 = no annealing, w = 1 (the reference's
 * `if self.annealed_temp`: a Sampling.yaml with `annealed_temp: 0` or `null`).  Applies to later pp_score / pp_sample calls. */
pp_status pp_plan_set_annealed_temp(pp_plan *plan, float annealed_temp);

/* No reference counterpart.  The split-f16 build rebalances every ReLU chain of the edge-level MLPs by powers of two when the
 * plan is created (W1, b1 times s; W2 divided by s: the same function, hidden activations of size O(1) whatever the checkpoint's
 * split of scale between consecutive layers; csrc/pp_api.hip rebalance_relu_chains): this returns how many of the 15 chains
 * were rescaled (0 for weights whose layer row norms lie within [1/8, 8], e.g. the seeded fixtures; always 0 in
 * libpackppi_hip.f32.so, which needs no such care), -1 for a null plan. */
int pp_plan_rebalanced_chains(const pp_plan *plan);
/* HOST helper, no device call: `out` [n_weights] = the weight vector as pp_plan_create packs it (the rebalanced one in the
 * split-f16 build, a copy in libpackppi_hip.f32.so), `chains` (may be NULL) = how many chains were rescaled.  For tests: the
 * rebalanced network must be the original function (tests/test_host.py runs both through the CPU oracle). */
pp_status pp_rebalance_weights_host(const float *weights, size_t n_weights, float *out, int *chains);

/* No reference counterpart.  Three LayerNorm outputs become f16 operands of the edge-level dense layers: h_E0 (encoder.norm_edges,
 * encoder.py:243-244), the h_E a layer writes and x1 (norm[3], norm[2]: layers.py:128-146).  Where a feature's gain and bias are
 * both far from 1 (a checkpoint that keeps the scale in the consuming weights) the split-f16 build multiplies that operand feature
 * by a power of two before the split and divides the consuming weight column by it when the plan is created -- exact, and only then
 * are the kernel instances with that multiply launched (csrc/pp_rebalance.h ln_operand_scales).  Returns how many of the 5 x 128
 * operand features carry a scale other than 1 (0 for the seeded fixtures; always 0 in libpackppi_hip.f32.so), -1 for a null plan. */
int pp_plan_ln_scaled_features(const pp_plan *plan);
/* HOST helper, no device call: `out` [5][128] = the operand scales pp_plan_create would choose for these weights, in the order
 * h_E0 | h_E after layer 0 | after layer 1 | x1 of layer 0 | of layer 1; `n_scaled` (may be NULL) = how many differ from 1. */
pp_status pp_ln_operand_scales_host(const float *weights, size_t n_weights, float *out, int *n_scaled);

/* HOST helper, no device call: idx_out[0..k-1] = torch.topk(values[0..n-1], k, largest=False) indices as ATen's CPU kernel
 * returns them -- the same code the neighbour search runs on the device for rows with ties.  Returns PP_OK / PP_ERR_INVALID. */
pp_status pp_topk_aten_host(const float *values, int n, int k, int32_t *idx_out);

/* Replaces the timestep-invariant part of ProteinEncoder.forward (encoder.py:198-246):
 * kNN graph, 468-d edge features, edge embedding + LayerNorm, backbone frames.  The batch
 * pointers must stay valid for the lifetime of the ctx.
 * A ctx keeps all its device workspaces in one allocation that pp_ctx_destroy hands back to the plan
 * (a pool of up to four) instead of freeing it: creating a ctx per batch costs no hipMalloc/hipFree.
 * pp_ctx_destroy does not wait for work already enqueued on the ctx's stream; the next ctx that takes
 * the workspace over either runs on that same stream (ordered behind it) or synchronises with it first.
 * pp_plan_destroy frees the pool: destroy the plan's contexts, and let their work finish, before it. */
pp_status pp_complex_prepare(pp_plan *plan, const pp_batch *batch, void *stream, pp_ctx **ctx);
void pp_ctx_destroy(pp_ctx *ctx);

/* The same for complexes of different lengths WITHOUT the padding rows of collate_fn
 * (complex_datamodule.py:196-226): the batch tensors are [1, sum of lengths, ...] with the complexes'
 * rows back to back, `seg_offsets` (DEVICE, int32 [n_seg + 1], caller-owned like the batch) gives the first
 * row of every complex and the total, min_len / max_len their shortest and longest length.  Neighbour
 * search, clash partners and E_idx numbering stay inside each complex; every other stage is per row or per
 * edge.  Results equal those of each complex prepared on its own (K = min(32, length): complexes shorter
 * than 32 residues cannot be mixed with longer ones -> PP_ERR_UNSUPPORTED).  pp_proximal needs one complex
 * per ctx.  CONTRACT: the device table must describe the batch (seg_offsets[0] = 0, seg_offsets[n_seg] = the number of rows,
 * every length within [min_len, max_len]): the library sizes its launches from min_len / max_len and cannot read the table
 * back without stalling the stream; a table that disagrees is clamped where that is cheap, but the behaviour is undefined. */
pp_status pp_complex_prepare_packed(pp_plan *plan, const pp_batch *batch, const int32_t *seg_offsets, int n_seg,
                                    int min_len, int max_len, void *stream, pp_ctx **ctx);

/* Inspection (tests): copy out E_idx [B,L,K] and the embedded edges h_E0 [B,L,K,128]; K = min(32,L). */
pp_status pp_ctx_get_graph(pp_ctx *ctx, int64_t *E_idx, float *hE0, void *stream);

/* Replace the ctx's neighbour lists by the caller's E_idx [B,L,K] (per-complex numbering, as
 * ProteinEncoder._dist returns them, encoder.py:105-118) and redo the edge embedding: for callers who want
 * another tie convention than the two the search offers (e.g. the lists of the reference's GPU path).
 * Validates the indices (one read-back: the call waits for `stream`). */
pp_status pp_ctx_set_graph(pp_ctx *ctx, const int64_t *E_idx, void *stream);

/* Replaces TDiffusionModule.network(batch, SC_D_noised, t) (TorsionalDiffusion.py:90-109) for a
 * timestep shared by all residues.  score [B,L,4]; hV [B,L,128] may be NULL. */
pp_status pp_score(pp_ctx *ctx, const float *chi, float t, float *score, float *hV, void *stream);

/* Replaces the loop of TDiffusionModule.sampling (TorsionalDiffusion.py:259-280):
 * chi [B,L,4] holds the initial noised angles on entry and the sample on exit.
 * `schedule` is a HOST array of n_schedule times (n_schedule-1 network evaluations); it is read before the
 * call returns, and the per-step scalars derived from it travel as kernel arguments (no staging copy, no wait).
 * mode PP_MODE_SDE needs `sde_noise` [n_schedule-1, 2, B*L, 4] (the two N(0,1) draws of
 * schedule.py:225 per step, 1pi schedule first); NULL is allowed for PP_MODE_ODE. */
pp_status pp_sample(pp_ctx *ctx, float *chi, const float *schedule, int n_schedule, int mode,
                    const float *sde_noise, void *stream);

/* Replaces get_atom14_coords(X, S, BB_D, SC_D) (components/__init__.py:76-120). xyz [B,L,14,3]. */
pp_status pp_atom14(pp_ctx *ctx, const float *chi, float *xyz, void *stream);

/* Replaces compute_residue_clash (clash.py:335-365) with the parameters last given to
 * pp_plan_set_clash_params.  per_res [B,L]; dchi, if not NULL, receives
 * d(mean over all B*L residues of per_res)/dchi [B,L,4] (the autograd of optimize.py:62-63). */
pp_status pp_clash(pp_ctx *ctx, const float *chi, float *per_res, float *dchi, void *stream);

/* Replaces proximal_optimizer(batch, SC_D, vtf, tol, lamda, num_steps) (optimize.py:21-73),
 * B must be 1.  chi_traj, if not NULL, receives the per-step angles [num_steps,1,L,4];
 * chi_last [1,L,4] the last of them; losses (DEVICE, [num_steps]) the pre-step loss values. */
pp_status pp_proximal(pp_ctx *ctx, const float *chi, float lamda, int num_steps, float *chi_traj,
                      float *chi_last, float *losses, void *stream);

/* Measurement aid, no reference counterpart: average duration (ms) of one launch of a hot kernel
 * (which: 0 = node-message kernel, 1 = edge-update kernel), timed with HIP events on `stream`
 * around `iters` launches.  Synchronises the stream. */
pp_status pp_time_kernel(pp_ctx *ctx, int which, int iters, float *avg_ms, void *stream);

/* Measurement aid, no reference counterpart: in-situ duration of a hot kernel.  After
 * pp_profile_kernel(ctx, which) (0 node message, 1 edge update, 2 node update; 3 the one launch per Adam step
 * inside pp_proximal: clash loss + gradient, the step, the reconstruction at the new angles) every launch of that
 * kernel made by pp_score / pp_sample / pp_proximal carries a start / stop HIP event pair on the launch stream
 * (hipExtLaunchKernelGGL: the dispatch's own begin and end, the interval rocprofv3's kernel trace reports);
 * pp_profile_read waits for the last one, returns the summed intervals (ms) and the number of
 * launches, and switches profiling off again. */
pp_status pp_profile_kernel(pp_ctx *ctx, int which);
pp_status pp_profile_read(pp_ctx *ctx, float *total_ms, int *launches);

/* Which edge kernels the library was built with (no reference counterpart; bench.py prices the roofline with it):
 * 1 = split-f16 MFMA with fp32-equivalent accuracy (default, csrc/pp_edge_f16.hip), 0 = exact-fp32 MFMA
 * (PACKPPI_EDGE=f32, csrc/pp_edge.hip). */
int pp_edge_variant(void);

/* f16 operand range check (no reference counterpart).  The default kernels run the dense layers on two-way f16 splits:
 * hidden activations saturate at 65504 and every other operand is assumed to be far below that.  A library built with
 * -DPP_CHECK_RANGE (libpackppi_hip.chk.so; `python -m packppi_amd.rangecheck`) counts every fp32 value at or beyond the
 * limit (or not finite) that its kernels were about to split: pp_range_check waits for the device, returns the events since
 * the last reset and optionally resets.  In any other build pp_has_range_check() is 0 and pp_range_check returns
 * PP_ERR_UNSUPPORTED.  (pp_plan_create rejects weights that are not finite or outside the f16 range in every build.) */
int pp_has_range_check(void);
pp_status pp_range_check(unsigned long long *events, int reset);
/* The same count by kernel family: the edge-level kernels and the node-level kernels.  libpackppi_hip.f32.so replaces BOTH
 * families by fp32 ones (fp32-MFMA edge kernels, csrc/pp_edge.hip; fp32 VALU node update, k_node_update_valu in
 * csrc/pp_node.hip): it has no f16 operand anywhere and is the build for a checkpoint that raises either count. */
pp_status pp_range_check_parts(unsigned long long *edge_events, unsigned long long *node_events, int reset);

/* Sticky saturation flag, every build (no reference counterpart).  The default kernels clamp hidden activations at the f16
 * maximum before splitting them; a context remembers that it happened: *flags bit 0 = in an edge-level kernel, bit 1 = in a
 * node-level kernel, 0 = never since pp_complex_prepare.  The call waits for `stream`.  A set bit means results of this
 * context are not fp32-equivalent for this checkpoint (run python -m packppi_amd.rangecheck for the details).
 * Bit 2 (value 4): a NaN or infinity ENTERED with the caller's tensors (backbone coordinates of an unmasked row at
 * pp_complex_prepare, an angle at pp_score / pp_sample).  The reference propagates it to its output (layers.py:22-33 has no
 * clamp); these kernels' clamps turn it into finite numbers that mean nothing -- the flag is how a caller learns of it. */
pp_status pp_ctx_saturated(pp_ctx *ctx, int *flags, void *stream);

/* Diagnostics -- ONLY in libpackppi_hip.dbg.so (built with -DPP_DIAG; same kernels and results as the default library):
 * single launches, internal-buffer copies and a prefix of one network evaluation, for tools/debug/ and the per-layer parity
 * test (tests/test_hip_layers.py).  Not part of the drop-in boundary; the default, .f32 and .chk libraries do not export them
 * and read no environment switches. */
#ifdef PP_DIAG
pp_status pp_debug_edge(pp_ctx *ctx, int layer, void *stream);     /* one edge-update launch (+ next node message) */
pp_status pp_debug_nm(pp_ctx *ctx, int layer, void *stream);       /* one node-message launch */
pp_status pp_debug_set_hE(pp_ctx *ctx, const float *src, size_t n);
/* which: 0 h_E [N,K,128], 1 S [N,128], 2 msum [N], 3 h_E0, 4 Z_em, 5 h_V [N,128]; waits for the device */
pp_status pp_debug_buffer(pp_ctx *ctx, int which, float *dst, size_t n);
/* the first n_launches launches of pp_score's schedule (embed, NM0, NU0, EU0, NU1, EU1, NU2) */
pp_status pp_debug_score_prefix(pp_ctx *ctx, const float *chi, float t, int n_launches, void *stream);
void pp_debug_set_dbg(float *p);          /* stamp buffer of the -DPP_LAB -DPP_X_TS builds */
void pp_debug_set_lds_pad(int bytes);     /* occupancy experiments */
void pp_debug_set_edge_R(int R);          /* residues per edge workgroup: 1, 2, 0 = automatic */
#endif

#ifdef __cplusplus
}
#endif
#endif /* PACKPPI_HIP_H */

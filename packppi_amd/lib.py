"""ctypes binding of ``csrc/libpackppi_hip.so`` (the C ABI declared in ``include/packppi_hip.h``).

There is no CPU fallback: if the library is missing or a call fails, a ``RuntimeError`` is
raised (the reference CLIs only catch/re-raise ``RuntimeError``, eval_diffusion.py:61-64).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import constants as rc
from .weights import check_state_dict

_LIB_PATH = os.environ.get("PACKPPI_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpackppi_hip.so")     # PACKPPI_LIB: A/B runs of build variants
_lib = None

SYMBOLS = ("pp_version", "pp_last_error", "pp_build_id", "pp_plan_set_knn_ties", "pp_plan_set_annealed_temp", "pp_plan_rebalanced_chains", "pp_rebalance_weights_host", "pp_plan_ln_scaled_features", "pp_ln_operand_scales_host", "pp_topk_aten_host", "pp_plan_create", "pp_plan_destroy", "pp_plan_set_clash_params",
           "pp_complex_prepare", "pp_complex_prepare_packed", "pp_ctx_destroy", "pp_ctx_get_graph", "pp_ctx_set_graph", "pp_score", "pp_sample", "pp_atom14",
           "pp_clash", "pp_proximal", "pp_time_kernel", "pp_profile_kernel", "pp_profile_read", "pp_edge_variant", "pp_has_range_check", "pp_range_check", "pp_range_check_parts", "pp_ctx_saturated")


KNN_TIES = {"lower_index": 0, "aten_cpu": 1, "aten_member": 2}


class PPTables(C.Structure):
    _fields_ = [("default_frames", C.c_void_p), ("atom14_to_group", C.c_void_p), ("atom14_mask", C.c_void_p),
                ("lit_positions", C.c_void_p), ("between_radius", C.c_void_p)]


class PPBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("X", C.c_void_p), ("atom_mask", C.c_void_p),
                ("residue_type", C.c_void_p), ("residue_mask", C.c_void_p), ("residue_index", C.c_void_p),
                ("chain_indices", C.c_void_p), ("BB_D", C.c_void_p), ("BB_D_sincos", C.c_void_p),
                ("SC_D", C.c_void_p), ("SC_D_mask", C.c_void_p), ("chi_1pi_periodic_mask", C.c_void_p),
                ("chi_2pi_periodic_mask", C.c_void_p)]


def load():
    """Load the shared library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} is missing: build it with `python -m packppi_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    vp, f, i = C.c_void_p, C.c_float, C.c_int
    try:
        lib.pp_build_id.restype = C.c_char_p
    except AttributeError:
        raise RuntimeError(f"{_LIB_PATH} is stale: it predates the build stamp (no pp_build_id); rebuild with "
                           "`python -m packppi_amd.build` (or `python __graft_entry__.py`)") from None
    # a prebuilt library must come from the sources on disk (content hash, packppi_amd/build.py), and from one of the four
    # product flag sets: a tagged laboratory build (-DPP_LAB -DPP_X_...: timing variants, most with wrong results) only loads
    # with PACKPPI_ALLOW_LAB_LIBRARY=1
    if not os.environ.get("PACKPPI_SKIP_BUILD_CHECK"):
        from .build import product_flag_stamps, source_hash
        have, _, have_flags = lib.pp_build_id().decode().partition("-")
        want = source_hash()
        if have != want:
            raise RuntimeError(f"{_LIB_PATH} is stale: built from sources {have}, csrc/ is now {want}; rebuild with "
                               "`python -m packppi_amd.build` (or `python __graft_entry__.py`)")
        if have_flags not in product_flag_stamps() and not os.environ.get("PACKPPI_ALLOW_LAB_LIBRARY"):
            raise RuntimeError(f"{_LIB_PATH} was built with flags (stamp {have_flags}) that are none of the product sets "
                               f"{sorted(product_flag_stamps().values())}: a laboratory variant; set PACKPPI_ALLOW_LAB_LIBRARY=1 "
                               "to load it for a measurement")
    lib.pp_plan_set_knn_ties.argtypes = [vp, i]
    lib.pp_plan_set_annealed_temp.argtypes = [vp, f]
    lib.pp_plan_rebalanced_chains.argtypes = [vp]
    lib.pp_rebalance_weights_host.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_int)]
    lib.pp_plan_rebalanced_chains.restype = C.c_int
    lib.pp_plan_ln_scaled_features.argtypes = [vp]
    lib.pp_plan_ln_scaled_features.restype = C.c_int
    lib.pp_ln_operand_scales_host.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_int)]
    lib.pp_topk_aten_host.argtypes = [vp, i, i, vp]
    lib.pp_version.restype = C.c_int
    lib.pp_last_error.restype = C.c_char_p
    lib.pp_plan_create.argtypes = [vp, C.c_size_t, C.POINTER(PPTables), i, C.POINTER(vp)]
    lib.pp_plan_destroy.argtypes = [vp]
    lib.pp_plan_destroy.restype = None
    lib.pp_plan_set_clash_params.argtypes = [vp, f, vp, vp, vp]
    lib.pp_complex_prepare.argtypes = [vp, C.POINTER(PPBatch), vp, C.POINTER(vp)]
    lib.pp_complex_prepare_packed.argtypes = [vp, C.POINTER(PPBatch), vp, i, i, i, vp, C.POINTER(vp)]
    lib.pp_ctx_destroy.argtypes = [vp]
    lib.pp_ctx_destroy.restype = None
    lib.pp_ctx_get_graph.argtypes = [vp, vp, vp, vp]
    lib.pp_ctx_set_graph.argtypes = [vp, vp, vp]
    lib.pp_score.argtypes = [vp, vp, f, vp, vp, vp]
    lib.pp_sample.argtypes = [vp, vp, vp, i, i, vp, vp]
    lib.pp_atom14.argtypes = [vp, vp, vp, vp]
    lib.pp_clash.argtypes = [vp, vp, vp, vp, vp]
    lib.pp_proximal.argtypes = [vp, vp, f, i, vp, vp, vp, vp]
    lib.pp_time_kernel.argtypes = [vp, i, i, C.POINTER(C.c_float), vp]
    lib.pp_profile_kernel.argtypes = [vp, i]
    lib.pp_profile_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.pp_edge_variant.argtypes = []
    lib.pp_has_range_check.argtypes = []
    lib.pp_has_range_check.restype = C.c_int
    lib.pp_range_check.argtypes = [C.POINTER(C.c_ulonglong), i]
    lib.pp_range_check_parts.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), i]
    lib.pp_ctx_saturated.argtypes = [vp, C.POINTER(C.c_int), vp]
    lib.pp_edge_variant.restype = C.c_int
    _lib = lib
    return lib


def _check(status, what):
    if status != 0:
        msg = load().pp_last_error()
        raise RuntimeError(f"{what} failed (pp_status {status}): {msg.decode() if msg else ''}")


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def ln_operand_scales(state_dict):
    """([5, 128] operand scales pp_plan_create would choose behind the edge-level LayerNorms -- rows: h_E0, h_E after layer 0,
    after layer 1, x1 of layer 0, of layer 1 --, how many differ from 1): host only, no device call."""
    lib = load()
    sd = check_state_dict(state_dict)
    flat = np.ascontiguousarray(torch.cat([v.reshape(-1) for v in sd.values()]).numpy(), dtype=np.float32)
    out = np.empty((5, 128), np.float32)
    n = C.c_int(0)
    _check(lib.pp_ln_operand_scales_host(flat.ctypes.data, flat.size, out.ctypes.data, C.byref(n)), "pp_ln_operand_scales_host")
    return torch.from_numpy(out), int(n.value)


def rebalanced_state_dict(state_dict):
    """(state_dict as pp_plan_create packs it, number of ReLU chains rescaled): host only, no device call.  The split-f16 build
    rescales the ReLU chains of the edge-level MLPs by powers of two (same function, hidden activations O(1))."""
    lib = load()
    sd = check_state_dict(state_dict)
    flat = np.ascontiguousarray(torch.cat([v.reshape(-1) for v in sd.values()]).numpy(), dtype=np.float32)
    out = np.empty_like(flat)
    n = C.c_int(0)
    _check(lib.pp_rebalance_weights_host(flat.ctypes.data, flat.size, out.ctypes.data, C.byref(n)), "pp_rebalance_weights_host")
    res, at = {}, 0
    for k, v in sd.items():
        res[k] = torch.from_numpy(out[at:at + v.numel()].reshape(tuple(v.shape)).copy())
        at += v.numel()
    return res, int(n.value)


class Plan:
    """Device-resident weights + chemistry tables (one per GPU)."""

    def __init__(self, state_dict, device):
        """``state_dict=None`` builds a geometry-only plan (atom14 / clash / proximal)."""
        lib = load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("packppi_amd runs on an MI355X HIP device only (got device '%s')" % device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.has_network = state_dict is not None
        if self.has_network:
            sd = check_state_dict(state_dict)
            flat = np.ascontiguousarray(torch.cat([v.reshape(-1) for v in sd.values()]).numpy(), dtype=np.float32)
            wptr, wn = flat.ctypes.data, flat.size
        else:
            wptr, wn = None, 0
        self._keep = [
            np.ascontiguousarray(rc.default_frames, np.float32), np.ascontiguousarray(rc.atom14_to_group, np.int32),
            np.ascontiguousarray(rc.atom14_mask, np.float32), np.ascontiguousarray(rc.lit_positions, np.float32),
            np.ascontiguousarray(rc.between_radius, np.float32)]
        tab = PPTables(*[a.ctypes.data for a in self._keep])
        h = C.c_void_p()
        _check(lib.pp_plan_create(wptr, wn, C.byref(tab), self.device.index, C.byref(h)), "pp_plan_create")
        self.handle = h
        self._clash_params = None
        self.knn_ties = "aten_cpu"
        if os.environ.get("PACKPPI_KNN_TIES"):
            self.set_knn_ties(os.environ["PACKPPI_KNN_TIES"])

    def set_knn_ties(self, mode):
        """What the neighbour search does on exactly equal CA distances: "aten_cpu" (default: the reference CPU path's
        torch.topk choice and order), "aten_member" (that choice only where membership depends on it) or "lower_index"."""
        if mode not in KNN_TIES:
            raise ValueError(f"knn ties mode must be one of {sorted(KNN_TIES)}")
        _check(load().pp_plan_set_knn_ties(self.handle, KNN_TIES[mode]), "pp_plan_set_knn_ties")
        self.knn_ties = mode

    def rebalanced_chains(self) -> int:
        """How many ReLU chains of the edge-level MLPs the split-f16 build rescaled by a power of two at plan creation."""
        return int(load().pp_plan_rebalanced_chains(self.handle))

    def ln_scaled_features(self) -> int:
        """How many of the 5 x 128 LayerNorm-output operand features of the edge kernels carry a power-of-two scale (split-f16 build;
        0 for weights whose LayerNorm gains and biases are of ordinary size)."""
        return int(load().pp_plan_ln_scaled_features(self.handle))

    def set_annealed_temp(self, T):
        """sample_cfg.annealed_temp (Sampling.yaml:4): the T of the annealed score weight in SO2VESchedule.step."""
        _check(load().pp_plan_set_annealed_temp(self.handle, float(T)), "pp_plan_set_annealed_temp")

    def set_clash_params(self, vtf, tol):
        key = (float(vtf), float(tol))
        if self._clash_params == key:
            return
        lo, up = rc.make_atom14_dists_bounds(overlap_tolerance=float(tol), bond_length_tolerance_factor=float(vtf))
        lo, up = np.ascontiguousarray(lo, np.float32), np.ascontiguousarray(up, np.float32)
        _check(load().pp_plan_set_clash_params(self.handle, float(tol), lo.ctypes.data, up.ctypes.data,
                                               _stream(self.device)), "pp_plan_set_clash_params")
        self._clash_params = key

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _lib is not None:
            _lib.pp_plan_destroy(h)
            self.handle = None


_BATCH_SPEC = (("X", torch.float32), ("atom_mask", torch.float32), ("residue_type", torch.int64),
               ("residue_mask", torch.float32), ("residue_index", torch.int64), ("chain_indices", torch.int64),
               ("BB_D", torch.float32), ("BB_D_sincos", torch.float32), ("SC_D", torch.float32),
               ("SC_D_mask", torch.float32), ("chi_1pi_periodic_mask", torch.bool),
               ("chi_2pi_periodic_mask", torch.bool))


def _get(batch, key):
    return batch.get(key) if hasattr(batch, "get") else getattr(batch, key, None)


class BatchKey:
    """What a cached ``Context`` is valid for: the batch tensors themselves (held, so that their storage cannot be handed to
    another batch while the key lives) and their version counters -- an in-place edit of ``batch.X``, ``residue_mask``,
    ``residue_index`` ... bumps ``_version`` and the graph, frames and edge embedding are rebuilt, as the reference recomputes
    them on every call (encoder.py:198-246).  A tensor the context had to copy (dtype / layout) is covered the same way: the
    key watches the caller's tensor, not the copy."""

    KEYS = tuple(k for k, _ in _BATCH_SPEC) + ("seg_offsets",)

    def __init__(self, batch):
        self.tensors = [t if isinstance(t, torch.Tensor) else None for t in (_get(batch, k) for k in self.KEYS)]
        self.meta = self._meta(self.tensors)

    @staticmethod
    def _meta(ts):
        # tensors made under torch.inference_mode() (Lightning's test / predict loops) track no version counter: reading
        # `_version` raises.  They get no version in the key -- and never match (below)
        return [None if t is None else ((None if t.is_inference() else t._version), tuple(t.shape), t.dtype) for t in ts]

    def matches(self, batch) -> bool:
        ts = [t if isinstance(t, torch.Tensor) else None for t in (_get(batch, k) for k in self.KEYS)]
        if any(t is not None and t.is_inference() for t in ts):
            # an in-place edit of an inference tensor cannot be seen (no version counter, same data_ptr): the context is
            # rebuilt on every call, as the reference recomputes graph and embedding on every call (encoder.py:198-246)
            return False
        return all(a is b for a, b in zip(ts, self.tensors)) and self._meta(ts) == self.meta


class Context:
    """One batch of complexes on one GPU: cached kNN graph, edge embedding, frames, workspaces."""

    def __init__(self, plan: Plan, batch):
        lib = load()
        self.plan = plan
        dev = plan.device
        B, L = batch["residue_type"].shape
        self.B, self.L, self.K = int(B), int(L), min(32, int(L))
        self._t = {}
        for key, dt in _BATCH_SPEC:
            t = _get(batch, key)
            if t is None:
                if plan.has_network or key in ("X", "residue_type", "BB_D"):
                    raise RuntimeError(f"batch.{key} is missing")
                self._t[key] = None
                continue
            if t.device.type != "cuda" or (t.device.index is not None and t.device.index != dev.index):
                raise RuntimeError(f"batch.{key} is on {t.device}, expected {dev} (call batch.to(device) first)")
            self._t[key] = t.to(dt).contiguous()
        pb = PPBatch(self.B, self.L, *[(self._t[k].data_ptr() if self._t[k] is not None else None)
                                       for k, _ in _BATCH_SPEC])
        h = C.c_void_p()
        seg = _get(batch, "seg_offsets")
        if seg is None:
            _check(lib.pp_complex_prepare(plan.handle, C.byref(pb), _stream(dev), C.byref(h)), "pp_complex_prepare")
        else:
            # ragged batch without padding rows (batch.pack): [1, sum of lengths, ...] + the complexes' first rows; the host
            # copy of the table that pack() carries along saves the read-back
            host = _get(batch, "seg_offsets_host")
            offs = [int(x) for x in (host if host is not None else seg.tolist())]
            # the device table is what the kernels index with, the host copy is what K and the launch sizes come from: they
            # must be the same table (a batch sliced or edited after pack() would only be clamped on the device)
            if len(offs) != seg.numel():
                raise RuntimeError(f"seg_offsets has {seg.numel()} entries, seg_offsets_host {len(offs)}")
            if host is not None and seg.device.type == "cpu" and [int(x) for x in seg.tolist()] != offs:
                raise RuntimeError("seg_offsets and seg_offsets_host disagree")
            lens = [b - a for a, b in zip(offs[:-1], offs[1:])]
            if self.B != 1 or len(offs) < 2 or offs[-1] != self.L or offs[0] != 0 or min(lens) < 1:
                raise RuntimeError("seg_offsets does not describe this batch")
            self._t["seg_offsets"] = seg.to(device=dev, dtype=torch.int32).contiguous()
            self.K = min(32, min(lens))
            _check(lib.pp_complex_prepare_packed(plan.handle, C.byref(pb), self._t["seg_offsets"].data_ptr(), len(lens),
                                                 min(lens), max(lens), _stream(dev), C.byref(h)),
                   "pp_complex_prepare_packed")
        self.handle = h

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.plan.device)

    def _chi(self, chi):
        chi = chi.to(device=self.plan.device, dtype=torch.float32).reshape(self.B, self.L, 4).contiguous()
        return chi

    def graph(self):
        E = self._new(self.B, self.L, self.K, dtype=torch.int64)
        hE = self._new(self.B, self.L, self.K, 128)
        _check(load().pp_ctx_get_graph(self.handle, _ptr(E), _ptr(hE), _stream(self.plan.device)), "pp_ctx_get_graph")
        return E, hE

    def set_graph(self, E_idx):
        """Use the caller's neighbour lists [B, L, K] (per-complex numbering) instead of the built-in search."""
        E = E_idx.to(device=self.plan.device, dtype=torch.int64).reshape(self.B, self.L, self.K).contiguous()
        _check(load().pp_ctx_set_graph(self.handle, _ptr(E), _stream(self.plan.device)), "pp_ctx_set_graph")

    def score(self, chi, t: float):
        chi = self._chi(chi)
        score, hV = self._new(self.B, self.L, 4), self._new(self.B, self.L, 128)
        _check(load().pp_score(self.handle, _ptr(chi), float(t), _ptr(score), _ptr(hV), _stream(self.plan.device)),
               "pp_score")
        return score, hV

    def sample(self, chi, schedule, mode="ode", sde_noise=None):
        chi = self._chi(chi).clone()
        sched = np.ascontiguousarray(torch.as_tensor(schedule, dtype=torch.float32).cpu().numpy())
        if mode not in ("ode", "sde"):
            raise NotImplementedError(mode)
        nz = None
        if mode == "sde":
            if sde_noise is None:
                raise RuntimeError("sde sampling needs the per-step noise tensor")
            nz = sde_noise.to(device=self.plan.device, dtype=torch.float32).contiguous()
            assert nz.numel() == (len(sched) - 1) * 2 * self.B * self.L * 4
        _check(load().pp_sample(self.handle, _ptr(chi), sched.ctypes.data, int(len(sched)), 0 if mode == "ode" else 1,
                                _ptr(nz), _stream(self.plan.device)), "pp_sample")
        return chi

    def atom14(self, chi):
        chi = self._chi(chi)
        xyz = self._new(self.B, self.L, 14, 3)
        _check(load().pp_atom14(self.handle, _ptr(chi), _ptr(xyz), _stream(self.plan.device)), "pp_atom14")
        return xyz

    def clash(self, chi, vtf=12.0, tol=0.5, need_grad=False):
        self.plan.set_clash_params(vtf, tol)
        chi = self._chi(chi)
        per_res = self._new(self.B, self.L)
        dchi = self._new(self.B, self.L, 4) if need_grad else None
        _check(load().pp_clash(self.handle, _ptr(chi), _ptr(per_res), _ptr(dchi), _stream(self.plan.device)), "pp_clash")
        return (per_res, dchi) if need_grad else per_res

    def proximal(self, chi, vtf, tol, lamda, num_steps, want_traj=True):
        self.plan.set_clash_params(vtf, tol)
        chi = self._chi(chi)
        traj = self._new(num_steps, self.B, self.L, 4) if want_traj else None
        last = self._new(self.B, self.L, 4)
        losses = self._new(num_steps)
        _check(load().pp_proximal(self.handle, _ptr(chi), float(lamda), int(num_steps), _ptr(traj), _ptr(last),
                                  _ptr(losses), _stream(self.plan.device)), "pp_proximal")
        return traj, last, losses

    def saturated(self) -> int:
        """Sticky flag word of this context: 0 = clean; bit 0 / bit 1 = a hidden activation was clamped at 65504 in an edge-level /
        node-level kernel; bit 2 (value 4) = a NaN or infinity entered with the caller's tensors (the reference would return NaN, the
        kernels' clamps return finite numbers that mean nothing).  Waits for the stream."""
        v = C.c_int(0)
        _check(load().pp_ctx_saturated(self.handle, C.byref(v), _stream(self.plan.device)), "pp_ctx_saturated")
        return int(v.value)

    def time_kernel(self, which: int, iters: int = 20) -> float:
        """Average ms per launch of the node-message (0) / edge-update (1) kernel, HIP events on the current stream."""
        ms = C.c_float(0.0)
        _check(load().pp_time_kernel(self.handle, int(which), int(iters), C.byref(ms), _stream(self.plan.device)),
               "pp_time_kernel")
        return float(ms.value)

    def profile_kernel(self, which: int):
        """Bracket every later launch of kernel `which` (0 node message, 1 edge update, 2 node update; inside proximal(): 3 clash
        loss + gradient, 4 Adam step + reconstruction) with HIP events."""
        _check(load().pp_profile_kernel(self.handle, int(which)), "pp_profile_kernel")

    def profile_read(self):
        """(average ms per launch, launches) since profile_kernel(); switches profiling off."""
        ms, n = C.c_float(0.0), C.c_int(0)
        _check(load().pp_profile_read(self.handle, C.byref(ms), C.byref(n)), "pp_profile_read")
        return (float(ms.value) / max(n.value, 1), int(n.value))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _lib is not None:
            _lib.pp_ctx_destroy(h)
            self.handle = None

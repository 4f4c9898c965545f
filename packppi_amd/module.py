"""``TDiffusionModule``: the reference Lightning module's inference surface on the HIP path.

Mirrors ``src/models/TorsionalDiffusion.py``: ``network`` (:90-109), ``add_sc_noise`` (:111-124),
``sampling`` (:254-298), ``compute_rmsd`` (:300-309), ``analyze_samples`` (:311-341), the plain
``schedule`` attribute (:77) and ``load_from_checkpoint(..., strict=False)`` as the CLIs use it
(eval_diffusion.py:29-41).  Training hooks are out of scope.
"""
import math
import pickle
import zipfile
from types import SimpleNamespace
from typing import Any, Dict, Optional

import numpy as np
import torch

from .functional import proximal_optimizer
from .lib import BatchKey, Context, Plan

SAMPLE_DEFAULTS = dict(eval_epochs=1, sample_during_training=True, annealed_temp=3, mode="ode", use_proximal=True,
                       violation_tolerance_factor=12., clash_overlap_tolerance=0.5, lamda=1., num_steps=50)

SIGMA_MIN, SIGMA_MAX = 0.01 * math.pi, math.pi


def _cfg(obj, defaults):
    out = dict(defaults)
    if obj is not None:
        src = obj if isinstance(obj, dict) else vars(obj)
        out.update({k: v for k, v in src.items() if k in defaults})
    return SimpleNamespace(**out)


class _Stub(dict):
    """Inert stand-in for any class the unpickler cannot import (OmegaConf nodes, Lightning callbacks, ...)."""

    def __init__(self, *a, **k):
        dict.__init__(self)

    def __setstate__(self, state):
        pass

    def __call__(self, *a, **k):
        return _Stub()

    def append(self, x):
        pass

    def extend(self, xs):
        pass

    def add(self, x):
        pass


class _TolerantUnpickler(pickle.Unpickler):
    """Reads a Lightning ``.ckpt`` without Lightning / OmegaConf: unknown classes become inert stubs."""

    def find_class(self, module, name):
        if module.split(".")[0] in ("torch", "collections", "numpy", "builtins", "_codecs"):
            return super().find_class(module, name)
        return _Stub


class _TolerantPickle:
    Unpickler = _TolerantUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def read_checkpoint_state_dict(path, map_location="cpu") -> Dict[str, torch.Tensor]:
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except Exception:
        ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_TolerantPickle)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


class TDiffusionModule:
    NUM_CHI_ANGLES = 4
    eps = 1e-6

    def __init__(self, state_dict: Dict[str, torch.Tensor], sample_cfg: Any = None, encoder_cfg: Any = None,
                 model_cfg: Any = None, device="cuda", knn_ties: Optional[str] = None, **kwargs):
        """``encoder_cfg`` / ``model_cfg`` / ``sample_cfg``: what eval_diffusion.py:33-40 instantiates from the YAML files (dicts
        or namespaces; ``config.load_hot_path_configs`` reads them).  The kernels are compiled for the reference's dimensions:
        any other ``hidden_dim`` / ``top_k`` / ``n_points`` / ``num_rbf`` ... raises a RuntimeError naming the key."""
        from .config import check_compiled_dims
        check_compiled_dims(encoder_cfg, model_cfg)
        self._state_dict = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.hparams = SimpleNamespace(sample_cfg=_cfg(sample_cfg, SAMPLE_DEFAULTS), encoder_cfg=encoder_cfg,
                                       model_cfg=model_cfg)
        if self.hparams.sample_cfg.mode not in ("ode", "sde"):
            raise NotImplementedError(self.hparams.sample_cfg.mode)
        # schedule.py:216-217 tests `if self.annealed_temp`: 0 and None (Sampling.yaml `annealed_temp: null`) mean weight 1
        if not self.hparams.sample_cfg.annealed_temp:
            self.hparams.sample_cfg.annealed_temp = 0.0
        if not np.isfinite(float(self.hparams.sample_cfg.annealed_temp)):
            raise RuntimeError(f"sample_cfg.annealed_temp = {self.hparams.sample_cfg.annealed_temp!r}: must be a finite number, 0 or null")
        self.schedule = torch.linspace(1, 0, 31)              # schedule.py:286-288
        self.device = torch.device("cpu")
        self._plan: Optional[Plan] = None
        self._ctx_key, self._ctx = None, None
        self._knn_ties = knn_ties         # None: the library default (the reference CPU path's torch.topk choice)
        self.to(device)

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=False, **kwargs):
        sd = read_checkpoint_state_dict(checkpoint_path)
        return cls(sd, device=map_location or "cuda", **kwargs)

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("packppi_amd.TDiffusionModule runs on the MI355X HIP device only; use the reference "
                               f"implementation for device '{device}'")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._plan is None or self._plan.device != device:
            self._plan = Plan(self._state_dict, device)
            if self._knn_ties is not None:
                self._plan.set_knn_ties(self._knn_ties)
            self._plan.set_annealed_temp(self.hparams.sample_cfg.annealed_temp)      # Sampling.yaml:4 -> SO2VESchedule
            self._ctx_key, self._ctx = None, None
        self.device = device
        return self

    def eval(self):
        return self

    def state_dict(self):
        return dict(self._state_dict)

    # ---- internals -------------------------------------------------------------------------
    def _key_matches(self, batch) -> bool:
        return self._ctx_key is not None and self._ctx_key.matches(batch)

    def _context(self, batch) -> Context:
        if not self._key_matches(batch):
            self._ctx_key, self._ctx = None, None          # drop the old workspace first: the new context takes it over
            self._ctx = Context(self._plan, batch)
            self._ctx_key = BatchKey(batch)
        return self._ctx

    def _geometry_context(self, batch) -> Context:
        """The batch's network context if it is the cached one, else a weight-free one (atom14 needs no graph or edge
        embedding: metrics of many small complexes must not pay a network preparation each)."""
        if self._key_matches(batch):
            return self._ctx
        from .functional import _ctx_for
        return _ctx_for(batch)

    @staticmethod
    def _t_to_sigma(t):
        lo, hi = np.log(SIGMA_MIN), np.log(SIGMA_MAX)
        return torch.exp(lo + (hi - lo) * t)

    # ---- reference surface -------------------------------------------------------------------
    def network(self, batch, SC_D_noised, t):
        """-> (pred_score [B,L,4], h_V [B,L,128]).  ``t`` [B*L] must hold one shared value."""
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
        t0 = float(t[0])
        if t.numel() > 1 and not bool((t == t[0]).all()):
            raise NotImplementedError("per-residue timesteps only occur in training (out of scope)")
        return self._context(batch).score(SC_D_noised, t0)

    @torch.no_grad()
    def add_sc_noise(self, batch, t):
        """Wrapped chi + sigma(t) N(0,1) on the 1pi then 2pi masks; two draws from the global generator of
        the batch's device, in the reference's order.  The second return (the lookup-table score the reference
        computes and sampling discards) is not reproduced: zeros."""
        x = batch.SC_D.reshape(-1, 4)
        sig = self._t_to_sigma(t.to(x.device)).unsqueeze(-1)
        n1 = torch.randn_like(x) * sig
        x = x + n1 * batch.chi_1pi_periodic_mask.reshape(-1, 4)
        n2 = torch.randn_like(x) * sig
        x = x + n2 * batch.chi_2pi_periodic_mask.reshape(-1, 4)
        x = (x + np.pi) % (2 * np.pi) - np.pi
        shape = (batch.num_proteins, -1, 4)
        return x.reshape(shape), torch.zeros_like(x).reshape(shape)

    def sampling(self, batch, use_proximal: bool = False, return_list: bool = False, sde_noise=None):
        cfg = self.hparams.sample_cfg
        t = torch.tensor([1.]).repeat_interleave(batch.max_size * batch.num_proteins).to(self.device)
        SC_D_sample, _ = self.add_sc_noise(batch, t)
        n_steps = len(self.schedule) - 1
        if cfg.mode == "sde" and sde_noise is None:
            # the reference draws torch.normal(size=[B*L, 4], device=...) inside the loop, once per schedule and step, the
            # 1pi schedule first (schedule.py:225, TorsionalDiffusion.py:271-274): the same calls in the same order, so a
            # seed gives the stream it gives the reference on this device (one [n, 2, N, 4] draw would not)
            shape = (batch.num_proteins * batch.max_size, 4)
            sde_noise = torch.stack([torch.stack([torch.normal(mean=0, std=1, size=shape, device=self.device)
                                                  for _ in range(2)]) for _ in range(n_steps)])
        SC_D_sample = self._context(batch).sample(SC_D_sample, self.schedule, cfg.mode, sde_noise)
        if not use_proximal:
            return SC_D_sample
        SC_D_resample_list, loss_list = proximal_optimizer(batch, SC_D_sample, cfg.violation_tolerance_factor,
                                                           cfg.clash_overlap_tolerance, cfg.lamda, cfg.num_steps)
        if return_list:
            return SC_D_sample, SC_D_resample_list, loss_list
        if loss_list[-1] < loss_list[0]:
            return SC_D_resample_list[-1]
        return SC_D_sample

    def sample_from(self, batch, SC_D_init, sde_noise=None):
        """The reverse-diffusion loop of ``sampling`` from given initial noised angles (parity runs inject the reference's
        own draw; the packed multi-complex path injects per-complex draws)."""
        return self._context(batch).sample(SC_D_init, self.schedule, self.hparams.sample_cfg.mode, sde_noise)

    def saturated(self) -> int:
        """Sticky flag word of the context of the last batch (0 = clean; see lib.Context.saturated).  Bits 0 / 1: a hidden
        activation reached the f16 maximum (results are not fp32-equivalent for this checkpoint); bit 2: a NaN / infinity entered
        with the batch or the angles (the reference would return NaN)."""
        return self._ctx.saturated() if self._ctx is not None else 0

    def compute_rmsd(self, true_coords, pred_coords, atom_mask, residue_mask):
        w = atom_mask * residue_mask[..., None]
        return (torch.sum((true_coords - pred_coords) ** 2, dim=-1) * w).sum() / (w + self.eps).sum()

    def analyze_samples(self, batch, SC_D_sample=None):
        true, m, pi1 = batch["SC_D"], batch["SC_D_mask"], batch["chi_1pi_periodic_mask"]
        metric = {}
        for i in range(self.NUM_CHI_ANGLES):
            n = m[..., i].sum()
            n = n if n != 0 else 1
            diff = (SC_D_sample[..., i] - true[..., i]).abs()
            acc = torch.logical_and(diff * 180 / np.pi < 20, diff > 0).float()
            ae = torch.minimum(diff, 2 * np.pi - diff)
            ae = torch.where(pi1[..., i], torch.minimum(ae, np.pi - ae), ae)
            metric[f"chi_{i}_ae_rad"] = ae.sum() / n
            metric[f"chi_{i}_ae_deg"] = (ae * 180 / np.pi).sum() / n
            metric[f"chi_{i}_acc"] = acc.sum() / n
        pred = self._geometry_context(batch).atom14(SC_D_sample)
        metric["atom_rmsd"] = self.compute_rmsd(batch.X, pred, batch.atom_mask, batch.residue_mask)
        return metric

    def get_atom14_coords(self, batch, SC_D):
        return self._geometry_context(batch).atom14(SC_D)

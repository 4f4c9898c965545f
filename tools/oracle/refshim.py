"""Container-only harness that makes the *unmodified* reference importable.

TEST INFRASTRUCTURE, NEVER SHIPPED TO THE GPU BOX AS A DEPENDENCY: nothing under
``packppi_amd/``, ``bench.py`` or ``tests/`` imports this module.  It is used only by
``tools/oracle/make_golden.py`` / ``make_constants.py`` (run by hand in the build
container, where ``/root/reference`` exists) to produce the data fixtures committed
under ``tests/golden/`` and ``packppi_amd/data/``.

It restates no reference arithmetic: it only registers stand-in modules for packages
the image lacks (Lightning, Hydra, torch_scatter, Biopython, ...) so that the
reference's own ``src.models.*`` / ``src.utils.*`` files import as they are
(recipe: SURVEY.md Appendix A).
"""
import argparse
import inspect
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("PACKPPI_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True            # never write __pycache__ into the reference tree
if REF not in sys.path:
    sys.path.insert(0, REF)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


import src  # noqa: E402  (src/__init__.py is empty)

# bare package: skips src/utils/__init__.py:5-16 (Lightning / Hydra / torchtyping imports)
_mod("src.utils").__path__ = [REF + "/src/utils"]


def _na(*a, **k):
    raise NotImplementedError("stub: not available in this container")


_mod("torch_scatter", scatter_add=_na)     # only used by the unused step_correct
_mod("omegaconf", DictConfig=dict)


class MeanMetric(nn.Module):
    def forward(self, x):
        return x

    def reset(self):
        pass


_mod("torchmetrics", MeanMetric=MeanMetric)


def rank_zero_only(fn):
    return fn


class LightningModule(nn.Module):
    """Just enough of the Lightning base class for TDiffusionModule.__init__/sampling."""

    def save_hyperparameters(self, logger=False):
        loc = inspect.currentframe().f_back.f_locals
        hp = {k: v for k, v in loc.items() if k not in ("self", "__class__", "kwargs")}
        hp.update(loc.get("kwargs", {}))
        self.hparams = argparse.Namespace(**hp)

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, *a, **k):
        pass


_pl = _mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object,
           Callback=object, Trainer=object)
_pl.__path__ = []
_mod("pytorch_lightning.utilities", rank_zero_only=rank_zero_only).__path__ = []
_mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=rank_zero_only)
_mod("freesasa")
_mod("Bio").__path__ = []
_mod("Bio.PDB", PDBParser=_na, NeighborSearch=_na, Selection=_na)


class Data(dict):
    """Stand-in for torch_geometric.data.Data (attribute + item access, .apply, .to)."""

    def __init__(self, **kw):
        super().__init__(**kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def apply(self, fn):
        for k in list(self.keys()):
            self[k] = fn(self[k])
        return self

    def to(self, device):
        for k in list(self.keys()):
            if isinstance(self[k], torch.Tensor):
                self[k] = self[k].to(device)
        return self


_tg = _mod("torch_geometric")
_tg.__path__ = []
_mod("torch_geometric.data", Data=Data).__path__ = []

ENC = argparse.Namespace(node_in=35, edge_in=468, node_features=128, edge_features=128,
                         time_embedding_type="sinusoidal", time_embedding_dim=16,
                         num_positional_embeddings=16, num_rbf=16, top_k=32, af2_relpos=True)
MDL = argparse.Namespace(hidden_dim=128, num_mpnn_layers=3, n_points=8, dropout=0.1, act="relu",
                         position_scale=1.0, use_ipmp=True, k_neighbors=32)
SMP = argparse.Namespace(eval_epochs=1, sample_during_training=True, annealed_temp=3, mode="ode",
                         use_proximal=True, violation_tolerance_factor=12.,
                         clash_overlap_tolerance=0.5, lamda=1., num_steps=50)


def build_reference_module(seed=0, mode="ode"):
    """Construct the real TDiffusionModule with seeded xavier weights.

    The 5001x5001 SO(2) score tables are training-only (sampling never reads them,
    SURVEY §0), so SO2Schedule.__init__ is replaced by a no-op that only records PI.
    """
    import numpy as np
    import src.models.components.schedule as sch
    import src.models.TorsionalDiffusion as TD

    def _light_init(self, PI, cache_folder):
        nn.Module.__init__(self)
        self.PI = PI

    sch.SO2Schedule.__init__ = _light_init
    # add_noise() calls self.score(); give it a harmless zero (its result is discarded by sampling)
    sch.SO2Schedule.score = lambda self, x, sigma: np.zeros_like(x)
    torch.manual_seed(seed)
    np.random.seed(seed)
    smp = argparse.Namespace(**vars(SMP))
    smp.mode = mode
    model = TD.TDiffusionModule(optimizer=None, scheduler=None, encoder_cfg=ENC, model_cfg=MDL,
                                sample_cfg=smp).eval()
    # biases/LayerNorm params keep nn defaults; give biases & LN affine non-trivial seeded values
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                if "norm" in name and name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return model

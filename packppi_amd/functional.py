"""Free functions of the reference's hot path, same names and argument meaning, HIP-backed.

    get_atom14_coords(X, S, BB_D, SC_D)                       components/__init__.py:76-120
    compute_residue_clash(batch, SC_D, vtf=12., tol=0.5)       clash.py:335-365
    find_clash_mask(batch, SC_D, vtf, tol)                     optimize.py:5-18
    proximal_optimizer(batch, SC_D, vtf, tol, lamda, steps)    optimize.py:21-73

All tensors must live on the MI355X; there is no CPU path here.
"""
from typing import List, Tuple

import torch

from .batch import Batch
from .lib import BatchKey, Context, Plan

_geometry_plans = {}


def geometry_plan(device) -> Plan:
    """Weight-free plan (chemistry tables only), one per device."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"packppi_amd needs tensors on the HIP device, got {device}")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _geometry_plans:
        _geometry_plans[idx] = Plan(None, torch.device("cuda", idx))
    return _geometry_plans[idx]


def _ctx_for(batch, plan=None) -> Context:
    """Context cached on the batch object (geometry only unless a network plan is given)."""
    plan = plan or geometry_plan(batch["X"].device)
    cache = batch.__dict__.setdefault("_pp_ctx", {}) if hasattr(batch, "__dict__") else {}
    hit = cache.get(id(plan))
    if hit is None or not hit[0].matches(batch):       # tensor identity + version counters (lib.BatchKey), not data_ptr
        cache.clear()
        hit = cache[id(plan)] = (BatchKey(batch), Context(plan, batch))
    return hit[1]


def get_atom14_coords(X, S, BB_D, SC_D):
    lead = X.shape[:-2]
    L = lead[-1]
    b = Batch(X=X.reshape(-1, L, 14, 3), residue_type=S.reshape(-1, L), BB_D=BB_D.reshape(-1, L, 3))
    ctx = Context(geometry_plan(X.device), b)
    return ctx.atom14(SC_D.reshape(-1, L, 4)).reshape(*lead, 14, 3)


def compute_residue_clash(batch, SC_D, violation_tolerance_factor=12., clash_overlap_tolerance=0.5):
    return _ctx_for(batch).clash(SC_D, violation_tolerance_factor, clash_overlap_tolerance)


def find_clash_mask(batch, SC_D, violation_tolerance_factor, clash_overlap_tolerance):
    pr = compute_residue_clash(batch, SC_D, violation_tolerance_factor, clash_overlap_tolerance)
    return (pr > pr.mean()).unsqueeze(-1).expand(-1, -1, 4)


def proximal_optimizer(batch, SC_D, violation_tolerance_factor, clash_overlap_tolerance, lamda,
                       num_steps=50) -> Tuple[List[torch.Tensor], List[float]]:
    assert batch.num_proteins == 1
    traj, _, losses = _ctx_for(batch).proximal(SC_D, violation_tolerance_factor, clash_overlap_tolerance, lamda,
                                               num_steps, want_traj=True)
    loss_list = [float(v) for v in losses.cpu()]          # the one host sync of the whole optimisation
    return [traj[i] for i in range(num_steps)], loss_list

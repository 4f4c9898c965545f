"""Weight contract of the PackPPI-MSC score network (112 tensors, 1 439 172 fp32 params).

Key names and ``nn.Linear`` ``[out, in]`` layouts are those of the reference Lightning
checkpoint's ``state_dict`` (TorsionalDiffusion.py:39-68, encoder.py:82-85,
layers.py:11-18,36-63); SURVEY.md §8b.
"""
from collections import OrderedDict

import numpy as np
import torch

H = 128          # node / edge / hidden width
NODE_IN = 51     # one-hot(21) + bb sincos(6) + sc sincos(8) + t-emb(16)
EDGE_IN = 468    # relpos(65) + 25*16 rbf + chain flag + 2 dihedrals
MSG_IN = 456     # h_V_i | h_E_ij | h_V_j | 72 geometric features
N_POINTS = 8
N_LAYERS = 3


def weight_spec():
    """Ordered (name, shape) list of every tensor the sampler reads."""
    s = []

    def lin(name, o, i):
        s.append((name + ".weight", (o, i)))
        s.append((name + ".bias", (o,)))

    def ln(name):
        s.append((name + ".weight", (H,)))
        s.append((name + ".bias", (H,)))

    lin("encoder.node_embedding", H, NODE_IN)
    ln("encoder.norm_nodes")
    lin("encoder.edge_embedding", H, EDGE_IN)
    ln("encoder.norm_edges")
    for l in range(N_LAYERS):
        p = f"mpnn.mpnn_layers.{l}."
        lin(p + "points_fn_node", 3 * N_POINTS, H)
        lin(p + "points_fn_edge", 3 * N_POINTS, H)
        for fn in ("node_message_fn", "edge_message_fn"):
            lin(p + fn + ".W_in", H, MSG_IN)
            lin(p + fn + ".W_inter.0", H, H)
            lin(p + fn + ".W_out", H, H)
        for k in range(4):
            ln(p + f"norm.{k}")
        for fn in ("node_dense", "edge_dense"):
            lin(p + fn + ".W_in", 4 * H, H)
            lin(p + fn + ".W_out", H, 4 * H)
    lin("decoder_score.0.W_in", H // 2, H)
    lin("decoder_score.0.W_out", H // 4, H // 2)
    lin("decoder_score.2.W_in", H // 8, H // 4)
    lin("decoder_score.2.W_out", 4, H // 8)
    return s


def make_random_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded stand-in weights (the trained checkpoint is not distributed with the reference).

    Matrices: xavier-uniform (as TorsionalDiffusion.py:80-82 initialises them); biases
    N(0, 0.1); LayerNorm gains 1 + N(0, 0.1).  Drawn from a CPU ``torch.Generator`` in
    spec order, so every machine with this torch build gets identical values.
    """
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    sd = OrderedDict()
    for name, shape in weight_spec():
        if len(shape) == 2:
            bound = float(np.sqrt(6.0 / (shape[0] + shape[1])))
            w = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif ".norm" in name and name.endswith("weight") or "norm_" in name and name.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:
            w = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        sd[name] = w
    return sd


def check_state_dict(sd, strict: bool = False):
    """Validate names/shapes; returns an OrderedDict of contiguous fp32 CPU tensors in spec order.

    ``strict=False`` mirrors ``load_from_checkpoint(strict=False)`` (eval_diffusion.py:33-40)
    for *extra* keys; every tensor the sampler reads must still be present.
    """
    out = OrderedDict()
    for name, shape in weight_spec():
        if name not in sd:
            raise RuntimeError(f"checkpoint is missing weight '{name}'")
        t = torch.as_tensor(sd[name]).detach().to(torch.float32).cpu().contiguous()
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"weight '{name}' has shape {tuple(t.shape)}, expected {shape}")
        out[name] = t
    if strict:
        extra = set(sd.keys()) - set(out.keys())
        if extra:
            raise RuntimeError(f"unexpected keys in state_dict: {sorted(extra)[:5]}")
    return out

"""Weight variants outside the statistics of the seeded xavier draw every other fixture uses (test infrastructure: shared by
tools/oracle/make_golden_envelope.py, which runs the unmodified reference on them, and tests/test_hip_parity.py, which runs the
HIP path on the same tensors).  Deterministic: a CPU ``torch.Generator`` in a fixed order."""
import math

import torch


def _is_linear(k, sd):
    return k.endswith("weight") and sd[k].dim() == 2


def envelope_variants(sd0):
    """{name: state_dict}: rescaled linear layers, large LayerNorm gains with shifted biases, heavy tails, and a "trained-like"
    draw (LayerNorm gains log-uniform in [0.1, 10], one log-uniform scale in [1/8, 8] per weight matrix, 1 % of the entries
    x30)."""
    g = torch.Generator().manual_seed(1)
    lin = [k for k in sd0 if _is_linear(k, sd0)]
    out = {}
    for name, f in (("linear x4", 4.0), ("linear x1/32", 1 / 32.)):
        out[name] = {k: (v * f if k in lin else v) for k, v in sd0.items()}
    v = dict(sd0)
    for k in sd0:
        if "norm" in k and k.endswith("weight"):
            v[k] = sd0[k] * 5.0
        elif k.endswith("bias") and "norm" not in k:
            v[k] = sd0[k] + (torch.rand(sd0[k].shape, generator=g) * 6 - 3)
    out["LN gain x5, biases +-3"] = v
    out["heavy tails"] = {k: (torch.where(torch.rand(v0.shape, generator=g) < 0.01, v0 * 30.0, v0) if k in lin else v0)
                          for k, v0 in sd0.items()}
    # (the decoder and the LayerNorm in front of it keep their values: they set the size of the score itself, and a score tens of
    # times larger makes the reverse process ill-conditioned in every arithmetic -- the variant is about the dynamic range
    # INSIDE the network, which the LayerNorm after every MLP re-normalises)
    g2 = torch.Generator().manual_seed(2)
    v = {}
    for k, v0 in sd0.items():
        if k.startswith("decoder_score") or k.startswith("mpnn.mpnn_layers.2.norm.1"):
            v[k] = v0
        elif "norm" in k and k.endswith("weight"):
            gain = torch.exp((torch.rand(v0.shape, generator=g2) * 2 - 1) * math.log(10.0))          # log-uniform 0.1 .. 10
            v[k] = torch.sign(v0) * gain
        elif k in lin:
            scale = float(torch.exp((torch.rand((), generator=g2) * 2 - 1) * math.log(8.0)))           # log-uniform 1/8 .. 8
            w = v0 * scale
            v[k] = torch.where(torch.rand(v0.shape, generator=g2) < 0.01, w * 30.0, w)
        else:
            v[k] = v0
    out["trained-like"] = v
    return out


def tiny_operand_variants(sd0):
    """Whole operand VECTORS of the edge kernels below 2^-4: the middle layer of a message MLP scaled by 1/1024 and the layer
    that consumes its (non-normalised) output by 1024 -- the function is unchanged up to rounding, but the hidden activations
    between the two are ~1e-3, where an unscaled f16 low part is subnormal and carries an absolute error of 2^-25."""
    out = {}
    for name, a, b in (("edge message, layer 0", "mpnn.mpnn_layers.0.edge_message_fn.W_inter.0", "mpnn.mpnn_layers.0.edge_message_fn.W_out"),
                       ("node message, layer 1", "mpnn.mpnn_layers.1.node_message_fn.W_inter.0", "mpnn.mpnn_layers.1.node_message_fn.W_out"),
                       ("edge FFN, layer 1", "mpnn.mpnn_layers.1.edge_dense.W_in", "mpnn.mpnn_layers.1.edge_dense.W_out")):
        v = dict(sd0)
        v[a + ".weight"] = sd0[a + ".weight"] / 1024.0
        v[a + ".bias"] = sd0[a + ".bias"] / 1024.0
        v[b + ".weight"] = sd0[b + ".weight"] * 1024.0
        out[name] = v
    return out


def small_ln_gain_variants(sd0):
    """Edge-level LayerNorms whose GAIN is tiny, with the scale put back into the weights that consume their output: the operand
    vectors h_E (encoder.norm_edges, norm.3 of a layer -> the W_B columns 128:256 of the next layer's two message MLPs) and x1
    (norm.2 -> the edge FFN's W_in) then sit at 1e-3 .. 1e-2 -- below 2^-4, where the edge kernels' unscaled f16 low part is
    subnormal -- while the dense products stay O(1).  (The residual paths see the small tensors as they are: another network than
    the seeded one, a valid one; the reference's fp32 / fp64 runs are the truth.)  "bias too": the LayerNorm biases shrink with the
    gains; "bias kept": they keep their size, so an operand feature is bias-dominated."""
    out = {}
    for name, shrink_bias in (("bias too", True), ("bias kept", False)):
        g = torch.Generator().manual_seed(3)
        v = dict(sd0)
        # a per-feature LayerNorm bias as well (the seeded draw has zeros): without one "bias kept" would be the same variant
        def gains():
            import math as _m
            return torch.exp(_m.log(1e-3) + torch.rand(128, generator=g) * (_m.log(1e-2) - _m.log(1e-3)))

        def apply(norm, consumers):
            gf = gains()
            b0 = (torch.rand(128, generator=g) * 2 - 1) * 0.5
            v[norm + ".weight"] = sd0[norm + ".weight"] * gf
            v[norm + ".bias"] = (sd0[norm + ".bias"] + b0) * (gf if shrink_bias else 1.0)
            for key, c0 in consumers:
                w = v[key].clone()
                w[:, c0:c0 + 128] = w[:, c0:c0 + 128] / gf
                v[key] = w
        apply("encoder.norm_edges", [("mpnn.mpnn_layers.0.node_message_fn.W_in.weight", 128),
                                     ("mpnn.mpnn_layers.0.edge_message_fn.W_in.weight", 128)])
        for l in (0, 1):
            apply(f"mpnn.mpnn_layers.{l}.norm.3", [(f"mpnn.mpnn_layers.{l + 1}.node_message_fn.W_in.weight", 128)]
                  + ([(f"mpnn.mpnn_layers.{l + 1}.edge_message_fn.W_in.weight", 128)] if l + 1 < 2 else []))
            apply(f"mpnn.mpnn_layers.{l}.norm.2", [(f"mpnn.mpnn_layers.{l}.edge_dense.W_in.weight", 0)])
        out["small LN gains, " + name] = v
    return out

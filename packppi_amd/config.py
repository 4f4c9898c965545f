"""The four hot-path YAML files of the reference, read without Hydra / OmegaConf.

The reference composes ``configs/eval_diffusion.yaml`` with Hydra and instantiates ``cfg.model.{encoder_cfg, model_cfg,
sample_cfg}`` into ``TDiffusionModule.load_from_checkpoint`` (eval_diffusion.py:22-41).  On this path only four files matter:

    configs/eval_diffusion.yaml                          ckpt_path, seed                  (:21-24)
    configs/model/encoder_cfg/ProteinEncoder.yaml        dimensions of the encoder
    configs/model/model_cfg/MpnnNet.yaml                 dimensions of the message-passing stack
    configs/model/sample_cfg/Sampling.yaml               mode, annealed_temp, proximal parameters

They are plain key/value YAML (``${...}`` interpolations only occur in files this path does not read and are kept as
strings).  The kernels are compiled for ONE set of dimensions (``PP_HIDDEN`` 128, ``PP_TOP_K`` 32, 8 points, 16 RBFs, 3 layers);
``check_compiled_dims`` raises a ``RuntimeError`` naming the key when a config asks for anything else, instead of loading a
checkpoint into kernels of another shape.
"""
import os
from types import SimpleNamespace
from typing import Any, Dict, Optional

ENCODER_YAML = os.path.join("model", "encoder_cfg", "ProteinEncoder.yaml")
MODEL_YAML = os.path.join("model", "model_cfg", "MpnnNet.yaml")
SAMPLE_YAML = os.path.join("model", "sample_cfg", "Sampling.yaml")
EVAL_YAML = "eval_diffusion.yaml"

# what libpackppi_hip.so is compiled for: the values of the reference's own YAML files
COMPILED_ENCODER = dict(node_in=35, edge_in=468, node_features=128, edge_features=128, time_embedding_type="sinusoidal",
                        time_embedding_dim=16, num_rbf=16, top_k=32, af2_relpos=True)
COMPILED_MODEL = dict(hidden_dim=128, num_mpnn_layers=3, n_points=8, act="relu", position_scale=1.0, use_ipmp=True)
# keys that do not change what the sampler computes and are therefore not checked: model_cfg.dropout (identity in eval()),
# model_cfg.k_neighbors (MpnnNet does not read it: the graph comes from the encoder's top_k), encoder_cfg.num_positional_embeddings
# (overridden to 65 whenever af2_relpos is true, encoder.py:93-94)
IGNORED_KEYS = ("dropout", "k_neighbors", "num_positional_embeddings")


def _as_dict(cfg) -> Dict[str, Any]:
    if cfg is None:
        return {}
    if isinstance(cfg, dict):
        return dict(cfg)
    return dict(vars(cfg))


def read_yaml(path) -> Dict[str, Any]:
    import yaml          # only a configuration TREE needs PyYAML: constructing a model from dicts / namespaces does not
    with open(path) as fh:
        data = yaml.safe_load(fh)
    if data is None:
        return {}
    if not isinstance(data, dict):
        raise RuntimeError(f"{path}: expected a mapping at the top level")
    return data


def load_hot_path_configs(config_dir) -> SimpleNamespace:
    """``config_dir`` = the reference checkout's ``configs/`` (or a directory laid out like it).  Returns a namespace with
    ``encoder_cfg``, ``model_cfg``, ``sample_cfg`` (dicts; a missing file gives ``None``), ``ckpt_path`` and ``seed``.
    ``seed`` is reported, not applied: the reference's eval_diffusion.py never reads ``cfg.seed`` (only train_diffusion.py:21-22 and
    train_affinity.py:21-22 call ``pl.seed_everything``), so its sampling draws are unseeded; ``cli.eval_diffusion --seed`` is this
    package's own switch."""
    config_dir = os.fspath(config_dir)
    if not os.path.isdir(config_dir):
        raise RuntimeError(f"config directory '{config_dir}' does not exist")

    def opt(rel):
        p = os.path.join(config_dir, rel)
        return read_yaml(p) if os.path.exists(p) else None

    top = opt(EVAL_YAML) or {}
    ckpt = top.get("ckpt_path")
    if isinstance(ckpt, str) and ("${" in ckpt or ckpt.startswith("/path/to/")):
        ckpt = None          # the placeholder the reference ships (eval_diffusion.yaml:24) or an unresolved interpolation
    return SimpleNamespace(encoder_cfg=opt(ENCODER_YAML), model_cfg=opt(MODEL_YAML), sample_cfg=opt(SAMPLE_YAML),
                           ckpt_path=ckpt, seed=top.get("seed"), config_dir=config_dir)


def _same(a, b) -> bool:
    if isinstance(b, bool) or isinstance(a, bool):
        return bool(a) == bool(b) and isinstance(a, (bool, int)) and isinstance(b, (bool, int))
    if isinstance(b, (int, float)) and isinstance(a, (int, float)):
        return float(a) == float(b)
    return a == b


def check_compiled_dims(encoder_cfg=None, model_cfg=None) -> None:
    """RuntimeError naming the first key whose value differs from what the kernels are compiled for.  Keys a config does not
    carry are not checked (the reference's defaults are the compiled values)."""
    for label, cfg, want in (("encoder_cfg", _as_dict(encoder_cfg), COMPILED_ENCODER), ("model_cfg", _as_dict(model_cfg), COMPILED_MODEL)):
        for key, value in want.items():
            if key in cfg and not _same(cfg[key], value):
                raise RuntimeError(f"{label}.{key} = {cfg[key]!r}, but libpackppi_hip.so is compiled for {key} = {value!r} "
                                   "(PP_HIDDEN / PP_TOP_K and the kernel tilings are compile-time constants): this checkpoint's "
                                   "architecture is not supported by this build")


def resolve_ckpt(explicit: Optional[str], cfgs: Optional[SimpleNamespace]) -> Optional[str]:
    """--ckpt_path, else $PACKPPI_CKPT, else ``ckpt_path`` of the config tree (the reference's only source)."""
    return explicit or os.environ.get("PACKPPI_CKPT") or (cfgs.ckpt_path if cfgs is not None else None)

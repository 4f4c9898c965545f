"""Container-only: run the UNMODIFIED reference on seeded inputs and store golden vectors.

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/make_golden.py [--only g2,g3,...]

Outputs (data only) go to tests/golden/*.npz.  Weights are not stored: they are the
deterministic ``packppi_amd.weights.make_random_state_dict(seed)`` values, loaded into the
reference module with ``strict=True`` (which also pins the key/shape contract).
Inputs are synthetic complexes from ``packppi_amd.synth`` (stored in the fixture, so the
tests do not depend on regenerating them bit-for-bit) and data/T1124_lig.pdb parsed by
``packppi_amd.pdb_io`` + ``featurize`` (featurisation itself is checked against the
reference's ``prot_to_data`` here and recorded in g0).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from packppi_amd import synth  # noqa: E402
from packppi_amd.batch import Batch, collate, TENSOR_KEYS  # noqa: E402
from packppi_amd.featurize import protein_to_data, protein_to_batch  # noqa: E402
from packppi_amd.pdb_io import from_pdb_file  # noqa: E402
from packppi_amd.weights import make_random_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 20251003


def ref_batch(b: Batch):
    """Same tensors in the shim's Data stand-in (what the reference code is handed)."""
    d = refshim.Data(**{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    return d


def pack_batch(b: Batch, prefix="batch."):
    out = {}
    for k in TENSOR_KEYS:
        out[prefix + k] = b[k].numpy()
    out[prefix + "num_proteins"] = np.int64(b["num_proteins"])
    out[prefix + "max_size"] = np.int64(b["max_size"])
    return out


def synth_batch(n_res, seed):
    return protein_to_batch(synth.make_complex(n_res, seed))


def padded_batch(sizes, seed0):
    return collate([protein_to_data(synth.make_complex(n, seed0 + i)) for i, n in enumerate(sizes)])


def seeded_init(model, batch, seed):
    """The reference's own add_sc_noise at t=1 under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    B, L = batch.residue_type.shape
    t = torch.tensor([1.]).repeat_interleave(B * L)
    x, _ = model.add_sc_noise(batch, t)
    return x


def run_sampling(model, batch, init, n_steps, **kw):
    model.schedule = torch.linspace(1, 0, n_steps + 1)
    orig = model.add_sc_noise
    model.add_sc_noise = lambda b, t: (init.clone(), None)
    try:
        # the proximal stage needs autograd; plain sampling gives identical values under no_grad
        with torch.set_grad_enabled(bool(kw.get("use_proximal", False))):
            out = model.sampling(batch, **kw)
    finally:
        model.add_sc_noise = orig
    return out


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--g7-blocks", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None

    def want(tag):
        return only is None or tag in only

    torch.set_num_threads(args.threads)
    model = refshim.build_reference_module(0)
    sd = make_random_state_dict(WEIGHT_SEED)
    model.load_state_dict(sd, strict=True)
    model.eval()
    from src.models.components import get_atom14_coords
    from src.models.components.clash import compute_residue_clash
    from src.models.components.optimize import proximal_optimizer
    from src.datamodules.components.complex_dataset import ComplexDataset

    # ---- g0: featurisation of the three shipped PDB files vs reference prot_to_data ------------
    if want("g0"):
        for tag in ("T1124_lig", "1BRS", "2FTL"):
            prot = from_pdb_file(os.path.join(refshim.REF, "data", tag + ".pdb"))
            mine = protein_to_data(prot)
            ref = ComplexDataset.prot_to_data({k: (v.copy() if hasattr(v, "copy") else v) for k, v in prot.items()},
                                              cache_processed_data=False)
            for k in TENSOR_KEYS:
                assert torch.equal(ref[k], mine[k]), (tag, k)
            if tag != "T1124_lig":
                save("g0_protein_" + tag, **{"prot." + k: v for k, v in prot.items()},
                     **{"ref." + k: ref[k] for k in TENSOR_KEYS})
        print("g0 ok")

    # ---- g2: per-op vectors -----------------------------------------------------------------------
    if want("g2"):
        cases = {"L8": synth_batch(8, 108), "L33": synth_batch(33, 133), "L64": synth_batch(64, 164),
                 "B3": padded_batch([20, 33, 27], 300)}
        for tag, b in cases.items():
            rb = ref_batch(b)
            B, L = b.residue_type.shape
            out = pack_batch(b)
            with torch.no_grad():
                init = seeded_init(model, rb, 7)
                out["init_chi_seed7"] = init
                # encoder internals
                enc = model.encoder
                _, E_idx, _ = enc._dist(rb.X[:, :, 1, :], rb.residue_mask)
                out["E_idx"] = E_idx
                for tval, tname in ((1.0, "t1"), (0.5, "t05"), (1.0 / 30, "t30")):
                    t = torch.tensor([tval]).repeat_interleave(B * L)
                    sincos = torch.stack((init.sin(), init.cos()), -1) * rb.SC_D_mask[..., None]
                    h_V0, h_E0, E2, _ = enc(rb.X, rb.residue_type, rb.BB_D_sincos, sincos, rb.chain_indices,
                                            rb.residue_mask, rb.residue_index, t.clone())
                    assert torch.equal(E2, E_idx)
                    score, h_V = model.network(rb, init, t.clone())
                    out[f"hV0_{tname}"] = h_V0
                    out[f"score_{tname}"] = score
                    out[f"hV_{tname}"] = h_V
                    if tname == "t1":
                        out["hE0"] = h_E0
                        # layer-by-layer states for kernel-level checks
                        from src.models.components import gather_nodes
                        ma = gather_nodes(rb.residue_mask.unsqueeze(-1), E_idx).squeeze(-1) * rb.residue_mask.unsqueeze(-1)
                        hv, he = h_V0, h_E0
                        for li, layer in enumerate(model.mpnn.mpnn_layers):
                            hv, he = layer(hv, he, E_idx, rb.X, rb.residue_mask, ma)
                            out[f"hV_l{li}_t1"] = hv
                            if li < 2 and tag in ("L8", "L33"):
                                out[f"hE_l{li}_t1"] = he
                if tag in ("L8", "L33"):
                    # raw 468-d edge features: re-trace the encoder's pieces
                    Ca, N, C, O = rb.X[:, :, 1], rb.X[:, :, 0], rb.X[:, :, 2], rb.X[:, :, 3]
                    Cb = enc._impute_CB(N, Ca, C)
                    X2 = torch.stack((N, Ca, C, O, Cb), dim=-2)
                    same = (rb.chain_indices[:, :, None] == rb.chain_indices[:, None, :]).float()
                    E = torch.cat((enc.embeddings(E_idx, rb.residue_index), enc._atomic_distances(X2, E_idx),
                                   (torch.gather(same, 2, E_idx) + 1).unsqueeze(-1),
                                   enc._pairwise_dihedrals(N, Ca, C, E_idx)), -1)
                    out["E_raw"] = E.float()
                xyz = get_atom14_coords(rb.X, rb.residue_type, rb.BB_D, init)
                out["atom14_init"] = xyz
                out["atom14_true"] = get_atom14_coords(rb.X, rb.residue_type, rb.BB_D, rb.SC_D)
                out["clash_init"] = compute_residue_clash(rb, init, 12., 0.5)
                out["clash_true_tol01"] = compute_residue_clash(rb, rb.SC_D, 12., 0.1)
                metric = model.analyze_samples(rb, SC_D_sample=init)
                for k, v in metric.items():
                    out["metric." + k] = np.float32(v)
            x = init.clone().requires_grad_(True)
            pr = compute_residue_clash(rb, x, 12., 0.5)
            pr.mean().backward()
            out["clash_grad_init"] = x.grad
            save("g2_ops_" + tag, **out)

    # ---- g3: end-to-end sampling ------------------------------------------------------------------
    if want("g3"):
        for tag, b, steps in (("L64", synth_batch(64, 164), (30, 100)), ("L300", synth_batch(300, 1300), (30, 100)),
                              ("B3", padded_batch([40, 64, 51], 400), (30,))):
            rb = ref_batch(b)
            out = pack_batch(b)
            init = seeded_init(model, rb, 11)
            out["init_chi_seed11"] = init
            for n in steps:
                t0 = time.time()
                out[f"chi_ode_{n}"] = run_sampling(model, rb, init, n)
                print(f"  {tag} ode {n} steps {time.time() - t0:.1f}s", flush=True)
            save("g3_sampling_" + tag, **out)
        # one SDE case: global generator seeded right before the loop
        b = synth_batch(33, 133)
        rb = ref_batch(b)
        sde_model = refshim.build_reference_module(0, mode="sde")
        sde_model.load_state_dict(sd, strict=True)
        init = seeded_init(sde_model, rb, 11)
        torch.manual_seed(99)
        chi = run_sampling(sde_model, rb, init, 30)
        save("g3_sampling_sde_L33", **pack_batch(b), init_chi_seed11=init, chi_sde_30_seed99=chi)

    # ---- g9: T1124 in SDE mode, 100 steps: the reference's own per-step draws from the global CPU generator seeded right
    # before the loop (only the seed and the result are stored; the test redraws the same torch.normal calls) -------------
    if want("g9"):
        g4 = np.load(os.path.join(GOLD, "g4_T1124.npz"))
        prot = from_pdb_file(os.path.join(refshim.REF, "data", "T1124_lig.pdb"))
        b = protein_to_batch(prot)
        rb = ref_batch(b)
        sde_model = refshim.build_reference_module(0, mode="sde")
        sde_model.load_state_dict(sd, strict=True)
        init = torch.from_numpy(g4["init_chi_seed1124"])
        torch.manual_seed(1124)
        t0 = time.time()
        chi = run_sampling(sde_model, rb, init, 100)
        print(f"  T1124 sde 100 steps {time.time() - t0:.1f}s", flush=True)
        save("g9_T1124_sde", chi_sde_100_seed1124=chi)

    # ---- g5: the 1500-residue synthetic complex of BASELINE config 3 (bench.py --workload s1500), 100 steps ----------
    if want("g5"):
        b = synth_batch(1500, 1500)
        rb = ref_batch(b)
        out = pack_batch(b)
        init = seeded_init(model, rb, 1500)
        out["init_chi_seed1500"] = init
        t0 = time.time()
        out["chi_ode_100"] = run_sampling(model, rb, init, 100)
        print(f"  S1500 ode 100 steps {time.time() - t0:.1f}s", flush=True)
        save("g5_S1500", **out)

    # ---- g7: BASELINE config 4 (256 synthetic ~300-residue complexes), 100 steps each, in blocks of 32 complexes
    # (g7_c5_rank{r} = complexes 32r .. 32r+31; keys carry the GLOBAL complex id).  --g7-blocks picks the blocks.
    # For every complex whose CA-distance rows hold an exact tie among the K+1 smallest values, the reference's own neighbour
    # lists (torch.topk on CPU, encoder.py:105-118) are stored too: E_idx_{i} int16 [L][K], tie_rows_{i}. --------------------
    if want("g7"):
        lens = synth.c5_lengths(256)
        for blk in [int(x) for x in args.g7_blocks.split(",")]:
            ids = list(range(32 * blk, 32 * blk + 32))
            out = {"lengths": np.array([lens[i] for i in ids], np.int64), "ids": np.array(ids, np.int64)}
            t0 = time.time()
            for i in ids:
                b = synth_batch(lens[i], 10000 + i)
                rb = ref_batch(b)
                init = seeded_init(model, rb, 20000 + i)
                out[f"init_{i}"] = init
                out[f"chi_ode_100_{i}"] = run_sampling(model, rb, init, 100)
                with torch.no_grad():
                    Dn, E_idx, _ = model.encoder._dist(rb.X[:, :, 1, :], rb.residue_mask)
                    # the K+1 smallest values of every row: any two equal -> order or membership is the topk's choice
                    X = rb.X[:, :, 1, :]
                    m2 = rb.residue_mask[:, None, :] * rb.residue_mask[:, :, None]
                    D = m2 * torch.sqrt(((X[:, None] - X[:, :, None]) ** 2).sum(3) + 1e-6)
                    Dadj = D + 2 * (1. - m2) * D.max(-1, keepdim=True)[0]
                    srt = torch.sort(Dadj, -1)[0][0, :, :33]
                    rows = torch.nonzero((srt[:, 1:] == srt[:, :-1]).any(-1)).flatten()
                if len(rows):
                    out[f"E_idx_{i}"] = E_idx[0].to(torch.int16)
                    out[f"tie_rows_{i}"] = rows.to(torch.int16)
                    member = [int(r) for r in rows if srt[r, 31] == srt[r, 32]]
                    print(f"  c5[{i}] tie rows {rows.tolist()} (membership: {member})", flush=True)
                print(f"  c5[{i}] L={lens[i]} {time.time() - t0:.0f}s", flush=True)
            save(f"g7_c5_rank{blk}", **out)

    # ---- g3p: proximal optimiser ------------------------------------------------------------------
    if want("g3p"):
        for tag, b in (("L64", synth_batch(64, 164)), ("L120", synth_batch(120, 1120))):
            rb = ref_batch(b)
            out = pack_batch(b)
            init = seeded_init(model, rb, 11)
            out["init_chi_seed11"] = init
            for n in (5, 50):
                chis, losses = proximal_optimizer(rb, init.clone(), 12., 0.5, 1., n)
                out[f"prox_chi_last_{n}"] = chis[-1]
                out[f"prox_chi_first_{n}"] = chis[0]
                out[f"prox_losses_{n}"] = np.array(losses, np.float64)
            # sampling(use_proximal=True) end-to-end, 30 steps
            res = run_sampling(model, rb, init, 30, use_proximal=True)
            out["chi_ode_30_proximal"] = res
            save("g3_proximal_" + tag, **out)

    # ---- g4: T1124 --------------------------------------------------------------------------------
    if want("g4"):
        prot = from_pdb_file(os.path.join(refshim.REF, "data", "T1124_lig.pdb"))
        b = protein_to_batch(prot)
        rb = ref_batch(b)
        out = pack_batch(b)
        out["prot.chain_id"] = prot["chain_id"]
        out["prot.residue_index"] = prot["residue_index"]
        out["prot.b_factors"] = prot["b_factors"].astype(np.float32)
        init = seeded_init(model, rb, 1124)
        out["init_chi_seed1124"] = init
        t0 = time.time()
        out["chi_ode_100"] = run_sampling(model, rb, init, 100)
        print(f"  T1124 100 steps (no_grad, 8 threads): {time.time() - t0:.1f}s", flush=True)
        with torch.no_grad():
            out["clash_final"] = compute_residue_clash(rb, out["chi_ode_100"], 12., 0.5)
            for k, v in model.analyze_samples(rb, SC_D_sample=out["chi_ode_100"]).items():
                out["metric." + k] = np.float32(v)
        save("g4_T1124", **out)


if __name__ == "__main__":
    main()

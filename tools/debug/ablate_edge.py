"""Timing-only ablations of the edge kernels (results are wrong by construction; only the clock matters)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import packppi_amd.build as b
base = list(b.FLAGS)
for tag, flags in [("baseline", []), ("noload", ["-DPP_X_NOLOAD"]), ("nostore", ["-DPP_X_NOSTORE"]),
                   ("nobarrier", ["-DPP_X_NOBARRIER"]), ("nomfma", ["-DPP_X_NOMFMA"]),
                   ("noload+nostore+nobarrier", ["-DPP_X_NOLOAD", "-DPP_X_NOSTORE", "-DPP_X_NOBARRIER"])]:
    b.FLAGS[:] = base + flags
    b.build_library(force=True, verbose=False)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/debug/time_vs_n.py"), "256", "739"],
                         capture_output=True, text=True).stdout.strip().splitlines()
    print(tag, "|", " | ".join(l for l in out if l.startswith("L=")), flush=True)
b.FLAGS[:] = base
b.build_library(force=True, verbose=False)

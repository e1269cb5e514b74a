"""Helpers for reading tests/golden/*.npz (fixtures produced by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

from oracle import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrays = {k: z[k] for k in z.files if k != "meta"}
    return meta, arrays


def names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def state_for(meta, overrides=None):
    """Rebuild the synthetic state dict of a fixture and verify the RNG recipe reproduced it."""
    st = synth.synth_state([tuple(s) for s in meta["shapes"]], meta["seed"])
    for k, v in (overrides or {}).items():
        st[k] = v
    if all(v is not None for v in st.values()):
        assert synth.checksum(st) == meta["checksum"], "synthetic-state recipe drifted from the fixture"
    return st


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))

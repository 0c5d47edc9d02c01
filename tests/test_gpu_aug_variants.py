"""GPU: the selectable variants of the tile kernels give the same bytes.  The switches are read once per process
(AADG_GENERIC_V1: the one-pass tile for down-scaling units instead of the horizontal + vertical pass; AADG_GEN_CHUNK: list slots per
chunk of the two-pass flow, whose intermediate lives in one chunk-sized slice of the workspace), so each variant runs in its own
interpreter on the same seeded units and writes a digest of its outputs; the default variant is also compared with the oracle."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, sys, os
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from aadg_amd import _lib
from helpers import random_units, synth_pool
rs = np.random.RandomState(77)
H, crop, N, P = 192, 192, 60, 10
imgs, msks = synth_pool(rs, P, H, H, vessel=True)
units = random_units(rs, N, P, H, H, crop, (0.5, 2.0))
for i in range(0, N, 3):                       # every third unit carries a Sharpness stencil (two of them every ninth)
    units[i]["n_ops"] = 2
    units[i]["op"][0] = 8; units[i]["farg"][0] = np.float32(1.7)
    if i % 9 == 0:
        units[i]["op"][1] = 8; units[i]["farg"][1] = np.float32(0.3)
oi, ol = _lib.aug_u8_forward(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda(), units, crop, 1)
torch.cuda.synchronize()
a, b = oi.cpu().numpy(), ol.cpu().numpy()
np.save(sys.argv[1], np.concatenate([a.reshape(-1), b.reshape(-1)]))
n_generic = int((((units["scaled_w"] < H) | (units["scaled_h"] < H)) & (2 * units["scaled_w"] >= H) & (2 * units["scaled_h"] >= H)).sum())
print("GENERIC", n_generic)
'''


def _run(tmp_path, tag, env_extra):
    out = os.path.join(str(tmp_path), tag + ".npy")
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    n_generic = int([l for l in p.stdout.decode().splitlines() if l.startswith("GENERIC")][-1].split()[1])
    return np.load(out), n_generic


def test_generic_flow_variants_agree(tmp_path, hip, oracle):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import random_units, synth_pool
    base, n_generic = _run(tmp_path, "default", {})
    assert n_generic >= 12                                      # chunks of 5 slots: at least three of them, the last one partial or full
    for tag, env in (("chunk5", {"AADG_GEN_CHUNK": "5"}), ("chunk1", {"AADG_GEN_CHUNK": "1"}), ("onepass", {"AADG_GENERIC_V1": "1"})):
        got, _ = _run(tmp_path, tag, env)
        assert got.shape == base.shape and np.array_equal(got, base), tag
    # and the default against the oracle (the same units, drawn with the same generator state)
    rs = np.random.RandomState(77)
    H, crop, N, P = 192, 192, 60, 10
    imgs, msks = synth_pool(rs, P, H, H, vessel=True)
    units = random_units(rs, N, P, H, H, crop, (0.5, 2.0))
    for i in range(0, N, 3):
        units[i]["n_ops"] = 2
        units[i]["op"][0] = 8; units[i]["farg"][0] = np.float32(1.7)
        if i % 9 == 0:
            units[i]["op"][1] = 8; units[i]["farg"][1] = np.float32(0.3)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 1)
    assert np.array_equal(base, np.concatenate([want_img.reshape(-1), want_lbl.reshape(-1)]))

"""GPU: the chunking of the two-pass flow of the down-scaling units does not change a byte.  The horizontally resampled rows of a
chunk of list slots share one slice of the workspace (csrc/aug_u8.hip: launch_tiles); `aadg_aug_lists.gen_chunk` (ABI 7) asks for
fewer slots per chunk than the library would take: chunks of 1 and 5 slots and the default on the same seeded units -- every third
with a Sharpness stencil (the stencil units are the tail of the list: a chunk can hold both kinds) -- and the default against the
oracle.  (Until round 3 this file also ran the one-pass tile of rounds 1-2 through an environment switch; that kernel and the three
getenv switches of the library are gone.)"""
import numpy as np
import pytest
import torch

from helpers import random_units, synth_pool

pytestmark = pytest.mark.gpu


def test_generic_flow_chunkings_agree(hip, oracle):
    rs = np.random.RandomState(77)
    H, crop, N, P = 192, 192, 60, 10
    imgs, msks = synth_pool(rs, P, H, H, vessel=True)
    units = random_units(rs, N, P, H, H, crop, (0.5, 2.0))
    for i in range(0, N, 3):                       # every third unit carries a Sharpness stencil (two of them every ninth)
        units[i]["n_ops"] = 2
        units[i]["op"][0] = 8; units[i]["farg"][0] = np.float32(1.7)
        if i % 9 == 0:
            units[i]["op"][1] = 8; units[i]["farg"][1] = np.float32(0.3)
    n_generic = hip.launch_hints(units, H, H, crop)[3][2]
    assert n_generic >= 12                                      # chunks of 5 slots: at least three of them
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    base = hip.aug_u8_forward(d_img, d_msk, units, crop, 1)
    for chunk in (5, 1, 7, 10 ** 6):                            # a value above the library's own is clamped to it
        got = hip.aug_u8_forward(d_img, d_msk, units, crop, 1, gen_chunk=chunk)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]), chunk
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 1)
    assert np.array_equal(base[0].cpu().numpy(), want_img) and np.array_equal(base[1].cpu().numpy(), want_lbl)
    with pytest.raises(hip.AadgError):
        hip.aug_u8_forward(d_img, d_msk, units, crop, 1, gen_chunk=-2)

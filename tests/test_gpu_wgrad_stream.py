"""GPU: weight gradients beside the backward chain (_lib.set_wgrad_stream, round 5).  The own convolutions' weight-gradient kernels run
on a second HIP stream and reach `.grad` at the end of the backward pass (a callback on the autograd engine) instead of through
autograd's accumulation.  Checked where the comparison is well conditioned: a chain of the convolution Functions themselves (a randomly
initialised ResNet with batch statistics amplifies the 1e-7 order noise of the split-K atomics to ~20 % between two IDENTICAL passes,
scripts/ab/wgrad_noise.py, so whole-network gradients cannot tell the two orders apart), plus the mechanics on the whole search step."""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["f32x3", "bf16"])
def test_weight_gradients_beside_the_chain_equal_gradients_in_line(hip, dtype):
    torch.manual_seed(5)
    x3 = dtype == "f32x3"
    act = torch.float32 if x3 else torch.bfloat16
    ws = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.1) for s in [(64, 32, 1, 1), (64, 64, 3, 3), (128, 64, 3, 3), (64, 128, 1, 1), (64, 64, 3, 3)]]
    x = torch.randn(4, 32, 64, 64, device="cuda").to(act)
    r = torch.randn(4, 64, 32, 32, device="cuda")

    def net():
        c1, c3, c3s2 = (hip.conv1x1_x3, hip.conv3x3_x3, hip.conv3x3s2_x3) if x3 else (hip.conv1x1, hip.conv3x3, hip.conv3x3s2)
        h = torch.relu(c1(x, ws[0]))
        h = torch.relu(c3(h, ws[1], 1))
        h = torch.relu(c3s2(h, ws[2]))
        h = torch.relu(c1(h, ws[3]))
        h = c3(h, ws[4], 2)
        return (h.float() * r).sum() * 1e-3

    def grads(mode, zero=True):
        old = hip.set_wgrad_stream(mode)
        try:
            if zero:
                for w in ws:
                    w.grad = None
            net().backward()
            torch.cuda.synchronize()
            assert all(w.grad is not None and w.grad.shape == w.shape and w.grad.is_contiguous() for w in ws)
            return [w.grad.detach().clone() for w in ws]
        finally:
            hip.set_wgrad_stream(old)

    line, side, side2 = grads(False), grads(True), grads(True)
    tol = 1e-5 if x3 else 1e-3                              # split-K atomics: the summation order differs from launch to launch
    for a, b, c in zip(line, side, side2):
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= tol * scale and float((a - c).abs().max()) <= tol * scale
    acc = grads(True, zero=False)                           # accumulation into an existing .grad
    for a, b in zip(side2, acc):
        assert float((b - 2 * a).abs().max()) <= 2 * tol * float(a.abs().max())


def test_search_step_with_the_side_stream(hip):
    """The whole policy-search step with the weight gradients beside the chain: the loader switches it on (one process, no DDP), every
    parameter of the segmentation model receives a finite gradient of its own shape, nothing is left pending, and the first step's loss
    -- computed before any weight moved -- equals the in-line run's."""
    sys.path.insert(0, ROOT)
    import bench
    from aadg_amd import _lib

    def run(no_side):
        for seed_fn in (random.seed, np.random.seed, torch.manual_seed):
            seed_fn(1023)
        a = bench.Args()
        a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), "resnet50", 2, 128
        a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout, a.no_wgrad_stream = "f32x3", True, "row", True, no_side
        cfg, st = bench.build_state(a, 0, 1)
        assert hip.wgrad_stream_enabled() == (not no_side)
        out = []
        for i in range(2):
            st.search_step(i, max_iters=1)
            torch.cuda.synchronize()
            assert not _lib._WG["pending"]
            out.append([p.grad.detach().clone() if p.grad is not None else None for p in st.model.parameters()])
        params = list(st.model.parameters())
        assert all(g is not None and g.shape == p.shape and bool(torch.isfinite(g).all()) for g, p in zip(out[-1], params) if p.requires_grad)
        return out

    try:
        side, line = run(False), run(True)
    finally:
        hip.set_wgrad_stream(False)
    # step 0: identical weights and batch on both sides -> the last layers' gradients (a short, well-conditioned path) agree
    for gs, gl in list(zip(side[0], line[0]))[-4:]:
        assert float((gs - gl).abs().max()) <= 2e-3 * float(gl.abs().max()) + 1e-9, (float((gs - gl).abs().max()), float(gl.abs().max()))

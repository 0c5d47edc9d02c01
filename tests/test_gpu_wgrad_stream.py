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


def test_on_load_batchnorm_operands_survive_on_the_side_stream(hip):
    """the weight-gradient kernels of a convolution that applies a BatchNorm + ReLU on operand load read pre_scale /
    pre_shift ON THE SIDE STREAM; those [K] tensors are freed to the launch stream's allocator as soon as the node's backward returns,
    and the very next node (the lazy BatchNorm's backward) allocates dw / db of the same size.  With the tensors recorded on the side
    stream the gradients beside the chain equal the in-line ones -- checked under allocator pressure: many same-sized allocations and
    writes on the launch stream right behind each backward node, side stream lagging behind a long kernel."""
    torch.manual_seed(9)
    N, C, H = 8, 64, 32
    x = torch.randn(N, C, H, H, device="cuda")
    w1 = torch.nn.Parameter(torch.randn(C, C, 1, 1, device="cuda") * 0.1)
    w2 = torch.nn.Parameter(torch.randn(C, C, 3, 3, device="cuda") * 0.05)
    w3 = torch.nn.Parameter(torch.randn(4 * C, C, 1, 1, device="cuda") * 0.1)
    bn = [(torch.nn.Parameter(torch.rand(C, device="cuda") + 0.5), torch.nn.Parameter(torch.randn(C, device="cuda") * 0.1)) for _ in range(2)]
    r = torch.randn(N, 4 * C, H, H, device="cuda")
    if not (hip.conv3x3_x3_pre_supported(x, w2, 1) and hip.conv1x1_x3_pre_supported(x, w3)):
        pytest.skip("shape outside the on-load tiles")

    class Churn(torch.autograd.Function):
        """identity whose backward hammers the launch stream's allocator with [C]-sized blocks (what the next node's dw / db would get)"""
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            for i in range(64):
                torch.full((C,), float(i + 1e6), device="cuda")           # allocated, written, freed: recycles the freed [K] blocks
            return g

    def run(mode):
        old = hip.set_wgrad_stream(mode)
        try:
            for p in [w1, w2, w3] + [t for pair in bn for t in pair]:
                p.grad = None
            c1 = hip.conv1x1_x3(x, w1, True)
            z1, s1, h1 = hip.batch_norm_lazy(c1, bn[0][0], bn[0][1], torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), 0.1, 1e-5, c1._aadg_bn_sums)
            c2 = hip.conv3x3_x3(Churn.apply(z1), w2, 1, True, pre=(s1, h1))
            z2, s2, h2 = hip.batch_norm_lazy(c2, bn[1][0], bn[1][1], torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), 0.1, 1e-5, c2._aadg_bn_sums)
            c3 = hip.conv1x1_x3(Churn.apply(z2), w3, False, pre=(s2, h2))
            if mode:
                # the side stream lags: a long kernel queued on it in front of the weight gradients
                side = hip._WG["stream"] or torch.cuda.Stream()
                hip._WG["stream"] = side
                with torch.cuda.stream(side):
                    torch.cuda._sleep(20_000_000)
            ((c3 * r).sum() * 1e-3).backward()
            torch.cuda.synchronize()
            return [p.grad.detach().clone() for p in (w1, w2, w3)]
        finally:
            hip.set_wgrad_stream(old)
    line = run(False)
    for rep in range(3):
        side = run(True)
        for a, b in zip(line, side):
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (rep, float((a - b).abs().max()), float(a.abs().max()))


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

"""GPU: fused BatchNorm (+ ReLU / ReLU6, + residual add) kernels vs torch.nn.functional.batch_norm, forward,
backward, running statistics, inference mode; and the backbone wired to them."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {0: lambda t: t, 1: F.relu, 2: F.relu6}


def _reference(x, res, w, b, rm, rv, training, momentum, eps, act):
    y = F.batch_norm(x.float(), rm, rv, w, b, training, momentum, eps)
    if x.dtype != torch.float32:
        y = y.to(x.dtype).float()          # the unfused graph stores bn's output in the activation dtype
    if res is not None:
        y = y + res.float()
    return ACTS[act](y)


@pytest.mark.parametrize("shape", [(4, 8, 16, 16), (3, 5, 7, 9), (6, 16, 1, 1), (2, 64, 64, 64), (5, 3, 33, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,with_res", [(0, False), (1, False), (2, False), (1, True), (0, True)])
def test_bn_training_matches_torch(hip, shape, dtype, act, with_res):
    torch.manual_seed(sum(shape) + act)
    N, C, H, W = shape
    x = (torch.randn(shape, device="cuda") * 2 + 0.5).to(dtype).requires_grad_(True)
    res = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True) if with_res else None
    w = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    b = (torch.randn(C, device="cuda") * (2.0 if act == 2 else 0.3)).requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm_r, rv_r = rm.clone(), rv.clone()
    y = hip.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, act, res)
    xr = x.detach().clone().requires_grad_(True)
    rr = res.detach().clone().requires_grad_(True) if with_res else None
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = _reference(xr, rr, wr, br, rm_r, rv_r, True, 0.1, 1e-5, act)
    lo = dtype == torch.bfloat16
    assert y.dtype == dtype
    assert (y.float() - yr).abs().max().item() <= (6e-2 if lo else 2e-5)
    assert torch.allclose(rm, rm_r, atol=1e-5) and torch.allclose(rv, rv_r, rtol=1e-4, atol=1e-5)
    g = torch.randn(shape, device="cuda").to(dtype)
    y.backward(g)
    yr.backward(g.float())
    n = N * H * W
    scale = max(1.0, n ** 0.5)
    tol_x = 8e-2 if lo else 2e-4
    # elements whose pre-activation is within rounding distance of the kink may take the other branch in bf16
    dxd = (x.grad.float() - xr.grad.float()).abs()
    if lo:
        assert (dxd > tol_x).float().mean().item() < 5e-3
    else:
        assert dxd.max().item() <= tol_x
    assert (w.grad - wr.grad).abs().max().item() <= (0.05 * scale if lo else 2e-3)
    assert (b.grad - br.grad).abs().max().item() <= (0.05 * scale if lo else 2e-3)
    if with_res:
        d = (res.grad.float() - rr.grad.float()).abs()
        assert (d > 1e-6).float().mean().item() < (5e-3 if lo else 1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_inference_uses_running_stats(hip, dtype):
    torch.manual_seed(3)
    x = torch.randn(3, 12, 10, 16, device="cuda").to(dtype)
    w, b = torch.rand(12, device="cuda") + 0.5, torch.randn(12, device="cuda")
    rm, rv = torch.randn(12, device="cuda"), torch.rand(12, device="cuda") + 0.2
    rm0, rv0 = rm.clone(), rv.clone()
    y = hip.batch_norm_act(x, w, b, rm, rv, False, 0.1, 1e-5, 1)
    yr = F.relu(F.batch_norm(x.float(), rm0, rv0, w, b, False, 0.1, 1e-5))
    assert torch.equal(rm, rm0) and torch.equal(rv, rv0)
    assert (y.float() - yr).abs().max().item() <= (5e-2 if dtype == torch.bfloat16 else 2e-5)


def test_bn_rejects_cpu_and_bad_layout(hip):
    x = torch.randn(2, 4, 8, 8)
    w = torch.ones(4)
    with pytest.raises(hip.AadgError):
        hip.batch_norm_act(x, w, w, w.clone(), w.clone(), True, 0.1, 1e-5)
    xc = torch.randn(2, 4, 8, 8, device="cuda").permute(0, 1, 3, 2)
    wc = w.cuda()
    with pytest.raises(hip.AadgError):
        hip.batch_norm_act(xc, wc, wc, wc.clone(), wc.clone(), True, 0.1, 1e-5)


@pytest.mark.parametrize("encoder", ["resnet50", "mobilenet_v2"])
def test_backbone_fused_bn_matches_module_path(hip, encoder):
    """The same DeepLabV3+ with bn_act forced onto the nn.Module path gives the same logits / gradients (fp32)."""
    from aadg_amd.models import deeplab
    torch.manual_seed(5)
    m = deeplab.DeepLabV3Plus(encoder, 2).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    x = torch.randn(4, 3, 64, 64, device="cuda")
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def run():
        m.load_state_dict(state)
        m.zero_grad(set_to_none=True)
        y, f = m(x)
        (y.square().mean() + f.square().mean()).backward()
        return y.detach(), f.detach(), m.classifier.weight.grad.clone(), m.encoder_first_grad()

    m.encoder_first_grad = lambda: next(m.encoder.parameters()).grad.clone()
    y1, f1, gc1, ge1 = run()
    rm1 = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    saved = deeplab.bn_act

    def module_path(bn, t, act=None, residual=None, handles=1, out=None, res_affine=None):      # (out: the caller copies the result into its slice)
        y = bn(t)
        if residual is not None:
            y = y + residual
        return F.relu(y) if act == "relu" else (F.relu6(y) if act == "relu6" else y)

    deeplab.bn_act = module_path
    try:
        y2, f2, gc2, ge2 = run()
    finally:
        deeplab.bn_act = saved
    rm2 = {k: v for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    assert (y1 - y2).abs().max().item() <= 2e-3 * max(1.0, y2.abs().max().item())
    assert (f1 - f2).abs().max().item() <= 2e-3 * max(1.0, f2.abs().max().item())
    assert (gc1 - gc2).abs().max().item() <= 2e-3 * max(1e-3, gc2.abs().max().item())
    assert (ge1 - ge2).abs().max().item() <= 2e-2 * max(1e-3, ge2.abs().max().item())
    for k in rm1:
        assert torch.allclose(rm1[k].float(), rm2[k].float(), rtol=1e-3, atol=1e-4), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_many_handles_sum_all_gradients(hip, dtype):
    """handles = 6 (the encoder output: five ASPP branches + the pooled feature): six gradients summed in the kernel."""
    torch.manual_seed(4)
    shape = (2, 8, 8, 16)
    x0 = torch.randn(shape, device="cuda").to(dtype)
    r0 = torch.randn(shape, device="cuda").to(dtype)
    w0, b0 = torch.rand(8, device="cuda") + 0.5, torch.randn(8, device="cuda") * 0.2
    gs = [torch.randn(shape, device="cuda").to(dtype) for _ in range(6)]
    res = []
    for k in (6, 1):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        out = hip.batch_norm_act(x, w0, b0, torch.zeros(8, device="cuda"), torch.ones(8, device="cuda"), True, 0.1, 1e-5, 1, r, handles=k)
        outs = list(out) if k > 1 else [out] * 6
        assert len(outs) == 6
        torch.autograd.backward(outs, gs)
        res.append((x.grad.float(), r.grad.float()))
    tol = 0.15 if dtype == torch.bfloat16 else 1e-5          # bf16: the reference rounds every partial sum
    assert (res[0][0] - res[1][0]).abs().max().item() <= tol and (res[0][1] - res[1][1]).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [True, False])
def test_bn_dual_output_sums_both_gradients(hip, dtype, with_res):
    """dual=True hands the output out twice (same storage); gradients arriving through the two handles are summed inside
    the backward kernel -- same result as using one output tensor twice."""
    torch.manual_seed(9)
    shape = (3, 16, 8, 16)
    C = shape[1]
    x0 = torch.randn(shape, device="cuda").to(dtype)
    r0 = torch.randn(shape, device="cuda").to(dtype) if with_res else None
    w0, b0 = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    ga, gb = torch.randn(shape, device="cuda").to(dtype), torch.randn(shape, device="cuda").to(dtype)

    def run(dual):
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if with_res else None
        w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = hip.batch_norm_act(x, w, b, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), True, 0.1, 1e-5, 1, r, dual=dual)
        if dual:
            ya, yb = out
            assert ya.data_ptr() == yb.data_ptr()
        else:
            ya = yb = out
        torch.autograd.backward([ya, yb], [ga, gb])
        return ya.detach().float(), x.grad.float(), w.grad, b.grad, (r.grad.float() if with_res else None)

    a, b = run(True), run(False)
    lo = dtype == torch.bfloat16
    assert torch.equal(a[0], b[0])
    assert (a[1] - b[1]).abs().max().item() <= (6e-2 if lo else 1e-5)
    assert (a[2] - b[2]).abs().max().item() <= (0.5 if lo else 1e-4) and (a[3] - b[3]).abs().max().item() <= (0.5 if lo else 1e-4)
    if with_res:
        assert (a[4] - b[4]).abs().max().item() <= (6e-2 if lo else 1e-6)
    # only one of the two handles used: the other gradient is absent
    x = x0.clone().requires_grad_(True)
    ya, yb = hip.batch_norm_act(x, w0, b0, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), True, 0.1, 1e-5, 1, r0, dual=True)
    yb.backward(gb)
    x2 = x0.clone().requires_grad_(True)
    y2 = hip.batch_norm_act(x2, w0, b0, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), True, 0.1, 1e-5, 1, r0)
    y2.backward(gb)
    assert (x.grad.float() - x2.grad.float()).abs().max().item() <= (6e-2 if lo else 1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_plane_constant_gradient_travels_as_one_value_per_plane(hip, dtype):
    """The output also feeds a global average pool: its gradient (a broadcast over each plane) is handed to the kernel as
    [N*C] floats -- same result as the materialised tensor."""
    from aadg_amd.models.deeplab import global_avg_pool_f32
    torch.manual_seed(11)
    shape = (3, 8, 8, 16)
    x0 = torch.randn(shape, device="cuda").to(dtype)
    r0 = torch.randn(shape, device="cuda").to(dtype)
    w0, b0 = torch.rand(8, device="cuda") + 0.5, torch.randn(8, device="cuda") * 0.2
    g = torch.randn(shape, device="cuda").to(dtype)
    gp = torch.randn(3, 8, device="cuda")
    res = []
    for custom in (True, False):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        ya, yb = hip.batch_norm_act(x, w0, b0, torch.zeros(8, device="cuda"), torch.ones(8, device="cuda"), True, 0.1, 1e-5, 1, r, handles=2)
        pooled = global_avg_pool_f32(yb) if custom else yb.float().mean(dim=(2, 3))
        torch.autograd.backward([ya, pooled], [g, gp])
        res.append((x.grad.float(), r.grad.float()))
    tol = 0.1 if dtype == torch.bfloat16 else 1e-5
    assert (res[0][0] - res[1][0]).abs().max().item() <= tol and (res[0][1] - res[1][1]).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_writes_into_concat_slices_and_reads_gradient_slices(hip, dtype):
    """batch_norm_act(out=slice of a concatenation buffer) + concat_from_slices == torch.cat of the separate outputs, forward and
    backward (the gradient slices are read in place: image stride of the wide tensor)."""
    torch.manual_seed(21)
    N, H, W = 3, 8, 16
    chans = [8, 16, 8]
    xs = [torch.randn(N, c, H, W, device="cuda").to(dtype) for c in chans]
    ws = [torch.rand(c, device="cuda") + 0.5 for c in chans]
    bs = [torch.randn(c, device="cuda") * 0.2 for c in chans]
    g = torch.randn(N, sum(chans), H, W, device="cuda").to(dtype)
    res = []
    for fused in (True, False):
        xin = [x.clone().requires_grad_(True) for x in xs]
        win = [w.clone().requires_grad_(True) for w in ws]
        if fused:
            buf, parts = hip.concat_slices(N, chans, H, W, dtype, xs[0].device)
            outs = [hip.batch_norm_act(x, w, b, torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), True, 0.1, 1e-5, 1, out=p)
                    for x, w, b, c, p in zip(xin, win, bs, chans, parts)]
            assert all(o.data_ptr() == p.data_ptr() for o, p in zip(outs, parts))
            y = hip.concat_from_slices(buf, outs)
        else:
            y = torch.cat([hip.batch_norm_act(x, w, b, torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), True, 0.1, 1e-5, 1)
                           for x, w, b, c in zip(xin, win, bs, chans)], dim=1)
        y.backward(g)
        res.append((y.detach().clone(), [x.grad.clone() for x in xin], [w.grad.clone() for w in win]))
    a, b = res
    assert torch.equal(a[0], b[0])
    for ga, gb in zip(a[1], b[1]):
        assert torch.equal(ga, gb)
    for ga, gb in zip(a[2], b[2]):
        assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-5)

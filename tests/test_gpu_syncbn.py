"""GPU: synchronised BatchNorm statistics (aadg_bn_sync_forward / aadg_bn_sync_backward) -- the all-reduce sits
between the HIP statistics and elementwise kernels.  Two data-parallel "ranks" are played in ONE process: the all-reduce
hook (_lib.BN_SYNC_REDUCE) first records each rank's local sums, then replays the other rank's record as the reduction.
The truth is the ordinary (single-device) kernel on the concatenated batch: same y, same dx, same running statistics;
parameter gradients are LOCAL sums (DDP averages them), so they must add up to the full-batch ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class TwoRanks(object):
    """Plays the collective for `n` ranks visited one after the other: pass 'record' leaves the local sums in place and
    remembers them; pass 'reduce' adds the other ranks' remembered sums of the same call."""

    def __init__(self, hip, n):
        self.hip, self.n = hip, n
        self.book = {}
        self.mode, self.rank, self.call = 'record', 0, 0

    def start(self, mode, rank):
        self.mode, self.rank, self.call = mode, rank, 0

    def __call__(self, t):
        key = self.call
        self.call += 1
        if self.mode == 'record':
            self.book[(self.rank, key)] = t.clone()
        else:
            for r in range(self.n):
                if r != self.rank:
                    t += self.book[(r, key)]


def _run(hip, fake, phase_fwd, phase_bwd, xs, ress, w, b, act, g_parts, momentum=0.1, eps=1e-5):
    outs = []
    for r, x in enumerate(xs):
        rm, rv = torch.zeros_like(w), torch.ones_like(w)
        xr = x.detach().clone().requires_grad_(True)
        rr = ress[r].detach().clone().requires_grad_(True) if ress is not None else None
        wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
        fake.start(phase_fwd, r)
        y = hip.batch_norm_act(xr, wr, br, rm, rv, True, momentum, eps, act, rr, sync=True)
        outs.append((xr, rr, wr, br, rm, rv, y))
    if phase_bwd is None:
        return outs
    for r, (xr, rr, wr, br, rm, rv, y) in enumerate(outs):
        fake.start(phase_bwd, r)
        fake.call = 1000                        # backward calls get their own keys
        y.backward(g_parts[r])
    return outs


@pytest.mark.parametrize("shape,cut", [((8, 16, 16, 16), 3), ((6, 5, 7, 9), 2), ((4, 64, 32, 32), 2)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,with_res", [(0, False), (1, False), (2, False), (1, True)])
def test_sync_bn_equals_full_batch(hip, shape, cut, dtype, act, with_res):
    torch.manual_seed(sum(shape) + act)
    N, C, H, W = shape
    x = (torch.randn(shape, device="cuda") * 2 + 0.5).to(dtype)
    res = torch.randn(shape, device="cuda").to(dtype) if with_res else None
    w = torch.rand(C, device="cuda") + 0.5
    b = torch.randn(C, device="cuda") * (2.0 if act == 2 else 0.3)
    g = torch.randn(shape, device="cuda").to(dtype)
    # truth: one device, whole batch
    xf = x.clone().requires_grad_(True)
    rf = res.clone().requires_grad_(True) if with_res else None
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_f, rv_f = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    yf = hip.batch_norm_act(xf, wf, bf, rm_f, rv_f, True, 0.1, 1e-5, act, rf)
    yf.backward(g)
    # two uneven ranks
    xs = [x[:cut].contiguous(), x[cut:].contiguous()]
    ress = [res[:cut].contiguous(), res[cut:].contiguous()] if with_res else None
    gs = [g[:cut].contiguous(), g[cut:].contiguous()]
    fake = TwoRanks(hip, 2)
    saved = hip.BN_SYNC_REDUCE
    hip.BN_SYNC_REDUCE = fake
    try:
        _run(hip, fake, 'record', None, xs, ress, w, b, act, gs)                  # forward sums of both ranks
        _run(hip, fake, 'reduce', 'record', xs, ress, w, b, act, gs)              # true forward; backward sums of both ranks
        outs = _run(hip, fake, 'reduce', 'reduce', xs, ress, w, b, act, gs)       # true forward and backward
    finally:
        hip.BN_SYNC_REDUCE = saved
    lo = dtype == torch.bfloat16
    y = torch.cat([o[6] for o in outs]).float()
    dx = torch.cat([o[0].grad for o in outs]).float()
    # identical statistics -> identical elementwise arithmetic; the sums differ only in float64 summation order
    assert (y - yf.float()).abs().max().item() <= (2e-2 if lo else 2e-5)
    assert ((y - yf.float()).abs() > 1e-6).float().mean().item() <= (2e-3 if lo else 1.0)
    d = (dx - xf.grad.float()).abs()
    assert d.max().item() <= (5e-2 if lo else 2e-4)
    for o in outs:                                                                  # every rank holds the GLOBAL running statistics
        assert torch.allclose(o[4], rm_f, atol=1e-6) and torch.allclose(o[5], rv_f, rtol=1e-5, atol=1e-6)
    dw = outs[0][2].grad + outs[1][2].grad
    db = outs[0][3].grad + outs[1][3].grad
    scale = max(1.0, (N * H * W) ** 0.5)
    assert (dw - wf.grad).abs().max().item() <= (0.02 * scale if lo else 2e-3)
    assert (db - bf.grad).abs().max().item() <= (0.02 * scale if lo else 2e-3)
    if with_res:
        dr = torch.cat([o[1].grad for o in outs]).float()
        assert ((dr - rf.grad.float()).abs() > 1e-6).float().mean().item() <= (2e-3 if lo else 1e-6)


def test_sync_bn_single_rank_is_the_plain_kernel(hip):
    """World size 1: the 'all-reduce' is the identity and the synchronised path must reproduce the plain one."""
    torch.manual_seed(0)
    x = torch.randn(5, 12, 24, 24, device="cuda").to(torch.bfloat16)
    w, b = torch.rand(12, device="cuda") + 0.5, torch.randn(12, device="cuda")
    g = torch.randn_like(x)
    outs = []
    saved = hip.BN_SYNC_REDUCE
    hip.BN_SYNC_REDUCE = lambda t: None
    try:
        for sync in (False, True):
            xr = x.clone().requires_grad_(True)
            wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            rm, rv = torch.zeros(12, device="cuda"), torch.ones(12, device="cuda")
            y = hip.batch_norm_act(xr, wr, br, rm, rv, True, 0.1, 1e-5, 1, None, sync=sync)
            y.backward(g)
            outs.append((y, xr.grad, wr.grad, br.grad, rm, rv))
    finally:
        hip.BN_SYNC_REDUCE = saved
    a, s = outs
    assert torch.equal(a[0], s[0]) and torch.equal(a[1], s[1])
    assert torch.allclose(a[2], s[2], rtol=1e-6, atol=1e-6) and torch.allclose(a[3], s[3], rtol=1e-6, atol=1e-6)
    assert torch.equal(a[4], s[4]) and torch.equal(a[5], s[5])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_layers_share_one_all_reduce_per_direction(hip, dtype):
    """Round 3: the five independent BatchNorm layers of the ASPP head travel in ONE all-reduce per direction
    (sync_batch_norm_act_group).  Same outputs, input / parameter gradients and running statistics as five separate synchronised
    layers -- including a member that writes into a channel slice of a concatenation buffer and a [N, C, 1, 1] member --, and the
    collective counter says 2 instead of 10."""
    torch.manual_seed(5)
    shapes = [(4, 16, 8, 8), (4, 16, 8, 8), (4, 24, 8, 8), (4, 16, 1, 1)]
    xs = [(torch.randn(s, device="cuda") * 1.5 + 0.3).to(dtype) for s in shapes]
    ws = [torch.rand(s[1], device="cuda") + 0.5 for s in shapes]
    bs = [torch.randn(s[1], device="cuda") * 0.1 for s in shapes]
    acts = [1, 1, 2, 1]
    gs = [torch.randn(s, device="cuda").to(dtype) for s in shapes]
    calls = []
    hip.BN_SYNC_REDUCE = lambda t: calls.append(t.numel())          # one rank: the sum over the ranks is the local sum
    try:
        def leaves():
            return ([x.detach().clone().requires_grad_(True) for x in xs], [w.detach().clone().requires_grad_(True) for w in ws],
                    [b.detach().clone().requires_grad_(True) for b in bs], [torch.zeros_like(w) for w in ws], [torch.ones_like(w) for w in ws])
        # separate layers
        x1, w1, b1, rm1, rv1 = leaves()
        hip.BN_SYNC_COLLECTIVES[0] = 0
        y1 = [hip.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, a, None, sync=True) for x, w, b, rm, rv, a in zip(x1, w1, b1, rm1, rv1, acts)]
        torch.autograd.backward(y1, gs)
        assert hip.BN_SYNC_COLLECTIVES[0] == 2 * len(shapes)
        # one group; the first two members write into the two halves of a [4, 32, 8, 8] buffer
        x2, w2, b2, rm2, rv2 = leaves()
        buf, parts = hip.concat_slices(4, [16, 16], 8, 8, dtype, xs[0].device)
        outs = [parts[0], parts[1], None, None]
        calls.clear()
        hip.BN_SYNC_COLLECTIVES[0] = 0
        y2 = hip.sync_batch_norm_act_group([(x, w, b, rm, rv, 0.1, 1e-5, a, o) for x, w, b, rm, rv, a, o in zip(x2, w2, b2, rm2, rv2, acts, outs)])
        assert y2[0].data_ptr() == parts[0].data_ptr() and y2[1].data_ptr() == parts[1].data_ptr()
        torch.autograd.backward(list(y2), gs)
        assert hip.BN_SYNC_COLLECTIVES[0] == 2
        assert calls == [sum(2 * s[1] + 1 for s in shapes), sum(2 * s[1] for s in shapes)]
    finally:
        hip.BN_SYNC_REDUCE = None
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for i in range(len(shapes)):
        assert torch.allclose(y1[i].float(), y2[i].float(), atol=tol), i
        assert torch.allclose(x1[i].grad.float(), x2[i].grad.float(), atol=tol), i
        assert torch.allclose(w1[i].grad, w2[i].grad, rtol=1e-4, atol=1e-4) and torch.allclose(b1[i].grad, b2[i].grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(rm1[i], rm2[i], atol=1e-6) and torch.allclose(rv1[i], rv2[i], atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_projection_shortcut_pair_shares_one_all_reduce_per_direction(hip, dtype):
    """relu(bn3(a) + bn_short(b)) -- the tail of a bottleneck with a projection shortcut: `sync_batch_norm_shortcut_pair` gives the
    outputs, gradients (two consumers of the output) and running statistics of two separate synchronised layers with 2 collectives
    instead of 4."""
    torch.manual_seed(11)
    shape = (6, 32, 8, 8)
    a0 = (torch.randn(shape, device="cuda") * 1.3 + 0.2).to(dtype)
    b0 = (torch.randn(shape, device="cuda") * 0.7 - 0.1).to(dtype)
    g0, g1 = torch.randn(shape, device="cuda").to(dtype), torch.randn(shape, device="cuda").to(dtype)
    calls = []
    hip.BN_SYNC_REDUCE = lambda t: calls.append(t.numel())
    try:
        def leaves():
            t = [a0.detach().clone().requires_grad_(True), b0.detach().clone().requires_grad_(True)]
            for k in range(2):
                t += [(torch.rand(32, device="cuda") * 0 + 0.5 + 0.1 * k).requires_grad_(True), torch.full((32,), 0.05 * (k + 1), device="cuda").requires_grad_(True),
                      torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")]
            return t
        a1, b1, wa1, ba1, rma1, rva1, wb1, bb1, rmb1, rvb1 = leaves()
        hip.BN_SYNC_COLLECTIVES[0] = 0
        idt = hip.batch_norm_act(b1, wb1, bb1, rmb1, rvb1, True, 0.1, 1e-5, 0, None, sync=True)
        y1 = hip.batch_norm_act(a1, wa1, ba1, rma1, rva1, True, 0.2, 1e-5, 1, idt, handles=2, sync=True)
        torch.autograd.backward(list(y1), [g0, g1])
        assert hip.BN_SYNC_COLLECTIVES[0] == 4
        a2, b2, wa2, ba2, rma2, rva2, wb2, bb2, rmb2, rvb2 = leaves()
        calls.clear()
        hip.BN_SYNC_COLLECTIVES[0] = 0
        y2 = hip.sync_batch_norm_shortcut_pair(a2, (wa2, ba2, rma2, rva2, 0.2, 1e-5), b2, (wb2, bb2, rmb2, rvb2, 0.1, 1e-5), 1, handles=2)
        torch.autograd.backward(list(y2), [g0, g1])
        assert hip.BN_SYNC_COLLECTIVES[0] == 2 and calls == [2 * (2 * 32 + 1), 4 * 32]
    finally:
        hip.BN_SYNC_REDUCE = None
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(y1[0].float(), y2[0].float(), atol=tol)
    for u, v in ((a1, a2), (b1, b2)):
        assert torch.allclose(u.grad.float(), v.grad.float(), atol=tol)
    for u, v in ((wa1, wa2), (ba1, ba2), (wb1, wb2), (bb1, bb2)):
        assert torch.allclose(u.grad, v.grad, rtol=1e-4, atol=1e-4)
    for u, v in ((rma1, rma2), (rva1, rva2), (rmb1, rmb2), (rvb1, rvb2)):
        assert torch.allclose(u, v, atol=1e-6)


class Jacobi(object):
    """The collective of `n` ranks played in one process for a CHAIN of layers (a layer's local sums are only right once every layer
    before it has been normalised with the global statistics): every pass records this pass's local sums and reduces with the other
    ranks' sums of the PREVIOUS pass; after as many passes as the chain is deep nothing changes any more (fixed point = the real job)."""

    def __init__(self, n):
        self.n, self.prev, self.cur, self.rank, self.call = n, {}, {}, 0, 0

    def start(self, rank):
        self.rank, self.call = rank, 0

    def next_pass(self):
        self.prev, self.cur = self.cur, {}

    def __call__(self, t):
        key, self.call = self.call, self.call + 1
        self.cur[(self.rank, key)] = t.clone()
        for r in range(self.n):
            if r != self.rank and (r, key) in self.prev:
                t += self.prev[(r, key)]


def test_on_load_bottlenecks_keep_their_fusion_under_synchronised_statistics(hip, monkeypatch):
    """Round 6: a ResNet-50 stage in the headline's arithmetic (f32x3) on two "ranks" with synchronised statistics takes the SAME fused
    paths as the single-device step -- bn1 / bn2 + ReLU on the consuming convolution's operand load (batch_norm_lazy(sync=True)), bn3 with
    the projection shortcut's BatchNorm in one pass each way (batch_norm_act_res_bn(sync=True), aadg_bn_sync_backward_res_bn_f32) -- and
    computes what one device computes on the whole batch: outputs, input gradients, running statistics; local parameter gradients add up."""
    import copy
    from aadg_amd.models import deeplab
    torch.manual_seed(5)
    enc = deeplab.ResNet50Encoder()
    stage = enc.layer1.cuda()                                  # 3 bottlenecks, the first with a 64 -> 256 projection shortcut
    for m in stage.modules():
        if isinstance(m, (deeplab.Conv1x1, deeplab.Conv3x3)):
            m.f32x3 = True
    deeplab.mark_bn_producers(stage)
    stage.train()
    x = torch.randn(5, 64, 32, 32, device="cuda")
    g = torch.randn(5, 256, 32, 32, device="cuda")
    cuts = [0, 3, 5]
    seen = {"lazy": 0, "lazy_sync": 0, "pair": 0, "pair_sync": 0}
    lazy0, pair0 = hip.batch_norm_lazy, hip.batch_norm_act_res_bn

    def lazy(*a, **k):
        seen["lazy_sync" if k.get("sync") else "lazy"] += 1
        return lazy0(*a, **k)

    def pair(*a, **k):
        seen["pair_sync" if k.get("sync") else "pair"] += 1
        return pair0(*a, **k)
    monkeypatch.setattr(hip, "batch_norm_lazy", lazy)
    monkeypatch.setattr(hip, "batch_norm_act_res_bn", pair)

    def run(mod, xin, gout):
        xin = xin.detach().clone().requires_grad_(True)
        y = mod(xin)
        y = y[0] if isinstance(y, tuple) else y
        y.backward(gout)
        return y.detach(), xin.grad.detach()
    # one device, whole batch, per-device statistics
    full = copy.deepcopy(stage)
    y_full, dx_full = run(full, x, g)
    assert seen["lazy"] >= 5 and seen["pair"] == 1, seen     # (conv2 of each block + conv3 of each block; the projection block's pair)
    # the yardstick: the SAME single-device pass once more.  The epilogue totals are float64 atomics, so mean / invstd repeat only to
    # their last bit; through nine ReLU layers that is ~6e-6 on the output and, by pre-activations that land on the other side of
    # zero, ~3e-3 on the gradients of two identical passes (scripts/r6/dbg_sync.py)
    again = copy.deepcopy(stage)
    y_again, dx_again = run(again, x, g)
    # two ranks, synchronised statistics
    fake = Jacobi(2)
    deeplab.set_bn_sync(True)
    hip.BN_SYNC_REDUCE = fake
    try:
        # 10 BatchNorm layers in sequence, forward then backward: after 2 x 10 + 2 passes every collective has seen its true totals
        for it in range(24):
            fake.next_pass()
            ranks = [copy.deepcopy(stage) for _ in range(2)]
            outs = []
            for r in range(2):
                fake.start(r)
                outs.append(run(ranks[r], x[cuts[r]:cuts[r + 1]], g[cuts[r]:cuts[r + 1]]))
            y = torch.cat([o[0] for o in outs])
            dx = torch.cat([o[1] for o in outs])
    finally:
        hip.BN_SYNC_REDUCE = None
        deeplab.set_bn_sync(False)
    assert seen["lazy_sync"] >= 5 and seen["pair_sync"] >= 1, seen
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()      # noqa: E731
    # (the yardstick is ONE draw of that noise: the factors and floors leave room for the draw -- a wrong fusion is off by 1e-1 .. 1;
    # 3x / 2e-3 on the gradients failed once in six runs of the whole suite)
    assert rel(y, y_full) < max(5 * rel(y_again, y_full), 5e-5), (rel(y, y_full), rel(y_again, y_full))
    assert rel(dx, dx_full) < max(5 * rel(dx_again, dx_full), 5e-3), (rel(dx, dx_full), rel(dx_again, dx_full))
    for (n, p), pa, p0, p1 in zip(full.named_parameters(), again.parameters(), ranks[0].parameters(), ranks[1].parameters()):
        s = p0.grad + p1.grad                                   # local sums (the reducer averages them; the loss carries the row weights)
        assert rel(s, p.grad) < max(5 * rel(pa.grad, p.grad), 1e-2), (n, rel(s, p.grad), rel(pa.grad, p.grad))
    for (n, b), b0 in zip(full.named_buffers(), ranks[0].buffers()):
        if b.dtype.is_floating_point:
            assert torch.allclose(b, b0, rtol=1e-4, atol=1e-5), n

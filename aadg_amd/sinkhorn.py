"""`SamplesLoss`-shaped callable over the fused Sinkhorn kernel -- drop-in for the reference's

    sinkhorn = SamplesLoss("sinkhorn", cost='( IntCst(1) - (X | Y) / ( Norm2(X) * Norm2(Y) ) )', backend='online')
    dist = sinkhorn(x, y)                                                    (search_dg.py:116,158-160)

Only what the reference uses is implemented: loss "sinkhorn", uniform weights, p = 2, debiased, cosine cost
(the KeOps formula above) -- anything else raises, loudly.  `sinkhorn(x [N,E], y [M,E])` returns a 0-d tensor;
`sinkhorn.pairs(feat, clouds, pairs)` evaluates many problems in ONE launch (what the reward loop needs).
The batched reward entry point is `aadg_amd._lib.sinkhorn_rewards`.
"""
import torch

from . import _lib

COSINE_COST = '( IntCst(1) - (X | Y) / ( Norm2(X) * Norm2(Y) ) )'


class SamplesLoss(object):
    def __init__(self, loss="sinkhorn", p=2, blur=.05, reach=None, diameter=None, scaling=.5, truncate=5, cost=None,
                 kernel=None, cluster_scale=None, debias=True, potentials=False, verbose=False, backend="auto"):
        if loss != "sinkhorn":
            raise NotImplementedError("only loss='sinkhorn' is on the AADG path")
        if p != 2 or reach is not None or diameter is not None or not debias or potentials:
            raise NotImplementedError("only p=2, balanced (reach=None), debiased Sinkhorn divergences are implemented")
        if cost is None or "".join(cost.split()) != "".join(COSINE_COST.split()):
            raise NotImplementedError("only the cosine cost formula used by the reference is implemented: " + COSINE_COST)
        self.blur, self.scaling = float(blur), float(scaling)

    def pairs(self, feat, clouds, pairs):
        """feat [rows,E] float32 GPU; clouds = list of 1-D index tensors/lists; pairs = list of (i, j) cloud ids.
        Returns a float32 tensor [len(pairs)]."""
        dev = feat.device
        sizes = [len(c) for c in clouds]
        rows = torch.cat([torch.as_tensor(c, dtype=torch.int32) for c in clouds]).to(dev)
        off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0).tolist()), dtype=torch.int32, device=dev)
        pxy = torch.tensor([v for ab in pairs for v in ab], dtype=torch.int32, device=dev)
        return _lib.sinkhorn_divergence(feat.contiguous().float(), rows, off, pxy, max(sizes), self.blur, self.scaling)

    def __call__(self, x, y):
        if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
            raise ValueError("expected x [N,E] and y [M,E]")
        feat = torch.cat([x, y], dim=0)
        n, m = x.shape[0], y.shape[0]
        return self.pairs(feat, [list(range(n)), list(range(n, n + m))], [(0, 1)])[0]

"""Domain discriminator producing the 128-d embeddings the Sinkhorn reward compares --
API mirror of models/discriminator.py:20-59 (`MomentumFeatureDiscriminator`) and :5-17
(`FeatureDiscriminator`).  Parameter names (`dis.0`, `fc`, `mom_dis.0`, `mom_fc`) match the
reference so its state_dicts load."""
import torch
import torch.nn as nn


def _embed(in_channels, width=128):
    return nn.Sequential(nn.Linear(in_channels, width), nn.LeakyReLU(0.2, inplace=True))


class FeatureDiscriminator(nn.Module):
    def __init__(self, num_classes, in_channels=1280):
        super().__init__()
        self.dis = nn.Sequential(nn.Linear(in_channels, 128), nn.LeakyReLU(0.2, inplace=True),
                                 nn.Linear(128, num_classes))

    def forward(self, x):
        return self.dis(x)


class MomentumFeatureDiscriminator(nn.Module):
    """Online branch (dis, fc) trained by back-prop; EMA twin (mom_dis, mom_fc), m = 0.999, produces the
    embeddings used for the reward under no_grad."""

    def __init__(self, num_classes, in_channels, m=0.999):
        super().__init__()
        self.m = m
        self.dis = _embed(in_channels)
        self.fc = nn.Linear(128, num_classes)
        self.mom_dis = _embed(in_channels)
        self.mom_fc = nn.Linear(128, num_classes)

    def _pairs(self):
        yield from zip(self.dis.parameters(), self.mom_dis.parameters())
        yield from zip(self.fc.parameters(), self.mom_fc.parameters())

    @torch.no_grad()
    def momentum_update(self):
        for q, k in self._pairs():
            k.data = k.data * self.m + q.data * (1. - self.m)

    @torch.no_grad()
    def synchronize_parameters(self):
        # The reference assigns `param_k.data = param_q.data` (models/discriminator.py:39-44): the EMA
        # twin ALIASES the online weights until the first momentum_update() re-materialises it, so
        # during the first search epoch the "momentum" embeddings track the online branch exactly.
        # Kept as is: cloning here would change the rewards of that epoch.
        for q, k in self._pairs():
            k.data = q.data

    def forward(self, x, momentum=False, return_feature=False, return_norm=False):
        """return_norm (momentum branch on the GPU only): also |fe[n]|_2, a by-product of the fused prologue that the Sinkhorn
        kernel takes as the denominators of its cosine cost (None when the fused kernel did not run)."""
        nrm = None
        if momentum:
            with torch.no_grad():
                if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] <= 4096:
                    # the reward's embedding prologue as one HIP launch (csrc/embed.hip)
                    from .. import _lib
                    lin = self.mom_dis[0]
                    res = _lib.embed_prologue(x, lin.weight, lin.bias, self.mom_fc.weight, self.mom_fc.bias,
                                              self.mom_dis[1].negative_slope, want_norm=return_norm)
                    out, fe = res[0], res[1]
                    nrm = res[2] if return_norm else None
                else:
                    fe = self.mom_dis(x)
                    out = self.mom_fc(fe)
        else:
            fe = self.dis(x)
            out = self.fc(fe)
        if return_norm:
            return out, fe, nrm
        return (out, fe) if return_feature else out

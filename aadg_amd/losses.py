"""Losses on the search path -- API mirror of the reference's losses.py:12-25,52-68,96-157.

`search_loss(cfg)` -> Reinforce | ProximalPolicyOptimization, objects with `.register_optimizer(opt)`
and `__call__(controller, policies, log_probs, entropies, reward) -> (loss, score_loss,
entropy_penalty)`; `task_loss(cfg)` -> BCE on probabilities; `CrossEntropy` = soft-target CE.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def search_loss(config):
    kind = config.CONTROLLER.LOSS
    if kind == 'reinforce':
        return Reinforce(config)
    if kind == 'ppo':
        return ProximalPolicyOptimization(config)
    raise NotImplementedError('{} is unavailable'.format(kind))


def task_loss(config):
    if config.DATASET.NAME in ['optic', 'rvs']:
        return nn.BCELoss()
    raise NotImplementedError('Task loss is unavailable for {}'.format(config.DATASET.NAME))


class CrossEntropy(nn.Module):
    """mean_n( -sum_c t_nc * log_softmax(x)_nc ) for soft targets t (losses.py:52-68)."""

    def __init__(self, reduction='mean'):
        super(CrossEntropy, self).__init__()
        self.reduction = reduction

    def forward(self, input, target):  # pylint: disable=redefined-builtin
        per_class = -target.detach() * F.log_softmax(input, dim=1)
        if self.reduction in ['avg', 'mean']:
            return per_class.sum(dim=1).mean()
        if self.reduction == 'sum':
            return per_class.sum()
        return per_class


class _ControllerLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.penalty = cfg.CONTROLLER.PENALTY
        self.optimizer = None

    def register_optimizer(self, optimizer):
        self.optimizer = optimizer

    def _step(self, loss, retain_graph=False):
        self.optimizer.zero_grad()
        loss.backward(retain_graph=retain_graph)
        self.optimizer.step()


class Reinforce(_ControllerLoss):
    """loss = mean(-log_prob * R) - penalty * mean(entropy); one optimiser step (losses.py:96-114)."""

    def forward(self, controller, policies, log_probs, entropies, reward):
        score_loss = (-log_probs * reward).mean()
        entropy_penalty = entropies.mean()
        loss = score_loss - self.penalty * entropy_penalty
        self._step(loss)
        return loss, score_loss, entropy_penalty


class ProximalPolicyOptimization(_ControllerLoss):
    """Clipped-surrogate PPO, 5 updates per call, clip 0.2; the entropy term is reported but NOT part of
    the optimised loss (losses.py:117-157).  Returned losses are means over the 5 updates."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.clip = 0.2
        self.n_updates_per_iteration = 5

    def forward(self, controller, policies, log_probs, entropies, reward):
        old = log_probs.detach()
        running_loss = 0
        running_score = 0
        entropy_penalty = entropies.mean()
        for _ in range(self.n_updates_per_iteration):
            ratios = torch.exp(controller.evaluate(policies, reward.size(0)) - old)
            clipped = torch.clamp(ratios, 1 - self.clip, 1 + self.clip)
            score_loss = (-torch.min(ratios * reward, clipped * reward)).mean()
            loss = score_loss
            self._step(loss, retain_graph=True)
            running_loss = running_loss + loss
            running_score = running_score + score_loss
        n = self.n_updates_per_iteration
        return running_loss / n, running_score / n, entropy_penalty

"""Policy application -- host side of the hot path; API mirror of the reference's data/policy.py.

Public names and call signatures follow the reference (`parse_policies` data/policy.py:64-97,
`Policy` :6-30, `MultiPolicy` :33-42, `DGMultiPolicy` :45-61) so `search_dg.py:340-341`-style
callers work unchanged.  The difference is what a call does: with `ImageRef` samples
(aadg_amd/data/basic.py) nothing touches pixels here -- each call only *draws* (in the reference's
RNG order) and *records*; one fused GPU launch per batch applies everything later.
"""
import random

import numpy as np

from . import basic
from .basic import *  # noqa: F401,F403  re-exported like the reference module does

_CUTMIX_QUEUE_DEPTH = 10


def _selectable_ops(config, logger=None):
    """augment_list() after CONTROLLER.EXCLUDE_OPS / EXCLUDE_OPS_NUM (data/policy.py:72-83)."""
    ops = basic.augment_list()
    banned = config.CONTROLLER.EXCLUDE_OPS
    if len(banned) > 0:
        if logger:
            logger.info(banned)
        return [entry for entry in ops if entry[0].__name__ not in banned]
    for _ in range(config.CONTROLLER.EXCLUDE_OPS_NUM):
        # the reference re-seeds `random` from numpy's stream times cfg.SEED, shuffles, drops the head
        # and remembers the dropped name in the (list-valued) config entry
        random.seed(np.random.randint(0, 65536) * config.SEED)
        random.shuffle(ops)
        dropped = ops.pop(0)[0].__name__
        config.CONTROLLER.EXCLUDE_OPS.append(dropped)
        if logger:
            logger.info(dropped)
    return ops


def parse_policies(policies, config, logger):
    """Controller output [M, Q*L*2] (op, mag, op, mag, ...) -> M x Q x L nested (op_name, level).

    Indexing law (bit-exact with the reference, data/policy.py:93): element (i, q, l) uses columns
    2L*q + 2l (op index into the filtered op list) and 2L*q + 2l + 1 (magnitude index);
    level = mag_index / (NUM_MAGS - 1) as a float64."""
    names = [entry[0].__name__ for entry in _selectable_ops(config, logger)]
    depth = config.CONTROLLER.L
    denom = config.CONTROLLER.NUM_MAGS - 1
    table = np.asarray(policies)
    n_policies, width = table.shape
    n_sub = width // (2 * depth)
    grid = table[:, :n_sub * 2 * depth].reshape(n_policies, n_sub, depth, 2)
    return [[[(names[grid[i, q, l, 0]], grid[i, q, l, 1] / denom) for l in range(depth)]
             for q in range(n_sub)] for i in range(n_policies)]


class Policy(object):
    """One policy = Q sub-policies of L (op_name, level) pairs; a call applies ONE random sub-policy."""

    def __init__(self, policy):
        self.policy = policy
        self.queue = []
        # (function, value) per step, resolved once: value = level * (hi - lo) + lo exactly as apply_augment
        # computes it (data/basic.py:258-260), so a call only draws and records
        self._compiled = None

    def _compile(self):
        out = []
        for sub in self.policy:
            steps = []
            for name, level in sub:
                if name == 'CutMix':
                    raise KeyError(name)  # not in augment_dict in the reference either (data/basic.py:253)
                fn, low, high = basic.get_augment(name)
                steps.append((fn, level * (high - low) + low))
            out.append(steps)
        return out

    def _cutmix_partner(self, img, mask):
        # The reference maintains a CutMix partner queue even though CutMix is unreachable; its
        # random.choice on every call (until the queue is full) advances the shared RNG stream, so it
        # is reproduced for seed-for-seed parity (data/policy.py:17-21).
        self.queue.append((img, mask))
        if len(self.queue) > _CUTMIX_QUEUE_DEPTH:
            return self.queue.pop(0)
        return random.choice(self.queue)

    def __call__(self, img, mask):
        self._cutmix_partner(img, mask)
        if self._compiled is None:
            self._compiled = self._compile()
        # random.choice(self.policy) in the reference: same draw (an index below len(policy))
        for fn, value in random.choice(self._compiled):
            img, mask = fn(img, mask, value)
        return img, mask


class MultiPolicy(object):
    def __init__(self, policies):
        self.policies = [Policy(p) for p in policies]

    def __call__(self, img):
        return [p(img) for p in self.policies]


class DGMultiPolicy(object):
    """M policies applied to one sample dict; adds 'aug_images' / 'aug_labels' (lists of M)."""

    def __init__(self, policies):
        self.policies = [Policy(p) for p in policies]

    def __call__(self, sample):
        pairs = [p(sample['image'], sample['label']) for p in self.policies]
        sample['aug_images'] = [img for img, _ in pairs]
        sample['aug_labels'] = [lbl for _, lbl in pairs]
        return sample

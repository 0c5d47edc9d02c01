"""Collective helpers -- API mirror of the reference's distributed.py:34-74 over torch.distributed
(backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests), plus the placement law the MI355X design
uses instead of DDP batch splitting.

Placement.  The unit of work is the (domain d, policy j) slice of a batch = B images
(collate rows (b*D + d)*M + j, b < B; search_dg.py:150-157 splits the embeddings exactly that way).
Units are numbered DOMAIN-MAJOR, u = d*M + j, and the unit sequence is cut into G contiguous pieces:

  law 'unit'  whole units per rank, sizes differ by at most one unit (D*M = 18 units: G = 3 -> 6/6/6 = one
              domain per GPU, G = 4 -> 5/5/4/4, G = 8 -> 3/3/2/2/2/2/2/2);
  law 'row'   the same domain-major ROW sequence (u*B + b) cut into G pieces that differ by at most one row
              (G = 3 -> still one domain per GPU; G = 4 -> 36 rows each; G = 8 -> 18 rows each): units may
              be split between two neighbouring ranks, the step time is set by 18 instead of 24 rows at G = 8.

(SURVEY 8e also writes the owner as "(d*M + j) mod G"; taken literally that law puts policies j, j+G, ... of
EVERY domain on one GPU and contradicts "G = 3 => one domain per GPU" in the same sentence, so the contiguous
domain-major cut is what is implemented)

Exchange pattern of one inner iteration on G GPUs:
  1. every rank augments + runs the backbone on its own rows only (every rank draws the SAME batch plan from
     the same seeds, so no plan is communicated),
  2. ONE all-gather of the local momentum embeddings, padded to the largest per-rank row count
     (<= 25 KB per rank: latency-bound, one hop on the fully connected xGMI mesh); `RowPlan.take` puts the
     gathered rows back into collate order, then every rank runs the Sinkhorn kernel redundantly on the full
     [N, 128] matrix, so rewards (and hence the replicated controller) stay bit-identical on all ranks
     without a second collective,
  3. gradient all-reduce of the segmentation model / discriminator: flat 32 MiB buckets of aadg_amd/reducer.py, each issued
     asynchronously from the stream its last member arrived on (the weight-gradient side stream for the convolution weights), on a
     communicator of its own, overlapped with backward.  The reducer averages over ranks; with n_r rows on rank r the local loss is the local mean times n_r*G/N
     (`RowPlan.loss_weight`), so the averaged gradient is the gradient of the mean over all N rows whatever
     the split (count-weighted mean),
  4. with --sync_bn the per-channel BatchNorm sums are all-reduced between the statistics and the
     normalisation kernels (aadg_bn_sync_*), forward and backward.
"""
import numpy as np
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


# ---- a communicator of its own for the small, latency-bound collectives -------------------------------------------------------
# The BatchNorm statistics all-reduces (108 per step, <= 32 KB each, every one between two dependent kernels), the embedding
# all-gather (<= 25 KB per rank) and the policy broadcasts would otherwise queue on the default process group's RCCL stream behind
# whatever 25 MB DDP gradient bucket is in flight there (head-of-line blocking in the backward pass).  `small_group()` is a second
# process group over the same ranks: with the nccl (= RCCL) backend it owns its own communicator and its own stream.  Created lazily
# on first use, collectively (every rank takes the same code path); `USE_SMALL_GROUP = False` (set identically on every rank before the
# first collective) keeps everything on the default group.
USE_SMALL_GROUP = True
_SMALL = {"group": None, "made": False}


def small_group():
    """process group for the small collectives (None = the default group: not distributed, disabled, or creation failed)"""
    if not is_dist() or not USE_SMALL_GROUP:
        return None
    if not _SMALL["made"]:
        _SMALL["made"] = True
        try:
            _SMALL["group"] = dist.new_group(ranks=list(range(dist.get_world_size())), backend=dist.get_backend())
        except Exception as e:  # noqa: BLE001 -- an old backend without sub-groups: stay on the default group
            import sys
            print("aadg_amd.distributed: new_group failed (%r); small collectives stay on the default group" % (e,), file=sys.stderr)
            _SMALL["group"] = None
    return _SMALL["group"]


_SIDE = {"group": None, "made": False}


def side_group():
    """Process group of the collectives issued from the controller's SIDE stream (the embedding all-gather, the policy broadcasts).
    A process group has one communicator and one internal stream, which orders everything issued on it: on the BatchNorm group the
    backward pass's statistics all-reduces (main stream) would queue behind an all-gather that is still waiting for the side stream's
    embedding kernels -- the head-of-line blocking the small group was created to avoid.  So the side-stream collectives get a
    communicator of their own.  Every rank issues the collectives of each group in the same order (the step is the same program on
    every rank), which is what concurrent communicators need.  Same opt-out and fallback as small_group()."""
    if not is_dist() or not USE_SMALL_GROUP:
        return None
    if not _SIDE["made"]:
        _SIDE["made"] = True
        try:
            _SIDE["group"] = dist.new_group(ranks=list(range(dist.get_world_size())), backend=dist.get_backend())
        except Exception as e:  # noqa: BLE001
            import sys
            print("aadg_amd.distributed: new_group failed (%r); side-stream collectives stay on the default group" % (e,), file=sys.stderr)
            _SIDE["group"] = None
    return _SIDE["group"]


_GRAD = {"group": None, "made": False}


def grad_group():
    """Process group of the gradient buckets (aadg_amd/reducer.py): a third communicator, so a 32 MiB bucket in flight never stands in
    front of a BatchNorm statistics all-reduce (small_group, main stream) or the embedding all-gather (side_group, controller stream).
    The buckets are issued from the weight-gradient stream / the backward chain's stream and wait for nothing but their members."""
    if not is_dist() or not USE_SMALL_GROUP:
        return None
    if not _GRAD["made"]:
        _GRAD["made"] = True
        try:
            _GRAD["group"] = dist.new_group(ranks=list(range(dist.get_world_size())), backend=dist.get_backend())
        except Exception as e:  # noqa: BLE001
            import sys
            print("aadg_amd.distributed: new_group failed (%r); gradient buckets stay on the default group" % (e,), file=sys.stderr)
            _GRAD["group"] = None
    return _GRAD["group"]


def reset_groups():
    """forget the cached groups (after destroy_process_group; tests that re-initialise the process group in one interpreter)"""
    _SMALL["group"], _SMALL["made"] = None, False
    _SIDE["group"], _SIDE["made"] = None, False
    _GRAD["group"], _GRAD["made"] = None, False


class _CollectiveTimer(object):
    """GPU-side duration of the small collectives (HIP events on the issuing stream around each call: what a dependent kernel waits
    for, including the wait for the slowest peer).  Off by default; bench.py switches it on for a few extra steps AFTER the timed
    region of a multi-rank run, so the headline is not perturbed by ~220 event records per step."""

    def __init__(self):
        self.enabled, self.pairs = False, []

    def run(self, kind, fn, *args, **kw):
        if not (self.enabled and torch.cuda.is_available()):
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        self.pairs.append((kind, e0, e1))
        return out

    def drain(self):
        """-> {kind: (count, total ms)} of the bracketed collectives since the last drain (synchronises the device)"""
        torch.cuda.synchronize()
        res = {}
        for kind, e0, e1 in self.pairs:
            c, t = res.get(kind, (0, 0.0))
            res[kind] = (c + 1, t + e0.elapsed_time(e1))
        self.pairs = []
        return res


COLLECTIVE_TIMER = _CollectiveTimer()


def small_all_reduce(t, kind="all_reduce"):
    """in-place SUM of a small tensor over the ranks on the small-collectives group"""
    return COLLECTIVE_TIMER.run(kind, dist.all_reduce, t, group=small_group())


def small_broadcast(t, src=0, kind="broadcast"):
    # the policy broadcasts are issued from the controller's side stream: their own communicator (side_group)
    return COLLECTIVE_TIMER.run(kind, dist.broadcast, t, src, group=side_group() if kind == "policy_broadcast" else small_group())


def all_gather(tensors, group=None):
    """list of tensors -> list of tensors concatenated over ranks on dim 0 (distributed.py:34-54).
    Uses one all_gather_into_tensor per entry (a single contiguous receive buffer, no per-rank list)."""
    if not is_dist():
        return [t.clone() for t in tensors]
    ws = dist.get_world_size()
    out = []
    for t in tensors:
        t = t.contiguous()
        buf = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if hasattr(dist, "all_gather_into_tensor") and dist.get_backend() != "gloo":
            COLLECTIVE_TIMER.run("all_gather", dist.all_gather_into_tensor, buf, t, group=group)
        else:
            COLLECTIVE_TIMER.run("all_gather", dist.all_gather, list(buf.chunk(ws, dim=0)), t, group=group)
        out.append(buf)
    return out


def all_reduce(tensors, average=True):
    """in-place sum (optionally mean) over ranks (distributed.py:57-74)."""
    if not is_dist():
        return tensors
    for t in tensors:
        dist.all_reduce(t)
    if average:
        inv = 1.0 / dist.get_world_size()
        for t in tensors:
            t.mul_(inv)
    return tensors


def describe():
    """what a reader of a bench line needs to know about the process group (bench.py: config.distributed)"""
    info = {"initialized": bool(is_dist()), "world_size": 1, "rank": 0, "backend": None, "rccl_version": None, "small_collectives_group": None}
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # noqa: BLE001 -- CPU-only build
        pass
    if is_dist():
        info.update(world_size=dist.get_world_size(), rank=dist.get_rank(), backend=str(dist.get_backend()))
        g = small_group()
        info["small_collectives_group"] = "own process group (own RCCL communicator + stream)" if g is not None else "default group"
        info["side_stream_collectives_group"] = ("own process group (embedding all-gather, policy broadcasts: issued from the controller's "
                                                 "stream)" if side_group() is not None else "default group")
        info["gradient_buckets_group"] = "own process group (aadg_amd.reducer.GradReducer)" if grad_group() is not None else "default group"
    return info


def balanced_cuts(n, parts):
    """parts + 1 cut points of range(n) into contiguous pieces whose sizes differ by at most one (larger first)."""
    base, extra = divmod(n, parts)
    cuts = [0]
    for r in range(parts):
        cuts.append(cuts[-1] + base + (1 if r < extra else 0))
    return cuts


def shard_rows(n_rows, rank=None, world_size=None):
    """Contiguous balanced [lo, hi) slice of `n_rows` rows for `rank` (used for the un-augmented images of a batch,
    which only feed the warm-up epochs)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    if n_rows < world_size:
        # an empty slice would give that rank a mean over zero images: NaN, which DDP then all-reduces into every replica
        raise ValueError("%d rows cannot be cut over %d ranks (every rank needs at least one row): raise TRAIN.BATCH_SIZE or use "
                         "fewer GPUs" % (n_rows, world_size))
    cuts = balanced_cuts(n_rows, world_size)
    return cuts[rank], cuts[rank + 1]


class RowPlan(object):
    """Which collate rows of the N = D*B*M augmented images this rank materialises, and how the all-gathered
    per-rank results go back into collate order.

    rows        int64 [n_local]  collate-row indices owned by this rank, in local (materialisation) order
    counts      per-rank row counts (every rank knows all of them: the plan needs no communication)
    take        int64 [N]        index into the flattened padded gather buffer [G * max(counts)]: gathered[take] is in
                                 collate order (row (b*D + d)*M + j)
    loss_weight n_local * G / N  factor on the local-mean loss so that DDP's average over ranks is the global mean
    """

    def __init__(self, D, B, M, rank=0, world_size=1, law='unit', force=False):
        if law not in ('unit', 'row'):
            raise ValueError("placement law must be 'unit' or 'row'")
        self.D, self.B, self.M, self.rank, self.world, self.law = D, B, M, int(rank), int(world_size), law
        N = D * B * M
        self.n_rows = N
        # force: a one-rank job that still takes the sharded code path (domain-major local order, padded all-gather,
        # index_select back into collate order) -- bench.py --force_dist: the first RCCL execution of this path on a one-GPU box
        self.force = bool(force)
        if self.world == 1 and not self.force:
            # single GPU: collate order itself (the reference's row law), nothing to gather
            self.order = np.arange(N, dtype=np.int64)
            self.cuts = [0, N]
        else:
            d, j, b = np.meshgrid(np.arange(D), np.arange(M), np.arange(B), indexing='ij')
            self.order = ((b * D + d) * M + j).reshape(-1).astype(np.int64)      # domain-major: position u*B + b
            if law == 'unit':
                if self.world > D * M:
                    raise ValueError("%d ranks for %d (domain, policy) units: use the 'row' law" % (self.world, D * M))
                self.cuts = [c * B for c in balanced_cuts(D * M, self.world)]
            else:
                if self.world > N:
                    raise ValueError("%d ranks for %d rows" % (self.world, N))
                self.cuts = balanced_cuts(N, self.world)
        self.counts = [self.cuts[r + 1] - self.cuts[r] for r in range(self.world)]
        self.max_count = max(self.counts)
        self.rows = self.order[self.cuts[self.rank]:self.cuts[self.rank + 1]]
        self.n_local = int(self.rows.shape[0])
        self.loss_weight = self.n_local * self.world / float(N)
        # gathered buffer position of domain-major position p = (owner r, offset p - cuts[r]) -> r * max_count + offset
        pos = np.empty(N, dtype=np.int64)
        for r in range(self.world):
            pos[self.cuts[r]:self.cuts[r + 1]] = r * self.max_count + np.arange(self.counts[r])
        take = np.empty(N, dtype=np.int64)
        take[self.order] = pos
        self.take = take
        self._dev = {}

    @property
    def sharded(self):
        return self.world > 1 or self.force

    def units(self, rank=None):
        """(domain, policy) units touched by `rank` (whole units under the 'unit' law)."""
        r = self.rank if rank is None else rank
        lo, hi = self.cuts[r], self.cuts[r + 1]
        if self.world == 1:
            return [(d, j) for d in range(self.D) for j in range(self.M)]
        return sorted({(int(p // self.B) // self.M, int(p // self.B) % self.M) for p in range(lo, hi)})

    def on(self, device):
        """(rows, take) as device tensors (cached per device)."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.rows).to(device), torch.from_numpy(self.take).to(device))
        return self._dev[key]

    def gather(self, local, emulate=False):
        """local [n_local, E] -> [N, E] in collate order on every rank: pad to the largest per-rank count, ONE all-gather,
        one index_select.  `emulate` (single process standing in for one rank of a G-rank job, bench.py --shard_of):
        the missing peers' rows are stood in by repetition of the local ones -- shapes and kernels as in the real job."""
        if not self.sharded:
            return local
        local = local.contiguous()
        _, take = self.on(local.device)
        if self.n_local < self.max_count:
            pad = local.new_zeros((self.max_count - self.n_local,) + tuple(local.shape[1:]))
            local = torch.cat([local, pad], dim=0)
        if emulate:
            if is_dist():
                all_gather([local], group=side_group())          # the call a rank of the real job makes (bench.py --shard_of G --force_dist)
            flat = local.repeat((self.world,) + (1,) * (local.dim() - 1))
            if self.n_local < self.max_count:        # never read a padding row: fold the index into the valid range
                r, o = take // self.max_count, take % self.max_count
                take = r * self.max_count + o % self.n_local
        else:
            flat = all_gather([local], group=side_group())[0]
        return flat.index_select(0, take)

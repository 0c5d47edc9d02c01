"""Gradient all-reduce of the data-parallel replicas, owned by this package (reference: models/__init__.py:39,104,165 wrap the modules
in torch's DistributedDataParallel; distributed.py:57-74 is the list all-reduce helper).

Why not DistributedDataParallel.  Its reducer listens to autograd's AccumulateGrad hooks and owns the stream discipline of the
gradients, so the weight-gradient kernels of the own convolutions -- which run on a second HIP stream beside the backward chain
(`_lib.set_wgrad_stream`) and bypass AccumulateGrad -- had to be switched off under it.  `GradReducer` is the same contract (rank 0's
parameters and buffers at construction, gradients averaged over the ranks when backward() returns, `.module` underneath) built for
that design:

  * the gradients live in a few flat float32 BUCKETS (default 32 MiB: xGMI rings are per-link bound, so few large messages);
    `param.grad` is a view into its bucket, the optimizer steps on the views;
  * a gradient ARRIVES either through a post-accumulate hook (BatchNorm / depthwise / bias parameters and every library layer: on the
    stream autograd ran the node on) or straight from `_lib._wgrad_beside` (the matrix-core weight gradients: on the SIDE stream, copied
    into the bucket there).  When the last member of a bucket has arrived, `all_reduce(async_op=True)` is issued from the arriving
    stream on a communicator of its own (`distributed.grad_group()`), i.e. a side-stream bucket never waits for the backward chain
    and the chain never waits for a bucket;
  * buckets are launched strictly in index order (every rank issues the same sequence of collectives whatever its arrival order);
    the order itself is the arrival order observed in the first backward pass (rebuilt once, as DDP does);
  * a callback on the autograd engine at the end of the backward pass launches what is incomplete (parameters that took no part get a
    zero contribution), makes the caller's stream wait for every bucket and scales by 1 / world -- so `loss.backward();
    optimizer.step()` reads averaged gradients exactly as under DDP (count-weighted mean: `RowPlan.loss_weight` on the loss).

Works on CPU tensors over gloo too (no streams there): tests/test_dist_cpu.py.

Several backward passes before one optimizer step accumulate as under DDP without `no_sync()`: every pass ends with its own all-reduce; a
gradient that is already in its bucket (the average of the earlier passes, equal on every rank) goes through the sum / world unchanged
and the new local contribution is added to it before the bucket leaves (tests/dist_worker.py: check_reducer, the two-pass step).  Every
rank must run the same program (the same set of parameters takes part in a pass on every rank).
"""
import torch
import torch.distributed as dist

from . import distributed as adist


class _Bucket(object):
    __slots__ = ("flat", "params", "views", "pending", "arrived", "bypassed", "streams", "work", "launched")

    def __init__(self, params, device):
        self.params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views, o = [], 0
        for p in params:
            self.views.append(self.flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        self.reset()

    def reset(self):
        self.pending = len(self.params)
        self.arrived = [False] * len(self.params)
        self.bypassed = [False] * len(self.params)
        self.streams = []
        self.work = None
        self.launched = False


class GradReducer(torch.nn.Module):
    """module -> data-parallel replica: see the module docstring.  `group`: process group of the gradient buckets (default:
    distributed.grad_group()); `bucket_bytes`: capacity of a bucket; `broadcast_buffers`: rank 0's buffers (BatchNorm running
    statistics) before every training forward, as DDP's default (off when the statistics are synchronised: identical everywhere)."""

    def __init__(self, module, group=None, bucket_bytes=32 << 20, broadcast_buffers=True):
        super().__init__()
        if not adist.is_dist():
            raise RuntimeError("GradReducer needs an initialised process group")
        self.module = module
        self.group = group if group is not None else adist.grad_group()
        self.world = dist.get_world_size(self.group)
        self.bucket_bytes = int(bucket_bytes)
        self.broadcast_buffers = bool(broadcast_buffers)
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._names = {id(p): n for n, p in module.named_parameters()}
        if not self._params:
            raise ValueError("GradReducer: the module has no parameter that requires a gradient")
        if any(p.dtype != torch.float32 for p in self._params):
            raise TypeError("GradReducer: float32 parameters only")
        self.device = self._params[0].device
        # rank 0's state everywhere (DDP does this at construction)
        self._broadcast_coalesced([p.data for p in module.parameters()] + [b.data for b in module.buffers()])
        self._buffers_dirty = False
        self._build(list(reversed(self._params)))
        self._rebuilt = False
        self._order = []                     # arrival order of the first backward pass
        self._task = -1                      # autograd graph task the current state belongs to
        self._next = 0                       # index of the next bucket to launch
        self.stats = {"buckets": len(self._buckets), "bytes": [b.flat.numel() * 4 for b in self._buckets], "launches": 0,
                      "side_stream_arrivals": 0, "hook_arrivals": 0}
        for p in self._params:
            p._aadg_grad_sink = self
            p.register_post_accumulate_grad_hook(self._hook)

    # ---- construction ----------------------------------------------------------------------------------------------------------
    def _build(self, order):
        self._buckets, self._slot = [], {}
        cur, size = [], 0
        for p in order:
            if cur and size + p.numel() * 4 > self.bucket_bytes:
                self._buckets.append(_Bucket(cur, self.device))
                cur, size = [], 0
            cur.append(p)
            size += p.numel() * 4
        if cur:
            self._buckets.append(_Bucket(cur, self.device))
        for bi, b in enumerate(self._buckets):
            for k, p in enumerate(b.params):
                self._slot[id(p)] = (bi, k)

    def _broadcast_coalesced(self, tensors):
        by_type = {}
        for t in tensors:
            by_type.setdefault((t.dtype, t.device), []).append(t)
        for ts in by_type.values():
            flat = torch.cat([t.reshape(-1) for t in ts]) if len(ts) > 1 else ts[0].reshape(-1).clone()
            dist.broadcast(flat, 0, group=self.group)
            o = 0
            for t in ts:
                t.copy_(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()

    # ---- forward ---------------------------------------------------------------------------------------------------------------
    def forward(self, *args, **kw):
        if self.broadcast_buffers and (self.module.training or self._buffers_dirty):
            bufs = [b.data for b in self.module.buffers()]
            if bufs:
                with torch.no_grad():
                    self._broadcast_coalesced(bufs)
            self._buffers_dirty = self.module.training
        return self.module(*args, **kw)

    # ---- arrivals --------------------------------------------------------------------------------------------------------------
    def _enter(self):
        """first arrival of a backward pass: drop whatever a pass cut short by an exception left behind, queue the finaliser"""
        task = torch._C._current_graph_task_id()
        if task == self._task and task != -1:
            return
        if task == -1:
            raise RuntimeError("GradReducer: gradient delivered outside a backward pass")
        self._task = task
        self._next = 0
        for b in self._buckets:
            b.reset()
        torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _hook(self, param):
        """post-accumulate hook: param.grad holds this pass's gradient (a fresh tensor after zero_grad(set_to_none=True), or the
        bucket view itself after zero_grad(set_to_none=False))"""
        g = param.grad
        if g is None:
            return                               # AccumulateGrad ran on an undefined gradient (torch still calls the hook): no arrival
        bi, k = self._slot[id(param)]
        if self._task == torch._C._current_graph_task_id() and self._buckets[bi].bypassed[k]:
            return                               # its gradient came in from the weight-gradient stream; the node only saw `None`
        self.stats["hook_arrivals"] += 1
        self._arrive(param, g)

    def deliver(self, param, grad):
        """from _lib._wgrad_beside, on the side stream: `grad` was produced on the CURRENT stream and bypasses AccumulateGrad"""
        self.stats["side_stream_arrivals"] += 1
        if grad.shape != param.shape:
            grad = grad.reshape(param.shape)
        self._arrive(param, grad, bypass=True)

    def _arrive(self, param, grad, bypass=False):
        self._enter()
        bi, k = self._slot[id(param)]
        b = self._buckets[bi]
        view = b.views[k]
        if b.arrived[k]:
            if b.launched:
                raise RuntimeError("GradReducer: the gradient of %s arrived twice in one backward pass (now %s) after its bucket left "
                                   "(a weight shared between two layers?)" % (self._names.get(id(param)), "from the weight-gradient stream" if bypass else "through AccumulateGrad"))
            view.add_(grad)
            return
        if bypass and param.grad is not None and param.grad.data_ptr() == view.data_ptr():
            view.add_(grad)                       # .grad was zeroed in place (or holds an earlier pass): what AccumulateGrad would do
        elif grad.data_ptr() != view.data_ptr():
            view.copy_(grad)
        param.grad = view
        b.arrived[k] = True
        b.bypassed[k] = bypass
        b.pending -= 1
        if not self._rebuilt:
            self._order.append(param)
        if view.is_cuda:
            s = torch.cuda.current_stream(view.device)
            if not any(s == t for t in b.streams):
                b.streams.append(s)
        self._launch_ready()

    def _launch_ready(self):
        while self._next < len(self._buckets) and self._buckets[self._next].pending == 0:
            self._launch(self._buckets[self._next])
            self._next += 1

    def _launch(self, b):
        if b.flat.is_cuda:
            cur = torch.cuda.current_stream(b.flat.device)
            for s in b.streams:
                if s != cur:
                    cur.wait_stream(s)          # members written on another stream (the chain's, or the weight-gradient stream)
        b.work = dist.all_reduce(b.flat, group=self.group, async_op=True)
        b.launched = True
        self.stats["launches"] += 1

    # ---- end of the backward pass ----------------------------------------------------------------------------------------------
    def _finalize(self):
        if self._task == -1:
            return
        for b in self._buckets[self._next:]:
            for k, done in enumerate(b.arrived):
                # took no part in this pass.  If .grad IS the view it holds what an earlier pass (or an in-place zero_grad) left there,
                # the same on every rank: sum / world gives it back, as under DDP.  Otherwise the slot is scratch: contribute zero
                if not done and not (b.params[k].grad is not None and b.params[k].grad.data_ptr() == b.views[k].data_ptr()):
                    b.views[k].zero_()
            self._launch(b)
        self._next = len(self._buckets)
        inv = 1.0 / self.world
        for b in self._buckets:
            b.work.wait()                        # the caller's stream waits for the collective (host-blocking only on gloo)
            if self.world > 1:
                b.flat.mul_(inv)
        self._task = -1
        if not self._rebuilt:
            # the order the gradients really arrive in (side-stream weights and hook-delivered parameters interleaved): buckets fill,
            # and leave, in that order from the second pass on.  Same graph on every rank -> same order on every rank.
            self._rebuilt = True
            seen = set(id(p) for p in self._order)
            order = self._order + [p for p in reversed(self._params) if id(p) not in seen]
            old = {id(p): p.grad for p in self._params}
            self._build(order)
            for p in self._params:
                if old[id(p)] is not None:
                    bi, k = self._slot[id(p)]
                    self._buckets[bi].views[k].copy_(old[id(p)])
                    p.grad = self._buckets[bi].views[k]
            self._order = []
            self.stats["buckets"] = len(self._buckets)
            self.stats["bytes"] = [b.flat.numel() * 4 for b in self._buckets]

    def describe(self):
        return {"buckets": self.stats["buckets"], "bucket_bytes": self.stats["bytes"], "capacity": self.bucket_bytes,
                "world": self.world, "group": "own process group" if self.group is not None else "default group"}

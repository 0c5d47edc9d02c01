"""Stream policy of the weight gradients: the own convolutions' weight-gradient kernels run on a second HIP stream beside the backward chain and reach `.grad`
(or a data-parallel replica's gradient bucket, aadg_amd/reducer.py) from there."""
import ctypes
import os

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# Weight gradients BESIDE the backward chain.  The input gradient of a convolution feeds the next BatchNorm backward of the chain; its
# weight gradient feeds nothing until the optimizer step.  With set_wgrad_stream(True) the convolution Functions below launch their
# weight-gradient kernels (matrix-core bound, 147 KB of LDS: one workgroup per CU) on a second HIP stream, where they overlap the
# chain's BatchNorm passes (HBM bound, no LDS) instead of standing in line with them, and hand the result over at the END of the
# backward pass: a callback queued on the autograd engine makes the launch stream wait for the side stream and puts the gradients
# into `.grad` (what AccumulateGrad would have done).  The gradients therefore bypass autograd's accumulation hooks: NOT for modules
# wrapped in torch's DistributedDataParallel (its reducer listens to those hooks) and not for torch.autograd.grad(); the package's own
# data-parallel wrapper (aadg_amd/reducer.py) takes them over on the side stream instead.  Off by default.
_WG = {"on": False, "stream": None, "pending": []}


def set_wgrad_stream(flag):
    """Weight-gradient kernels of the own convolutions on a side stream, gradients written to `.grad` at the end of the backward pass
    (not under torch's DistributedDataParallel; aadg_amd.reducer.GradReducer is built for it).  Returns the previous setting."""
    old = _WG["on"]
    _WG["on"] = bool(flag)
    return old


def wgrad_stream_enabled():
    return _WG["on"]


def _flush_wgrads():
    side = _WG["stream"]
    pending, _WG["pending"] = _WG["pending"], []
    if not pending:
        return
    task = torch._C._current_graph_task_id()
    main = torch.cuda.current_stream()
    main.wait_stream(side)
    for stamp, weight, dw in pending:
        if stamp != task:
            continue            # left behind by a backward pass that raised before its callbacks ran: not this pass's gradient
        dw.record_stream(main)
        if dw.shape != weight.shape or dw.stride() != weight.stride():
            dw = dw.reshape(weight.shape).contiguous()           # the layout AccumulateGrad would have given it
        if weight.grad is None:
            weight.grad = dw
        else:
            weight.grad.add_(dw)


def _wgrad_beside(weight, fn, *reads):
    """dw = fn() for the parameter `weight`, reading the tensors `reads` (produced on the current stream).  Side stream off (or `weight`
    is no leaf that accumulates into .grad): runs fn() here and returns dw.  On: launches fn() on the side stream and returns None --
    the gradient reaches weight.grad in _flush_wgrads() when the backward pass ends, or, for a parameter of a data-parallel replica
    (aadg_amd/reducer.py: `weight._aadg_grad_sink`), goes into its gradient bucket ON the side stream, from where the bucket's
    all-reduce is issued as soon as its last member is in."""
    if not (_WG["on"] and weight.is_leaf and weight.requires_grad and weight.is_cuda):
        return fn()
    if _WG["stream"] is None:
        _WG["stream"] = torch.cuda.Stream(device=weight.device)
    side = _WG["stream"]
    sink = getattr(weight, "_aadg_grad_sink", None)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dw = fn()
        if sink is not None:
            sink.deliver(weight, dw)
    for t in reads:
        t.record_stream(side)                 # the caching allocator must not hand these out again before the side stream has read them
    if sink is not None:
        return None
    _WG["pending"].append((torch._C._current_graph_task_id(), weight, dw))
    # one callback per deferred gradient (the first to run delivers everything pending, the others find nothing)
    torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)
    return None

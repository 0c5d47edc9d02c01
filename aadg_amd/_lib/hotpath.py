"""The rest of the hot path: Sinkhorn rewards (csrc/sinkhorn*.hip; search_dg.py:150-162), per-policy BCE + Dice (csrc/seg_loss.hip; search_dg.py:140-165), the
float tensor ops of data/functional.py (csrc/tensor_ops.hip), the fused controller calls (csrc/controller.hip; models/controller.py:73-145, losses.py:117-157) and the
discriminator's embedding prologue (csrc/embed.hip; models/discriminator.py:46-59)."""
import ctypes
import os

import numpy as np
import torch

from .binding import AadgError, _check, _ptr, _ptr_array, _require_cuda, _stream, _zeroed_workspace, load, workspace


# ------------------------------------------------------------------------------------------------
def sinkhorn_rewards(fe, D, B, M, blur=0.05, scaling=0.5, rewards=None, row_norm=None):
    """fe f32 [D*B*M, E] in collate order (row (b*D+d)*M + j). rewards[j] += sum_pairs S(x_d1, x_d2).
    row_norm (optional, f32 [D*B*M]): |fe[n]| from embed_prologue(..., want_norm=True); the kernel then skips its norm pass."""
    lib = load()
    _require_cuda(fe, rewards, row_norm)
    if fe.dtype != torch.float32 or fe.dim() != 2 or fe.shape[0] != D * B * M:
        raise AadgError("fe must be float32 [D*B*M, E]")
    if rewards is None:
        rewards = torch.zeros(M, dtype=torch.float32, device=fe.device)
    P = D * (D - 1) // 2
    nb = lib.aadg_sinkhorn_workspace_bytes(M * P, B, fe.shape[1])
    ws = workspace(nb, fe.device, "sinkhorn")
    if row_norm is not None:
        if row_norm.dtype != torch.float32 or row_norm.numel() != fe.shape[0]:
            raise AadgError("row_norm must be float32 [D*B*M]")
        rc = lib.aadg_sinkhorn_rewards_norm_f32(fe.data_ptr(), row_norm.data_ptr(), D, B, M, fe.shape[1], blur, scaling,
                                                rewards.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    else:
        rc = lib.aadg_sinkhorn_rewards_f32(fe.data_ptr(), D, B, M, fe.shape[1], blur, scaling, rewards.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_sinkhorn_rewards_f32")
    return rewards


def sinkhorn_divergence(feat, cloud_rows, cloud_off, prob_xy, max_cloud, blur=0.05, scaling=0.5):
    """General form: index tables (int32 device tensors) into feat f32 [rows, E]; returns f32 [n_prob]."""
    lib = load()
    _require_cuda(feat, cloud_rows, cloud_off, prob_xy)
    if feat.dtype != torch.float32 or feat.dim() != 2:
        raise AadgError("feat must be float32 [rows, E]")
    for t in (cloud_rows, cloud_off, prob_xy):
        if t.dtype != torch.int32:
            raise AadgError("index tables must be int32")
    n_prob = prob_xy.numel() // 2
    out = torch.empty(n_prob, dtype=torch.float32, device=feat.device)
    nb = lib.aadg_sinkhorn_workspace_bytes(n_prob, int(max_cloud), feat.shape[1])
    ws = workspace(nb, feat.device, "sinkhorn")
    rc = lib.aadg_sinkhorn_divergence_f32(feat.data_ptr(), feat.stride(0), feat.shape[1], cloud_rows.data_ptr(),
                                          cloud_off.data_ptr(), prob_xy.data_ptr(), n_prob, int(max_cloud), blur,
                                          scaling, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_sinkhorn_divergence_f32")
    return out


def sinkhorn_divergence_phases(feat, cloud_rows, cloud_off, prob_xy, max_cloud, phases, blur=0.05, scaling=0.5, out=None):
    """Measurement (bench.py): the large-cloud path in halves -- phases 1 = cost build into the workspace, 2 = sweeps over it + result,
    3 = both."""
    lib = load()
    _require_cuda(feat, cloud_rows, cloud_off, prob_xy)
    n_prob = prob_xy.numel() // 2
    if out is None:
        out = torch.empty(n_prob, dtype=torch.float32, device=feat.device)
    nb = lib.aadg_sinkhorn_workspace_bytes(n_prob, int(max_cloud), feat.shape[1])
    ws = workspace(nb, feat.device, "sinkhorn")
    _check(lib.aadg_sinkhorn_divergence_phases_f32(feat.data_ptr(), feat.stride(0), feat.shape[1], cloud_rows.data_ptr(), cloud_off.data_ptr(),
                                                   prob_xy.data_ptr(), n_prob, int(max_cloud), blur, scaling, out.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), int(phases), _stream()), "aadg_sinkhorn_divergence_phases_f32")
    return out


def normalize_rewards(rewards):
    lib = load()
    _require_cuda(rewards)
    out = torch.empty_like(rewards)
    rc = lib.aadg_normalize_rewards_f32(rewards.data_ptr(), rewards.numel(), out.data_ptr(), _stream())
    _check(rc, "aadg_normalize_rewards_f32")
    return out


# ------------------------------------------------------------------------------------------------
def seg_bce_dice(logits, labels, M, want_grad=False, grad_scale=1.0):
    """logits/labels f32 [N,K,H,W] -> (bce [M], dice [K], grad or None); grad = d(grad_scale * mean_j bce_j)/d logits."""
    lib = load()
    _require_cuda(logits, labels)
    if logits.dtype != torch.float32 or labels.dtype != torch.float32 or logits.shape != labels.shape or logits.dim() < 3:
        raise AadgError("logits and labels must be float32 tensors of the same [N,K,...] shape")
    N, K = logits.shape[:2]
    HW = logits[0, 0].numel()
    if N % M:
        raise AadgError("N must be a multiple of M")
    bce = torch.empty(M, dtype=torch.float32, device=logits.device)
    dice = torch.empty(K, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    nb = lib.aadg_seg_loss_workspace_bytes(N, K, HW)
    ws = _zeroed_workspace(nb, logits.device, "segloss")         # accumulators + arrival counter: zero on entry, left zeroed by the kernel
    rc = lib.aadg_seg_bce_dice_scaled_f32(logits.data_ptr(), labels.data_ptr(), N, K, HW, M, float(grad_scale), bce.data_ptr(),
                                          dice.data_ptr(), _ptr(grad), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_seg_bce_dice_scaled_f32")
    return bce, dice, grad


class _PolicyBCE(torch.autograd.Function):
    """loss = mean_j BCE(sigmoid(z)[j::M], y[j::M]); forward and backward share one fused pass."""

    @staticmethod
    def forward(ctx, logits, labels, M):
        bce, dice, grad = seg_bce_dice(logits.contiguous(), labels.contiguous(), M, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(dice)
        return bce.mean(), bce.detach(), dice

    @staticmethod
    def backward(ctx, g_loss, g_bce, g_dice):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None


def policy_bce_loss(logits, labels, M):
    """Drop-in for search_dg.py:140-142 (+ the Dice monitor of :164-165): returns (seg_loss, bce[M], dice[K])."""
    return _PolicyBCE.apply(logits, labels, M)


def policy_bce_backward(logits, labels, M, scale=1.0):
    """The segmentation loss of search_dg.py:140-142 AND its backward pass in one call: computes scale * mean_j BCE_j, the Dice monitor
    and d loss / d logits in ONE fused pass over logits / labels, then starts the backward pass of the graph behind `logits` from
    that gradient (`logits.backward(grad)`).  Equivalent to `(scale * policy_bce_loss(...)[0]).backward()` without autograd's
    `grad * d loss` product -- a read + write of the whole [N,K,H,W] gradient (0.6 GB at 144 x 2 x 512 x 512) that multiplied it by
    a scalar the kernel can apply itself.  Returns (scale * loss (detached), bce [M], dice [K])."""
    z = logits if logits.dtype == torch.float32 else logits.float()         # a differentiable cast under autocast
    z = z if z.is_contiguous() else z.contiguous()
    bce, dice, grad = seg_bce_dice(z.detach(), labels.contiguous(), M, want_grad=True, grad_scale=scale)
    if z.requires_grad:
        z.backward(grad)
    return bce.mean() * scale, bce, dice


# ------------------------------------------------------------------------------------------------
FOP = {name: i for i, name in enumerate([
    "invert", "solarize", "posterize", "gray", "contrast", "auto_contrast", "saturate", "brightness", "hue",
    "sample_pairing", "equalize", "sharpness", "gaussian_blur3x3", "shear_x", "shear_y", "translate_x",
    "translate_y", "rotate", "hflip", "vflip"])}


def fop(name, img, mag=None, kernel=None, perm=None):
    """One float tensor op of data/functional.py on a [B,3,H,W] float32 GPU tensor (output clamped to [0,1])."""
    lib = load()
    _require_cuda(img, mag, kernel, perm)
    if img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3:
        raise AadgError("img must be float32 [B,3,H,W]")
    B, C, H, W = img.shape
    out = torch.empty_like(img)
    mag_n = 0
    if mag is not None:
        mag = mag.to(torch.float32).reshape(-1).contiguous()
        mag_n = mag.numel()
    if kernel is not None:
        kernel = kernel.to(torch.float32).reshape(-1).contiguous()
        if kernel.numel() != 9:
            raise AadgError("kernel must be 3x3")
    if perm is not None:
        perm = perm.to(torch.int32).contiguous()
    nb = lib.aadg_fop_workspace_bytes(B, H, W)
    ws = workspace(nb, img.device, "fop")
    rc = lib.aadg_fop_f32(FOP[name], img.data_ptr(), out.data_ptr(), _ptr(mag), mag_n, _ptr(kernel), _ptr(perm),
                          B, C, H, W, ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_fop_f32(%s)" % name)
    return out


def controller_dims(controller, M):
    """(M, Q, S, E, H, n_ops, n_mags) of a Controller module."""
    return (int(M), int(controller.Q), 2 * int(controller.L), int(controller.embedding_dim), int(controller.hidden_dim),
            int(controller.NUM_OPS), int(controller.NUM_MAGS))


def controller_supported(controller, M):
    ps = list(controller.parameters())
    return (len(ps) == 9 and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps) and
            bool(load().aadg_controller_supported(*controller_dims(controller, M))))


def controller_workspace(controller, M):
    need = load().aadg_controller_workspace_bytes(*controller_dims(controller, M))
    return torch.zeros(need, dtype=torch.uint8, device=next(controller.parameters()).device)


def controller_pointers(controller, exp_avg=None, exp_avg_sq=None):
    """The pointer arrays of the fused controller calls, built once (models/graphed.py keeps them while the tensors keep their storage):
    (parameters, exp_avg, exp_avg_sq, the data_ptr they were built from)."""
    params = list(controller.parameters())
    return (_ptr_array(params), _ptr_array(exp_avg) if exp_avg is not None else None, _ptr_array(exp_avg_sq) if exp_avg_sq is not None else None,
            tuple(p.data_ptr() for p in params))


def controller_sample(controller, M, uniforms, ws, ptrs=None, dims=None):
    """Fused controller.sample(M): returns (policies int64 [M, Q*2L], mean op probs, mean mag probs, log_probs, entropies)."""
    lib = load()
    dims = dims or controller_dims(controller, M)
    dev = uniforms.device
    policies = torch.empty((M, dims[1] * dims[2]), dtype=torch.int64, device=dev)
    op_probs = torch.empty(dims[5], dtype=torch.float32, device=dev)
    mag_probs = torch.empty(dims[6], dtype=torch.float32, device=dev)
    log_probs = torch.empty(M, dtype=torch.float32, device=dev)
    entropies = torch.empty(M, dtype=torch.float32, device=dev)
    params = ptrs[0] if ptrs is not None else _ptr_array(list(controller.parameters()))
    rc = lib.aadg_controller_sample_f32(params, *dims, float(controller.C) / float(controller.T), uniforms.data_ptr(),
                                        policies.data_ptr(), op_probs.data_ptr(), mag_probs.data_ptr(), log_probs.data_ptr(),
                                        entropies.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_controller_sample_f32")
    return policies, op_probs, mag_probs, log_probs, entropies


def controller_ppo_update(controller, M, exp_avg, exp_avg_sq, policies, old_log_probs, reward, clip, n_updates, step0, lr,
                          betas, eps, ws, ptrs=None, dims=None):
    """n_updates PPO epochs (evaluate -> clipped surrogate -> backward -> Adam) in place; returns loss terms [n_updates, M]."""
    lib = load()
    dims = dims or controller_dims(controller, M)
    losses = torch.empty((n_updates, M), dtype=torch.float32, device=policies.device)
    if ptrs is None:
        ptrs = (_ptr_array(list(controller.parameters())), _ptr_array(exp_avg), _ptr_array(exp_avg_sq))
    rc = lib.aadg_controller_ppo_update_f32(ptrs[0], ptrs[1], ptrs[2],
                                            *dims, float(controller.C) / float(controller.T), policies.data_ptr(),
                                            old_log_probs.data_ptr(), reward.data_ptr(), float(clip), int(n_updates), int(step0),
                                            float(lr), float(betas[0]), float(betas[1]), float(eps), losses.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_controller_ppo_update_f32")
    return losses


# ------------------------------------------------------------------------------------------------
def embed_prologue(x, w1, b1, w2=None, b2=None, slope=0.2, want_norm=False):
    """(out [N, D] or None, fe [N, E]): fe = LeakyReLU(x W1^T + b1), out = fe W2^T + b2 -- the no-grad EMA branch of the
    domain discriminator, one launch.  want_norm: returns (out, fe, |fe[n]|_2 [N]) for sinkhorn_rewards(row_norm=...)."""
    lib = load()
    _require_cuda(w1, b1, w2, b2)
    if not x.is_cuda:
        raise AadgError("aadg_amd kernels need GPU tensors (got %s); there is no CPU path" % x.device)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise AadgError("embed_prologue: expected a float32 [N, C] matrix with unit column stride")
    w1, b1 = w1.detach().contiguous(), b1.detach().contiguous()
    N, C = x.shape
    E = w1.shape[0]
    fe = torch.empty((N, E), dtype=torch.float32, device=x.device)
    out = None
    D = 0
    if w2 is not None:
        w2, b2 = w2.detach().contiguous(), b2.detach().contiguous()
        D = w2.shape[0]
        out = torch.empty((N, D), dtype=torch.float32, device=x.device)
    if want_norm:
        nrm = torch.empty(N, dtype=torch.float32, device=x.device)
        rc = lib.aadg_embed_prologue_norm_f32(x.data_ptr(), x.stride(0), N, C, w1.data_ptr(), b1.data_ptr(), E, _ptr(w2), _ptr(b2), D,
                                              float(slope), fe.data_ptr(), _ptr(out), nrm.data_ptr(), _stream())
        _check(rc, "aadg_embed_prologue_norm_f32")
        return out, fe, nrm
    rc = lib.aadg_embed_prologue_f32(x.data_ptr(), x.stride(0), N, C, w1.data_ptr(), b1.data_ptr(), E, _ptr(w2), _ptr(b2), D,
                                     float(slope), fe.data_ptr(), _ptr(out), _stream())
    _check(rc, "aadg_embed_prologue_f32")
    return out, fe

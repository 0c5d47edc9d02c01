"""Optimisers / LR schedules of the search path (reference: scheduler.py:5-34).

controller: Adam(lr 3.5e-4), no schedule; segmentation model: Adam(TRAIN.LR, TRAIN.WD) whose LR drops x0.1 when the
warm-up ends; discriminator: Adam(TRAIN.LR) with a constant LR (a cosine schedule only for the unused image
discriminator).  Function names and return tuples follow the reference."""

from torch.optim import Adam
from torch.optim.lr_scheduler import CosineAnnealingLR, MultiStepLR

CONTROLLER_LR = 0.00035


def _step_at_warmup_end(optimizer, cfg, gamma):
    return MultiStepLR(optimizer, milestones=[cfg.TRAIN.WARMUP_EPOCH], gamma=gamma, last_epoch=-1)


FUSED_ADAM = True


def _fused(params):
    """torch's single-launch Adam on device tensors (same update rule as the default multi-tensor path: ~10 elementwise passes over
    the 40 M parameters -> one; 0.8 -> 0.3 ms per step at every batch size).  `scheduler.FUSED_ADAM = False` keeps the default."""
    return FUSED_ADAM and len(params) > 0 and all(p.is_cuda and p.is_floating_point() for p in params)


def _adam(params, **kw):
    params = list(params)
    return Adam(params, fused=True, **kw) if _fused(params) else Adam(params, **kw)


def _model_adam(model, cfg):
    return _adam(model.parameters(), lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)


def get_optimizer_scheduler(controller, model, cfg):
    """-> (model optimiser, model LR scheduler, controller optimiser)"""
    model_opt = _model_adam(model, cfg)
    return model_opt, _step_at_warmup_end(model_opt, cfg, 0.1), Adam(controller.parameters(), lr=CONTROLLER_LR)


def get_optimizer_scheduler2(model, cfg):
    model_opt = _model_adam(model, cfg)
    return model_opt, CosineAnnealingLR(model_opt, T_max=cfg.TRAIN.END_EPOCH)


def get_dis_optimizer_scheduler(discriminator, cfg):
    """-> (discriminator optimiser, LR scheduler)"""
    dis_opt = _adam([p for p in discriminator.parameters() if p.requires_grad], lr=cfg.TRAIN.LR)
    cosine = cfg.TRAIN.WARMUP_EPOCH > 0 and cfg.DISCRIMINATOR.NAME == 'image'
    sched = CosineAnnealingLR(dis_opt, T_max=cfg.TRAIN.WARMUP_EPOCH) if cosine else _step_at_warmup_end(dis_opt, cfg, 1)
    return dis_opt, sched

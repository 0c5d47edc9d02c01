"""Optimisers / LR schedules of the search path, as in the reference's scheduler.py:5-34:
controller Adam(lr 3.5e-4); model Adam(TRAIN.LR, TRAIN.WD) + MultiStepLR([WARMUP_EPOCH], 0.1);
discriminator Adam(TRAIN.LR) + MultiStepLR([WARMUP_EPOCH], gamma 1) (cosine only for the unused image
discriminator)."""
from torch.optim import Adam
from torch.optim.lr_scheduler import CosineAnnealingLR, MultiStepLR


def get_optimizer_scheduler(controller, model, cfg):
    controller_optimizer = Adam(controller.parameters(), lr=0.00035)
    optimizer = Adam(model.parameters(), lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    scheduler = MultiStepLR(optimizer, [cfg.TRAIN.WARMUP_EPOCH], gamma=0.1, last_epoch=-1)
    return optimizer, scheduler, controller_optimizer


def get_optimizer_scheduler2(model, cfg):
    optimizer = Adam(model.parameters(), lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    return optimizer, CosineAnnealingLR(optimizer, T_max=cfg.TRAIN.END_EPOCH)


def get_dis_optimizer_scheduler(discriminator, cfg):
    optimizer = Adam(discriminator.parameters(), lr=cfg.TRAIN.LR)
    if cfg.TRAIN.WARMUP_EPOCH > 0 and cfg.DISCRIMINATOR.NAME == 'image':
        return optimizer, CosineAnnealingLR(optimizer, T_max=cfg.TRAIN.WARMUP_EPOCH)
    return optimizer, MultiStepLR(optimizer, [cfg.TRAIN.WARMUP_EPOCH], gamma=1, last_epoch=-1)

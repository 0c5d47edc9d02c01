"""Model factory -- API mirror of the reference's models/__init__.py (load_ddp_model :8-53,
load_ddp_controller :73-118, load_ddp_discriminator :134-179, class/domain/channel parsers :205-222).

Differences that are design, not omission:
  * the backbone is our own torch.nn DeepLabV3+ (deeplab.py) -- smp is not in this image;
    `MODEL.BACKBONE: resnet50` is accepted in addition to `mobilenet_v2`, `MODEL.NAME: unet` for config 0;
  * data parallelism is NOT "split TRAIN.BATCH_SIZE across DDP replicas" but row sharding of the N =
    D*B*M augmented images (aadg_amd/distributed.py); the model is still wrapped in DDP (RCCL bucketed
    all-reduce overlapped with backward), the controller is replicated deterministically (identical
    rewards on every rank), so it needs no collective at all.
"""
import torch

from .controller import Controller
from .deeplab import DeepLabV3Plus, UNetSmall
from .discriminator import FeatureDiscriminator, MomentumFeatureDiscriminator


def class_parser(dataset):
    return {'rvs': 1, 'optic': 2}[dataset]


def domain_parser(dataset):
    return {'optic': 3, 'rvs': 3}[dataset]


def channel_parser(backbone):
    return {'mobilenet_v2': 1280, 'resnet50': 2048, 'unet': 128}[backbone]


def _device(args):
    if torch.cuda.is_available() and getattr(args, 'gpu', None) is not None:
        torch.cuda.set_device(args.gpu)
        return torch.device('cuda', args.gpu)
    return torch.device('cpu')


def _wrap(module, args, dev):
    if getattr(args, 'distributed', False) and any(p.requires_grad for p in module.parameters()):
        ids = [dev.index] if dev.type == 'cuda' else None
        return torch.nn.parallel.DistributedDataParallel(module, device_ids=ids, gradient_as_bucket_view=True)
    return module


def build_model(cfg):
    name, backbone = cfg.MODEL.NAME, cfg.MODEL.BACKBONE
    classes = class_parser(cfg.DATASET.NAME)
    if name == 'deeplabv3+':
        assert backbone in ['mobilenet_v2', 'resnet50']
        return DeepLabV3Plus(backbone, classes, aux_pooling='feature' in cfg.DISCRIMINATOR.NAME)
    if name == 'unet':
        return UNetSmall(classes)
    raise NotImplementedError(name + ' has not been implemented!')


def load_ddp_model(ngpus_per_node, args, cfg):
    print("=> creating model '{}' with '{}".format(cfg.MODEL.NAME, cfg.MODEL.BACKBONE))
    dev = _device(args)
    model = build_model(cfg).to(dev)
    if getattr(args, 'distributed', False) and getattr(args, 'sync_bn', False):
        # The reference's single-GPU batch mixes all domains in every BatchNorm batch; row-sharded replicas see
        # only their slice.  --sync_bn reduces the per-channel statistics over the ranks (SURVEY.md 8e caveat 1).
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model = _wrap(model, args, dev)
    return model, cfg.TRAIN.BATCH_SIZE, args.workers


def load_ddp_controller(ngpus_per_node, args, cfg):
    print("=> creating controller '{}'".format(cfg.CONTROLLER.NAME))
    dev = _device(args)
    if cfg.CONTROLLER.NAME != 'controller':
        raise NotImplementedError(cfg.CONTROLLER.NAME + ' has not been implemented!')
    # replicated, never wrapped: every rank sees identical rewards and runs identical updates
    return Controller(cfg).to(dev), cfg.CONTROLLER.M, args.workers


def load_ddp_discriminator(ngpus_per_node, args, cfg):
    name = cfg.DISCRIMINATOR.NAME
    print("=> creating discriminator '{}'".format(name))
    dev = _device(args)
    num_classes = domain_parser(cfg.DATASET.NAME)
    in_channels = channel_parser('unet' if cfg.MODEL.NAME == 'unet' else cfg.MODEL.BACKBONE)
    if name == 'feature':
        model = FeatureDiscriminator(num_classes, in_channels)
    elif name == 'momentum_feature':
        model = MomentumFeatureDiscriminator(num_classes, in_channels)
        # the EMA twin is only ever written by momentum_update()/synchronize_parameters(); marking it
        # non-trainable keeps DDP from waiting for gradients that never come
        for p in list(model.mom_dis.parameters()) + list(model.mom_fc.parameters()):
            p.requires_grad_(False)
    else:
        raise NotImplementedError(name + ' has not been implemented!')
    return model.to(dev), cfg.TRAIN.BATCH_SIZE, args.workers

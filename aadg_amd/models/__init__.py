"""Model factory -- API mirror of the reference's models/__init__.py (load_ddp_model :8-53,
load_ddp_controller :73-118, load_ddp_discriminator :134-179, class/domain/channel parsers :205-222).

Differences that are design, not omission:
  * the backbone is our own torch.nn DeepLabV3+ (deeplab.py) -- smp is not in this image;
    `MODEL.BACKBONE: resnet50` is accepted in addition to `mobilenet_v2`, `MODEL.NAME: unet` for config 0,
    `MODEL.NAME: segformer` / `MODEL.BACKBONE: mit_b2` (models/segformer.py) for config 4;
  * data parallelism is NOT "split TRAIN.BATCH_SIZE across DDP replicas" but row sharding of the N =
    D*B*M augmented images (aadg_amd/distributed.py); the model is wrapped in the package's own gradient reducer
    (aadg_amd/reducer.py: bucketed asynchronous RCCL all-reduce overlapped with backward, issued from the weight-gradient
    stream), the controller is replicated deterministically (identical
    rewards on every rank), so it needs no collective at all.
"""
import torch

from .controller import Controller
from .deeplab import DeepLabV3Plus, UNetSmall
from .discriminator import FeatureDiscriminator, MomentumFeatureDiscriminator
from .segformer import SegFormer


def class_parser(dataset):
    return {'rvs': 1, 'optic': 2}[dataset]


def domain_parser(dataset, cfg=None):
    """Number of source domains = width of the discriminator head and of the soft domain codes.  The reference hard-codes 3
    for both datasets (models/__init__.py:211-216); with a config it follows len(DATASET.DG.TRAIN) (BASELINE configs[4] merges
    Fundus + RVS into 8 synthetic source domains)."""
    if cfg is not None and len(cfg.DATASET.DG.TRAIN) > 0:
        return max(len(cfg.DATASET.DG.TRAIN), 3 if dataset in ('optic', 'rvs') else 2)
    return {'optic': 3, 'rvs': 3}[dataset]


def channel_parser(backbone):
    return {'mobilenet_v2': 1280, 'resnet50': 2048, 'unet': 128, 'mit_b2': 512}[backbone]


def _device(args):
    if torch.cuda.is_available() and getattr(args, 'gpu', None) is not None:
        torch.cuda.set_device(args.gpu)
        return torch.device('cuda', args.gpu)
    return torch.device('cpu')


def _wrap(module, args, dev, broadcast_buffers=True):
    if getattr(args, 'distributed', False) and any(p.requires_grad for p in module.parameters()):
        # the package's own data-parallel wrapper (reducer.py) instead of torch's DistributedDataParallel: flat gradient buckets,
        # all-reduced asynchronously from the stream their members arrive on -- the weight-gradient side stream stays on
        from ..reducer import GradReducer
        return GradReducer(module, broadcast_buffers=broadcast_buffers)
    return module


def build_model(cfg):
    name, backbone = cfg.MODEL.NAME, cfg.MODEL.BACKBONE
    classes = class_parser(cfg.DATASET.NAME)
    if name == 'deeplabv3+':
        assert backbone in ['mobilenet_v2', 'resnet50']
        return DeepLabV3Plus(backbone, classes, aux_pooling='feature' in cfg.DISCRIMINATOR.NAME)
    if name == 'unet':
        return UNetSmall(classes)
    if name == 'segformer':
        # BASELINE configs[4]; the reference's wrapper picks the variant from MODEL.PRETRAINED_WEIGHTS (models/segformer.py:12-55)
        assert backbone in ['mit_b2'] or 'mit_b2' in cfg.MODEL.PRETRAINED_WEIGHTS
        return SegFormer(classes, aux_pooling='feature' in cfg.DISCRIMINATOR.NAME)
    raise NotImplementedError(name + ' has not been implemented!')


def load_ddp_model(ngpus_per_node, args, cfg):
    print("=> creating model '{}' with '{}".format(cfg.MODEL.NAME, cfg.MODEL.BACKBONE))
    dev = _device(args)
    model = build_model(cfg).to(dev)
    if dev.type == 'cuda':
        from . import deeplab
        # weight casts and BatchNorm counters: two launches per forward; backbone_dtype 'f32x3' = float32 tensors, convolutions on the
        # own float32-precision matrix-core kernels ('fp32': the library's float32 convolutions, 'bf16': bfloat16 autocast)
        deeplab.batch_step_bookkeeping(model, f32x3=getattr(args, 'backbone_dtype', 'f32x3') == 'f32x3')
    sync = bool(getattr(args, 'distributed', False) and getattr(args, 'sync_bn', False))
    if sync:
        # The reference's single-GPU batch mixes all domains in every BatchNorm batch; sharded replicas see only their rows.
        # --sync_bn all-reduces the per-channel sums over the ranks (SURVEY.md 8e caveat 1).  On the GPU the modules stay
        # nn.BatchNorm2d and keep the HIP kernels: the all-reduce sits between the statistics and the normalisation
        # kernels (aadg_bn_sync_*, models/deeplab.py: bn_act); elsewhere (CPU, unsupported layouts) torch's SyncBatchNorm.
        from . import deeplab
        if dev.type == 'cuda':
            deeplab.set_bn_sync(True)
        else:
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    # with synchronised statistics the running buffers are identical on every rank: no per-forward buffer broadcast
    model = _wrap(model, args, dev, broadcast_buffers=not sync)
    if dev.type == 'cuda':
        # weight-gradient kernels of the own convolutions on a second stream, beside the backward chain (_lib.set_wgrad_stream); in a
        # data-parallel job the GradReducer takes them over there and issues the bucket all-reduces from that stream
        from .. import _lib
        _lib.set_wgrad_stream(bool(getattr(args, 'wgrad_stream', True)))
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
    num_classes = domain_parser(cfg.DATASET.NAME, cfg)
    in_channels = channel_parser({'unet': 'unet', 'segformer': 'mit_b2'}.get(cfg.MODEL.NAME, cfg.MODEL.BACKBONE))
    if name == 'feature':
        model = FeatureDiscriminator(num_classes, in_channels)
    elif name == 'momentum_feature':
        model = MomentumFeatureDiscriminator(num_classes, in_channels)
        # the EMA twin is only ever written by momentum_update()/synchronize_parameters(); marking it
        # non-trainable keeps the reducer from carrying (and all-reducing) gradients that never come
        for p in list(model.mom_dis.parameters()) + list(model.mom_fc.parameters()):
            p.requires_grad_(False)
    else:
        raise NotImplementedError(name + ' has not been implemented!')
    # data-parallel as in the reference (models/__init__.py:165 wraps it in DDP): the online branch's gradients are averaged over the ranks, so the
    # online and (through momentum_update) the EMA weights stay identical everywhere -- the all-gathered embeddings come
    # from ONE network.  The search loop reaches the EMA branch / momentum_update through `.module` (search_dg._bare).
    return _wrap(model.to(dev), args, dev), cfg.TRAIN.BATCH_SIZE, args.workers

"""Search driver and inner loop -- API mirror of the reference's search_dg.py (pretrain :24-99,
train :102-214, validate :217-286, search_seg_dg_policy :289-407; search_dg_2d.py is the same loop for
the single-class RVS task and is served by this module too).

What differs from the reference, by design:
  * `sample['aug_images']` etc. come from the fused GPU augmentation call (no DataLoader workers, no H2D);
  * sigmoid + M per-policy BCE + M redundant F1 passes (search_dg.py:140-144,164-165) are ONE fused kernel
    (`_lib.policy_bce_loss`, forward and backward in a single pass over logits/labels);
  * the 3 x M geomloss calls with ~790 launches and 18 host syncs (search_dg.py:150-162) are ONE kernel
    (`_lib.sinkhorn_rewards`) -- no `.item()` inside the iteration; meters are read once per PRINT_FREQ;
  * multi-GPU: rows are sharded over ranks, embeddings are all-gathered once (distributed.py), rewards are
    computed redundantly and identically on every rank.
"""
import contextlib
import json
import os
import time

import numpy as np
import torch

from . import _lib, utils
from . import distributed as adist
from .data.dataloader import get_seg_dg_dataloader
from .data.policy import DGMultiPolicy, parse_policies
from .data import transform as T
from .losses import CrossEntropy, search_loss, task_loss
from .models import load_ddp_controller, load_ddp_discriminator, load_ddp_model
from .scheduler import get_dis_optimizer_scheduler, get_optimizer_scheduler


def _autocast(args):
    dt = getattr(args, 'backbone_dtype', 'f32x3')
    if dt == 'bf16' and torch.cuda.is_available():
        return torch.autocast('cuda', dtype=torch.bfloat16)
    return torch.autocast('cuda', enabled=False) if torch.cuda.is_available() else torch.autocast('cpu', enabled=False)


LAST_RAW_REWARDS = None


def seed_everything(seed):
    import random
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _bare(module):
    return module.module if hasattr(module, 'module') else module


def _dice_monitor(seg_output, mask_gt):
    """Samplewise Dice per class (the reference's torchmetrics F1[1], search_dg.py:57-58) via the fused kernel."""
    _, dice, _ = _lib.seg_bce_dice(seg_output.float().contiguous(), mask_gt.contiguous(), 1, want_grad=False)
    return dice


def pretrain(config, train_loader, model, discriminator, model_criterion, dis_criterion, model_optimizer,
             dis_optimizer, epoch, writer_dict, logger, args=None):
    """Warm-up epoch on the un-augmented images (search_dg.py:24-99)."""
    batch_time, seg_losses, dis_losses = utils.AverageMeter(), utils.AverageMeter(), utils.AverageMeter()
    model.train()
    discriminator.train()
    end = time.time()
    for i, sample in enumerate(train_loader):
        input, mask_gt = sample['image'], sample['label']
        domain_gt = sample.get('dc_image', sample['dc'])
        with _autocast(args):
            seg_output, feature = model(input)
        dis_output = discriminator(feature.detach().float())
        seg_loss = model_criterion(torch.sigmoid(seg_output.float()), mask_gt)
        dis_loss = dis_criterion(dis_output, domain_gt)
        if 'image_rows' in sample and adist.is_dist():
            # local means -> global mean under DDP's average over ranks (count-weighted: slices may differ by a row)
            lo_s, hi_s, S = sample['image_rows']
            w = (hi_s - lo_s) * adist.world()[1] / float(S)
            seg_loss, dis_loss = seg_loss * w, dis_loss * w
        model_optimizer.zero_grad(set_to_none=True)
        seg_loss.backward()
        model_optimizer.step()
        dis_optimizer.zero_grad(set_to_none=True)
        dis_loss.backward()
        dis_optimizer.step()
        if i % config.PRINT_FREQ == 0 and logger:
            seg_losses.update(seg_loss.item(), input.size(0))
            dis_losses.update(dis_loss.item(), input.size(0))
            batch_time.update(time.time() - end)
            logger.info('Epoch: [{0}][{1}/{2}]\tTime {3:.3f}s\tSpeed {4:.1f} samples/s\tSeg Loss {5:.5f}\tDis Loss {6:.5f}'.format(
                epoch, i, len(train_loader), batch_time.val, input.size(0) / max(batch_time.val, 1e-9), seg_losses.val, dis_losses.val))
        end = time.time()


def inner_iteration(config, sample, model, discriminator, dis_criterion, model_optimizer, dis_optimizer, M, rewards,
                    args=None, n_domains=3, after_rewards=None):
    """One pass of search_dg.py:123-176 on one collated batch.  Accumulates into `rewards` ([M], device).
    Returns device scalars (seg_loss, dis_loss, diversity_ot, dice[K]) -- no host sync in here.
    `after_rewards()` (optional) runs once this batch's rewards are accumulated, before the backward passes."""
    input, mask_gt, domain_gt = sample['aug_images'], sample['aug_labels'], sample['dc']
    plan = sample.get('plan')
    sharded = plan is not None and plan.sharded
    n_rows = plan.n_rows if plan is not None else input.size(0)
    with _autocast(args):
        seg_output, feature = model(input)
    feature = feature.detach().float()
    # The reward branch -- EMA-branch embeddings (no grad, outside the DDP wrapper: nothing to reduce) -> all-gather -> Sinkhorn reward ->
    # (last batch) PPO update -- feeds only the controller.  Its kernels are tiny (k_embed 86 us on 9 workgroups, 18 Sinkhorn problems = 18
    # workgroups, the collective's latency), so with the fused controller path it runs on the controller's stream, beside the loss and the
    # backward passes of the segmentation model instead of in front of them.
    side = getattr(args, '_side_stream', None) if feature.is_cuda else None
    main = torch.cuda.current_stream() if side is not None else None
    if side is not None:
        side.wait_stream(main)                            # the features are complete (and last step's EMA update is)
        feature.record_stream(side)
        domain_gt.record_stream(side)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        dis_output, domain_feature, fe_norm = _bare(discriminator)(feature, momentum=True, return_feature=True, return_norm=True)
        with torch.no_grad():
            logp = torch.log_softmax(dis_output, dim=1)
            dis_loss = -(domain_gt * logp).sum(dim=1).mean()        # mean_j CE(dis_output[j::M], gt[j::M]) = mean over rows
        # reward: all-gather the local [n_local, 128] embeddings once (back into collate order), then ONE kernel for all M x P problems
        fe_all = domain_feature.contiguous()
        if sharded:
            emulate = bool(getattr(args, 'emulate_shards', 0))
            if not emulate and not adist.is_dist():
                raise RuntimeError("row-sharded batch without an initialised process group")
            fe_all = plan.gather(fe_all, emulate=emulate)
            fe_norm = None                                   # the gathered rows' norms are recomputed by the reward kernel
        before = rewards.clone()
        B = n_rows // (M * n_domains)
        if n_domains >= 2:                               # a single source domain has no domain pair to compare (BASELINE configs[0])
            _lib.sinkhorn_rewards(fe_all, n_domains, B, M, rewards=rewards, row_norm=fe_norm)
        diversity_ot = (rewards - before).sum()
        if after_rewards is not None:
            after_rewards()
        # bp: online branch trained on the soft domain codes.  It sees only the detached features, so on the side stream its forward,
        # backward and Adam step (a dozen small launches) also run beside the segmentation model's backward instead of behind it
        dis_loss_bp = dis_criterion(discriminator(feature, momentum=False), domain_gt)
        if sharded:
            dis_loss_bp = dis_loss_bp * plan.loss_weight     # DDP averages the ranks' gradients: count-weighted mean (RowPlan)
        if side is not None:
            dis_optimizer.zero_grad(set_to_none=True)
            dis_loss_bp.backward()
            dis_optimizer.step()
    # sigmoid + per-policy BCE + Dice + d loss / d logits, one fused pass; the backward pass starts from the kernel's gradient
    # (_lib.policy_bce_backward).  mean_j BCE_j == mean over all rows (every policy owns N/M rows), so a rank whose local rows are not
    # policy-interleaved takes the plain mean of its rows.  DDP averages the ranks' gradients: the local mean is weighted by
    # n_local * G / N (count-weighted mean, RowPlan) -- inside the kernel.
    model_optimizer.zero_grad(set_to_none=True)
    seg_loss, _, dice = _lib.policy_bce_backward(seg_output, mask_gt, 1 if sharded else M, plan.loss_weight if sharded else 1.0)
    model_optimizer.step()
    if side is None:
        dis_optimizer.zero_grad(set_to_none=True)
        dis_loss_bp.backward()
        dis_optimizer.step()
    else:
        main.wait_stream(side)                            # rewards, dis_loss, diversity_ot, the discriminator's update come from there
    return seg_loss.detach(), dis_loss, diversity_ot, dice


def train(config, train_loader, model, discriminator, model_criterion, dis_criterion, model_optimizer, dis_optimizer,
          M, epoch, writer_dict, logger, args=None, max_iters=None, on_last_rewards=None):
    """One search epoch (search_dg.py:102-214): returns the normalised rewards [M].
    `on_last_rewards(normalised rewards)` (optional) is called as soon as the epoch's rewards are complete -- after the forward
    pass of the LAST batch, before its backward passes: nothing after that point changes them."""
    batch_time = utils.AverageMeter()
    model.train()
    discriminator.train()
    dev = next(model.parameters()).device
    rewards = torch.zeros(M, device=dev)
    n_domains = len(config.DATASET.DG.TRAIN)
    length = len(train_loader)
    last = (length if max_iters is None else min(length, max_iters)) - 1
    end = time.time()
    for i, sample in enumerate(train_loader):
        if max_iters is not None and i >= max_iters:
            break
        hook = None
        if on_last_rewards is not None and i == last:
            hook = lambda: on_last_rewards(_lib.normalize_rewards(rewards))
        seg_loss, dis_loss, div_ot, dice = inner_iteration(config, sample, model, discriminator, dis_criterion,
                                                           model_optimizer, dis_optimizer, M, rewards, args, n_domains, hook)
        if i % config.PRINT_FREQ == 0 and logger:
            n_img = sample['plan'].n_rows if 'plan' in sample else sample['aug_images'].size(0)
            vals = torch.stack([seg_loss, dis_loss, div_ot]).tolist()        # the only host sync, every PRINT_FREQ
            batch_time.update(time.time() - end)
            logger.info('Epoch: [{0}][{1}/{2}]\tTime {3:.3f}s\tSpeed {4:.1f} samples/s\tSeg Loss {5:.5f}\t'
                        'Dis Loss {6:.5f}\tOT {7:.5f}'.format(epoch, i, length, batch_time.val,
                                                              n_img / max(batch_time.val, 1e-9), *vals))
            if writer_dict:
                writer = writer_dict['writer']
                steps = writer_dict['train_global_steps']
                writer.add_scalar('train_seg_loss', vals[0], steps)
                writer.add_scalar('train_dis_loss', vals[1], steps)
                writer.add_scalar('diversity_ot_distance', vals[2], steps)
                writer_dict['train_global_steps'] = steps + 1
        end = time.time()
    global LAST_RAW_REWARDS
    LAST_RAW_REWARDS = rewards                      # diagnostics / tests: the epoch's accumulated Sinkhorn sums before normalisation
    if M < 2:
        return torch.zeros_like(rewards)            # one policy (fixed-policy plumbing case): the unbiased std of one value is undefined
    return _lib.normalize_rewards(rewards)          # (r - mean) / (std + 1e-5), search_dg.py:214


@torch.no_grad()
def validate(config, val_loader, model, epoch, writer_dict, logger, args=None):
    """Thresholded Dice on the held-out domain (search_dg.py:217-286 / search_dg_2d.py:216-281).  hd95 (medpy) is not
    available in this image and is reported as 0 (out of scope).  Every rank scores the WHOLE test set
    (test batches are not sharded), so best_dsc / is_best agree on all ranks without a collective."""
    model.eval()
    K = 2 if config.DATASET.NAME == 'optic' else 1
    sums = torch.zeros(K, device=next(model.parameters()).device)
    count = 0
    # optic: sigmoid(z) > 0.75 <=> z - log(3) > 0 (search_dg.py:243); rvs: the F1 argmax over [1 - p, p], i.e. p > 0.5
    # (search_dg_2d.py:252)
    shift = float(np.log(0.75 / 0.25)) if config.DATASET.NAME == 'optic' else 0.0
    for sample in val_loader:
        input, mask_gt = sample['image'], sample['label']
        with _autocast(args):
            seg_output, _ = model(input)
        sums += _dice_monitor(seg_output.float() - shift, mask_gt) * input.size(0)
        count += input.size(0)
    dsc = (sums / max(count, 1)).tolist()
    if logger:
        logger.info('Test Epoch {} dsc: {}'.format(epoch, ' '.join('%.4f' % d for d in dsc)))
    if K == 2:
        return dsc[0], dsc[1], 0.0, 0.0
    return dsc[0], dsc[0], 0.0, 0.0


class SearchState(object):
    """Everything one policy-search step needs (built once by search_seg_dg_policy / bench.py)."""

    def __init__(self, gpu, ngpus_per_node, config, args):
        self.config, self.args = config, args
        # Row sharding relies on every rank drawing the SAME batch plan and holding the SAME controller / discriminator:
        # seed python / numpy / torch identically on all ranks (the reference never seeds, SURVEY fact 8, and is single-GPU)
        seed_everything(config.SEED if config.SEED is not None else 1023)
        self.model, self.batch_size, workers = load_ddp_model(ngpus_per_node, args, config)
        self.controller, self.M, _ = load_ddp_controller(ngpus_per_node, args, config)
        self.discriminator, _, _ = load_ddp_discriminator(ngpus_per_node, args, config)
        rank, world = adist.world()
        # The augmentation call's helper stream (ABI 12: the late units' statistics chain beside the tile kernel) stays off in a
        # data-parallel job: beside RCCL's communicator streams it cost 7-26 ms per step on the boxes measured (one rank over RCCL: 189-209
        # against 182 ms, scripts/r6/exp_dist_knobs.py; more hardware queues change nothing) for 0.02 ms of a 180 ms step.
        _lib.AUG_FORK = not adist.is_dist()
        T.set_row_shard(rank, world, getattr(args, 'placement', 'row'), force=bool(getattr(args, 'force_sharded', False)))
        if adist.is_dist():
            # the controller is replicated, not wrapped (identical rewards -> identical updates); DDP broadcast the wrapped
            # modules' state at construction, do the same for it
            for t in list(self.controller.parameters()) + list(self.controller.buffers()):
                torch.distributed.broadcast(t.data, 0)             # once, at construction: the default group
        _, self.train_loader, self.test_loader = get_seg_dg_dataloader(config, args, self.batch_size, workers)
        self.model_optimizer, self.model_lrscheduler, self.controller_optimizer = \
            get_optimizer_scheduler(self.controller, self.model, config)
        self.dis_optimizer, self.dis_lrscheduler = get_dis_optimizer_scheduler(self.discriminator, config)
        self.model_criterion = task_loss(config)
        self.controller_criterion = search_loss(config)
        self.dis_criterion = CrossEntropy()
        self.controller_criterion.register_optimizer(self.controller_optimizer)
        # launch-bound controller sample / PPO update: fused HIP kernels (csrc/controller.hip) for PPO, else replayed as
        # two HIP graphs (models/graphed.py)
        self.graphed = None
        if torch.cuda.is_available() and getattr(args, 'controller_graphs', True) and not getattr(args, 'fixed_policy', False):
            from .models.graphed import make_controller_step
            self.graphed = make_controller_step(self.controller, self.controller_criterion, self.controller_optimizer, self.M,
                                                fused=getattr(args, 'controller_fused', True))

    def _sample_policies(self, async_host=False):
        """Sample M policies from the controller (rank 0's draw is authoritative in a distributed run) and fetch them to the
        host: (policies, op_probs, mag_probs, log_probs, entropies, policies as a numpy array -- or, with async_host, a callable
        that waits for the asynchronous copy and returns it)."""
        if self.graphed is not None:
            try:
                policies, op_probs, mag_probs, log_probs, entropies = self.graphed.sample()
            except RuntimeError as e:                      # capture refused (e.g. by another thread's HIP call)
                import sys
                print("aadg_amd: controller graph capture failed (%s); using the eager controller" % e, file=sys.stderr)
                self.graphed = None
        if self.graphed is None:
            policies, op_probs, mag_probs, log_probs, entropies = self.controller(self.M)
        if adist.is_dist():
            # the controller is replicated, not wrapped: make rank 0's draw authoritative and re-derive the
            # log-probs for it -- evaluate() equals sample()'s log-prob for the same actions
            adist.small_broadcast(policies, 0, kind="policy_broadcast")
            if self.graphed is not None:
                # the replicas' parameters are identical, so rank 0's log-probs of rank 0's draw are THE log-probs:
                # ship the [M] vector instead of re-evaluating the controller on every rank
                adist.small_broadcast(self.graphed.old_log_probs, 0, kind="policy_broadcast")
            else:
                log_probs = self.controller.evaluate(policies, self.M)
        if async_host:
            # device-to-host copy without blocking the host: the caller enqueues more work first and fetches later
            pinned = torch.empty(policies.shape, dtype=policies.dtype, device='cpu', pin_memory=True)
            pinned.copy_(policies.detach(), non_blocking=True)
            done = torch.cuda.Event()
            done.record()

            def fetch():
                done.synchronize()
                return pinned.numpy()
            return policies, op_probs, mag_probs, log_probs, entropies, fetch
        return policies, op_probs, mag_probs, log_probs, entropies, policies.cpu().detach().numpy()

    def search_step(self, epoch, writer_dict=None, logger=None, max_iters=None):
        """The epoch body of search_dg.py:338-347: sample M policies -> inject -> train -> EMA -> PPO.

        With the fused controller kernels the PPO update and the NEXT epoch's sampling are issued as soon as this epoch's
        rewards are complete, i.e. before the backward passes of its last batch: the device-to-host copy of the sampled
        policies (a full pipeline drain when it sat between two epochs) then waits only for the forward pass, and the host
        parses / draws the next batch plan while the GPU is still busy with the backward.  Same arithmetic, same random
        streams (the backward consumes no random numbers)."""
        if getattr(self.args, 'fixed_policy', False):
            return self._fixed_policy_step(epoch, writer_dict, logger, max_iters)
        self.controller.train()
        nxt, self._prefetched = getattr(self, '_prefetched', None), None
        policies, op_probs, mag_probs, log_probs, entropies, host_policies = nxt if nxt is not None else self._sample_policies()
        if callable(host_policies):
            host_policies = host_policies()                # the copy was enqueued behind the early update of the previous step
        parsed = parse_policies(host_policies, self.config, logger)
        self.train_loader.dataset.transforms.transforms[0] = DGMultiPolicy(parsed)
        early = {}
        hook = None
        side = None
        self.args._side_stream = None
        if getattr(self.graphed, 'fused', False) and getattr(self.args, 'early_controller_update', True):
            # The controller kernels (5 PPO epochs + the next sampling: ~0.6 ms on 6 workgroups) touch nothing the backbone touches:
            # they run on a stream of their own, beside the backward passes instead of in front of them (0.6 ms per step on every
            # rank, 3 % of an 18-row step), and the sampled policies come back through a pinned buffer that the host reads only after
            # it has enqueued those backward passes.
            main = torch.cuda.current_stream()
            side = self._controller_stream(main) if (main.device.type == 'cuda' and getattr(self.args, 'controller_stream', True)) else None

            self.args._side_stream = side if getattr(self.args, 'reward_stream', True) else None

            def hook(normalized):
                if side is None:
                    early['losses'] = self.graphed.update(normalized, entropies)
                    self._prefetched = self._sample_policies()
                    return
                side.wait_stream(main)                     # the rewards are complete
                normalized.record_stream(side)
                with torch.cuda.stream(side):
                    early['losses'] = self.graphed.update(normalized, entropies)
                    self._prefetched = self._sample_policies(async_host=True)
        normalized_rewards = train(self.config, self.train_loader, self.model, self.discriminator, self.model_criterion,
                                   self.dis_criterion, self.model_optimizer, self.dis_optimizer, self.M, epoch,
                                   writer_dict, logger, self.args, max_iters, hook)
        _bare(self.discriminator).momentum_update()
        losses = early.get('losses')
        if losses is None and self.graphed is not None:
            try:
                losses = self.graphed.update(normalized_rewards, entropies)
            except RuntimeError as e:
                import sys
                print("aadg_amd: controller update graph capture failed (%s); using the eager criterion" % e, file=sys.stderr)
                self.graphed = None
                policies = policies.clone()
                _, lps, ents, _, _ = self.controller._rollout(self.M, forced=policies, want_entropy=True)
                log_probs, entropies = torch.stack(lps, -1).sum(-1), torch.stack(ents, -1).sum(-1)
        if losses is None:
            losses = self.controller_criterion(self.controller, policies, log_probs, entropies, normalized_rewards)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)  # what this step returns (losses, probabilities) was produced there
        if getattr(self.args, 'predraw', True):
            # everything of this step is enqueued: draw the policy-independent part of the next epoch's first batch (python's
            # generator: sub-policy choices, geometry, soft codes) while the GPU works -- the next step then only completes the records
            # once the sampled policies have reached the host.  The next epoch injects a new DGMultiPolicy (above): fresh CutMix queues.
            self.train_loader.predraw(fresh_policies=True)
        return parsed, op_probs, mag_probs, normalized_rewards, losses

    def _controller_stream(self, main):
        side = getattr(self, '_ctrl_stream', None)
        if side is None:
            side = self._ctrl_stream = torch.cuda.Stream(device=main.device)
        return side


FIXED_POLICY = [('Contrast', 0.5), ('Sharpness', 0.5)]      # the fixed sub-policy of the no-search plumbing case


def _fixed_policy_step(self, epoch, writer_dict=None, logger=None, max_iters=None):
    """BASELINE configs[0] ("fixed policy, no controller search"): the epoch body without the controller -- every one of the M
    policies is the single sub-policy FIXED_POLICY, no sampling, no PPO update; the inner loop itself is unchanged."""
    parsed = [[list(FIXED_POLICY)] for _ in range(self.M)]
    self.train_loader.dataset.transforms.transforms[0] = DGMultiPolicy(parsed)
    normalized_rewards = train(self.config, self.train_loader, self.model, self.discriminator, self.model_criterion,
                               self.dis_criterion, self.model_optimizer, self.dis_optimizer, self.M, epoch, writer_dict, logger,
                               self.args, max_iters)
    _bare(self.discriminator).momentum_update()
    zero = torch.zeros((), device=normalized_rewards.device)
    return parsed, None, None, normalized_rewards, (zero, zero, zero)


SearchState._fixed_policy_step = _fixed_policy_step


def search_seg_dg_policy(gpu, ngpus_per_node, config, args):
    # validate() runs between two epochs and its pipeline draws from python's `random` too (DGRandomCrop, SoftLable): drawing the next
    # epoch's first batch ahead (search_step's predraw) would take those numbers BEFORE validate's instead of after, i.e. change what a
    # seeded run draws relative to the reference order (search_dg.py:338-356).  Off here unless the caller asks for it.
    if not hasattr(args, 'predraw'):
        args.predraw = False
    st = SearchState(gpu, ngpus_per_node, config, args)
    # models, data pool and kernel handles live for the whole run: take them out of the cyclic garbage collector's scans
    # (a full collection is a host pause of tens of milliseconds in the middle of a step)
    import gc
    gc.collect()
    gc.freeze()
    rank, _ = adist.world()
    main = rank == 0
    logger, final_output_dir, writer_dict = None, None, None
    if main:
        logger, final_output_dir, tb_log_dir = utils.create_logger(config, args.cfg, 'train')
        writer_dict = {'writer': utils.make_summary_writer(tb_log_dir), 'train_global_steps': 0, 'valid_global_steps': 0}
    best_dsc = 0
    best_metric = {'epoch': 0, 'avg_dsc': 0, 'cup_dsc': 0, 'disc_dsc': 0, 'avg_hd': 0, 'cup_hd': 0, 'disc_hd': 0}
    mag_probs_trajectory, op_probs_trajectory = [], []
    end_epoch = min(config.TRAIN.END_EPOCH, getattr(args, 'max_epochs', None) or config.TRAIN.END_EPOCH)
    for epoch in range(config.TRAIN.BEGIN_EPOCH, end_epoch):
        searching = config.TRAIN.WARMUP_EPOCH <= epoch
        if not searching:
            pretrain(config, st.train_loader, st.model, st.discriminator, st.model_criterion, st.dis_criterion,
                     st.model_optimizer, st.dis_optimizer, epoch, writer_dict, logger, args)
        else:
            if config.TRAIN.WARMUP_EPOCH == epoch:
                _bare(st.discriminator).synchronize_parameters()
            parsed, op_probs, mag_probs, _, (controller_loss, score_loss, entropy_penalty) = \
                st.search_step(epoch, writer_dict, logger)
        st.model_lrscheduler.step()
        st.dis_lrscheduler.step()
        cup_dsc, disc_dsc, cup_hd, disc_hd = validate(config, st.test_loader, st.model, epoch, writer_dict, logger, args)
        dsc, hd = (cup_dsc + disc_dsc) / 2, (cup_hd + disc_hd) / 2
        is_best = dsc > best_dsc
        if is_best:
            best_dsc = dsc
            best_metric = {'epoch': epoch + 1, 'avg_dsc': dsc, 'cup_dsc': cup_dsc, 'disc_dsc': disc_dsc,
                           'avg_hd': hd, 'cup_hd': cup_hd, 'disc_hd': disc_hd}
        if main and searching:
            if op_probs is not None:                                    # None: fixed policy, no controller statistics
                mag_probs_trajectory.append(mag_probs.detach().cpu().numpy())
                op_probs_trajectory.append(op_probs.detach().cpu().numpy())
            logger.info('Train Epoch {}: controller loss:{:.4f} score loss:{:.4f} entropy penalty:{:.4f}'.format(
                epoch, controller_loss.item(), score_loss.item(), entropy_penalty.item()))
            utils.save_checkpoint({"state_dict": _bare(st.model), "epoch": epoch + 1, "best_dsc": best_dsc,
                                   "optimizer": st.model_optimizer.state_dict(), "policies": parsed},
                                  is_best, final_output_dir, 'checkpoint_{}.pth'.format(epoch))
    if main:
        torch.save(_bare(st.model).state_dict(), os.path.join(final_output_dir, 'final_model_state.pth'))
        torch.save(st.controller.state_dict(), os.path.join(final_output_dir, 'final_controller_state.pth'))
        np.save(os.path.join(final_output_dir, 'mag_probs_trajectory.npy'), np.array(mag_probs_trajectory))
        np.save(os.path.join(final_output_dir, 'op_probs_trajectory.npy'), np.array(op_probs_trajectory))
        with open(os.path.join(final_output_dir, 'final_result.json'), 'w') as f:
            f.write(json.dumps(best_metric))
        if writer_dict and writer_dict['writer'] is not None:
            writer_dict['writer'].close()
    return best_metric


search_seg2d_dg_policy = search_seg_dg_policy      # the RVS entry point of the reference (search_dg_2d.py:284)

"""Per-process worker of `run.py --mode search` (reference search.py:10-35): one process per GPU,
torch.distributed over RCCL (`--dist_backend nccl` is RCCL on ROCm), then the dataset-specific driver."""
import builtins
import os

import torch
import torch.distributed as dist

from .search_dg import search_seg2d_dg_policy, search_seg_dg_policy


def search_worker(gpu, ngpus_per_node, config, args):
    args.gpu = gpu
    if args.multiprocessing_distributed and args.gpu != 0:
        builtins.print = lambda *a, **k: None          # only the master prints
    if args.gpu is not None:
        print("Use GPU: {} for training".format(args.gpu))
    if args.distributed:
        if args.dist_url == "env://" and args.rank == -1:
            args.rank = int(os.environ["RANK"])
        if args.multiprocessing_distributed:
            args.rank = args.rank * ngpus_per_node + gpu
        backend = args.dist_backend if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank)
    if config.DATASET.NAME in ['optic']:
        return search_seg_dg_policy(gpu, ngpus_per_node, config, args)
    if config.DATASET.NAME in ['rvs']:
        return search_seg2d_dg_policy(gpu, ngpus_per_node, config, args)
    raise NotImplementedError(config.DATASET.NAME)


def lanuch_mp_worker(main_worker, config, args):
    """Launcher with the reference's (misspelt) name and contract (distributed.py:15-31)."""
    import torch.multiprocessing as mp
    if args.dist_url == "env://" and args.world_size == -1:
        args.world_size = int(os.environ["WORLD_SIZE"])
    args.distributed = args.world_size > 1 or args.multiprocessing_distributed
    ngpus_per_node = max(torch.cuda.device_count(), 1)
    if args.multiprocessing_distributed:
        args.world_size = ngpus_per_node * max(args.world_size, 1)
        mp.spawn(main_worker, nprocs=ngpus_per_node, args=(ngpus_per_node, config, args))
    else:
        return main_worker(args.gpu, ngpus_per_node, config, args)

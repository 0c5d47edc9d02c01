"""Entry point with the reference's CLI (run.py:14-67):  python run.py --cfg experiments/<...>.yaml
[--mode search] [--gpu N] [--multiprocessing_distributed] [--dist_backend nccl] [--dist_url ...] ...

Extra, optional flags (not in the reference): --crop_size (the reference hard-codes 256),
--backbone_dtype {f32x3,fp32,bf16} (default f32x3: the own float32-precision convolution kernels), --max_epochs / --epoch_items (short synthetic runs), --sync_bn / --placement (multi-GPU),
--fixed_policy (BASELINE configs[0])."""
import argparse
import sys

from aadg_amd.config.defaults import _C as config
from aadg_amd.config.defaults import update_config
from aadg_amd.search import lanuch_mp_worker, search_worker


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Adversarial AutoAugment (MI355X-native hot path)')
    parser.add_argument('-j', '--workers', default=4, type=int, metavar='N')
    parser.add_argument('--world_size', default=-1, type=int)
    parser.add_argument('--rank', default=-1, type=int)
    parser.add_argument('--dist_url', default='tcp://localhost:10001', type=str)
    parser.add_argument('--dist_backend', default='nccl', type=str)
    parser.add_argument('--gpu', default=0, type=int)
    parser.add_argument('--gpus', default=1, type=int)
    parser.add_argument('--multiprocessing_distributed', action='store_true')
    parser.add_argument('--smoke_test', action='store_true')
    parser.add_argument('--mode', default='search')
    parser.add_argument('--cfg', required=True, type=str)
    parser.add_argument('--output_dir', default='output', type=str)
    parser.add_argument('--vis_dir', default='vis', type=str)
    parser.add_argument('--output_type', default='image', type=str)
    parser.add_argument('--seed', default=1023, type=int)
    parser.add_argument('--crop_size', default=256, type=int)
    parser.add_argument('--no_wgrad_stream', dest='wgrad_stream', action='store_false',
                        help='weight-gradient kernels in line with the backward chain (default: on a second HIP stream beside it, multi-GPU included)')
    parser.add_argument('--backbone_dtype', default='f32x3', choices=['fp32', 'f32x3', 'bf16'],
                        help="f32x3 (default, the measured path of bench.py): float32 tensors, the own float32-precision matrix-core convolution "
                             "kernels (products as three bfloat16 MFMA products of split operands, float32 accumulation; inside north_star's 1e-4 "
                             "contract, tests/test_gpu_precision.py) -- layers / backbones they do not cover take the library's float32 convolution "
                             "and are listed on stderr after the first step; fp32: the library's float32 convolutions everywhere (2.2x slower on "
                             "ResNet-50); bf16: bfloat16 autocast (narrower than the reference)")
    parser.add_argument('--sync_bn', action='store_true', help='SyncBatchNorm over the row-sharded ranks')
    parser.add_argument('--placement', default='row', choices=['unit', 'row'],
                        help='multi-GPU cut of the domain-major (domain, policy) unit sequence: balanced to the row (default, as bench.py: 18 rows per rank at 8 GPUs) or whole units per rank (SURVEY 8e as written: 3/3/2/... units = 24 rows on the slowest rank); one domain per GPU at 3 GPUs either way')
    parser.add_argument('--fixed_policy', action='store_true',
                        help='no controller search: every policy is [Contrast .5, Sharpness .5] (BASELINE configs[0])')
    parser.add_argument('--max_epochs', default=None, type=int)
    parser.add_argument('--epoch_items', default=32, type=int)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    update_config(config, args)
    if args.mode == 'search':
        return lanuch_mp_worker(search_worker, config, args)
    raise NotImplementedError("Only --mode search is on the hot path (train/test are out of scope, SURVEY.md section 2).")


if __name__ == '__main__':
    main()

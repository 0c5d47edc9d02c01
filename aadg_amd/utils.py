"""Logging / checkpoint helpers with the reference's semantics (utils.py:18-38,181-241): AverageMeter,
create_logger (file + console, output/<dataset>/<cfg>_<time>/train.log, TensorBoard dir under LOG_DIR),
save_checkpoint (model_best.pth on improvement), load_checkpoint (strips 'module.')."""
import logging
import os
import time
from collections import OrderedDict
from pathlib import Path

import torch


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def create_logger(cfg, cfg_name, phase='train'):
    root = Path(cfg.OUTPUT_DIR)
    root.mkdir(parents=True, exist_ok=True)
    dataset = cfg.DATASET.NAME
    stem = os.path.basename(cfg_name).split('.')[0]
    stamp = time.strftime('%Y-%m-%d-%H-%M')
    final_output_dir = root / dataset / '{}_{}'.format(stem, stamp)
    final_output_dir.mkdir(parents=True, exist_ok=True)
    log_file = final_output_dir / '{}.log'.format(phase)
    logger = logging.getLogger('aadg_amd.%s.%s' % (stem, stamp))
    logger.setLevel(logging.INFO)
    logger.handlers = []
    fmt = logging.Formatter('%(asctime)-15s %(message)s')
    for h in (logging.FileHandler(str(log_file)), logging.StreamHandler()):
        h.setFormatter(fmt)
        logger.addHandler(h)
    tb_dir = Path(cfg.LOG_DIR) / dataset / '{}_{}'.format(stem, stamp)
    tb_dir.mkdir(parents=True, exist_ok=True)
    return logger, str(final_output_dir), str(tb_dir)


def make_summary_writer(log_dir):
    """TensorBoard is optional in this image; scalars are dropped when it is missing."""
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:  # noqa: BLE001
        class _Null(object):
            def add_scalar(self, *a, **k):
                pass

            def close(self):
                pass
        return _Null()


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth'):
    """The reference only refreshes a latest.pth symlink and stores the whole model object as
    model_best.pth when the Dice improved (utils.py:217-224)."""
    latest = os.path.join(output_dir, 'latest.pth')
    if os.path.islink(latest) or os.path.exists(latest):
        os.remove(latest)
    os.symlink(os.path.join(output_dir, filename), latest)
    if is_best and 'state_dict' in states:
        torch.save(states['state_dict'], os.path.join(output_dir, 'model_best.pth'))


def load_checkpoint(model_path, model):
    """Argument order of the reference (utils.py:225): load_checkpoint(model_path, model)."""
    state = torch.load(model_path, map_location='cpu')
    if hasattr(state, 'state_dict'):
        state = state.state_dict()
    clean = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state.items())
    model.load_state_dict(clean)
    return model

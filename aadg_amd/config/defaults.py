"""Configuration tree -- same keys, defaults and `update_config` contract as the reference's
config/defaults.py:8-73, so experiments/*.yaml run unchanged.  yacs is not in this image, so a small
CfgNode with the subset of yacs semantics the reference uses (attribute access, nested merge from a
yaml file with unknown-key rejection, defrost/freeze) is provided here."""
import copy

import yaml


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode._FROZEN] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode._FROZEN]:
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def _set_frozen(self, flag):
        self.__dict__[CfgNode._FROZEN] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self.__dict__[CfgNode._FROZEN]

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        out.__dict__[CfgNode._FROZEN] = self.__dict__[CfgNode._FROZEN]
        return out

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: {}".format(full))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("Type mismatch for config key {}".format(full))
                self[k]._merge(v, path + [k])
            else:
                old = self[k]
                if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                    v = float(v)
                if isinstance(old, tuple) and isinstance(v, list):
                    v = tuple(v)
                if old is not None and v is not None and type(old) is not type(v) and \
                        not (isinstance(old, (list, tuple)) and isinstance(v, (list, tuple))):
                    raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(v), full))
                self[k] = v

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            self._merge(yaml.safe_load(f) or {}, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def __str__(self):
        return yaml.safe_dump(_to_plain(self), default_flow_style=None)


def _to_plain(node):
    return {k: _to_plain(v) if isinstance(v, CfgNode) else v for k, v in node.items()}


CN = CfgNode

# Same tree, keys and default values as the reference's config/defaults.py:8-66 (written as one literal).
_C = CN({
    'OUTPUT_DIR': 'output', 'LOG_DIR': 'log', 'PRINT_FREQ': 100, 'SEED': 0,
    'MODEL': {'NAME': 'deeplabv3+', 'BACKBONE': 'mobilenet_v2', 'PRETRAINED_WEIGHTS': ''},
    'CONTROLLER': {'NAME': 'controller', 'LOSS': 'ppo', 'PENALTY': 0.00001, 'L': 2, 'M': 6, 'T': 2, 'C': 2.5,
                   'NUM_MAGS': 10, 'EXCLUDE_OPS_NUM': 0, 'EXCLUDE_OPS': []},
    'DISCRIMINATOR': {'NAME': 'momentum_feature'},
    'DATASET': {'ROOT': './dataset', 'NAME': 'cifar10', 'TRAINSET': '', 'TESTSET': '',
                'DG': {'TRAIN': [1, 2, 3], 'TEST': [4]}},
    'TRAIN': {'LR': 0.1, 'WD': 0.0004, 'BEGIN_EPOCH': 0, 'WARMUP_EPOCH': 0, 'END_EPOCH': 200, 'BATCH_SIZE': 8,
              'SHUFFLE': True},
    'TEST': {'BATCH_SIZE': 8, 'MODEL_DIR': ''},
})


def update_config(cfg, args):
    """yaml merge, then the two CLI overrides, then freeze (config/defaults.py:68-73)."""
    cfg.defrost()
    cfg.merge_from_file(args.cfg)
    cfg.OUTPUT_DIR = args.output_dir
    cfg.SEED = args.seed
    cfg.freeze()


def get_default_config():
    return _C.clone()

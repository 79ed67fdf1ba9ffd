"""Reading the reference's checkpoints (lib/trainer.py:184-193): a pickled dict
{'state_dict', 'config': EasyDict, 'epoch', ...}.  `easydict` is not installed here, so a minimal
stand-in is registered for unpickling; torch >= 2.6 needs weights_only=False for such files."""
import sys
import types

import torch


class _EasyDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _ensure_easydict():
    try:
        import easydict  # noqa: F401
    except ImportError:
        mod = types.ModuleType("easydict")
        mod.EasyDict = _EasyDict
        sys.modules["easydict"] = mod


class Config(_EasyDict):
    """The seven fields inference reads (generate_desc.py:161-186)."""
    DEFAULTS = dict(model="ResUNetBN2C", model_n_out=32, normalize_feature=True, conv1_kernel_size=5,
                    voxel_size=0.025, image_H=120, image_W=160, bn_momentum=0.05)

    def __init__(self, **kw):
        super().__init__({**self.DEFAULTS, **kw})


def load_checkpoint(path, map_location="cpu"):
    """Returns (state_dict, config).  Accepts the older key prefix 'perceiver_io' for the fusion block
    (lib/Test.py:17-19 renames it to 'attention_fusion')."""
    _ensure_easydict()
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    sd = {k.replace("perceiver_io", "attention_fusion"): v for k, v in sd.items()}
    cfg = ckpt.get("config", None) if isinstance(ckpt, dict) else None
    if cfg is None:
        cfg = Config()
    elif not hasattr(cfg, "model"):                    # plain dict / Namespace -> attribute access
        cfg = Config(**(cfg if isinstance(cfg, dict) else vars(cfg)))
    return sd, cfg

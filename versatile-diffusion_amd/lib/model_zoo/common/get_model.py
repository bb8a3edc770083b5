"""Name -> class registry with the reference's surface (`get_model()(cfg)`, `@register(name)`;
lib/model_zoo/common/get_model.py:37-104 there).  Implementing modules are imported lazily by type prefix, weights
are loaded from `ckpt` / `pth` when the config names a file that exists."""
import copy
import importlib
import os

import torch

from ...log_service import print_log

_LAZY = (
    ("autoencoderkl", "autokl"),
    ("clip", "clip"),
    ("vd", "vd"),
    ("openai_unet", "openaimodel"),
    ("optimus", "optimus"),
)


class _Registry(object):
    def __init__(self):
        self.model = {}

    def register(self, model, name):
        self.model[name] = model

    def _import_for(self, t):
        for prefix, mod in _LAZY:
            if t.startswith(prefix):
                importlib.import_module("lib.model_zoo." + mod)
                return

    def __call__(self, cfg, verbose=True):
        t = cfg["type"] if isinstance(cfg, dict) else cfg.type
        if t not in self.model:
            self._import_for(t)
        if t not in self.model:
            raise KeyError("model type '%s' is not registered" % t)
        args = copy.deepcopy(cfg.get("args", {}))
        net = self.model[t](**args)
        map_location = cfg.get("map_location", "cpu")
        strict_sd = cfg.get("strict_sd", True)
        for key in ("ckpt", "pth"):
            path = cfg.get(key, None)
            if path is None:
                continue
            if not os.path.exists(path):
                # offline environments have no pretrained files: keep the random init and say so
                print_log("[get_model] weight file %s not found, keeping initialised weights" % path)
                break
            sd = torch.load(path, map_location=map_location)
            if key == "ckpt":
                sd = sd["state_dict"]
            net.load_state_dict(sd, strict=strict_sd)
            if verbose:
                print_log("Load %s from %s" % (key, path))
            break
        if "hfm" in cfg and "pth" not in cfg and "ckpt" not in cfg:
            raise RuntimeError("hfm (huggingface hub) weights need network access; provide a local `pth` instead")
        if verbose:
            n = sum(p.numel() for p in net.parameters())
            print_log("Load %s with total %d parameters" % (t, n))
        return net


_instance = _Registry()


def get_model():
    """Singleton accessor, called as `get_model()(cfg)` like the reference."""
    return _instance


def register(name):
    def wrapper(cls):
        _instance.register(cls, name)
        return cls
    return wrapper

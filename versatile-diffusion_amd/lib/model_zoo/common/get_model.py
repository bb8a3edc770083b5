"""Name -> class registry with the reference's surface (`get_model()(cfg)`, `@register(name)`;
lib/model_zoo/common/get_model.py:19-104 there).  Implementing modules are imported lazily by type prefix; weights come
from `ckpt` / `pth` / `hfm` exactly as there: a config that names a weight file which does not exist raises (the
reference fails in torch.load), unless the caller opts into keeping the initialised weights with
`cfg.allow_missing_weights = True` or VD_ALLOW_MISSING_WEIGHTS=1 (offline development without checkpoints)."""
import copy
import functools
import importlib
import os
import re

import torch
import torch.nn as nn

from ...log_service import print_log


class _Units(object):
    """Layer-unit factory of the reference (`get_unit`, common/utils.py:41-87): 'name' or 'name(k=v, k2=(a,b))' ->
    class / functools.partial.  Only the torch.nn units are registered; none of them is on the sampling path."""

    def __init__(self):
        self.unit = {"none": None, "conv": nn.Conv2d, "bn": nn.BatchNorm2d, "relu": nn.ReLU, "relu6": nn.ReLU6,
                     "lrelu": nn.LeakyReLU, "dropout": nn.Dropout, "dropout2d": nn.Dropout2d}

    def register(self, name, unitf):
        self.unit[name] = unitf

    @staticmethod
    def _value(v):
        v = v.strip()
        if v in ("True", "False"):
            return v == "True"
        if v == "None":
            return None
        for cast in (int, float):
            try:
                return cast(v)
            except ValueError:
                pass
        return v

    def __call__(self, name):
        if name is None:
            return None
        m = re.match(r"^\s*([A-Za-z_0-9]+)\s*(?:\((.*)\))?\s*$", name)
        if m is None or m.group(1) not in self.unit:
            raise KeyError("unknown layer unit '%s'" % name)
        f, args = self.unit[m.group(1)], (m.group(2) or "").strip()
        if not args:
            return f
        kwargs = {}
        for k, v in re.findall(r"([A-Za-z_0-9]+)\s*=\s*(\([^)]*\)|\[[^\]]*\]|[^,]+)", args):
            v = v.strip()
            if v[0] in "([":
                items = [self._value(i) for i in v[1:-1].split(",") if i.strip()]
                kwargs[k] = tuple(items) if v[0] == "(" else items
            else:
                kwargs[k] = self._value(v)
        return functools.partial(f, **kwargs)


_units = _Units()


def get_unit():
    return _units


def preprocess_model_args(args):
    """`layer_units` names -> unit factories, `backbone` config -> constructed model (reference get_model.py:19-35)."""
    args = copy.deepcopy(args)
    if "layer_units" in args:
        args["layer_units"] = [get_unit()(i) for i in args["layer_units"]]
    if "backbone" in args:
        args["backbone"] = get_model()(args["backbone"])
    return args

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
        args = preprocess_model_args(cfg.get("args", {}))
        net = self.model[t](**args)
        map_location = cfg.get("map_location", "cpu")
        strict_sd = cfg.get("strict_sd", True)
        allow_missing = bool(cfg.get("allow_missing_weights", False)) or os.environ.get("VD_ALLOW_MISSING_WEIGHTS") == "1"
        for key in ("ckpt", "pth", "hfm"):   # same precedence as the reference's if / elif chain
            if cfg.get(key, None) is None:
                continue
            if key == "hfm":
                from huggingface_hub import hf_hub_download  # needs network or a populated hub cache, like the reference
                path = hf_hub_download(cfg["hfm"][0], cfg["hfm"][1])
                map_location = "cpu"
            else:
                path = cfg[key]
                if not os.path.exists(path):
                    if not allow_missing:
                        raise FileNotFoundError(
                            "[get_model] %s names the weight file '%s' which does not exist (cwd %s); set "
                            "cfg.allow_missing_weights / VD_ALLOW_MISSING_WEIGHTS=1 to keep the initialised weights"
                            % (t, path, os.getcwd()))
                    print_log("[get_model] weight file %s not found, keeping initialised weights (opt-in)" % path)
                    break
            sd = torch.load(path, map_location=map_location)
            if key == "ckpt":
                sd = sd["state_dict"]
            net.load_state_dict(sd, strict=strict_sd)
            if verbose:
                print_log("Load %s from %s" % (key, cfg[key] if key != "hfm" else "/".join(cfg["hfm"])))
            break
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

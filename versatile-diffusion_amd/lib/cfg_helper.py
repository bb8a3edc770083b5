"""`model_cfg_bank` with the reference's semantics (lib/cfg_helper.py:21-165 there): YAML files under
configs/model chosen by name prefix, `super_cfg` inheritance that MERGES `args`, `delete_args`, and the
MODEL(name) / SAME(a.b) / SEARCH(a.b) macros.  Only the model bank is provided; the experiment/dataset banks and
CLI of the reference feed its (dead) training launcher and are out of scope."""
import copy
import os
import os.path as osp

import yaml


class CfgDict(dict):
    """Attribute-access dict with recursive wrapping (the reference uses easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __deepcopy__(self, memo):
        return CfgDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


edict = CfgDict


def _walk(root, path):
    zoom = root
    for p in [q.strip() for q in path.split(".")]:
        try:
            p = int(p)
        except ValueError:
            pass
        zoom = zoom[p]
    return zoom


def cfg_solvef(cmd, root):
    if not isinstance(cmd, str):
        return cmd
    if cmd.startswith("SAME"):
        try:
            return cfg_solvef(_walk(root, cmd[4:].strip("()")), root)
        except (KeyError, IndexError, TypeError):
            return cmd
    if cmd.startswith("SEARCH"):
        try:
            return cfg_solvef(_walk(root, cmd[6:].strip("()")), root)
        except (KeyError, IndexError, TypeError):
            children = root.values() if isinstance(root, dict) else (root if isinstance(root, list) else [])
            for child in children:
                rv = cfg_solvef(cmd, child)
                if rv != cmd:
                    return rv
            return cmd
    if cmd.startswith("MODEL"):
        return model_cfg_bank()(cmd[5:].strip("()"))
    return cmd


def cfg_solve(cfg, cfg_root):
    if isinstance(cfg, list):
        for i in range(len(cfg)):
            cfg[i] = cfg_solve(cfg[i], cfg_root) if isinstance(cfg[i], (list, dict)) else cfg_solvef(cfg[i], cfg_root)
    if isinstance(cfg, dict):
        for k in list(cfg):
            cfg[k] = cfg_solve(cfg[k], cfg_root) if isinstance(cfg[k], (list, dict)) else cfg_solvef(cfg[k], cfg_root)
    return cfg


_YAML_BY_PREFIX = (("openai_unet", "openai_unet.yaml"), ("clip", "clip.yaml"), ("openclip", "clip.yaml"),
                   ("vd", "vd.yaml"), ("optimus", "optimus.yaml"), ("autokl", "autokl.yaml"))


class model_cfg_bank(object):
    """`model_cfg_bank()(name)` -> resolved config (type, args, pth, ...).  Paths are CWD-relative like the
    reference (`configs/model`); set VD_CONFIG_DIR to point elsewhere (e.g. at the reference's own YAMLs)."""

    def __init__(self):
        default = osp.join("configs", "model")
        if not osp.isdir(default):
            default = osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), "configs", "model")
        self.cfg_dir = os.environ.get("VD_CONFIG_DIR", default)
        self.cfg_bank = CfgDict()

    def __call__(self, name):
        if name not in self.cfg_bank:
            with open(self.get_yaml_path(name), "r") as f:
                self.cfg_bank.update(CfgDict(yaml.load(f, Loader=yaml.FullLoader)))
        cfg = self.cfg_bank[name]
        cfg.name = name
        if "super_cfg" in cfg:
            sup = self.__call__(cfg.super_cfg)
            if "args" in cfg:
                if "args" in sup:
                    sup.args.update(cfg.args)
                else:
                    sup.args = cfg.args
                cfg.pop("args")
            sup.update(cfg)
            sup.pop("super_cfg")
            cfg = sup
            for dropped in cfg.pop("delete_args", []):
                cfg.args.pop(dropped)
        cfg = cfg_solve(cfg, cfg)
        self.cfg_bank[name] = cfg
        return copy.deepcopy(cfg)

    def get_yaml_path(self, name):
        for prefix, fn in _YAML_BY_PREFIX:
            if name.startswith(prefix):
                return osp.join(self.cfg_dir, fn)
        raise ValueError("no config file known for model '%s'" % name)

"""Import the READ-ONLY reference (/root/reference) in this CPU container.  TEST INFRASTRUCTURE ONLY.

Used by oracle/gen_golden.py to produce tests/golden/*.npz and by the optional `-m "not gpu"` tests that
cross-check the oracle against the live reference when it is present.  /root/reference does not exist on the
GPU box; nothing at run time there may import this module.

Shims: stub torchvision + easydict (absent here), rank helpers for a 0-GPU host,
DDIMSampler.register_buffer (hard-codes "cuda"), CWD-relative config paths.
"""
import contextlib
import os
import sys
import types

REF_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "model_zoo"))


class _EasyDict(dict):
    """Minimal attr-dict with recursive wrapping (stand-in for easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = {} if d is None else dict(d)
        d.update(kw)
        for k, v in d.items():
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
        d = {} if d is None else dict(d)
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    def pop(self, k, *a):
        return super().pop(k, *a)


_loaded = {}


def load_reference():
    """Returns a namespace with the reference modules; idempotent."""
    if _loaded:
        return _loaded["ns"]
    if not reference_available():
        raise RuntimeError("reference checkout not present at %s" % REF_ROOT)
    import transformers  # noqa: F401  (must be imported BEFORE the torchvision stub)
    from transformers import CLIPModel  # noqa: F401  force the lazy import now

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvm.VGG16_Weights = type("VGG16_Weights", (), {"IMAGENET1K_V1": None})
        tvm.vgg16 = lambda *a, **k: None
        tvt = types.ModuleType("torchvision.transforms")
        tvt.ToPILImage = lambda *a, **k: (lambda img: img)  # clip.py:90-91; the fake processor ignores the images
        tv.models, tv.transforms = tvm, tvt
        sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt})
    if "easydict" not in sys.modules:
        ed = types.ModuleType("easydict")
        ed.EasyDict = _EasyDict
        sys.modules["easydict"] = ed

    # make `import lib...` resolve to the reference (and not to this repo's own lib package)
    for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
        del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        import lib.sync as rsync
        rsync.get_rank = lambda type="local": 0
        rsync.get_world_size = lambda type="local": 1
        import lib.log_service as rlog
        rlog.print_log = lambda *a, **k: None
        from lib.model_zoo import get_model
        import lib.model_zoo.vd as rvd
        rvd.print_log = lambda *a, **k: None
        import lib.model_zoo.common.get_model as rgm
        rgm.print_log = lambda *a, **k: None
        import lib.model_zoo.openaimodel as rom
        import lib.model_zoo.attention as ratt
        import lib.model_zoo.ddim as rddim
        import lib.model_zoo.autokl as rautokl
        import lib.model_zoo.autokl_modules as rakm
        import lib.model_zoo.clip as rclip
        import lib.model_zoo.diffusion_utils as rdu
        import lib.model_zoo.distributions as rdist
        import lib.cfg_helper as rcfg
        rddim.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    finally:
        sys.path.remove(REF_ROOT)
    ns = types.SimpleNamespace(get_model=get_model, vd=rvd, openaimodel=rom, attention=ratt, ddim=rddim,
                               autokl=rautokl, autokl_modules=rakm, clip=rclip, diffusion_utils=rdu,
                               distributions=rdist, cfg_helper=rcfg, edict=_EasyDict)
    _loaded["ns"] = ns
    _loaded["modules"] = {k: v for k, v in sys.modules.items() if k == "lib" or k.startswith("lib.")}
    # leave the reference's `lib` out of sys.modules so the repo's own `lib` package can be imported afterwards
    for k in list(_loaded["modules"]):
        del sys.modules[k]
    return ns


@contextlib.contextmanager
def reference_cwd():
    """Config paths in the reference are CWD-relative (lib/cfg_helper.py:104)."""
    old = os.getcwd()
    saved = {k: v for k, v in sys.modules.items() if k == "lib" or k.startswith("lib.")}
    for k in saved:
        del sys.modules[k]
    sys.modules.update(_loaded.get("modules", {}))
    os.chdir(REF_ROOT)
    try:
        yield
    finally:
        os.chdir(old)
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
            del sys.modules[k]
        sys.modules.update(saved)

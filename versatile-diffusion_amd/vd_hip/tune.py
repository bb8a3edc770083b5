"""Tuned GEMM launch table (vd_gemm_tune_set): per problem shape the tile configuration / split-K factor that measured
fastest INSIDE a UNet forward on MI355X (tools/tune_forward.py writes configs/gemm_tune_gfx950.json).  Shapes that are not
in the table go through the library's cost model, so the table only ever refines; VD_GEMM_TUNE=0 ignores it (A/B runs),
VD_GEMM_TUNE=/path/file.json loads another one."""
import json
import os

from .loader import lib

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_TABLE = os.path.join(os.path.dirname(_HERE), "configs", "gemm_tune_gfx950.json")
_loaded = None


def load_table(path=None, clear=True):
    """Install the entries of `path` (default: the shipped table).  Returns the number of entries installed."""
    global _loaded
    path = DEFAULT_TABLE if path is None else path
    h = lib()
    if clear:
        h.vd_gemm_tune_clear()
    if not os.path.exists(path):
        _loaded = (path, 0)
        return 0
    with open(path) as f:
        tab = json.load(f)
    names = {h.vd_gemm_config_name(i).decode(): i for i in range(h.vd_gemm_num_configs())}
    n = 0
    for e in tab.get("entries", []):
        cfg = names.get(e["kernel"])       # entries name the instantiation, not its index: robust to table reordering
        if cfg is None:
            continue
        if h.vd_gemm_tune_set(int(e["M"]), int(e["N"]), int(e["K"]), int(e["ks"]), int(e["cls"]), cfg, int(e.get("nsplit", 1))) == 0:
            n += 1
    _loaded = (path, n)
    return n


def ensure_loaded():
    """Called once from ops on first use: honours VD_GEMM_TUNE (0 = off, path = that file, unset = shipped table)."""
    if _loaded is not None:
        return _loaded[1]
    env = os.environ.get("VD_GEMM_TUNE")
    if env == "0":
        globals()["_loaded"] = ("", 0)
        return 0
    return load_table(env if env else None)

"""TEST INFRASTRUCTURE: resolve model names through the REFERENCE's own lib/cfg_helper.model_cfg_bank (its own YAMLs, its
own super_cfg / MODEL() logic) and print {name: resolved config} as JSON.  Separate process: the reference package is
also called `lib`.  Needs /root/reference (CPU container only).    usage: python oracle/ref_cfg_dump.py name [name ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402


def plain(o):
    if isinstance(o, dict):
        return {str(k): plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    return o


def main():
    refshim.load_reference()
    out = {}
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        from lib.cfg_helper import model_cfg_bank
        for name in sys.argv[1:]:
            out[name] = plain(model_cfg_bank()(name))
    json.dump(out, sys.stdout, sort_keys=True)


if __name__ == "__main__":
    main()

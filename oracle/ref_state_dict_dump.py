"""TEST INFRASTRUCTURE: build REFERENCE modules by registry name at FULL size on the meta device (no memory, no weights) and
print {name: {state-dict key: shape}} as JSON.  Separate process: the reference package is also called `lib`.
Needs /root/reference (CPU container only).    usage: python oracle/ref_state_dict_dump.py model_name [...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402


def main():
    refshim.load_reference()
    out = {}
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        from lib.cfg_helper import model_cfg_bank
        from lib.model_zoo import get_model
        import lib.model_zoo.common.get_model as gm
        gm.get_total_param_sum = lambda net: 0     # the verbose parameter checksum calls .item(): not on meta tensors
        for name in sys.argv[1:]:
            cfg = model_cfg_bank()(name)

            def strip(c):   # no weight files offline: construct only
                if isinstance(c, dict):
                    for k in ("pth", "ckpt", "hfm"):
                        c.pop(k, None)
                    for v in c.values():
                        strip(v)
                elif isinstance(c, (list, tuple)):
                    for v in c:
                        strip(v)
            strip(cfg)
            with torch.device("meta"):
                net = get_model()(cfg, verbose=False)
            out[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()

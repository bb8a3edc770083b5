"""TEST INFRASTRUCTURE: the REFERENCE's lib/model_zoo/diffusion_utils schedule helpers evaluated over a grid, as JSON
(hex-encoded float bits so the comparison is exact).  Separate process; needs /root/reference (CPU container only)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402


def enc(a):
    a = np.asarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape), "hex": a.tobytes().hex()}


def main():
    grid = json.load(sys.stdin)   # {"steps": [...], "etas": [...], "schedules": [[name, n, start, end], ...]}
    refshim.load_reference()
    out = {"betas": {}, "ddim": {}}
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        import lib.model_zoo.diffusion_utils as du
        for name, n, a, b in grid["schedules"]:
            betas = du.make_beta_schedule(name, n, linear_start=a, linear_end=b)
            out["betas"]["%s/%d/%r/%r" % (name, n, a, b)] = enc(np.asarray(betas))
        betas = np.asarray(du.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012))
        import torch
        # what DDIMSampler.make_schedule passes (ddim.py:33-41 there): the model's fp32 buffer as a CPU torch tensor
        alphacums = torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)
        for method in grid["methods"]:
            for s in grid["steps"]:
                try:
                    ts = du.make_ddim_timesteps(method, s, 1000, verbose=False)
                except Exception as e:  # noqa
                    out["ddim"]["%s/%d" % (method, s)] = {"error": type(e).__name__}
                    continue
                ent = {"timesteps": enc(ts)}
                for eta in grid["etas"]:
                    try:
                        sig, al, alp = du.make_ddim_sampling_parameters(alphacums, ts, eta, verbose=False)
                        ent["eta%r" % eta] = [enc(np.asarray(sig, dtype=np.float64)), enc(np.asarray(al, dtype=np.float64)),
                                              enc(np.asarray(alp, dtype=np.float64))]
                    except Exception as e:  # noqa
                        ent["eta%r" % eta] = {"error": type(e).__name__}
                out["ddim"]["%s/%d" % (method, s)] = ent
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()

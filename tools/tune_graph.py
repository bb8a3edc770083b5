"""In-forward tile A/B on the GRAPH-REPLAYED forward (round 5): for one GEMM problem at a time, every candidate instantiation is
pinned through vd_gemm_tune_set, the whole forward is captured into a HIP graph and replayed; the number is the device time of the
forward (what bench.py's step metric measures), not per-launch events of an eager run.  Greedy over the problems in the order
given: a winner (>= --min-gain ms) stays installed while the next problem is tried.

    python tools/tune_graph.py [--workload t2i|i2v|dual|triple] [--batch B] [--cfgs 0,1,2,...] [--problems "M,N,K,ks,cls;..."] [--min-gain 0.01]

Prints one line per candidate and the final pins in VD_FWD_TUNE syntax (tools/unet_forward.py)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
os.environ["VD_GEMM_TUNE"] = "0"
import torch  # noqa: E402
import bench  # noqa: E402
from vd_hip import ops  # noqa: E402
from vd_hip.loader import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="0,1,2,3,4,5,6,13,14,15,25,26")
    ap.add_argument("--problems", default="")
    ap.add_argument("--min-gain", type=float, default=0.01)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--top", type=int, default=14)
    ap.add_argument("--workload", default="t2i", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per forward before CFG doubling (default: the workload's per-GPU share)")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")]
    problems = [tuple(int(v) for v in p.split(",")) for p in args.problems.split(";") if p]
    dev = torch.device("cuda:0")
    net = bench.build_model(dev)
    wl = bench.WORKLOADS[args.workload]
    x, t, cs = bench.forward_inputs(wl, args.batch or bench.default_per_gpu(wl), dev)
    for c in cs:
        c["kv_cache"] = {}
    h = lib()
    names = [h.vd_gemm_config_name(i).decode() for i in range(h.vd_gemm_num_configs())]

    def fwd():
        with torch.no_grad():
            return bench.run_forward(net, x, t, cs)

    def install(pins):
        h.vd_gemm_tune_clear()
        for (M, N, K, ks, cls), cfg in pins.items():
            assert h.vd_gemm_tune_set(M, N, K, ks, cls, cfg, 1) == 0

    def measure():
        for _ in range(2):
            out = fwd()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                out = fwd()
        torch.cuda.current_stream().wait_stream(s)
        ops.drop_workspaces(s.cuda_stream)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(args.reps):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.reps)
        return best, out.float().clone()

    pins = {}
    install(pins)
    if not args.problems:   # the GEMM problems of this forward, by device time (per-launch events of one eager forward)
        fwd(); torch.cuda.synchronize()
        ops.PROFILE_SHAPES = True
        ops.profile_begin()
        fwd()
        tm = {}
        for name, fl, by, ms in ops.profile_end():
            if "gemm_f16_kernel" in name:
                f = dict(kv.split("=") for kv in name.split(">", 1)[1].strip().split(" split=")[0].split())
                key = (int(f["M"]), int(f["N"]), int(f["K"]), int(f["ks"]), int(f["cls"]))
                tm[key] = tm.get(key, 0.0) + ms
        problems = [k for k, v in sorted(tm.items(), key=lambda kv: -kv[1]) if k[0] >= 96 and k[1] >= 96 and k[3] == 1][:args.top]
        ops.PROFILE_SHAPES = False
        for k in problems:
            print("problem M=%d N=%d K=%d ks=%d cls=%d: %.3f ms per forward (eager events)" % (k + (tm[k],)))
    base, ref = measure()
    print("planner forward: %.3f ms" % base)
    cur = base
    for prob in problems:
        res = []
        for cfg in cfgs:
            trial = dict(pins)
            trial[prob] = cfg
            install(trial)
            try:
                ms, out = measure()
            except Exception as e:  # noqa
                res.append((cfg, None, str(e)[:60]))
                continue
            err = float((out - ref).norm() / ref.norm())
            res.append((cfg, ms, err))
        install(pins)
        ok = [(ms, cfg) for cfg, ms, err in res if ms is not None and err < 5e-3]
        line = " ".join("%d:%s" % (cfg, ("%.3f" % ms) if ms is not None else "x") for cfg, ms, err in res)
        bms, bcfg = min(ok) if ok else (cur, None)
        keep = bcfg is not None and cur - bms >= args.min_gain
        print("M=%d N=%d K=%d ks=%d cls=%d | cur %.3f | %s | best %s %.3f %s" % (prob + (cur, line, names[bcfg] if bcfg is not None else "-", bms,
                                                                              "KEPT" if keep else "")))
        sys.stdout.flush()
        if keep:
            pins[prob] = bcfg
            cur = bms
    install(pins)
    final, _ = measure()
    print("final forward with pins: %.3f ms (planner %.3f)" % (final, base))
    print("VD_FWD_TUNE=" + ";".join("%d,%d,%d,%d,%d,%d,1" % (p + (c,)) for p, c in pins.items()))


if __name__ == "__main__":
    main()

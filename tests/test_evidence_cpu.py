"""The committed measurement evidence is self-consistent: the bench line of the round, the PMC traffic file and the
kernel-trace file `bench.py` reads back (digest-matched, see roofline_leg) were taken with ONE build of libvd_hip.so, and
the roofline numbers of the bench line are derivable from them.  No GPU, no library needed."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _bench_module():
    src = open(os.path.join(ROOT, "bench.py")).read()
    consts = {}
    for line in src.split("\n"):   # only the two file-name constants (importing bench.py pulls in the model stack)
        if line.startswith("PMC_TRAFFIC_FILE") or line.startswith("TRACE_FILE") or line.startswith("MFMA_FP16_PEAK_TFLOPS"):
            exec(line.split("#")[0], consts)
    return consts


def _load(name):
    with open(os.path.join(PROF, name)) as f:
        txt = f.read().strip()
    try:
        return json.loads(txt)
    except ValueError:   # a bench log: the JSON line is the last one
        return json.loads(txt.split("\n")[-1])


def test_round_evidence_files_share_one_library_digest():
    c = _bench_module()
    rnd = c["PMC_TRAFFIC_FILE"][:3]
    assert c["TRACE_FILE"].startswith(rnd)
    bench = _load(rnd + "_bench.json")
    traffic = _load(c["PMC_TRAFFIC_FILE"])
    trace = _load(c["TRACE_FILE"])
    dg = bench["library"]["digest"]
    assert len(dg) == 16
    assert traffic["library_digest"] == dg and trace["library_digest"] == dg
    recheck = os.path.join(PROF, rnd + "_bench_recheck.json")
    if os.path.exists(recheck):
        assert _load(rnd + "_bench_recheck.json")["library"]["digest"] == dg


def test_bench_line_roofline_follows_from_the_trace_and_counter_files():
    c = _bench_module()
    rnd = c["PMC_TRAFFIC_FILE"][:3]
    bench = _load(rnd + "_bench.json")
    roof = bench["roofline"]
    dom = roof["kernel"]
    peak = c["MFMA_FP16_PEAK_TFLOPS"]
    assert roof["peak"] == peak and roof["bound"] == "mfma"
    # achieved / frac of the stated clock
    assert roof["frac"] == pytest.approx(roof["achieved"] / peak, abs=2e-4)
    assert roof["achieved"] == pytest.approx(roof["algorithmic_gflop_per_launch"] / roof["avg_launch_us"] * 1e3, rel=2e-3)
    # kernel-trace figure next to it = the committed trace file
    trace = _load(c["TRACE_FILE"])
    assert trace["kernel"] == dom
    kt = roof["kernel_trace"]
    assert kt["avg_launch_us"] == pytest.approx(trace["avg_us"])
    assert kt["achieved"] == pytest.approx(roof["algorithmic_gflop_per_launch"] / trace["avg_us"] * 1e3, rel=2e-3)
    assert kt["frac"] == pytest.approx(kt["achieved"] / peak, abs=2e-4) and kt["frac"] <= 1.0
    # traffic = the counter file's bytes per launch; ratio against the algorithmic bytes
    ent = _load(c["PMC_TRAFFIC_FILE"])["kernels"][dom]
    assert roof["traffic"] == ent["hbm_side_bytes_per_launch"]
    assert roof["traffic_ratio"] == pytest.approx(roof["traffic"] / roof["algorithmic_bytes_per_launch"], abs=6e-3)
    # the whole-path numbers: value = images / time, forward fraction from the step time
    assert bench["value"] == pytest.approx(bench["config"]["global_batch"] * 1e3 / bench["ms_per_step"], rel=1e-3)
    fwd_ms = bench["unet_forward_ms_per_ddim_step_bs4"]
    assert bench["unet_forward_frac_of_mfma_peak"] == pytest.approx(roof["forward_algorithmic_tflop"] / fwd_ms / peak * 1e3, abs=3e-4)
    assert roof["launches_per_forward"] * roof["avg_launch_us"] * 1e-3 <= fwd_ms + 1.5   # dominant kernel fits in the forward (events carry launch gaps)

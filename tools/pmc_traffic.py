"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/unet_forward.py) into the per-kernel
HBM-side traffic table bench.py reads (profiles/rNN_pmc_traffic.json).

usage: python tools/pmc_traffic.py <fetch_results.db> <write_results.db> > profiles/r01_pmc_traffic.json

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): gfx950 rocprofv3 reports half of the bytes of
a wide coalesced read -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB.  GEMM instantiations are keyed
like the kernel names of bench.py's per-kernel table (vd_gemm_config_name)."""
import collections, json, re, sqlite3, sys


def norm(name):
    m = re.search(r"(conv3x3_halo_kernel)<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(2).split(",")]
        if len(args) == 7:   # the 7th template argument (SKIP: folded skip convolution) is "skip" / absent in the library's names
            args = args[:6] + (["skip"] if args[6] in ("true", "1") else [])
        return "%s<%s>" % (m.group(1), ",".join(args))
    if "ff_geglu_kernel" in name:
        return "ff_geglu_kernel"
    m = re.search(r"ff_chain_kernel<(\w+), (\w+)>", name)
    if m:
        return "ff_chain_kernel<%d,%d>" % (m.group(1) == "true", m.group(2) == "true")
    m = re.search(r"(gemm_f16_kernel|attn_fwd_kernel)<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(2).split(",")]
        if m.group(1) == "gemm_f16_kernel":
            args = args[:7]   # BM, BN, WM, WN, threads, stages, K depth (the 8th, the occupancy hint, is not part of bench.py's names)
        return "%s<%s>" % (m.group(1), ",".join(args))
    m = re.search(r"(gn_partial_kernel|gn_apply_table_kernel|gn_apply_kernel|gn_slab_kernel|gn_from_stats_kernel|gn_table_kernel|"
                  r"layernorm_kernel|splitk_reduce_stats_kernel|splitk_reduce_kernel|conv3x3_wstream_kernel|conv3x3_wsk_kernel|gemm_wstream_kernel|rowchain320_kernel|"
                  r"rowgemm320_kernel|xattn_kernel)", name)
    return m.group(1) if m else None


def collect(path, counter):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    acc = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        if r[ci["counter_name"]] != counter:
            continue
        k = norm(r[ci["kernel_name"]])
        if k:
            acc[k].append(r[ci["value"]])
    return acc


import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "versatile-diffusion_amd"))
from vd_hip.loader import lib_digest

fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"library_digest": lib_digest(),   # of the libvd_hip.so the passes ran with: bench.py ignores the file when it differs
       "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python tools/unet_forward.py 3",
       "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 rocprofv3 reports half of a wide coalesced read "
               "(MI355X_MICROARCH.md, HBM section); counters sit on the L2's fabric side, so Infinity-Cache hits are "
               "included (upper bound on HBM bytes); WRITE_SIZE x 1.000 and FETCH_SIZE x 2.04 calibrated on a known 84 MB stream on this "
               "pool (profiles/r05_pmc_calibration.txt)",
       "kernels": {}}
for k in sorted(fetch):
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
    out["kernels"][k] = {"launches": len(fetch[k]), "fetch_size_kb_mean": round(f, 1), "write_size_kb_mean": round(w, 1),
                         "hbm_side_bytes_per_launch": int((2 * f + w) * 1024)}
print(json.dumps(out, indent=1, sort_keys=True))

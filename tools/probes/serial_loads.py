"""Static check of the device code (development probe, round 4): loads that hipcc follows with `s_waitcnt vmcnt(0)` right away.

    python tools/probes/serial_loads.py            (compiles every csrc/*.hip to gfx950 assembly under /tmp)

`if (cond) x = *ptr;` inside an unrolled loop puts every load into its own conditional block and the compiler waits for it
before leaving the block: N requests become N SERIAL round trips (0.3-0.6 us each from L2).  That was 10 + 20 round trips in
conv3x3_halo_kernel's epilogue and 16 in the split-K reduce kernels.  The fix is to make the requests
branch-free (buffer loads whose out-of-range offsets return zeros, or clamped addresses) so that they leave back to back.
This script lists, per kernel, how many loads are followed by a full wait within two instructions."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "versatile-diffusion_amd", "csrc")
VGPR_FORM = {"attention", "xattn_fused", "ff_fused", "gemm_row320", "gemm", "conv_halo"}


def demangle(n):
    try:
        out = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        out = n
    return re.sub(r"\(anonymous namespace\)::", "", out)


def main():
    procs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        name = os.path.basename(src)[:-4]
        out = "/tmp/dev_%s.s" % name
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
        if name in VGPR_FORM:
            cmd += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
        procs.append((out, subprocess.Popen(cmd + ["--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)))
    rows = []
    for out, p in procs:
        p.wait()
        cur, prev_load, n, res = None, -10, 0, {}
        for l in open(out):
            if l.startswith("_Z") and "@" in l:
                cur = l.split(":")[0]
                res[cur], prev_load, n = 0, -10, 0
                continue
            t = l.strip()
            if cur is None or not t or t[0] in ".;":
                continue
            n += 1
            if re.match(r"(global_load|buffer_load|flat_load)", t) and " lds" not in t:
                prev_load = n
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and n - prev_load <= 2:
                res[cur] += 1
        rows += [(v, demangle(k), os.path.basename(out)) for k, v in res.items() if v >= (int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
    for v, k, f in sorted(rows, reverse=True):
        print("%4d  %-120s %s" % (v, k[:120], f))


if __name__ == "__main__":
    main()

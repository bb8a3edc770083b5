"""Offline (no GPU) study of hipcc code-generation flags on the kernels of libvd_hip.so.

    python tools/probes/flag_study.py <tag> [--files a.hip,b.hip] -- <extra hipcc flags>

Compiles each source device-only for gfx950 with build.py's flags + the extra flags, collects the compiler's
kernel-resource-usage remarks (VGPRs, AGPRs, scratch, occupancy) and the code size of every kernel (llvm-readelf), and writes
versatile-diffusion_amd/build/flags/<tag>.json.  `--diff base` prints the kernels whose numbers differ from another tag.
Nothing here is measured on hardware: it filters the flag sets worth a GPU A/B (spills / occupancy loss rule a set out).
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "versatile-diffusion_amd")
sys.path.insert(0, PKG)
import importlib.util
spec = importlib.util.spec_from_file_location("vd_build", os.path.join(PKG, "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
OUT = os.path.join(PKG, "build", "flags")
ARCH = os.environ.get("VD_STUDY_ARCH", "gfx950")   # e.g. gfx950:xnack-
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def demangle(names):
    p = subprocess.run([FILT], input="\n".join(names), capture_output=True, text=True)
    out = p.stdout.strip().split("\n")
    short = []
    for s in out:
        s = s.replace("(anonymous namespace)::", "").replace("void ", "")
        s = re.sub(r"\(.*\)$", "", s)
        short.append(s.replace(" ", ""))
    return short


def study(tag, files, extra):
    os.makedirs(OUT, exist_ok=True)
    res = {}
    procs = []
    for s in files:
        o = os.path.join(OUT, "%s_%s.co" % (s.replace(".hip", ""), tag))
        flags = [f.replace("gfx950", ARCH) if f.startswith("--offload-arch") else f for f in B.FLAGS]
        cmd = ["/opt/rocm/bin/hipcc"] + flags + B.EXTRA_FLAGS.get(s, []) + extra + [
            "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, s), "-o", o]
        procs.append((s, o, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, o, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(log[-3000:])
            raise SystemExit("hipcc failed on %s" % s)
        cur = None
        kern = {}
        for line in log.split("\n"):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = kern.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
            if m and cur is not None:
                cur[{"ScratchSize [bytes/lane]": "Scratch", "Occupancy [waves/SIMD]": "Occupancy", "VGPRs Spill": "Spill", "LDS Size [bytes/block]": "LDS"}.get(m.group(1), m.group(1))] = int(m.group(2))
        elf = o.replace(".co", ".elf")
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH, "--input=" + o, "--output=" + elf], check=True)
        nm = subprocess.run([READELF, "-s", "-W", elf], capture_output=True, text=True).stdout
        for line in nm.split("\n"):
            f = line.split()   # Num: Value Size Type Bind Vis Ndx Name
            if len(f) == 8 and f[3] == "FUNC" and f[7] in kern:
                kern[f[7]]["code"] = int(f[2])
        names = list(kern)
        for n, sh in zip(names, demangle(names)):
            res["%s:%s" % (s.replace(".hip", ""), sh)] = kern[n]
    with open(os.path.join(OUT, tag + ".json"), "w") as f:
        json.dump({"extra": extra, "kernels": res}, f, indent=1, sort_keys=True)
    return res


def show(res, base=None):
    keys = ("VGPRs", "AGPRs", "Spill", "Scratch", "Occupancy", "code")
    n = 0
    for k in sorted(res):
        r = res[k]
        if base is not None:
            b = base.get(k)
            if b is None or all(r.get(x) == b.get(x) for x in keys):
                continue
            print("%-110s %s" % (k[:110], "  ".join("%s %s->%s" % (x[:4], b.get(x), r.get(x)) for x in keys if r.get(x) != b.get(x))))
        else:
            print("%-110s %s" % (k[:110], "  ".join("%s=%s" % (x[:4], r.get(x)) for x in keys)))
        n += 1
    print("%d kernels %s" % (n, "differ" if base is not None else ""))


if __name__ == "__main__":
    a = sys.argv[1:]
    extra = []
    if "--" in a:
        i = a.index("--")
        a, extra = a[:i], a[i + 1:]
    tag = a[0]
    files = B.SOURCES
    base = None
    for i, x in enumerate(a):
        if x == "--files":
            files = a[i + 1].split(",")
        if x == "--diff":
            base = json.load(open(os.path.join(OUT, a[i + 1] + ".json")))["kernels"]
    show(study(tag, files, extra), base)

"""Development build of libvd_hip.so with extra hipcc flags on SOME source files (compiler-flag A/Bs through VD_HIP_LIB).

    python tools/probes/build_variant.py <name> "<file.hip>[,<file.hip>...]=<flags>" ["<files>=<flags>" ...]

Files without a spec re-use the product objects of versatile-diffusion_amd/build/ (same flags, same bytes); the variant is
linked to versatile-diffusion_amd/build/<name>/libvd_hip_<name>.so.  `all=<flags>` applies to every source.  The product
library and build.py are not touched.
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "versatile-diffusion_amd")
spec = importlib.util.spec_from_file_location("vd_build", os.path.join(PKG, "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main():
    name = sys.argv[1]
    extra = {}
    for sp in sys.argv[2:]:
        files, flags = sp.split("=", 1)
        for f in (B.SOURCES if files == "all" else files.split(",")):
            assert f in B.SOURCES, f
            extra.setdefault(f, []).extend(flags.split())
    B.build()   # product objects up to date
    odir = os.path.join(PKG, "build", name)
    os.makedirs(odir, exist_ok=True)
    objs, procs = [], []
    for s in B.SOURCES:
        if s not in extra:
            objs.append(os.path.join(PKG, "build", s.replace(".hip", ".o")))
            continue
        o = os.path.join(odir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(s, []) + extra[s] + ["-c", os.path.join(B.CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise SystemExit("hipcc failed on %s" % s)
    so = os.path.join(odir, "libvd_hip_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so, {k: " ".join(v) for k, v in extra.items()})


if __name__ == "__main__":
    main()

"""Build libvd_hip.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc.

    python versatile-diffusion_amd/build.py [--force] [--verbose]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only dev container; the
resulting .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# VD_BUILD_OUT: development builds of variants (VD_EXTRA_DEFS) next to the product library, for VD_HIP_LIB A/B runs
OUT = os.environ.get("VD_BUILD_OUT") or os.path.join(HERE, "libvd_hip.so")
SOURCES = ["gemm.hip", "conv_halo.hip", "conv_wstream.hip", "ff_fused.hip", "ff_chain.hip", "gemm_row320.hip", "norm.hip", "gn_fused.hip", "attention.hip", "xattn_fused.hip", "elementwise.hip", "preprocess.hip", "lowrank.hip"]
# every header under csrc/ (a kernel header that is not hashed would let a stale library pass the stamp check)
import glob
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "vd_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
# keep MFMA results in VGPRs where VALU code consumes them right away (softmax on the S tile): avoids the
# v_accvgpr_read/write shuffle and lowers the register footprint (2 -> 3 waves/SIMD for head dim 40)
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "xattn_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # GEMM: without it the accumulators bounce AGPR<->VGPR (64 reads + 64 writes) every K tile
               # (+ the AMDGPU register-pressure trackers where they were measured: halves ff_chain_kernel's spills, takes
               # splitk_reduce_stats_kernel<8> from 3 to 5 waves per SIMD; bit-identical outputs, profiles/r05_compiler_flags.txt)
               "gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-use-amdgpu-trackers"],
               "conv_halo.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "ff_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-use-amdgpu-trackers"],
               "ff_chain.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-use-amdgpu-trackers"],
               "gemm_row320.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _digest():
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stamp = OUT + ".stamp"
    dg = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == dg:
        return OUT
    if not os.path.exists(hipcc):
        if os.path.exists(OUT):
            # GPU box without the need to rebuild: trust the shipped library
            return OUT
        raise RuntimeError("hipcc not found at %s and no prebuilt libvd_hip.so" % hipcc)
    objs = []
    objdir = os.path.join(HERE, "build", os.path.basename(OUT).replace(".so", "") if os.environ.get("VD_BUILD_OUT") else "")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + os.environ.get("VD_EXTRA_DEFS", "").split() + ["-c", os.path.join(CSRC, s), "-o", o]
        objs.append(o)
        # per-object stamp (source + every header + the command line): an edit of one kernel file recompiles that file only
        h = hashlib.sha256(" ".join(cmd).encode())
        for p in [os.path.join(CSRC, s)] + HEADERS:
            with open(p, "rb") as f:
                h.update(f.read())
        ostamp = o + ".stamp"
        if not force and os.path.exists(o) and os.path.exists(ostamp) and open(ostamp).read().strip() == h.hexdigest():
            continue
        if verbose:
            print(" ".join(cmd))
        # the old stamp goes BEFORE hipcc overwrites the object: an interrupted or failed compile must not leave a stamp that
        # vouches for a truncated / newer object
        if os.path.exists(ostamp):
            os.remove(ostamp)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), ostamp, h.hexdigest()))
    failed = []
    for s, p, ostamp, odg in procs:   # reap EVERY compiler process before reporting a failure
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            failed.append(s)
            continue
        with open(ostamp, "w") as f:
            f.write(odg)
        if verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("hipcc failed on %s" % ", ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dg)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))

"""The boundary is a C ABI: include/vd_hip.h must be consumable by a plain C compiler (no HIP headers, no C++), and the
struct layout a C caller sees must be the one the ctypes binding (vd_hip/loader.py) uses.  A small C program is compiled
with gcc against the header, dlopen()s libvd_hip.so and exercises the entry points that need no device."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))

C_SRC = r"""
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "vd_hip.h"

typedef int (*abi_fn)(void);
typedef int (*plan_fn)(const VdGemmDesc*, int*, int*);
typedef size_t (*ws_fn)(const VdGemmDesc*);
typedef const char* (*err_fn)(void);
typedef const char* (*name_fn)(int);

int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    abi_fn abi = (abi_fn)dlsym(h, "vd_abi_version");
    plan_fn plan = (plan_fn)dlsym(h, "vd_gemm_plan");
    ws_fn ws = (ws_fn)dlsym(h, "vd_gemm_workspace_bytes");
    err_fn err = (err_fn)dlsym(h, "vd_last_error");
    name_fn name = (name_fn)dlsym(h, "vd_gemm_config_name");
    if (!abi || !plan || !ws || !err || !name) return 3;
    VdGemmDesc d;
    memset(&d, 0, sizeof d);
    d.a0 = d.w = (const void*)16; d.out = (void*)16; d.ws = (float*)16;
    d.M = 2048; d.N = 1280; d.K = 11520; d.ksize = 3; d.stride = 1; d.pad = 1; d.c0 = 1280;
    d.Hin = d.Win = d.Hout = d.Wout = 16;
    int cfg = -1, ns = -1;
    int rc = plan(&d, &cfg, &ns);
    printf("abi=%d version_macro=%d sizeof=%zu off_M=%zu off_stride_a=%zu off_colsum=%zu off_sync=%zu off_ln_stats=%zu off_out_stats=%zu off_stat_img_rows=%zu off_gn_gamma=%zu off_gn_eps=%zu off_skip_a0=%zu off_skip_w=%zu off_skip_c0=%zu off_skip_ldw=%zu off_row_sums=%zu off_stat_sums=%zu\n",
           abi(), VD_HIP_ABI_VERSION, sizeof(VdGemmDesc), offsetof(VdGemmDesc, M), offsetof(VdGemmDesc, stride_a),
           offsetof(VdGemmDesc, colsum), offsetof(VdGemmDesc, sync), offsetof(VdGemmDesc, ln_stats),
           offsetof(VdGemmDesc, out_stats), offsetof(VdGemmDesc, stat_img_rows), offsetof(VdGemmDesc, gn_gamma),
           offsetof(VdGemmDesc, gn_eps), offsetof(VdGemmDesc, skip_a0), offsetof(VdGemmDesc, skip_w),
           offsetof(VdGemmDesc, skip_c0), offsetof(VdGemmDesc, skip_ldw), offsetof(VdGemmDesc, row_sums), offsetof(VdGemmDesc, stat_sums));
    printf("plan rc=%d cfg=%d ns=%d name=%s ws=%zu\n", rc, cfg, ns, name(cfg), ws(&d));
    d.K = 12;   /* not a multiple of 8: rejected with a message, no device touched */
    rc = plan(&d, &cfg, &ns);
    printf("bad rc=%d msg=%s\n", rc, err());
    return 0;
}
"""


def test_header_compiles_as_plain_c_and_layout_matches_ctypes(tmp_path):
    from vd_hip.loader import VdGemmDesc, lib_path
    src = tmp_path / "capi.c"
    src.write_text(C_SRC)
    exe = tmp_path / "capi"
    cc = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-ldl"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe), lib_path()], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    lines = run.stdout.strip().splitlines()
    kv = dict(t.split("=") for t in lines[0].split())
    assert int(kv["abi"]) == int(kv["version_macro"]) == 8
    assert int(kv["sizeof"]) == ctypes.sizeof(VdGemmDesc)
    for field, key in (("M", "off_M"), ("stride_a", "off_stride_a"), ("colsum", "off_colsum"), ("sync", "off_sync"), ("ln_stats", "off_ln_stats"),
                       ("out_stats", "off_out_stats"), ("stat_img_rows", "off_stat_img_rows"), ("gn_gamma", "off_gn_gamma"),
                       ("gn_eps", "off_gn_eps"), ("skip_a0", "off_skip_a0"), ("skip_w", "off_skip_w"), ("skip_c0", "off_skip_c0"),
                       ("skip_ldw", "off_skip_ldw"), ("row_sums", "off_row_sums"), ("stat_sums", "off_stat_sums")):
        assert int(kv[key]) == getattr(VdGemmDesc, field).offset, field
    # same plan as the ctypes path (tests/test_gemm_planner_cpu.py): the 16x16-level 3x3 conv runs on the halo kernel, 4-way split
    assert lines[1].startswith("plan rc=0 cfg=29 ns=4 name=conv3x3_halo_kernel<256,160,32,160,512,2>")
    assert "bad rc=-" in lines[2] and "multiple of 8" in lines[2]

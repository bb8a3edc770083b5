"""vd_gemm_plan (tile shape / split-K cost model) is host code: its decisions for the UNet's shapes are pinned here so a
retune that silently changes them shows up in the CPU suite.  No GPU work is launched."""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))

T128x128, T128x64, T64x64, T128x128w8, T128x320 = 0, 1, 2, 3, 7
T128x64d, T64x64d = 14, 15                                   # 3-stage ring for grids that do not fill the chip
T128x128q, T128x128w8q, T128x320q, T128x160q = 16, 19, 20, 21   # 32-deep K tiles, 4-stage ring: VD_GEMM_VARIANT=q|h only
T128x64w8, T256x256 = 4, 24
NCFG = 27                                                    # vd_gemm_num_configs(): classic gemm_f16_kernel instantiations
H160p, H128p = NCFG + 2, NCFG + 5                            # conv3x3_halo_kernel<256,160,...,2> / <256,128,...,2>


def plan(M, N, K, ks=1, B=8, ws=True, act=0, halo=-1):
    from vd_hip.loader import VdGemmDesc, lib
    assert lib().vd_conv_halo_set_variant(halo) == 0        # 0: 3x3 convs stay on gemm_f16_kernel (the classic planner)
    d = VdGemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a0 = d.w = d.out = 16
    if ks == 3:
        h = int(round((M // B) ** 0.5))
        d.Hin = d.Win = d.Hout = d.Wout = h
        d.ksize, d.stride, d.pad, d.c0 = 3, 1, 1, K // 9
    d.act = act
    d.ws = 16 if ws else None
    cfg, ns = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
    return cfg.value, ns.value


def test_round_quantisation_drives_the_split():
    # classic planner (halo kernel off): 160 tiles of 128x128: split 3 = 480 blocks (one round of 512 slots), not split 4 = 640
    assert plan(2048, 1280, 11520, ks=3, halo=0) == (T128x128, 3)
    assert plan(2048, 1280, 23040, ks=3, halo=0) == (T128x128, 3)
    # 320 tiles already cover most of a round: no split
    assert plan(8192, 640, 5760, ks=3, halo=0) == (T128x128, 1)
    # 40 tiles at the 8x8 level: deep split (also with the halo kernel on: 16 patches x column tiles are too few for it; its
    # small-M variant -- 128-pixel patches x 32 columns, whole K per block -- is opt-in / forced only)
    for halo in (0, -1):
        cfg, ns = plan(512, 1280, 11520, ks=3, halo=halo)
        assert cfg in (T128x128, T128x64, T128x64d) and 5 <= ns <= 12
    from vd_hip.loader import lib
    # development variants of the halo kernel (whole-K small-M blocks, other barrier placements, ...) are no longer instantiated
    for removed in (1, 2, 4, 5, 7, 8, 9, 10, 11, 12):
        assert lib().vd_conv_halo_set_variant(removed) != 0
    assert lib().vd_conv_halo_set_variant(-1) == 0


def test_halo_kernel_takes_the_3x3_convolutions():
    from vd_hip.loader import lib
    assert lib().vd_gemm_num_configs() == NCFG
    assert lib().vd_gemm_config_name(H160p) == b"conv3x3_halo_kernel<256,160,32,160,512,2>"
    # 64x64 level: 128 patches x 2 column tiles = one block per CU, no split
    for K in (2880, 5760, 8640):
        assert plan(32768, 320, K, ks=3) == (H160p, 1)
    # 32x32 level: 32 patches x 4 column tiles: split the channel chunks in two to cover the chip
    assert plan(8192, 640, 5760, ks=3) == (H160p, 2)
    assert plan(8192, 640, 2880, ks=3) == (H160p, 2)
    # 16x16 level: 8 patches (one image each) x 8 column tiles: four ways
    assert plan(2048, 1280, 11520, ks=3) == (H160p, 4)
    assert plan(2048, 1280, 23040, ks=3) == (H160p, 4)
    # no workspace, no split; widths that 160 does not divide but 128 does (VAE): 64x64 wave tiles
    assert plan(8192, 640, 5760, ks=3, ws=False) == (H160p, 1)
    assert plan(4 * 64 * 64, 512, 9 * 512, ks=3, B=4) == (H128p, 1)
    # the 4-channel output head and plain GEMMs never go there
    assert plan(32768, 4, 2880, ks=3)[0] < NCFG
    assert plan(32768, 320, 320)[0] < NCFG


def test_wide_tile_for_the_64x64_level():
    for K, ks in ((2880, 3), (5760, 3), (1280, 1)):
        assert plan(32768, 320, K, ks=ks, halo=0) == (T128x320, 1)     # one block spans all of N: A is read from L2 once
    assert plan(32768, 320, 320) == (T128x64w8, 1)            # short K, many rows: small tiles, 4 waves per SIMD
    assert plan(8192, 640, 640) == (T128x128w8, 1)            # round 5: N = 640 at M >= 8192 takes the 128x128 tile (in-forward A/B)
    assert plan(4096, 640, 640) == (T128x64d, 1)              # ... below that (CFG batch 4) the 4-wave 128x64 tile with three stages
    assert plan(16384, 320, 320) == (T128x64w8, 1)
    assert plan(2048, 1280, 1280) == (T128x64d, 1)            # 640 blocks of 64x64 -> 320 co-resident 128x64 blocks, three stages
    assert plan(512, 1280, 1280)[0] == T64x64d                # the 8x8 level keeps its (already deep) 64x64 grid
    assert plan(32768, 960, 320) == (T128x320, 1)             # N > 640: the wide tile
    # 64 tiles of 128x320 would leave 3/4 of the CUs idle
    assert plan(8192, 640, 5760, ks=3, halo=0)[0] not in (T128x320, T128x320q, T128x160q)


def test_no_split_without_workspace_and_for_geglu():
    assert plan(2048, 1280, 11520, ks=3, ws=False)[1] == 1
    assert plan(2048, 1280, 11520, ks=3, ws=False, halo=0)[1] == 1
    assert plan(32768, 2560, 320, act=1) == (T128x128w8, 1)    # VD_ACT_GEGLU = 1: 8 waves, value / gate tile pairs
    assert plan(2048, 10240, 1280, act=1) == (T128x128w8, 1)
    assert plan(512, 10240, 1280, act=1) == (T128x128w8, 1)


def test_layernorm_fold_uses_fold_instances():
    from vd_hip.loader import VdGemmDesc, lib
    fold_ok = {T128x128, T128x64, T64x64, T128x128w8, T128x64w8, T128x320, T64x64d, T256x256}
    for (M, N, K, act) in ((32768, 960, 320, 0), (32768, 2560, 320, 1), (512, 3840, 1280, 0), (2048, 1280, 1280, 0), (8, 1280, 320, 0)):
        d = VdGemmDesc()
        d.M, d.N, d.K, d.act = M, N, K, act
        d.a0 = d.w = d.out = d.colsum = d.ln_stats = 16
        d.flags = 32                                          # VD_EPI_LNFOLD
        d.ln_eps = 1e-5
        cfg, ns = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
        assert cfg.value in fold_ok and ns.value == 1, (M, N, K, cfg.value, ns.value)
    # q|k|v of the 32x32 level (K = 640) at every workload's row count: 8 waves on the 128x128 tile; the 16x16 level keeps 4 waves
    for (M, N, K, want) in ((8192, 1920, 640, T128x128w8), (4096, 1920, 640, T128x128w8), (16384, 1920, 640, T128x128w8), (2048, 3840, 1280, T128x128)):
        d = VdGemmDesc()
        d.M, d.N, d.K = M, N, K
        d.a0 = d.w = d.out = d.colsum = d.ln_stats = 16
        d.flags, d.ln_eps = 32, 1e-5
        cfg, ns = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
        assert (cfg.value, ns.value) == (want, 1), (M, N, K, cfg.value)


def test_small_m_weight_streaming_splits_k():
    cfg, ns = plan(8, 5120, 5120)
    assert cfg in (T64x64, T64x64d) and ns >= 8
    assert plan(8, 1280, 320) == (T64x64, 1)


def test_plan_rejects_bad_descriptors():
    from vd_hip.loader import VdGemmDesc, lib
    d = VdGemmDesc()
    d.M, d.N, d.K = 128, 128, 12   # K not a multiple of 8
    d.a0 = d.w = d.out = 16
    cfg, ns = ctypes.c_int(0), ctypes.c_int(0)
    assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) != 0
    assert b"multiple of 8" in lib().vd_last_error()


def test_tuned_entry_with_split_is_honoured_on_the_relaunch():
    """ops.gemm plans first (split_k = 0), then launches with split_k = the planned factor: the second plan must pick the
    tuned tile again instead of the cost model's tile for that split."""
    from vd_hip.loader import VdGemmDesc, lib
    M, N, K = 2048, 1280, 5120
    try:
        assert lib().vd_gemm_tune_set(M, N, K, 1, 0, T128x64, 4) == 0
        for split_k, want in ((0, (T128x64, 4)), (4, (T128x64, 4))):
            d = VdGemmDesc()
            d.M, d.N, d.K, d.split_k = M, N, K, split_k
            d.a0 = d.w = d.out = d.ws = 16
            cfg, ns = ctypes.c_int(-1), ctypes.c_int(-1)
            assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
            assert (cfg.value, ns.value) == want, (split_k, cfg.value, ns.value)
        d = VdGemmDesc()                      # a different caller-fixed split: the entry does not apply
        d.M, d.N, d.K, d.split_k = M, N, K, 2
        d.a0 = d.w = d.out = d.ws = 16
        cfg, ns = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
        assert ns.value == 2
    finally:
        lib().vd_gemm_tune_clear()


def stat_rows(M, N, K, ks=1, B=8, ws=True, flags=0, act=0, img_rows=0, conv1x1=False, sync=False, colsum=False):
    from vd_hip.loader import VdGemmDesc, lib
    assert lib().vd_conv_halo_set_variant(-1) == 0
    d = VdGemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a0 = d.w = d.out = 16
    if ks == 3 or conv1x1:
        h = int(round((M // B) ** 0.5))
        d.Hin = d.Win = d.Hout = d.Wout = h
        d.ksize, d.stride, d.pad, d.c0 = (3, 1, 1, K // 9) if ks == 3 else (1, 1, 0, K)
    d.act, d.flags, d.stat_img_rows = act, flags, img_rows
    d.ws = 16 if ws else None
    d.sync = 16 if sync else None
    if colsum:
        d.colsum = d.ln_stats = 16
    rows = ctypes.c_int(-1)
    assert lib().vd_gemm_stat_rows(ctypes.byref(d), ctypes.byref(rows)) == 0
    return rows.value


def test_which_launches_emit_groupnorm_statistics():
    """vd_gemm_stat_rows: rows per out_stats partial of the launch the planner picks (round 4, csrc/gn_fused.hip)."""
    # halo conv, one block per 256-pixel patch: the epilogue emits one partial per patch
    assert stat_rows(32768, 320, 2880, ks=3) == 256
    # 32x32 / 16x16 / 8x8 levels run split over K: the reduce kernel emits one partial per 64 rows
    assert stat_rows(8192, 640, 5760, ks=3) == 64
    assert stat_rows(2048, 1280, 11520, ks=3) == 64
    assert stat_rows(512, 1280, 11520, ks=3) == 64
    # ... or, with ticket counters, the halo conv's own epilogue (run by the last block of a tile): one partial per patch
    assert stat_rows(2048, 1280, 11520, ks=3, sync=True) == 256
    assert stat_rows(512, 1280, 11520, ks=3, sync=True) == 0      # gemm_f16_kernel's last-arriver fix-up emits none
    # SpatialTransformer.proj_out: 1x1 conv on gemm_f16_kernel, one partial per tile (128 rows), or per image where a
    # tile spans several 8x8 images
    assert stat_rows(32768, 320, 320, conv1x1=True) == 128
    assert stat_rows(512, 1280, 1280, conv1x1=True) == 64
    # plain matrices need the rows of one sample; without it the whole matrix is one "image"
    assert stat_rows(32768, 320, 64, img_rows=4096) in (64, 128)
    assert stat_rows(32768, 320, 64) in (64, 128)
    # GEGLU / fp32 output / LayerNorm-fold launches never feed a GroupNorm
    assert stat_rows(8192, 5120, 640, act=1) == 0
    assert stat_rows(8192, 640, 640, flags=16) == 0
    assert stat_rows(8192, 1920, 640, flags=32, colsum=True) == 0


def test_which_convolutions_take_the_folded_skip_convolution():
    """vd_gemm_skip_ok: ResBlock's skip 1x1 convolution rides as extra K of the second 3x3 conv where the halo-resident kernel
    takes the launch (its SKIP instance, variant 12); the 8x8 level and the upsampling / strided convs do not -- there the caller
    runs the 1x1 convolution itself (the weight-streaming kernel has its own entry, vd_conv3x3_wstream_f16)."""
    from vd_hip.loader import VdGemmDesc, lib
    assert lib().vd_conv_halo_set_variant(-1) == 0

    def ok(B, h, c, cs0, cs1, ups=0, stride=1, res=False):
        d = VdGemmDesc()
        d.M, d.N, d.K = B * (h << ups) ** 2 // (stride * stride), c, 9 * c
        d.a0 = d.w = d.out = d.ws = 16
        d.Hin = d.Win = h
        d.Hout = d.Wout = (h << ups) // stride
        d.ksize, d.stride, d.pad, d.ups, d.c0 = 3, stride, 1, ups, c
        d.skip_a0, d.skip_w, d.skip_c0 = 16, 16, cs0
        if cs1:
            d.skip_a1, d.skip_c1 = 16, cs1
        if res:
            d.res, d.flags = 16, 8   # VD_EPI_RESIDUAL
        return lib().vd_gemm_skip_ok(ctypes.byref(d))

    assert ok(8, 64, 320, 640, 320) == 1      # output block at the 64x64 level
    assert ok(8, 32, 640, 1280, 640) == 1     # 32x32: split over chunks, the skip chunks shared between the splits
    assert ok(8, 16, 1280, 1280, 1280) == 1
    assert ok(8, 32, 640, 320, 0) == 1        # input block 4: single source
    assert ok(8, 8, 1280, 1280, 1280) == 0    # 8x8 level: gemm_f16_kernel / the weight stream, not the halo kernel
    assert ok(8, 32, 640, 640, 0, ups=1) == 0   # Upsample conv: no skip path in the upsampling instance
    assert ok(8, 64, 320, 100, 0) == 0        # channels not a multiple of 64
    cfg = ctypes.c_int(-1); ns = ctypes.c_int(-1)
    d = VdGemmDesc()
    d.M, d.N, d.K = 32768, 320, 2880
    d.a0 = d.w = d.out = d.ws = 16
    d.Hin = d.Win = d.Hout = d.Wout = 64
    d.ksize, d.stride, d.pad, d.c0 = 3, 1, 1, 320
    d.skip_a0, d.skip_w, d.skip_c0 = 16, 16, 640
    assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
    assert cfg.value == NCFG + 12 and lib().vd_gemm_config_name(cfg.value) == b"conv3x3_halo_kernel<256,160,32,160,512,2,skip>"


def test_which_projections_take_the_weight_streaming_gemm():
    """vd_gemm_wstream_supported: plain single-source GEMMs with whole 128-row / 256-column tiles and 64-deep chunks of K and a
    plain fp16 epilogue (the FF-out projection of the deep levels); everything else stays on gemm_f16_kernel."""
    from vd_hip.loader import VdGemmDesc, lib

    def ok(M, N, K, **kw):
        d = VdGemmDesc()
        d.M, d.N, d.K = M, N, K
        d.a0 = d.w = d.out = d.ws = 16
        for k, v in kw.items():
            setattr(d, k, v)
        return lib().vd_gemm_wstream_supported(ctypes.byref(d))

    assert ok(2048, 1280, 5120) == 1 and ok(512, 1280, 5120) == 1 and ok(128, 256, 64) == 1
    assert ok(2048, 640, 2560) == 0          # N = 640: no whole 256-column tiles (the 32x32 level stays on gemm_f16_kernel)
    assert ok(2000, 1280, 5120) == 0         # ragged rows
    assert ok(2048, 1280, 5100) == 0         # K not in 64-deep chunks
    assert ok(2048, 1280, 5120, act=1) == 0  # GEGLU epilogue
    assert ok(2048, 1280, 5120, a1=16, c1=64) == 0   # two-source A
    assert ok(2048, 1280, 5120, ksize=3, stride=1, pad=1) == 0


def test_which_8x8_convolutions_run_without_a_split_and_which_launches_take_row_sums(monkeypatch):
    """Round 5, host decisions only: vd_conv3x3_wstream_plan answers 0 (whole-K kernel: no workspace, no reduce launch) for the
    8x8-level convolutions whose 128-pixel x 32-channel tiles cover half the chip, the launcher's split otherwise (folded skip
    convolution, too few image groups, explicit split, VD_WSK=0); vd_gemm_row_sums_ok admits VdGemmDesc.row_sums exactly for
    unsplit gemm_f16_kernel launches with vector-aligned fp16 output and power-of-two segment counts per tile row."""
    from vd_hip.loader import VdGemmDesc, lib

    def conv8(B, cin, n=1280, skip=0, split=0, flags=0):
        d = VdGemmDesc()
        d.M, d.N, d.K = B * 64, n, 9 * cin
        d.a0 = d.w = d.out = 16
        d.Hin = d.Win = d.Hout = d.Wout = 8
        d.ksize, d.stride, d.pad, d.c0 = 3, 1, 1, cin
        d.split_k, d.flags = split, flags
        if skip:
            d.skip_a0 = d.skip_w = 16
            d.skip_c0 = skip
        ns = ctypes.c_int(-1)
        assert lib().vd_conv3x3_wstream_plan(ctypes.byref(d), ctypes.byref(ns)) == 0
        return ns.value

    monkeypatch.delenv("VD_WSK", raising=False)
    monkeypatch.delenv("VD_WSK_MIN_BLOCKS", raising=False)
    assert conv8(8, 1280) == 0 and conv8(8, 2560) == 0         # bench shape: 4 image groups x 40 column tiles = 160 blocks
    assert conv8(8, 1280, skip=2560) == 10                      # the folded skip convolution stays on the split kernel
    assert conv8(4, 1280) == 20                                 # 80 whole-K tiles < 128: 10 split-kernel tiles, one chunk per split
    assert conv8(8, 1280, split=5) == 5                         # an explicit split factor is honoured
    assert conv8(8, 128) == 2                                   # 2 chunks: below the whole-K kernel's pipeline depth
    monkeypatch.setenv("VD_WSK", "0")
    assert conv8(8, 1280) == 10

    def rs_ok(M, N, K, res=False, rowvec=False, act=0, f32=False, ks=1, ws=False):
        d = VdGemmDesc()
        d.M, d.N, d.K = M, N, K
        d.a0 = d.w = d.out = 16
        d.act = act
        d.flags = (4 if res else 0) | (2 if rowvec else 0) | (16 if f32 else 0)
        if res:
            d.res = 16
        if rowvec:
            d.rowvec, d.rows_per_batch = 16, M
        if ks == 3:
            h = int(round((M // 8) ** 0.5))
            d.Hin = d.Win = d.Hout = d.Wout = h
            d.ksize, d.stride, d.pad, d.c0 = 3, 1, 1, K // 9
        d.ws = 16 if ws else None
        return lib().vd_gemm_row_sums_ok(ctypes.byref(d))

    for (M, C) in ((8192, 640), (2048, 1280), (512, 1280)):     # to_out / proj_in of the 32x32 .. 8x8 levels
        assert rs_ok(M, C, C, res=True) == 1 and rs_ok(M, C, C) == 1
    assert rs_ok(8192, 5120, 640, act=1) == 0                   # GEGLU output is not the row a LayerNorm normalises
    assert rs_ok(8192, 640, 640, f32=True) == 0
    assert rs_ok(8192, 640, 640, res=True, rowvec=True) == 0    # both operands: the element-wise write-out path
    assert rs_ok(32768, 320, 2880, ks=3) == 0                   # halo-resident convolution: another epilogue
    assert rs_ok(2048, 1280, 5120, ws=True) in (0, 1)           # (whatever the planner picks, the answer is consistent with ...)
    assert rs_ok(8192, 644, 640) == 0                           # N % 8 != 0

"""Index arithmetic of conv3x3_halo_kernel (versatile-diffusion_amd/csrc/conv_halo_kernel.h) replayed on the CPU.

The kernel's correctness rests on four address maps that no compiler checks: (1) which input pixel / 16-byte slot every
lane of every LDS-DMA piece fetches (halo image, XOR swizzle applied on the source side, zero padding through out-of-range
offsets), (2) the fragment address a lane reads for (tap, k-step), (3) the weight-tile image and its fragment offsets,
(4) the patch-order -> pixel-order output row map.  This test restates them with the kernel's own expressions (same names)
and checks, for every geometry the launcher accepts in the UNet / VAE, that the MFMA operands a lane would receive are
exactly X[pixel + tap][k-slice] and W[n][tap][k-slice], and that the output map is a bijection onto the M rows.
No GPU work; numpy only.
"""
import numpy as np
import pytest

BM = 256
OOB = -1


def geometry(nimg, Hin, Win, ups, BM=BM):
    """halo_geometry() of conv_halo.hip"""
    Hv, Wv = Hin << ups, Win << ups
    tw = 32 if Wv % 32 == 0 else 16 if Wv % 16 == 0 else 8 if Wv % 8 == 0 else 0
    if tw == 0:
        return None
    th = BM // tw
    if Hv % th == 0:
        ngrp, rg = 1, th
    elif th % Hv == 0 and tw == Wv and (Hv & (Hv - 1)) == 0 and nimg % (th // Hv) == 0:
        ngrp, rg = th // Hv, Hv
    else:
        return None
    g = dict(tw=tw, ltw=tw.bit_length() - 1, rg=rg, ngrp=ngrp, lgsz=(tw * rg).bit_length() - 1, pitch=tw + 2, Hv=Hv, Wv=Wv,
             Hin=Hin, Win=Win, ups=ups, nimg=nimg)
    g["gpx"] = (rg + 2) * g["pitch"]
    g["hpx"] = ngrp * g["gpx"]
    if g["hpx"] > BM * 100 // 64 + 16:
        return None
    g["mg_pitch"] = (1 << 20) // g["pitch"] + 1
    g["mg_gpx"] = (1 << 20) // g["gpx"] + 1
    g["tiles_x"] = Wv // tw
    g["tiles_y"] = Hv // rg if ngrp == 1 else 1
    g["halo_bytes"] = ((g["hpx"] + 7) // 8) * 1024
    g["tiles_m"] = nimg * Hv * Wv // BM
    return g


def patch_origin(g, tm):
    if g["ngrp"] == 1:
        tpi = g["tiles_x"] * g["tiles_y"]
        img0 = tm // tpi
        r = tm - img0 * tpi
        ty = r // g["tiles_x"]
        return img0, ty * g["rg"], (r - ty * g["tiles_x"]) * g["tw"]
    return tm * g["ngrp"], 0, 0


def build_halo_image(g, tm, x, NW):
    """LDS halo buffer (as [halo pixel][8 slots] of (pixel id, logical slot) or zeros) as the DMA pieces write it."""
    img0, y0, x0 = patch_origin(g, tm)
    npieces = g["halo_bytes"] // 1024
    lds = np.full((npieces * 8, 8, 2), -7, dtype=np.int64)      # -7 = never written
    HPXMAX = BM * 100 // 64 + 16
    NHP = (HPXMAX + 7) // 8
    HPW = (NHP + NW - 1) // NW
    HPT = (HPW + 7) // 8
    MAXHP = HPT * 8
    for wave in range(NW):
        for j in range(MAXHP):
            q = j * NW + wave
            if not (q * 8 < g["hpx"]):
                continue
            assert j // HPT < 8, "piece must be issued in taps 0..7"
            for lane in range(64):
                hp = q * 8 + (lane >> 3)
                grp = (hp * g["mg_gpx"]) >> 20
                rem = hp - grp * g["gpx"]
                hy = (rem * g["mg_pitch"]) >> 20
                hx = rem - hy * g["pitch"]
                if hp < g["hpx"]:
                    assert grp == hp // g["gpx"] and hy == rem // g["pitch"]
                vy, vx = y0 + hy - 1, x0 + hx - 1
                ok = hp < g["hpx"] and 0 <= vy < g["Hv"] and 0 <= vx < g["Wv"]
                pix = ((img0 + grp) * g["Hin"] + (vy >> g["ups"])) * g["Win"] + (vx >> g["ups"])
                slot = (lane & 7) ^ ((hp >> 1) & 7)
                # destination: piece base + lane * 16  ->  halo pixel hp, physical slot lane & 7
                assert q * 1024 + lane * 16 == hp * 128 + (lane & 7) * 16
                lds[hp, lane & 7] = (pix, slot) if ok else (OOB, slot)
    return lds, (img0, y0, x0)


@pytest.mark.parametrize("nimg,Hin,Win,ups", [(8, 64, 64, 0), (8, 32, 32, 0), (8, 16, 16, 0), (8, 8, 8, 0), (4, 32, 32, 1),
                                               (8, 16, 16, 1), (8, 8, 8, 1), (2, 96, 96, 0), (2, 48, 48, 0), (1, 512, 512, 0),
                                               (1, 256, 256, 0), (4, 128, 128, 0), (4, 64, 64, 1), (3, 40, 24, 0), (2, 16, 16, 0)])
@pytest.mark.parametrize("NW,WM", [(8, 32), (8, 64), (4, 64)])
def test_halo_and_fragment_addresses(nimg, Hin, Win, ups, NW, WM):
    g = geometry(nimg, Hin, Win, ups)
    if g is None:
        pytest.skip("geometry not accepted by the halo launcher (falls back to gemm_f16_kernel)")
    _check_halo_and_fragments(g, nimg, Hin, ups, NW, WM)


@pytest.mark.parametrize("nimg,Hin,Win,ups", [(8, 8, 8, 0), (2, 8, 8, 0), (4, 16, 16, 0), (2, 32, 32, 0), (6, 8, 8, 0)])
def test_small_m_variant_addresses(nimg, Hin, Win, ups, monkeypatch):
    """conv3x3_halo_kernel<128,32,32,32,256,4> (variant 10): 128-pixel patches (two whole 8x8 images, 16 x 8 at 16x16, 4 rows x 32
    columns at 32x32), four waves of 32 pixels -- same index maps with BM = 128."""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "BM", 128)
    g = geometry(nimg, Hin, Win, ups, BM=128)
    assert g is not None and g["tiles_m"] == nimg * Hin * Win // 128
    _check_halo_and_fragments(g, nimg, Hin, ups, 4, 32)


def _check_halo_and_fragments(g, nimg, Hin, ups, NW, WM):
    MI = WM // 32
    waves_m = BM // WM
    rng = np.random.RandomState(nimg * 1000 + Hin + ups)
    for tm in sorted(set([0, g["tiles_m"] - 1, g["tiles_m"] // 2] + list(rng.randint(0, g["tiles_m"], 3)))):
        lds, (img0, y0, x0) = build_halo_image(g, tm, None, NW)
        assert (lds[: g["hpx"], :, 0] != -7).all(), "every halo pixel of the patch must be written by some piece"
        for wm in range(waves_m):
            for i in range(MI):
                for l31 in range(32):
                    m = wm * WM + i * 32 + l31
                    grp = m >> g["lgsz"]
                    r = m - (grp << g["lgsz"])
                    hp_base = grp * g["gpx"] + (r >> g["ltw"]) * g["pitch"] + (r & (g["tw"] - 1))
                    oy, ox = y0 + (r >> g["ltw"]), x0 + (r & (g["tw"] - 1))       # output pixel of this lane
                    for t in range(9):
                        ky, kx = t // 3, t % 3
                        tapoff = ky * g["pitch"] + kx
                        for hi in range(2):
                            hp = hp_base + tapoff
                            xk = ((hp >> 1) & 7) ^ hi
                            a0 = (hp << 7) + (xk << 4)
                            for ks in range(4):
                                addr = a0 ^ (ks << 5)
                                pixel, phys = addr >> 7, (addr >> 4) & 7
                                src, slot = lds[pixel, phys]
                                assert slot == 2 * ks + hi, "lane must receive k-slice 2 ks + hi of the 64-channel chunk"
                                vy, vx = oy + ky - 1, ox + kx - 1
                                if 0 <= vy < g["Hv"] and 0 <= vx < g["Wv"]:
                                    want = ((img0 + grp) * g["Hin"] + (vy >> ups)) * g["Win"] + (vx >> ups)
                                else:
                                    want = OOB
                                assert src == want, (tm, m, t, hi, ks)


@pytest.mark.parametrize("BN,NW,WN", [(160, 8, 160), (128, 8, 64), (160, 4, 160)])
def test_weight_tile_image_and_fragments(BN, NW, WN):
    NPW = BN // 8
    WPW = (NPW + NW - 1) // NW
    lds = np.full((BN, 8), -7, dtype=np.int64)
    for wave in range(NW):
        for j in range(WPW):
            q = j * NW + wave
            if q >= NPW:
                continue
            for lane in range(64):
                r = q * 8 + (lane >> 3)
                slot = (lane & 7) ^ ((r >> 1) & 7)
                lds[r, lane & 7] = r * 8 + slot          # (tile row, logical slot)
    assert (lds >= 0).all()
    for wn in range(BN // WN):
        for j in range(WN // 32):
            for l31 in range(32):
                for hi in range(2):
                    for ks in range(4):
                        r = wn * WN + l31
                        s = ks * 2 + hi
                        off = r * 128 + ((s ^ ((r >> 1) & 7)) << 4) + j * 32 * 128      # lds_off_kb<64> + fragment immediate
                        row, phys = off >> 7, (off >> 4) & 7
                        assert lds[row, phys] == (wn * WN + j * 32 + l31) * 8 + s


@pytest.mark.parametrize("nimg,Hin,Win,ups", [(8, 64, 64, 0), (8, 16, 16, 0), (8, 8, 8, 0), (4, 32, 32, 1), (2, 96, 96, 0), (3, 40, 24, 0)])
def test_output_row_map_is_a_bijection(nimg, Hin, Win, ups):
    g = geometry(nimg, Hin, Win, ups)
    if g is None:
        pytest.skip("geometry not accepted")
    M = nimg * g["Hv"] * g["Wv"]
    seen = np.zeros(M, dtype=np.int32)
    for tm in range(g["tiles_m"]):
        img0, y0, x0 = patch_origin(g, tm)
        m = np.arange(BM)
        grp = m >> g["lgsz"]
        r = m - (grp << g["lgsz"])
        row = ((img0 + grp) * g["Hv"] + y0 + (r >> g["ltw"])) * g["Wv"] + x0 + (r & (g["tw"] - 1))
        seen[row] += 1
    assert (seen == 1).all()


def test_lds_budget_of_the_accepted_geometries():
    for (nimg, H, W, ups) in [(8, 64, 64, 0), (8, 32, 32, 0), (8, 16, 16, 0), (8, 8, 8, 0), (1, 512, 512, 0), (2, 96, 96, 0)]:
        g = geometry(nimg, H, W, ups)
        for bn, wst in ((160, 3), (160, 2), (128, 3)):
            assert 2 * g["halo_bytes"] + wst * bn * 128 <= 160 * 1024, (H, bn, wst)

"""CPU-only checks of the host side: config bank, registry, state-dict layout, schedules, the C ABI surface,
and that the product path refuses to run without the GPU (no silent fallback)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from vdtest_util import GOLD, load_gold, meta, tiny_vd_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VD_QUIET", "1")


def test_capi_exports_every_declared_symbol():
    """libvd_hip.so loads on a CPU host and exports exactly what include/vd_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "vd_hip.h")).read()
    declared = set(re.findall(r"\b(vd_[a-z0-9_]+)\s*\(", hdr))
    from vd_hip.loader import PROTOTYPES, lib
    h = lib()
    assert declared == set(PROTOTYPES), (declared ^ set(PROTOTYPES))
    for name in declared:
        assert hasattr(h, name), name
    assert h.vd_abi_version() == 8
    # argument validation works without a device
    from vd_hip.loader import VdGemmDesc
    d = VdGemmDesc()
    d.M, d.N, d.K = 4, 4, 68
    assert h.vd_gemm_f16(ctypes.byref(d), None) < 0 and b"multiple of 8" in h.vd_last_error()
    assert h.vd_groupnorm_workspace_bytes(8, 4096, 320, 32) > 0


def test_gemm_desc_struct_matches_header():
    from vd_hip.loader import VdGemmDesc
    skip = 3 * 8 + 6 * 4 + 8 + 8   # folded skip convolution (ABI 5): three pointers, five ints + one reserved; row_sums (ABI 6); stat_sums (ABI 7)
    assert ctypes.sizeof(VdGemmDesc) == 8 * 8 + 24 * 4 + 4 * 8 + 8 + 2 * 4 + 8 + 8 + 8 + 2 * 4 + 2 * 8 + 2 * 4 + skip
    assert VdGemmDesc.stride_a.offset == 8 * 8 + 24 * 4
    assert VdGemmDesc.colsum.offset == 8 * 8 + 24 * 4 + 4 * 8   # LayerNorm-fold fields (ABI 2)
    assert VdGemmDesc.sync.offset == ctypes.sizeof(VdGemmDesc) - 56 - skip   # split-K arrival counters (ABI 2)
    assert VdGemmDesc.ln_stats.offset == ctypes.sizeof(VdGemmDesc) - 48 - skip   # LayerNorm-fold row statistics (ABI 2)
    assert VdGemmDesc.out_stats.offset == ctypes.sizeof(VdGemmDesc) - 40 - skip   # producer-emitted GroupNorm statistics (ABI 5)
    assert VdGemmDesc.gn_gamma.offset == ctypes.sizeof(VdGemmDesc) - 24 - skip    # GroupNorm fused into the split-K reduce (ABI 5)
    assert VdGemmDesc.skip_a0.offset == ctypes.sizeof(VdGemmDesc) - skip
    assert VdGemmDesc.skip_c0.offset == ctypes.sizeof(VdGemmDesc) - 6 * 4 - 16
    assert VdGemmDesc.row_sums.offset == ctypes.sizeof(VdGemmDesc) - 16   # producer-accumulated LayerNorm row sums (ABI 6)
    assert VdGemmDesc.stat_sums.offset == ctypes.sizeof(VdGemmDesc) - 8   # producer-accumulated GroupNorm sums (ABI 7)


def test_model_cfg_bank_resolves_four_flow():
    from lib.cfg_helper import model_cfg_bank
    cfg = model_cfg_bank()("vd_four_flow_v1-0")
    assert cfg.type == "vd_v2_0" and cfg.args.beta_linear_start == 0.00085 and cfg.args.beta_linear_end == 0.012
    assert cfg.args.timesteps == 1000 and cfg.args.global_layer_ptr == "image"
    assert cfg.args.latent_scale_factor["image"] == 0.18215
    unet = dict(cfg.args.diffuser_cfg_list)["image"]
    assert unet.type == "openai_unet_2d_next" and unet.args.model_channels == 320
    assert list(unet.args.channel_mult) == [1, 2, 4, 4] and list(unet.args.attention_resolutions) == [4, 2, 1]
    unet0 = dict(cfg.args.diffuser_cfg_list)["text"]
    assert unet0.type == "openai_unet_0d_next" and list(unet0.args.parts) == ["data", "context"]  # args merged, parts replaced
    assert unet0.args.input_channels == 768
    vae = dict(cfg.args.vae_cfg_list)["image"]
    assert vae.type == "autoencoderkl" and vae.pth == "pretrained/kl-f8.pth" and vae.args.ddconfig.ch == 128
    assert dict(cfg.args.ctx_cfg_list)["text"].type == "clip_text_context_encoder"
    assert dict(cfg.args.vae_cfg_list)["text"].type == "optimus_vae_next"


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs/model"), reason="reference checkout not present")
def test_reference_yaml_files_load_unchanged(monkeypatch):
    """The reference's own configs/model/*.yaml resolve to the same model definitions through our bank."""
    from lib.cfg_helper import model_cfg_bank
    mine = model_cfg_bank()("vd_four_flow_v1-0")
    monkeypatch.setenv("VD_CONFIG_DIR", "/root/reference/configs/model")
    ref = model_cfg_bank()("vd_four_flow_v1-0")
    for key in ("diffuser_cfg_list", "ctx_cfg_list"):
        for (n1, c1), (n2, c2) in zip(mine.args[key], ref.args[key]):
            assert n1 == n2 and c1.type == c2.type
            assert json.dumps(c1.args, sort_keys=True) == json.dumps(c2.args, sort_keys=True)
    v1, v2 = dict(mine.args.vae_cfg_list)["image"], dict(ref.args.vae_cfg_list)["image"]
    assert json.dumps(v1.args, sort_keys=True) == json.dumps(v2.args, sort_keys=True)
    for k in ("beta_linear_start", "beta_linear_end", "timesteps", "global_layer_ptr"):
        assert mine.args[k] == ref.args[k]


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs/model"), reason="reference checkout not present")
def test_every_model_name_resolves_like_the_reference_bank():
    """All 27 model names of the reference's configs/model/*.yaml, resolved by the REFERENCE's own model_cfg_bank from its
    own files (separate process, oracle/ref_cfg_dump.py) and by this package's bank from this package's YAML files:
    identical resolved trees (type, args after super_cfg merging, MODEL() expansion, pth / hfm sources)."""
    import subprocess
    import sys
    import yaml
    from lib.cfg_helper import model_cfg_bank
    names = []
    for fn in sorted(os.listdir("/root/reference/configs/model")):
        if fn.endswith(".yaml"):
            names += list(yaml.safe_load(open(os.path.join("/root/reference/configs/model", fn))))
    assert len(names) == 27
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_cfg_dump.py")] + names, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("{"):])

    def plain(o):
        if isinstance(o, dict):
            return {str(k): plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return o

    for name in names:
        mine = json.loads(json.dumps(plain(model_cfg_bank()(name)), sort_keys=True))
        assert mine == ref[name], name


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib/model_zoo"), reason="reference checkout not present")
def test_full_size_state_dict_layouts_equal_the_reference_modules():
    """FULL-size modules of the checkpoint (2-D UNet, 0-D UNet with data+context / context-only blocks, KL-f8 VAE, Optimus
    BERT encoder and GPT-2 decoder), built on the meta device by the REFERENCE's registry from its own configs (separate
    process, oracle/ref_state_dict_dump.py) and by this package: same state-dict keys, same shapes -- what loading
    vd-four-flow-v1-0[-fp16].pth / kl-f8.pth / optimus-vae.pth relies on."""
    import subprocess
    import sys
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    names = ["openai_unet_2d_v1", "openai_unet_0d_v1_dc", "openai_unet_0d_v1_c", "autokl_v1", "optimus_bert_encoder", "optimus_gpt2_decoder"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_state_dict_dump.py")] + names, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index('{"'):])
    for name in names:
        cfg = model_cfg_bank()(name)
        for k in ("pth", "ckpt", "hfm"):
            cfg.pop(k, None)
        with torch.device("meta"):
            net = get_model()(cfg, verbose=False)
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert mine == ref[name], (name, sorted(set(mine) ^ set(ref[name]))[:5])
    assert len(ref["openai_unet_2d_v1"]) == 686


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib/model_zoo"), reason="reference checkout not present")
def test_schedule_helpers_bit_exact_against_live_reference_over_a_grid():
    """make_beta_schedule / make_ddim_timesteps / make_ddim_sampling_parameters of this package against the REFERENCE's own
    functions (separate process, oracle/ref_schedules.py): every DDIM step count 1..120 plus 200 / 250 / 500 / 1000, both
    discretisations, eta in {0, 0.25, 1}, the reference fed its model's fp32 alphas_cumprod buffer as its sampler does.
    Betas, timesteps, alphas and alphas_prev bit-exact; sigmas (eta > 0 only) to 2e-5 relative: the reference evaluates
    that one formula on a mix of an fp32 torch tensor and float64 numpy arrays whose promotion even depends on the array
    length, this package evaluates it in float64; the same exception type where the reference raises."""
    import subprocess
    import sys
    from lib.model_zoo import diffusion_utils as du
    grid = {"steps": list(range(1, 121)) + [200, 250, 500, 1000], "etas": [0.0, 0.25, 1.0], "methods": ["uniform", "quad"],
            "schedules": [["linear", 1000, 0.00085, 0.012], ["linear", 1000, 1e-4, 2e-2], ["linear", 250, 0.0015, 0.0195],
                          ["cosine", 1000, 1e-4, 2e-2], ["sqrt_linear", 1000, 1e-4, 2e-2], ["sqrt", 1000, 1e-4, 2e-2]]}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_schedules.py")], input=json.dumps(grid),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index('{"'):])

    def same(mine, e, what):
        a = np.asarray(mine)
        assert str(a.dtype) == e["dtype"] and list(a.shape) == e["shape"] and a.tobytes().hex() == e["hex"], what

    for name, n, a, b in grid["schedules"]:
        key = "%s/%d/%r/%r" % (name, n, a, b)
        same(np.asarray(du.make_beta_schedule(name, n, linear_start=a, linear_end=b)), ref["betas"][key], key)
    betas = np.asarray(du.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012))
    alphacums = np.cumprod(1.0 - betas, axis=0).astype(np.float32)

    def dec(e):
        return np.frombuffer(bytes.fromhex(e["hex"]), dtype=e["dtype"]).reshape(e["shape"])

    for method in grid["methods"]:
        for s_ in grid["steps"]:
            e = ref["ddim"]["%s/%d" % (method, s_)]
            if "error" in e:
                with pytest.raises(Exception) as ei:
                    du.make_ddim_timesteps(method, s_, 1000, verbose=False)
                assert type(ei.value).__name__ == e["error"], (method, s_)
                continue
            ts = du.make_ddim_timesteps(method, s_, 1000, verbose=False)
            same(ts, e["timesteps"], (method, s_))
            for eta in grid["etas"]:
                ee = e["eta%r" % eta]
                if isinstance(ee, dict):
                    with pytest.raises(Exception) as ei:
                        du.make_ddim_sampling_parameters(alphacums, ts, eta, verbose=False)
                    assert type(ei.value).__name__ == ee["error"], (method, s_, eta)
                    continue
                sig, al, alp = du.make_ddim_sampling_parameters(alphacums, ts, eta, verbose=False)
                same(al, ee[1], (method, s_, eta, "alpha"))
                same(alp, ee[2], (method, s_, eta, "alpha_prev"))
                assert np.allclose(sig, dec(ee[0]), rtol=2e-5 if s_ > 1 else 2e-3, atol=0), (method, s_, eta, "sigma")


def test_state_dict_layout_matches_reference():
    """Keys and shapes equal those of the reference modules (recorded by oracle/gen_golden.py)."""
    from lib.model_zoo import get_model
    net = get_model()(tiny_vd_cfg(), verbose=False)
    g = load_gold("unet_tiny.npz")
    ref = {str(k): tuple(json.loads(str(s))) for k, s in zip(g["state_keys"], g["state_shapes"])}
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == ref
    assert net.diffuser["image"].layer_order == net.diffuser["text"].layer_order
    assert net.to("cpu") is None and net.device == "cpu"  # reference quirk: .to() returns None


def test_half_batch_fork_region_of_the_unet_walk():
    """vd._fork_region: the 16x16 / 8x8 levels and the middle block of openai_unet_2d_v1 (from the Downsample that enters the 16x16
    level up to the Upsample that leaves it), skip tensors balanced inside; nothing for the 0-D net or a threshold no level meets."""
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    from lib.model_zoo import vd
    from lib.model_zoo.openaimodel import Downsample, Upsample
    with torch.device("meta"):
        net = get_model()(model_cfg_bank()("openai_unet_2d_v1"), verbose=False)
    d_iter, c_iter, steps = iter(enumerate(net.data_blocks)), iter(net.context_blocks), []
    for lt in list(net.i_order) + list(net.m_order) + list(net.o_order):
        if lt == "d":
            di, blk = next(d_iter)
            steps.append(("d", di, blk))
        elif lt == "c":
            steps.append(("c", [next(c_iter)], [None], [1.0]))
        else:
            steps.append(("save",) if lt == "save_hidden_feature" else ("load",))
    a, b = vd._fork_region(steps, 64 * 64, 256)
    assert isinstance(steps[a][2][0], Downsample) and isinstance(steps[b][2][0], Upsample)
    downs = [i for i, st in enumerate(steps) if st[0] == "d" and isinstance(st[2][0], Downsample)]
    ups = [i for i, st in enumerate(steps) if st[0] == "d" and isinstance(st[2][0], Upsample)]
    assert a == downs[1] and b == ups[1]          # 64 -> 32 -> [16 -> 8 ... 8 -> 16] -> 32 -> 64
    inside = steps[a:b]
    assert sum(st[0] == "save" for st in inside) == sum(st[0] == "load" for st in inside) == 6
    assert sum(st[0] == "c" for st in inside) == 6   # 2 + 3 transformer blocks of the 16x16 level, the middle block's, none at 8x8
    a2, b2 = vd._fork_region(steps, 64 * 64, 1024)
    assert a2 == downs[0] and b2 == ups[2]
    assert vd._fork_region(steps, 64 * 64, 16) is None        # no level that small
    assert vd._fork_region(steps, 16 * 16, 256) is None       # a 16x16 latent never leaves the range again: no region


def test_full_unet_structure():
    """Block inventory of openai_unet_2d_v1 without allocating it."""
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    cfg = model_cfg_bank()("openai_unet_2d_v1")
    with torch.device("meta"):
        net = get_model()(cfg, verbose=False)
    assert len(net.data_blocks) == 30 and len(net.context_blocks) == 16
    assert sum(p.numel() for p in net.parameters()) == 859520964
    assert net.i_order.count("save_hidden_feature") == 12 and net.o_order.count("load_hidden_feature") == 12
    from oracle import vd_oracle as O
    plan = O.unet_plan()
    assert net.i_order == plan["i_order"] and net.m_order == plan["m_order"] and net.o_order == plan["o_order"]
    cfg0 = model_cfg_bank()("openai_unet_0d_v1_dc")
    with torch.device("meta"):
        net0 = get_model()(cfg0, verbose=False)
    assert net0.layer_order == net.layer_order
    assert sum(p.numel() for p in net0.context_blocks.parameters()) == 267239360
    assert sum(p.numel() for p in net0.parameters()) == 1706797888


def test_schedule_buffers_and_ddim_schedule_bit_exact():
    from lib.model_zoo import get_model
    from lib.model_zoo.ddim import DDIMSampler
    net = get_model()(tiny_vd_cfg(), verbose=False)
    g = load_gold("schedule.npz")
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.array_equal(getattr(net, k).numpy(), g[k]), k
    s = DDIMSampler(net)
    for steps in (50, 10, 5, 4):
        s.make_schedule(steps, ddim_eta=0.0, verbose=False)
        assert np.array_equal(s.ddim_timesteps, g["ddim%d_timesteps" % steps])
        assert np.array_equal(s.ddim_alphas, g["ddim%d_alphas" % steps])
        assert np.array_equal(s.ddim_alphas_prev, g["ddim%d_alphas_prev" % steps])
        assert np.array_equal(s.ddim_sigmas, g["ddim%d_sigmas" % steps])
        assert np.array_equal(s.ddim_sqrt_one_minus_alphas, g["ddim%d_sqrt_one_minus_alphas" % steps])
    s.make_schedule(10, ddim_eta=0.7, verbose=False)
    assert np.allclose(s.ddim_sigmas, g["ddim10_eta07_sigmas"], rtol=1e-6, atol=0)
    with pytest.raises(IndexError):
        s.make_schedule(3, verbose=False)


def test_clip_state_dict_is_hf_compatible():
    """Key layout of the CLIP sub-tree equals transformers.CLIPModel's (so ctx.*.model.* checkpoint tensors load)."""
    from transformers import CLIPConfig, CLIPModel
    from lib.model_zoo.clip import CLIPModelHIP
    m = meta()["clip"]
    tc, vc = m["text_config"], m["vision_config"]
    cfg = dict(text={k: tc[k] for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers",
                                        "num_attention_heads", "max_position_embeddings")},
               vision={k: vc[k] for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                          "image_size", "patch_size")}, projection_dim=m["projection_dim"])
    mine = {k: tuple(v.shape) for k, v in CLIPModelHIP(cfg).state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in CLIPModel(CLIPConfig(**m)).state_dict().items()}
    assert mine == ref


def test_product_path_fails_loudly_without_gpu():
    from lib.model_zoo import get_model
    from vd_hip import VdHipError
    net = get_model()(tiny_vd_cfg(), verbose=False)
    x = torch.zeros(1, 4, 16, 16)
    with pytest.raises((RuntimeError, VdHipError), match="GPU"):
        net.apply_model({"type": "image", "x": x}, torch.tensor([1]), {"type": "text", "c": torch.zeros(1, 77, 128)})
    with pytest.raises((RuntimeError, VdHipError), match="GPU"):
        net.vae_decode(torch.zeros(1, 4, 4, 4), which="image")


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "versatile-diffusion_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dp, fn)


def test_registry_weight_sources_and_model_args(tmp_path, monkeypatch):
    """get_model (reference common/get_model.py:19-87): a missing weight file raises unless the caller opts in, `pth` /
    `ckpt` load, `backbone` / `layer_units` arguments are resolved before construction."""
    import torch.nn as nn
    from lib.cfg_helper import CfgDict
    from lib.model_zoo import get_model
    from lib.model_zoo.common.get_model import get_unit, preprocess_model_args, register
    monkeypatch.delenv("VD_ALLOW_MISSING_WEIGHTS", raising=False)
    m = meta()
    cfg = CfgDict(type="autoencoderkl", args=m["vae"], pth=str(tmp_path / "nope.pth"))
    with pytest.raises(FileNotFoundError, match="nope.pth"):
        get_model()(cfg, verbose=False)
    cfg.allow_missing_weights = True
    net = get_model()(cfg, verbose=False)
    monkeypatch.setenv("VD_ALLOW_MISSING_WEIGHTS", "1")
    cfg.pop("allow_missing_weights")
    get_model()(cfg, verbose=False)
    # ckpt: {'state_dict': ...}
    path = str(tmp_path / "kl.ckpt")
    sd = {k: v + 1 for k, v in net.state_dict().items()}
    torch.save({"state_dict": sd}, path)
    net2 = get_model()(CfgDict(type="autoencoderkl", args=m["vae"], ckpt=path), verbose=False)
    assert all(torch.equal(v, sd[k]) for k, v in net2.state_dict().items())

    @register("_test_wrapper")
    class Wrapper(nn.Module):
        def __init__(self, backbone, layer_units):
            super().__init__()
            self.backbone = backbone
            self.act = layer_units[0]()

    w = get_model()(CfgDict(type="_test_wrapper", args=CfgDict(
        backbone=CfgDict(type="autoencoderkl", args=m["vae"]), layer_units=["lrelu(negative_slope=0.2)", "none"])), verbose=False)
    assert type(w.backbone).__name__ == "AutoencoderKL" and w.act.negative_slope == 0.2
    assert get_unit()("conv(kernel_size=(3,3), padding=1)")(4, 8).kernel_size == (3, 3)
    assert preprocess_model_args({"a": 1}) == {"a": 1}


# ---- round 4 -------------------------------------------------------------------------------------------------------

def test_bench_never_reports_one_gpu_for_a_multi_gpu_request():
    """`python bench.py --gpus 8` outside torchrun must launch 8 ranks or fail -- never print an n_gpus: 1 line (round-3
    review).  On this GPU-less host: non-zero exit, no JSON on stdout, for both the bare and the WORLD_SIZE-mismatch form."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "8" in r.stderr and "GPU" in r.stderr
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=2" in r.stderr


def test_draw_initial_latent_device_generator_is_the_reference_draw():
    """device_generator=True: torch.manual_seed(seed + 100) then torch.randn(shape, device, dtype) -- the draw of the
    reference's unsharded call (app.py:309 -> ddim.py:105); the default host-generator draw differs and is seed-stable."""
    import torch
    from lib.model_zoo import sharded
    shape = (3, 4, 8, 8)
    a = sharded.draw_initial_latent(shape, 7, dtype=torch.float32, device_generator=True, device="cpu")
    torch.manual_seed(107)
    assert torch.equal(a, torch.randn(shape))
    b = sharded.draw_initial_latent(shape, 7)
    assert torch.equal(b, sharded.draw_initial_latent(shape, 7)) and b.shape == a.shape


def test_graft_entry_build_runs():
    """__graft_entry__.build() is what the driver runs on the CPU every round: library built (or re-used when the sources are
    unchanged), every declared symbol exported, ABI version as the header says, host package importable."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=root, capture_output=True, text=True,
                       timeout=3000)
    assert r.returncode == 0 and "build ok" in r.stdout, r.stderr[-2000:]


def test_sampler_static_state_is_keyed_by_geometry_flow_and_weight_versions():
    """DDIMSampler._static_state (the buffers a kept step graph reads): one entry per (weights, latent shape, flow, contexts),
    at most two kept, most recently used last; an in-place weight update or a moved parameter gets a NEW entry (the kept
    graph would otherwise replay against weight packs of the old values)."""
    import torch
    from lib.model_zoo.ddim import DDIMSampler

    class Net(torch.nn.Module):
        num_timesteps = 1000

        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(4, 4)

    net = Net()
    s = DDIMSampler(net)
    x = torch.zeros(2, 4, 8, 8, dtype=torch.float16)
    ctx = lambda L: [{"type": "text", "c": torch.zeros(4, L, 768), "ratio": 1.0}]
    a = s._static_state(x, {"type": "image"}, ctx(77), True, True)
    assert a["xs"].shape == x.shape and a["ts"].shape == (4,) and a["c"][0].shape == (4, 77, 768) and a["graph"] is None
    assert s._static_state(x, {"type": "image"}, ctx(77), True, True) is a
    b = s._static_state(x, {"type": "image"}, ctx(257), True, True)          # another context length
    assert b is not a and len(s._static) == 2
    assert s._static_state(x, {"type": "image"}, ctx(77), True, True) is a    # a is now the most recently used
    c = s._static_state(x[:1], {"type": "image"}, ctx(77)[:1], False, True)   # third geometry: the oldest (b) goes
    assert c["ts"].shape == (1,) and len(s._static) == 2
    assert s._static_state(x, {"type": "image"}, ctx(77), True, True) is a
    assert s._static_state(x, {"type": "image"}, ctx(257), True, True) is not b
    with torch.no_grad():
        net.lin.weight.add_(1.0)                                              # in-place update: version bump
    assert s._static_state(x, {"type": "image"}, ctx(77), True, True) is not a
    s.release_graphs()
    assert len(s._static) == 0
    s.graph_cache = False
    assert s._static_state(x, {"type": "image"}, ctx(77), True, True) is None


def test_coef_table_rows_are_the_ddim_update_of_the_reference():
    """DDIMSampler._coef_table feeds cfg_ddim_dev_kernel: {scale, 1/sqrt(a_t), sqrt(a_prev), sqrt(1 - a_prev - sigma^2), sigma,
    sqrt(1 - a_t)} per DDIM index -- the coefficients of p_sample_ddim (reference ddim.py:150-171), from the bit-exact schedule."""
    import numpy as np
    import torch
    from lib.model_zoo.ddim import DDIMSampler
    from lib.model_zoo.diffusion_utils import make_beta_schedule

    class Net(object):
        num_timesteps = 1000

        def host_schedule(self, name):
            betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
            return np.cumprod(1.0 - np.asarray(betas, dtype=np.float64), axis=0)

    for steps, eta in ((50, 0.0), (20, 0.7)):
        s = DDIMSampler(Net())
        s.make_schedule(steps, ddim_eta=eta, verbose=False)
        n = s.ddim_timesteps.shape[0]
        tab = s._coef_table(n, 7.5, torch.device("cpu")).numpy()
        assert tab.shape == (n, 6) and tab.dtype == np.float32
        a_t, a_prev, sig = s.ddim_alphas.astype(np.float64), s.ddim_alphas_prev.astype(np.float64), s.ddim_sigmas.astype(np.float64)
        x, e = 0.3, -1.1   # one element through the update, reference order of operations in fp64
        for i in range(n):
            pred_x0 = (x - float(s.ddim_sqrt_one_minus_alphas[i]) * e) / np.sqrt(a_t[i])
            x_prev = np.sqrt(a_prev[i]) * pred_x0 + np.sqrt(max(1.0 - a_prev[i] - sig[i] ** 2, 0.0)) * e
            sc, r_at, s_ap, dirc, sg, s1m = [float(v) for v in tab[i]]
            assert sc == 7.5 and sg == np.float32(sig[i])
            got = s_ap * ((x - s1m * e) * r_at) + dirc * e
            assert abs(got - x_prev) < 2e-6 * max(1.0, abs(x_prev))
        if eta == 0.0:
            assert not sig.any()

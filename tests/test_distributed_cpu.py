"""N > 1 path on CPU: world-size-2 gloo processes run the batch-sharding helper; the compute engine in this test is
the CPU oracle on the tiny fixture model (tests may use the oracle; the product path itself needs the GPU)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, out_dir):
    for p in (ROOT, os.path.join(ROOT, "versatile-diffusion_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VD_QUIET="1")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from lib.model_zoo import sharded
        from oracle import synth, vd_oracle as O
        from vdtest_util import load_gold, meta
        m = meta()
        g = load_gold("unet_tiny.npz")
        shapes = {str(k): tuple(json.loads(str(s))) for k, s in zip(g["state_keys"], g["state_shapes"])}
        sd = synth.synth_state_dict(shapes, m["seed"])
        sd.update(O.register_schedule())
        plan = O.unet_plan(**m["unet2d"])
        gen = torch.Generator().manual_seed(5)
        c = torch.randn((B, 77, 128), generator=gen) * 0.5
        u = torch.randn((1, 77, 128), generator=gen).repeat(B, 1, 1) * 0.5
        ctx = [{"type": "text", "conditioning": c, "unconditional_conditioning": u}]

        def sample_fn(x_T, ctxs):
            with torch.no_grad():
                return O.ddim_sample(sd, plan, sd["alphas_cumprod"], x_T, ctxs, 4, 7.5, global_ptr="image")[0]

        def decode_fn(z):
            with torch.no_grad():
                return O.vae_decode(sd, "vae.image", z / 0.18215, ch_mult=m["vae"]["ddconfig"]["ch_mult"],
                                    num_res_blocks=m["vae"]["ddconfig"]["num_res_blocks"])

        imgs = sharded.sample_sharded(sample_fn, decode_fn, [B, 4, 8, 8], ctx, seed=23, device="cpu")
        torch.save(imgs, os.path.join(out_dir, "rank%d.pt" % rank))
        if rank == 0:  # single-process result for comparison
            x_T = sharded.draw_initial_latent([B, 4, 8, 8], 23)
            torch.save(decode_fn(sample_fn(x_T, ctx)), os.path.join(out_dir, "single.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 3])
def test_batch_sharding_world2_gloo(tmp_path, B):
    port = 29500 + (os.getpid() % 500) + B
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    single = torch.load(tmp_path / "single.pt")
    assert r0.shape == (B, 3, 16, 16)
    assert torch.equal(r0, r1), "all_gather must leave the same full batch on every rank"
    # sharded == unsharded: samples are independent and the latent is drawn once and sliced
    assert torch.allclose(r0, single, atol=1e-5)


def _worker_fidelity(rank, world, port, B, out_dir):
    """vd_sample_sharded(images=..., fidelity > 0) with the product's sharding / noise-slicing logic and the oracle standing in
    for the GPU model: `net` / `sampler` stubs with the product objects' call signatures."""
    for p in (ROOT, os.path.join(ROOT, "versatile-diffusion_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VD_QUIET="1")
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from lib.model_zoo import sharded
        from oracle import synth, vd_oracle as O
        from vdtest_util import load_gold, meta
        m = meta()
        g = load_gold("unet_tiny.npz")
        shapes = {str(k): tuple(json.loads(str(s))) for k, s in zip(g["state_keys"], g["state_shapes"])}
        sd = synth.synth_state_dict(shapes, m["seed"])
        sd.update(O.register_schedule())
        plan = O.unet_plan(**m["unet2d"])
        vkw = dict(ch_mult=m["vae"]["ddconfig"]["ch_mult"], num_res_blocks=m["vae"]["ddconfig"]["num_res_blocks"])

        class Net(object):
            device = "cpu"

            def vae_encode(self, x, which, noise=None):
                assert noise is not None, "the sharded path must inject the posterior noise"
                with torch.no_grad():
                    return O.diag_gaussian_sample(O.vae_encode_moments(sd, "vae." + which, x, **vkw), noise) * 0.18215

            def vae_decode(self, z, which):
                with torch.no_grad():
                    return O.vae_decode(sd, "vae." + which, z / 0.18215, **vkw)

        class Sampler(object):
            def sample(self, steps, shape, x_info, c_info, eta=0., verbose=False):
                assert x_info.get("x0_noise") is not None, "the sharded path must inject the forward-process noise"
                k = x_info["x0_forward_timesteps"]
                sched = O.ddim_schedule(sd["alphas_cumprod"], steps, eta)
                ts = torch.full((shape[0],), int(sched["timesteps"][k]), dtype=torch.long)
                with torch.no_grad():
                    xk = O.q_sample(sd, x_info["x0"], ts, x_info["x0_noise"])
                    return O.ddim_sample(sd, plan, sd["alphas_cumprod"], xk, [c_info], steps, c_info["unconditional_guidance_scale"],
                                         global_ptr="image", forward_steps=k)

        gen = torch.Generator().manual_seed(9)
        imgs = torch.rand((B, 3, 16, 16), generator=gen)
        c = torch.randn((B, 77, 128), generator=gen) * 0.5
        ctx = [{"type": "text", "conditioning": c, "unconditional_conditioning": torch.zeros_like(c)}]
        lo, hi = sharded.shard_bounds(B, world, rank)
        out = sharded.vd_sample_sharded(Net(), Sampler(), 6, [B, 4, 8, 8], ctx, seed=31, images=imgs[lo:hi], fidelity=0.5)
        torch.save(out, os.path.join(out_dir, "fid_w%d_r%d.pt" % (world, rank)))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_image_variation_with_fidelity_is_world_size_invariant(tmp_path):
    """the fidelity branch used to draw q_sample / posterior noise from each rank's own generator.  Now both are
    slices of seeded full-batch draws: world size 1 and 2 give the same images, and the two halves of a batch differ."""
    B = 4
    _worker_fidelity(0, 1, 0, B, str(tmp_path))
    port = 29500 + (os.getpid() % 500) + 11
    mp.spawn(_worker_fidelity, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / "fid_w1_r0.pt")
    r0, r1 = torch.load(tmp_path / "fid_w2_r0.pt"), torch.load(tmp_path / "fid_w2_r1.pt")
    assert one.shape == (B, 3, 16, 16) and torch.equal(r0, r1)
    assert torch.allclose(r0, one, atol=1e-5)
    assert not torch.allclose(one[:2], one[2:], atol=1e-3)


def test_batch_smaller_than_world_raises_on_every_rank(tmp_path):
    """B < world size is detected from the arguments on every rank (no rank is left hanging in the all_gather)."""
    port = 29500 + (os.getpid() % 500) + 7
    with pytest.raises(Exception, match="batch 1 < world size 2"):
        mp.spawn(_worker, args=(2, port, 1, str(tmp_path)), nprocs=2, join=True)


def test_shard_bounds():
    sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
    from lib.model_zoo.sharded import draw_initial_latent, shard_bounds
    for total in (1, 7, 8, 16, 33):
        for world in (1, 2, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    a, b = draw_initial_latent([4, 4, 8, 8], 23), draw_initial_latent([4, 4, 8, 8], 23)
    assert torch.equal(a, b)

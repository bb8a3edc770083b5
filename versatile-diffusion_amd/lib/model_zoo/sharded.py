"""Batch-axis sharding of the sampling path across the GPUs of a node: one process per GPU (torchrun), full weight
replica per rank, NO collective inside the DDIM loop, one all_gather of the decoded images at the end (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests).

The reference has no inference-time multi-GPU path (app.py:279-283 uses one device); samples are independent
(GroupNorm / LayerNorm / attention are per sample), so the batch axis shards with the CFG pair of a sample kept on one
rank.  The full initial latent is drawn ONCE with the reference's seed rule (`torch.manual_seed(seed + 100)`,
app.py:309) and sliced, never re-seeded per rank, so 1-GPU and N-GPU runs produce the same images.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous, balanced [lo, hi) slice of `total` samples for `rank` (earlier ranks take the remainder)."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def draw_initial_latent(shape, seed, dtype=torch.float32, device_generator=False, device=None):
    """Full-batch x_T with the reference's seed convention (`seed + 100`, app.py:309).

    Default: from a HOST generator -- identical on every rank and for every world size, but NOT the numbers the
    reference's unsharded call draws (it seeds the global generator and draws `torch.randn(shape, device=cuda,
    dtype=fp16)` on the device, ddim.py:105).  device_generator=True reproduces exactly that draw (global
    `torch.manual_seed(seed + 100)`, then randn on `device` in `dtype`): a single-rank sharded run then starts from the
    same x_T as `DDIMSampler.sample(x_info={'type': 'image'})` after the app's seeding.  Device RNG streams depend on
    the tensor's shape, so this mode is only offered for the whole batch on one rank (world size 1)."""
    if device_generator:
        torch.manual_seed(int(seed) + 100)
        return torch.randn(tuple(shape), device=device, dtype=dtype)
    g = torch.Generator(device="cpu").manual_seed(int(seed) + 100)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32).to(dtype)


def _slice_ctx(c_info, lo, hi):
    out = dict(c_info)
    for k in ("conditioning", "unconditional_conditioning"):
        out[k] = c_info[k][lo:hi]
    return out


def sample_sharded(sample_fn, decode_fn, shape, c_info_list, seed, device, group=None, gather=True, device_generator=False):
    """Run `sample_fn(x_T_local, c_info_list_local) -> latents` and `decode_fn(latents) -> images` on this rank's slice
    of the batch and all_gather the images.

    shape: full-batch latent shape [B, C, h, w]; c_info_list: reference-style context dicts holding the FULL batch.
    Returns [B, 3, H, W] images on every rank (or the local slice when gather=False)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = shape[0]
    if B < world:  # deterministic from the arguments: every rank raises, nobody is left waiting in the all_gather
        raise ValueError("sample_sharded: batch %d < world size %d (every rank needs at least one sample)" % (B, world))
    if torch.device(device).type == "cuda":
        torch.cuda.set_device(device)  # kernels and the RCCL communicator of this rank live on its own GPU
    lo, hi = shard_bounds(B, world, rank)
    if device_generator:
        if world != 1:
            raise ValueError("sample_sharded: device_generator=True reproduces the unsharded sampler's x_T and needs world size 1")
        x_T = draw_initial_latent(shape, seed, dtype=torch.float16, device_generator=True, device=device)
    else:
        x_T = draw_initial_latent(shape, seed)[lo:hi].to(device)
    local_ctx = [_slice_ctx(ci, lo, hi) for ci in c_info_list]
    images = decode_fn(sample_fn(x_T, local_ctx))
    if world == 1 or not gather:
        return images
    # ragged slices: pad to the largest slice so a single fixed-size all_gather does it
    max_n = shard_bounds(B, world, 0)[1]
    pad = images
    if images.shape[0] < max_n:
        pad = torch.cat([images, images.new_zeros((max_n - images.shape[0],) + tuple(images.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    parts = []
    for r, b in enumerate(bufs):
        rlo, rhi = shard_bounds(B, world, r)
        parts.append(b[: rhi - rlo])
    return torch.cat(parts)


def vd_sample_sharded(net, sampler, steps, shape, c_info_list, seed, guidance_scale=7.5, eta=0., group=None,
                      images=None, fidelity=0., device_generator=False):
    """t2i / image-variation / multi-context sampling + kl-f8 decode of a full batch, sharded over the process group.

    images + fidelity > 0: image variation with fidelity (reference app.py:355-371) -- `images` is THIS RANK's slice of
    the input images [n_local, 3, H, W] in [0, 1]; x0 = vae_encode(images) and only the first int(steps * (1 - fidelity))
    DDIM steps run (ddim.py:97-103).  Posterior and forward-process noise are slices of seeded full-batch draws (see
    sample_fn), so the result does not depend on the world size."""
    def sample_fn(x_T, ctxs):
        for ci in ctxs:
            ci["unconditional_guidance_scale"] = guidance_scale
        lshape = [x_T.shape[0]] + list(shape[1:])
        if images is not None and fidelity > 0.:
            # every random number of this branch comes from a seeded FULL-batch draw sliced to this rank's samples, like
            # x_T: the forward-process noise of q_sample is the rank's slice of the x_T draw itself (the `x0_noise`
            # extension of DDIMSampler), the VAE posterior noise a second draw (seed + 1) -- never the rank's own device
            # generator, so a 1-GPU and an N-GPU run give the same images (and two ranks never repeat each other's noise)
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            lo, hi = shard_bounds(shape[0], world, rank)
            post = draw_initial_latent(shape, int(seed) + 1)[lo:hi].to(x_T.device)
            x0 = net.vae_encode(images, which="image", noise=post)
            x_info = {"type": "image", "x0": x0, "x0_forward_timesteps": int(steps * (1 - fidelity)), "x0_noise": x_T}
        else:
            x_info = {"type": "image", "xt": x_T}
        if len(ctxs) == 1:
            z, _ = sampler.sample(steps=steps, shape=lshape, x_info=x_info, c_info=ctxs[0], eta=eta, verbose=False)
        else:
            z, _ = sampler.sample_multicontext(steps=steps, shape=lshape, x_info=x_info, c_info_list=ctxs, eta=eta,
                                               verbose=False)
        return z

    return sample_sharded(sample_fn, lambda z: net.vae_decode(z, which="image"), shape, c_info_list, seed, net.device,
                          group=group, device_generator=device_generator)

"""DDIM sampler with the reference's API (lib/model_zoo/ddim.py:10-298 there): `DDIMSampler(model).sample(...)`
and `.sample_multicontext(...)` with the same arguments, dict protocol and return values.

What changed underneath: the CFG combine and the DDIM update are ONE elementwise kernel (the reference issues ~10
tiny kernels and three device->host syncs per step), the CFG-doubled context batch is assembled once instead of every
step, the step-invariant context K/V projections of all 16 cross-attention layers are computed once per sample() call,
and -- because every launch goes to torch's current stream with static shapes -- the whole step (UNet forward + update,
~450 kernel launches) is captured once into a HIP graph and replayed for the remaining steps; between replays the host
only refreshes a 6-float coefficient vector and the timestep tensor.  Set VD_DDIM_GRAPH=0 to run every step eagerly.
"""
import os
import threading

import numpy as np
import torch

from vd_hip import ops

from .diffusion_utils import make_ddim_sampling_parameters, make_ddim_timesteps


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = os.environ.get("VD_DDIM_GRAPH", "1") != "0"
        self.graph_cache = os.environ.get("VD_DDIM_GRAPH_CACHE", "1") != "0"   # keep captured steps across sample() calls
        self.replay_first = os.environ.get("VD_DDIM_REPLAY_FIRST", "1") != "0"   # with a kept graph step 0 is replayed too
        # the t-only part of the UNet (time-embedding MLP + every ResBlock's emb_layers projection) for all steps at once,
        # outside the step graph (VD_v2_0.precompute_step_emb); 0 = recompute it inside every step like the reference
        self.emb_hoist = os.environ.get("VD_EMB_HOIST", "1") != "0"
        self._static = {}
        # one request at a time per sampler: the kept step graphs read and write static buffers (the reference's sampler is
        # not re-entrant either, but it has no captured state to corrupt; app.py runs Gradio workers unlocked)
        self._lock = threading.RLock()

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def release_graphs(self):
        """Drop the kept step graphs with their static latent / context / K-V buffers and private memory pools (up to two
        geometries are kept alive per sampler; at 768x768, batch 32 that is a sizeable HBM reservation).  The next
        sample() call captures again.  Serialised with running sample() calls by the sampler's lock."""
        with self._lock:
            self._static.clear()

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize,
                                                  num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        if hasattr(self.model, "host_schedule"):
            alphas_cumprod = self.model.host_schedule("alphas_cumprod")
        else:
            alphas_cumprod = self.model.alphas_cumprod.detach().float().cpu().numpy()
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        self.alphas_cumprod = alphas_cumprod
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(
            alphacums=alphas_cumprod, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        self.ddim_sigmas = np.asarray(sigmas, dtype=np.float32)
        self.ddim_alphas = np.asarray(alphas, dtype=np.float32)
        self.ddim_alphas_prev = np.asarray(alphas_prev, dtype=np.float64)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(np.float32(1.) - self.ddim_alphas)

    # ---- single context ---------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, steps, shape, x_info, c_info, eta=0., temperature=1., noise_dropout=0., verbose=True,
               log_every_t=100):
        with self._lock:   # the schedule attributes and the kept step graphs are per-sampler state
            self.make_schedule(ddim_num_steps=steps, ddim_eta=eta, verbose=verbose)
            if verbose:
                print("Data shape for DDIM sampling is {}, eta {}".format(shape, eta))
            return self.ddim_sampling_multicontext(shape, x_info, [c_info], noise_dropout=noise_dropout,
                                                   temperature=temperature, log_every_t=log_every_t, _single=True)

    @torch.no_grad()
    def ddim_sampling(self, shape, x_info, c_info, noise_dropout=0., temperature=1., log_every_t=100):
        return self.ddim_sampling_multicontext(shape, x_info, [c_info], noise_dropout=noise_dropout,
                                               temperature=temperature, log_every_t=log_every_t, _single=True)

    # ---- multi context ----------------------------------------------------------------------------
    @torch.no_grad()
    def sample_multicontext(self, steps, shape, x_info, c_info_list, eta=0., temperature=1., noise_dropout=0.,
                            verbose=True, log_every_t=100):
        with self._lock:
            self.make_schedule(ddim_num_steps=steps, ddim_eta=eta, verbose=verbose)
            if verbose:
                print("Data shape for DDIM sampling is {}, eta {}".format(shape, eta))
            return self.ddim_sampling_multicontext(shape, x_info, c_info_list, noise_dropout=noise_dropout,
                                                   temperature=temperature, log_every_t=log_every_t)

    @torch.no_grad()
    def ddim_sampling_multicontext(self, shape, x_info, c_info_list, noise_dropout=0., temperature=1.,
                                   log_every_t=100, _single=False):
        with self._lock:
            return self._ddim_sampling_multicontext(shape, x_info, c_info_list, noise_dropout, temperature, log_every_t, _single)

    def _ddim_sampling_multicontext(self, shape, x_info, c_info_list, noise_dropout, temperature, log_every_t, _single):
        device = self.model.device
        dtype = c_info_list[0]["conditioning"].dtype
        bs = shape[0]
        timesteps = self.ddim_timesteps
        if ("xt" in x_info) and (x_info["xt"] is not None):
            x_info["x"] = x_info["xt"].to(device=device, dtype=dtype)
        elif ("x0" in x_info) and (x_info["x0"] is not None):
            x0 = x_info["x0"].to(device=device, dtype=dtype)
            k = x_info["x0_forward_timesteps"]
            ts = torch.full((bs,), int(timesteps[k]), device=device, dtype=torch.long)
            timesteps = timesteps[:k]
            # `x0_noise` (extension): inject the forward-process noise instead of drawing it, for reproducible runs
            x_info["x"] = self.model.q_sample(x0, ts, noise=x_info.get("x0_noise"))
        else:
            x_info["x"] = torch.randn(shape, device=device, dtype=dtype)

        scale = c_info_list[0]["unconditional_guidance_scale"]
        for ci in c_info_list:
            assert ci["unconditional_guidance_scale"] == scale, \
                "A different unconditional guidance scale between different context is not allowed!"
        guided = scale != 1.
        # the loop works on shallow copies: 'c' (the CFG batch, possibly a static buffer the captured graph reads) and
        # 'kv_cache' never appear in the caller's dicts
        c_info_list = [dict(ci) for ci in c_info_list]
        # CFG batch [uncond ; cond] assembled ONCE; K/V projections of it cached for the whole loop
        for ci in c_info_list:
            if guided:
                ci["c"] = torch.cat([ci["unconditional_conditioning"], ci["conditioning"]]).to(device)
            else:
                ci["c"] = ci["conditioning"].to(device)
            ci["kv_cache"] = {}

        intermediates = {"pred_xt": [], "pred_x0": []}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        x = x_info["x"].to(torch.float16).contiguous()
        eta_zero = bool(np.all(self.ddim_sigmas[:total_steps] == 0.))
        # noise_dropout > 0 (reference ddim.py:167-169 / :294-296: F.dropout on the step noise) draws a mask from the device
        # generator on every step, between the noise draws: that order only exists in the eager loop
        if x.is_cuda and eta_zero and total_steps > 0 and not noise_dropout > 0.:
            x, pred_x0 = self._loop_static(x, x_info, c_info_list, time_range, total_steps, guided, scale, _single,
                                           log_every_t, intermediates, dtype)
        else:
            pred_x0 = None
            for i, step in enumerate(time_range):
                index = total_steps - i - 1
                x, pred_x0 = self._step(x, x_info, c_info_list, int(step), index, guided, scale, temperature, _single,
                                        noise_dropout=noise_dropout)
                if index % log_every_t == 0 or index == total_steps - 1:
                    intermediates["pred_xt"].append(x.to(dtype))
                    intermediates["pred_x0"].append(pred_x0.to(dtype))
        x_info["x"] = x.to(dtype)
        return x_info["x"], intermediates

    def _coef_table(self, total_steps, scale, device):
        """[S, 6] fp32 device table of {scale, 1/sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma, sqrt(1-a_t)}."""
        a_t = self.ddim_alphas[:total_steps].astype(np.float64)
        a_prev = self.ddim_alphas_prev[:total_steps].astype(np.float64)
        sig = self.ddim_sigmas[:total_steps].astype(np.float64)
        tab = np.stack([np.full_like(a_t, float(scale)), 1.0 / np.sqrt(a_t), np.sqrt(a_prev),
                        np.sqrt(np.maximum(1.0 - a_prev - sig ** 2, 0.0)), sig,
                        self.ddim_sqrt_one_minus_alphas[:total_steps].astype(np.float64)], axis=1)
        return torch.from_numpy(tab.astype(np.float32)).to(device)

    def _static_state(self, x, x_info, c_info_list, guided, single):
        """Buffers the captured step reads and writes, kept ACROSS sample() calls per (model weights, shapes, flow): a
        second call with the same geometry re-uses the instantiated HIP graph instead of capturing again (capture =
        one host-bound pass over ~400 launches with the GPU idle + instantiation: 10-15 ms per batch of 680).  Everything
        the graph dereferences lives here: latent buffers, step scalars, the CFG context batches and their K/V
        projections (refreshed in place by the eager first step of every call)."""
        if not self.graph_cache:
            return None
        # in-place updates / load_state_dict bump a parameter's version, .half() / .to() move its storage: the key hashes the
        # ORDERED (storage, version) pairs of every parameter and buffer (an additive checksum would let two parameters
        # that swap storages collide)
        wv = hash(tuple((t.data_ptr(), t._version) for t in list(self.model.parameters()) + list(self.model.buffers())))
        # (emb_hoist is part of the key: a step graph captured with the hoisted time embedding reads st["embrow"], one captured
        # without it computes the embedding inside the step -- replaying either under the other setting would be silently wrong)
        key = (id(self.model), wv, str(x.device), tuple(x.shape), x_info["type"], bool(guided), bool(single), bool(self.emb_hoist),
               tuple((ci["type"], tuple(ci["c"].shape), float(ci.get("ratio", 1.0))) for ci in c_info_list))
        st = self._static.get(key)
        if st is None:
            while len(self._static) >= 2:                      # shapes seen long ago: let their graphs go
                self._static.pop(next(iter(self._static)))
            nb = (2 if guided else 1) * x.shape[0]
            st = {"xs": torch.empty_like(x), "x_next": torch.empty_like(x), "p0": torch.empty_like(x),
                  "ts": torch.empty((nb,), device=x.device, dtype=torch.long),
                  "coef": torch.empty((6,), device=x.device, dtype=torch.float32),
                  "c": [torch.empty(ci["c"].shape, device=x.device, dtype=torch.float16) for ci in c_info_list],
                  "kv": [dict() for _ in c_info_list], "graph": None}
            self._static[key] = st
        else:
            self._static[key] = self._static.pop(key)          # most recently used last
        return st

    def _loop_static(self, x, x_info, c_info_list, time_range, total_steps, guided, scale, single, log_every_t,
                     intermediates, dtype):
        """eta = 0 loop on static buffers: step 0 runs eagerly (fills weight-pack and K/V caches), is then captured
        into a HIP graph, and the graph is replayed for the remaining steps -- and, through _static_state, by later
        sample() calls of the same geometry."""
        dev = x.device
        b = x.shape[0]
        nb = 2 * b if guided else b
        st = self._static_state(x, x_info, c_info_list, guided, single)
        if st is None:
            xs = x.clone()
            x_next, p0 = torch.empty_like(xs), torch.empty_like(xs)
            ts = torch.empty((nb,), device=dev, dtype=torch.long)
            coef = torch.empty((6,), device=dev, dtype=torch.float32)
            graph = None
            replay_first = False
        else:
            xs, x_next, p0, ts, coef, graph = st["xs"], st["x_next"], st["p0"], st["ts"], st["coef"], st["graph"]
            replay_first = False
            xs.copy_(x)
            for ci, cbuf, kv in zip(c_info_list, st["c"], st["kv"]):
                cbuf.copy_(ci["c"])
                ci["c"] = cbuf
                kv["_stale"] = set(k for k in kv if not (isinstance(k, str) and k.startswith("_")))   # K/V of the previous call's context: recomputed in place
                ci["kv_cache"] = kv
            if graph is not None and self.use_graph and self.replay_first and all("_modules" in kv for kv in st["kv"]):
                # a kept graph: the context K/V projections of this call are refreshed in place right here (16 small GEMMs per
                # context), so step 0 is replayed like every other step instead of running its ~370 launches eagerly
                for ci, kv in zip(c_info_list, st["kv"]):
                    cc = self.model._prep(ci["c"])
                    for mid in list(kv["_stale"]):
                        kv[mid].copy_(kv["_modules"][mid][0].project_context(cc))
                    kv["_stale"].clear()
                replay_first = True
        table = self._coef_table(total_steps, scale, dev)
        steps_dev = torch.from_numpy(np.ascontiguousarray(time_range).astype(np.int64)).to(dev)
        # every sample of the batch is at the same timestep in every step, and all steps are known now: the t-only part of the
        # forward (reference vd.py:339-349 -> openaimodel.py:2627-2633, :263) is computed here for all of them (M = steps
        # instead of `steps` times M = batch) and the step reads row i from a static buffer
        emb_tab = emb_rows = embrow = None
        if self.emb_hoist and hasattr(self.model, "precompute_step_emb"):
            pre = self.model.precompute_step_emb(x_info["type"], steps_dev, multicontext=not single)
            if pre is not None:
                emb_tab, layout = pre
                embrow = st.get("embrow") if st is not None else None
                if embrow is None or embrow.numel() != emb_tab.shape[1]:
                    assert graph is None, "the kept step graph reads another time-embedding buffer"
                    embrow = torch.empty((emb_tab.shape[1],), device=dev, dtype=torch.float16)
                    if st is not None:
                        st["embrow"] = embrow
                emb_rows = {di: embrow[o:o + c] for di, (o, c) in layout.items()}

        def body():
            # guided: the UNet batch is [xs; xs] (ddim.py:144-149).  It is handed over as (xs, repeat=2) so the data blocks in
            # front of the first context block run once (extension key of this package's apply_model*)
            xi = {"type": x_info["type"], "x": xs, "repeat": 2 if guided else 1}
            if emb_rows is not None:
                xi["emb_rows"] = emb_rows
            if single:
                eps = self.model.apply_model(xi, ts, c_info_list[0])
            else:
                eps = self.model.apply_model_multicontext(xi, ts, c_info_list)
            ops.cfg_ddim_step_dev(xs, eps.contiguous(), coef, guided=guided, x_prev=x_next, pred_x0=p0)
            xs.copy_(x_next)

        # RNG contract: the reference draws noise_like(x) = torch.randn_like(x) on every step even when sigma = 0
        # (ddim.py:167 / :294 there), so the device generator ends a sample() call advanced by one latent-sized draw per
        # step.  The draws are consumed here, up front, and the generator state is pinned to that point after the loop
        # (graph capture / replay bookkeeping must not leak into it), so code that keeps drawing from the default
        # generator after sample() sees the reference's stream.
        for _ in range(total_steps):
            torch.randn_like(xs)
        rng_after = torch.cuda.get_rng_state(dev)
        for i in range(total_steps):
            index = total_steps - i - 1
            ts.copy_(steps_dev[i].expand(nb))       # device-side refresh, no host sync
            coef.copy_(table[index])
            if embrow is not None:
                embrow.copy_(emb_tab[i])
            if graph is not None and st is not None and st.get("graph_embrow", embrow is not None) != (embrow is not None):
                # the kept graph was captured with / without the hoisted embedding row and this call has it the other way round
                # (precompute_step_emb started / stopped returning a table): capture again instead of replaying stale conditioning
                graph = st["graph"] = None
            if (i == 0 and not replay_first) or not self.use_graph:
                body()
            elif graph is None:
                graph = self._capture(body)
                if graph is None:
                    body()
                else:
                    if st is not None:
                        st["graph"] = graph
                        st["graph_embrow"] = embrow is not None
                    graph.replay()
            else:
                graph.replay()
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["pred_xt"].append(xs.to(dtype).clone())
                intermediates["pred_x0"].append(p0.to(dtype).clone())
        torch.cuda.set_rng_state(rng_after, dev)
        return xs.clone() if st is not None else xs, p0.clone() if st is not None else p0

    def _capture(self, body):
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    body()
            torch.cuda.current_stream().wait_stream(s)
            ops.drop_workspaces(s.cuda_stream)
            return g
        except Exception as e:  # stay on the (same) HIP kernels, just launched eagerly
            print("[DDIMSampler] HIP graph capture unavailable (%s); running steps eagerly" % e)
            self.use_graph = False
            return None

    def _step(self, x, x_info, c_info_list, step, index, guided, scale, temperature, single, noise_dropout=0.):
        """One p_sample_ddim (reference ddim.py:129-171 / 244-298) on the fp16 device latent `x` [B,C,H,W]."""
        b = x.shape[0]
        nb = 2 * b if guided else b
        t_in = torch.full((nb,), step, device=x.device, dtype=torch.long)
        xi = {"type": x_info["type"], "x": x, "repeat": 2 if guided else 1}   # [x; x] of the reference, see _loop_static
        if single:
            eps = self.model.apply_model(xi, t_in, c_info_list[0])
        else:
            eps = self.model.apply_model_multicontext(xi, t_in, c_info_list)
        sigma = float(self.ddim_sigmas[index])
        # drawn on every step like the reference's noise_like(x) (ddim.py:167 there): same generator consumption, and for
        # eta > 0 the same noise values as a reference running in fp16 on this device
        noise = torch.randn_like(x)
        if noise_dropout > 0.:
            # reference: noise = dropout(sigma_t * noise_like(x) * temperature, p) -- the mask (and its 1 / (1 - p) scale)
            # commutes with the scalar factors, so it is applied to the unit noise; drawn even at sigma = 0, like there
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        if sigma != 0.:
            noise = noise if temperature == 1. else (noise.float() * temperature).to(torch.float16)
        else:
            noise = None
        return ops.cfg_ddim_step(x, eps.contiguous(), guided=guided, guidance_scale=float(scale),
                                 a_t=float(self.ddim_alphas[index]), a_prev=float(self.ddim_alphas_prev[index]),
                                 sigma=sigma, sqrt_one_minus_at=float(self.ddim_sqrt_one_minus_alphas[index]),
                                 noise=noise)

    @torch.no_grad()
    def p_sample_ddim(self, x_info, c_info, t, index, repeat_noise=False, use_original_steps=False,
                      noise_dropout=0., temperature=1.):
        """Reference-compatible single step: returns (x_prev, pred_x0)."""
        assert not use_original_steps and not repeat_noise
        scale = c_info["unconditional_guidance_scale"]
        guided = scale != 1.
        ci = dict(c_info)
        ci["c"] = torch.cat([c_info["unconditional_conditioning"], c_info["conditioning"]]) if guided else c_info["conditioning"]
        x = x_info["x"]
        xp, p0 = self._step(x.to(torch.float16).contiguous(), x_info, [ci], int(t[0]), index, guided, scale,
                            temperature, True, noise_dropout=noise_dropout)
        return xp.to(x.dtype), p0.to(x.dtype)

    @torch.no_grad()
    def p_sample_ddim_multicontext(self, x_info, c_info_list, t, index, repeat_noise=False, use_original_steps=False,
                                   noise_dropout=0., temperature=1.):
        assert not use_original_steps and not repeat_noise
        scale = c_info_list[0]["unconditional_guidance_scale"]
        guided = scale != 1.
        cis = []
        for c_info in c_info_list:
            assert c_info["unconditional_guidance_scale"] == scale
            ci = dict(c_info)
            ci["c"] = torch.cat([c_info["unconditional_conditioning"], c_info["conditioning"]]) if guided else c_info["conditioning"]
            cis.append(ci)
        x = x_info["x"]
        xp, p0 = self._step(x.to(torch.float16).contiguous(), x_info, cis, int(t[0]), index, guided, scale,
                            temperature, False, noise_dropout=noise_dropout)
        return xp.to(x.dtype), p0.to(x.dtype)

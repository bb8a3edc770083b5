"""Noise schedules (host, float64 numpy like the reference) and the timestep embedding (device kernel).

Reference: lib/model_zoo/diffusion_utils.py:8-59 (schedules), :131-151 (timestep_embedding), :79-82
(extract_into_tensor).  The schedules are tiny one-off host computations; their values -- in particular the DDIM
index schedule -- must be bit-identical to the reference, so they stay in numpy float64 exactly as there."""
import numpy as np
import torch

from vd_hip import ops


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu") ** 2
    elif schedule == "cosine":
        ts = torch.arange(n_timestep + 1, dtype=torch.float64, device="cpu") / n_timestep + cosine_s
        alphas = torch.cos(ts / (1 + cosine_s) * np.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = torch.clamp(1 - alphas[1:] / alphas[:-1], min=0, max=0.999)
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64, device="cpu")
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64, device="cpu") ** 0.5
    else:
        raise ValueError("schedule '%s' unknown." % schedule)
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError('There is no ddim discretization method called "%s"' % ddim_discr_method)
    steps_out = ddim_timesteps + 1  # +1: final alpha values line up with the first scale-to-data step
    if verbose:
        print("Selected timesteps for ddim sampler: %s" % steps_out)
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphacums: float array [T] (fp32 values).  Returns float64 numpy (sigmas, alphas, alphas_prev).
    An index of T (e.g. 3 DDIM steps over 1000) raises IndexError exactly as the reference does."""
    ac = np.asarray(alphacums, dtype=np.float32)
    alphas = ac[ddim_timesteps].astype(np.float64)
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist(), dtype=np.float64)
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print("Selected alphas for ddim sampler: a_t: %s; a_(t-1): %s" % (alphas, alphas_prev))
        print("For the chosen value of eta, which is %s, this results in the following sigma_t schedule for ddim "
              "sampler %s" % (eta, sigmas))
    return sigmas, alphas, alphas_prev


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[N] int64 device tensor -> [N, dim] fp16 [cos | sin] (vd_timestep_embedding_f16, fp32 math)."""
    assert not repeat_only
    return ops.timestep_embedding(timesteps.to(torch.int64).contiguous(), dim, float(max_period))


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def count_params(model, verbose=False):
    return sum(p.numel() for p in model.parameters())

"""Deterministic synthetic weights and inputs.

There is no checkpoint and no dataset offline, so every parity test and the
benchmark use weights from ``fill_state_dict`` (all parameters non-zero,
including the zero-initialised ``SpatialTransformer.proj_out`` of
layers/SpatialTransformer.py:204, otherwise attention bugs are invisible) and
inputs from ``make_inputs``.  All randomness comes from
``np.random.RandomState(seed)``, the reference's own choice for its start code
(LatentDiffusionInpainting.py:302), so fixtures are regenerable anywhere.
"""
import numpy as np
import torch


def _fan_in(shape):
    n = 1
    for s in shape[1:]:
        n *= s
    return n


def fill_state_dict(sd, seed=0, fast=False):
    """Overwrite every tensor of ``sd`` in key order, in place. Returns sd.

    weights (ndim>=2): N(0,1)/sqrt(fan_in); 1-D ``*.weight`` (norm gains): 1 + 0.1 N(0,1);
    biases: 0.05 N(0,1).  ``fast`` switches to the PCG64 float32 generator (benchmark-only:
    not used for committed fixtures).
    """
    rs = np.random.default_rng(seed) if fast else np.random.RandomState(seed)

    def randn(shape):
        if fast:
            return rs.standard_normal(size=shape, dtype=np.float32)
        return rs.standard_normal(size=shape).astype(np.float32)

    with torch.no_grad():
        for k, v in sd.items():
            if not torch.is_floating_point(v):
                continue
            shape = tuple(v.shape)
            if k.endswith("weight") and v.ndim >= 2:
                a = randn(shape) * np.float32(1.0 / np.sqrt(_fan_in(shape)))
            elif k.endswith("weight"):
                a = np.float32(1.0) + np.float32(0.1) * randn(shape)
            else:
                a = np.float32(0.05) * randn(shape)
            v.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).to(v.device))
    return sd


def make_inputs(b, L, steps, seed=100, ctx_len=77, ctx_dim=1024):
    """Synthetic sampler inputs of SURVEY.md 8(d): x_T, mask latent (centred rectangle, 25 % area),
    masked-image latent, cond / uncond context, per-step noise (drawn in loop order)."""
    rs = np.random.RandomState(seed)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    x_T = f32(rs.randn(b, 4, L, L))
    mask = np.zeros((b, 1, L, L), dtype=np.float32)
    q = L // 4
    mask[:, :, q:L - q, q:L - q] = 1.0
    masked = f32(rs.randn(b, 4, L, L) * (0.18215 * 5.0))
    cond = f32(rs.randn(b, ctx_len, ctx_dim))
    uncond = f32(rs.randn(b, ctx_len, ctx_dim))
    noises = [f32(rs.randn(b, 4, L, L)) for _ in range(steps)]
    c_concat = torch.cat([f32(mask), masked], dim=1)
    return {"x_T": x_T, "c_concat": c_concat, "cond": cond, "uncond": uncond, "noises": noises}

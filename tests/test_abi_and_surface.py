"""CPU: the C-ABI library loads and exports exactly what include/sdmi.h declares; the drop-in modules expose the
reference's state_dict keys and signatures; the product path fails loudly without a GPU (no CPU fallback)."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from stable_diffusion_amd import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "sdmi.h")).read()
    return sorted(set(re.findall(r"\b(sdmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.sdmi_version() >= 100
    assert lib.sdmi_last_error() is not None


def test_argument_validation_returns_error_codes_without_gpu():
    lib = _lib.load()
    assert lib.sdmi_groupnorm_ws_bytes(2, 4096, 320) > 0
    # null pointers / bad shapes are rejected before any launch
    assert lib.sdmi_gemm(None, 8, 0, None, 8, 0, None, 8, 0, 4, 4, 8, 1, None, None, 0, 0, None, 0, 1.0, 0, None, 0, None) == -1
    assert lib.sdmi_conv3x3_ws_bytes(2, 1280, 1280, 12, 12) > 0 and lib.sdmi_conv3x3_ws_bytes(1, 128, 128, 768, 768) == 0
    assert b"null" in lib.sdmi_last_error()
    assert lib.sdmi_attention(None, 64, 0, None, 64, 0, None, 64, 0, None, 64, 0, 1, 1, 4, 4, 0.125, None) == -1
    with pytest.raises(_lib.SdmiError):
        _lib.call("sdmi_layernorm", 0, 8, 0, 0, 1e-5, 0, 8, 1, 8, None)


def test_state_dict_surface_matches_reference(golden):
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    g = golden("g6_keys")
    u, v = Unet(), AutoencoderKL()
    assert list(u.state_dict().keys()) == list(g["unet_keys"])
    assert [str(tuple(t.shape)) for t in u.state_dict().values()] == list(g["unet_shapes"])
    assert list(v.state_dict().keys()) == list(g["vae_keys"])
    assert [str(tuple(t.shape)) for t in v.state_dict().values()] == list(g["vae_shapes"])
    assert sum(p.numel() for p in u.parameters()) == 865925124


def test_signatures():
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    from layers.ResBlock import ResBlock, Upsample, Downsample
    from layers.SpatialTransformer import (SpatialTransformer, MemoryEfficientAttnBlock, MemoryEfficientCrossAttention,
                                           BasicTransformerBlock)
    P = lambda f: list(inspect.signature(f).parameters)
    assert P(Unet.forward) == ["self", "x", "timesteps", "context"]
    assert P(AutoencoderKL.decode) == ["self", "z"] and P(AutoencoderKL.encode) == ["self", "x"]
    assert P(ResBlock.__init__) == ["self", "in_channels", "out_channels", "time_emb_channels"]
    assert P(ResBlock.forward) == ["self", "x", "t_emb", "context"]
    assert P(Upsample.__init__) == ["self", "in_channels", "out_channels", "padding"] == P(Downsample.__init__)
    assert P(SpatialTransformer.__init__) == ["self", "in_channels", "n_heads"]
    assert P(SpatialTransformer.forward) == ["self", "x", "t_emb", "context"]
    assert P(MemoryEfficientAttnBlock.__init__) == ["self", "in_channels"]
    assert P(MemoryEfficientCrossAttention.__init__) == ["self", "query_dim", "context_dim", "heads", "dim_head", "dropout"]
    assert P(BasicTransformerBlock.__init__) == ["self", "dim", "n_heads", "context_dim"]
    st = SpatialTransformer(128, 2)
    assert float(st.proj_out.weight.abs().max()) == 0.0      # zero_module, reference layers/SpatialTransformer.py:204


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback():
    from layers.ResBlock import ResBlock
    m = ResBlock(64, 64, 0)
    with pytest.raises(_lib.SdmiError):
        m(torch.randn(1, 64, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "stable-diffusion_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "sd_oracle" not in src and "import oracle" not in src, f


def test_sampler_schedule_matches_golden(golden):
    from stable_diffusion_amd.sampler import ddim_schedule
    g = golden("g8_schedule")
    for n in (10, 45, 50):
        ts, a_t, a_prev, sig, s1m = ddim_schedule(n)
        np.testing.assert_array_equal(ts, g["n%d_timesteps" % n])
        for k, v in (("alphas", a_t), ("alphas_prev", a_prev), ("sigmas", sig), ("sqrt_one_minus_alphas", s1m)):
            np.testing.assert_array_equal(np.asarray(v, dtype=np.float64), g["n%d_%s" % (n, k)])
    with pytest.raises(IndexError):
        ddim_schedule(3)

"""CPU: the oracle reproduces every committed golden vector (generated from the reference by oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

import sd_oracle as O
from stable_diffusion_amd.synth import fill_state_dict, make_inputs

torch.set_grad_enabled(False)
T = torch.from_numpy


def _sd(module, seed, prefix="b."):
    sd = fill_state_dict(module.state_dict(), seed=seed)
    return {prefix + k: v for k, v in sd.items()}


def _psum(sd):
    return np.array([float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))])


def test_resblocks(golden):
    from layers.ResBlock import ResBlock
    g = golden("g1_resblock")
    for name, (cin, cout, temb) in {"rb_a": (64, 128, 256), "rb_b": (128, 128, 0)}.items():
        sd = _sd(ResBlock(cin, cout, temb), 11)
        np.testing.assert_allclose(_psum(sd), g[name + "_psum"], rtol=1e-9)
        t = T(g[name + "_t"]) if temb else None
        y = O.resblock(sd, "b", T(g[name + "_x"]), t)
        np.testing.assert_allclose(y.numpy(), g[name + "_y"], rtol=0, atol=2e-5)


def test_spatial_transformer(golden):
    from layers.SpatialTransformer import SpatialTransformer
    g = golden("g2_spatial_transformer")
    sd = _sd(SpatialTransformer(128, 2), 12)
    taps = {}
    y = O.spatial_transformer(sd, "b", T(g["x"]), T(g["ctx"]), taps)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)
    for k in ("attn1", "attn2", "ff"):
        np.testing.assert_allclose(taps[k].numpy(), g[k], rtol=0, atol=2e-5)


def test_vae_attn_and_updown(golden):
    from layers.SpatialTransformer import MemoryEfficientAttnBlock
    from layers.ResBlock import Upsample, Downsample
    g = golden("g3_vae_attn")
    y = O.vae_attn_block(_sd(MemoryEfficientAttnBlock(64), 13), "b", T(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)
    g = golden("g4_updown")
    x = T(g["x"])
    np.testing.assert_allclose(O.upsample(_sd(Upsample(64), 14), "b", x).numpy(), g["up_y"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(O.downsample(_sd(Downsample(64), 15), "b", x).numpy(), g["down_y"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(O.downsample(_sd(Downsample(64, padding=(0, 1, 0, 1)), 16), "b", x, (0, 1, 0, 1)).numpy(),
                               g["down_enc_y"], rtol=0, atol=2e-5)


def test_attention_kats(golden):
    g = golden("g5_attention")
    for name in ("cross", "self", "vae"):
        o = O.attention(T(g[name + "_q"]), T(g[name + "_k"]), T(g[name + "_v"]))
        np.testing.assert_allclose(o.numpy(), g[name + "_o"], rtol=0, atol=2e-6)


def test_timestep_embedding_and_schedule(golden):
    g = golden("g0_timestep_embedding")
    np.testing.assert_array_equal(O.timestep_embedding(T(g["t"]), 320).numpy(), g["emb"])
    g = golden("g8_schedule")
    for n in (10, 45, 50):
        sch = O.ddim_schedule(n)
        np.testing.assert_array_equal(sch["timesteps"], g["n%d_timesteps" % n])
        for k in ("alphas", "alphas_prev", "sigmas", "sqrt_one_minus_alphas"):
            np.testing.assert_array_equal(np.asarray(sch[k], dtype=np.float64), g["n%d_%s" % (n, k)])
    assert len(O.ddim_schedule(45)["timesteps"]) == 46 and len(O.ddim_schedule(50)["timesteps"]) == 50


def test_vae_full(golden):
    from AutoencoderKL import AutoencoderKL
    g = golden("g6_vae")
    sd = fill_state_dict(AutoencoderKL().state_dict(), seed=1)
    np.testing.assert_allclose(_psum(sd), g["psum"], rtol=1e-9)
    np.testing.assert_allclose(O.vae_decode(sd, T(g["z"])).numpy(), g["y"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(O.vae_encode_moments(sd, T(g["img"])).numpy(), g["moments"], rtol=0, atol=5e-5)


def test_unet_full_and_ddim(golden):
    """Full checkpoint-shaped UNet at L=8 + the 4-step DDIM trajectory of the reference's own sample()."""
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    g = golden("g6_unet")
    sd_u = fill_state_dict(Unet().state_dict(), seed=0)
    np.testing.assert_allclose(_psum(sd_u), g["psum"], rtol=1e-9)
    taps = {}
    y = O.unet_forward(sd_u, T(g["x"]), T(g["t"]), T(g["ctx"]), taps)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=5e-5)
    for k, v in taps.items():
        np.testing.assert_allclose([v.double().mean().item(), v.double().abs().mean().item()], g["tap_" + k], rtol=1e-4, atol=1e-6)
    g7 = golden("g7_ddim")
    steps, L, b, scale = int(g7["steps"]), int(g7["L"]), int(g7["b"]), float(g7["scale"])
    inp = make_inputs(b, L, steps, seed=100)
    sd_v = fill_state_dict(AutoencoderKL().state_dict(), seed=1)
    trace = []
    _, img = O.sample(sd_u, sd_v, steps, inp["cond"], inp["uncond"], inp["c_concat"], inp["x_T"], inp["noises"], scale, trace=trace)
    for i in range(steps):
        np.testing.assert_allclose(trace[i].numpy(), g7["traj"][i], rtol=0, atol=2e-3)
    assert np.abs(img.astype(np.int32) - g7["image"].astype(np.int32)).max() <= 1


def test_cfg_ddim_numpy_restatement_equals_torch_restatement():
    rs = np.random.RandomState(8)
    x, eu, ec, nz = (T(rs.randn(2, 4, 8, 8).astype(np.float32)) for _ in range(4))
    sch = O.ddim_schedule(50)
    for index in (0, 17, 49):
        args = (x, eu, ec, 9.0, sch["alphas"][index], sch["alphas_prev"][index], sch["sigmas"][index], sch["sqrt_one_minus_alphas"][index], nz)
        a, a0 = O.cfg_ddim_update(*args)
        b, b0 = O.cfg_ddim_update_np(*args)
        assert np.abs(a.numpy() - b).max() <= 1e-6 * np.abs(b).max() and np.abs(a0.numpy() - b0).max() <= 1e-6 * np.abs(b0).max()


def test_preprocess_restatement_equals_reference_fixture(golden):
    """G9: pad_image / make_batch_sd / the mask branch of the c_cat loop, outputs of the reference's own functions
    (oracle/make_golden.py compiles them from its source) against the oracle's restatement: identical."""
    from PIL import Image
    g = golden("g9_preprocess")
    pim, pmk = O.pad_image(Image.fromarray(g["image_u8"])), O.pad_image(Image.fromarray(g["mask_u8"]))
    assert np.array_equal(np.array(pim), g["padded_image_u8"]) and np.array_equal(np.array(pmk), g["padded_mask_u8"])
    b = O.make_batch_sd(pim, pmk, "p", num_samples=int(g["num_samples"]))
    for k in ("image", "mask", "masked_image"):
        assert b[k].shape[0] == 2 and np.array_equal(b[k][0].numpy(), g[k][0]) and torch.equal(b[k][0], b[k][1]), k
    ml = torch.nn.functional.interpolate(b["mask"], size=(b["mask"].shape[2] // 8, b["mask"].shape[3] // 8))
    assert np.array_equal(ml[:1].numpy(), g["mask_latent"])
    # host-side helpers of the product mirror the same padding rule
    from stable_diffusion_amd.inpaint import padded_size, pad_image
    assert padded_size(150, 97) == (192, 128) and padded_size(64, 64) == (128, 128) and padded_size(768, 768) == (768, 768)
    assert np.array_equal(np.array(pad_image(Image.fromarray(g["image_u8"]))), g["padded_image_u8"])


def test_clip_text_restatement_equals_torch_multihead_attention():
    """The oracle's restatement of open_clip's text tower (parity unpinned: open_clip is absent) against the same blocks
    built from torch.nn.MultiheadAttention in (L, N, D) layout with the additive causal mask - the module and call open_clip's
    ResidualAttentionBlock uses - on a small width; plus causality: later tokens do not change earlier positions."""
    from stable_diffusion_amd.clip_text import ClipTextEncoder
    torch.manual_seed(0)
    w, heads, layers, T_, vocab = 128, 2, 3, 77, 500
    m = ClipTextEncoder(vocab_size=vocab, context_length=T_, width=w, heads=heads, layers=layers)
    sd = {k: v.clone() for k, v in fill_state_dict(m.state_dict(), seed=4).items()}
    rs = np.random.RandomState(2)
    tokens = torch.from_numpy(rs.randint(0, vocab, size=(2, T_)))
    got = O.clip_text_encode(sd, tokens, heads=heads)
    x = (sd["token_embedding.weight"][tokens] + sd["positional_embedding"]).permute(1, 0, 2)       # NLD -> LND
    mask = torch.full((T_, T_), float("-inf")).triu_(1)
    for i in range(layers - 1):
        p = "transformer.resblocks.%d." % i
        mha = torch.nn.MultiheadAttention(w, heads)
        mha.load_state_dict({"in_proj_weight": sd[p + "attn.in_proj_weight"], "in_proj_bias": sd[p + "attn.in_proj_bias"],
                             "out_proj.weight": sd[p + "attn.out_proj.weight"], "out_proj.bias": sd[p + "attn.out_proj.bias"]})
        ln = lambda t, q: torch.nn.functional.layer_norm(t, (w,), sd[p + q + ".weight"], sd[p + q + ".bias"], 1e-5)
        h = ln(x, "ln_1")
        x = x + mha(h, h, h, need_weights=False, attn_mask=mask)[0]
        h = torch.nn.functional.linear(ln(x, "ln_2"), sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        x = x + torch.nn.functional.linear(torch.nn.functional.gelu(h), sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    want = torch.nn.functional.layer_norm(x.permute(1, 0, 2), (w,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    assert float((got - want).abs().max()) < 2e-5
    t2 = tokens.clone()
    t2[:, 40:] = torch.from_numpy(rs.randint(0, vocab, size=(2, T_ - 40)))
    got2 = O.clip_text_encode(sd, t2, heads=heads)
    assert torch.equal(got2[:, :40], got[:, :40]) and not torch.equal(got2[:, 40:], got[:, 40:])
    # open_clip's key set for the text tower (what the reference checkpoint's cond_stage_model.* entries are named)
    keys = set(m.state_dict().keys())
    for k in ("token_embedding.weight", "positional_embedding", "ln_final.weight", "text_projection", "logit_scale",
              "transformer.resblocks.0.attn.in_proj_weight", "transformer.resblocks.2.mlp.c_proj.bias",
              "transformer.resblocks.1.attn.out_proj.weight", "transformer.resblocks.0.ln_2.bias"):
        assert k in keys, k
    assert "attn_mask" not in keys

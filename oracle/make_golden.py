"""Pin the oracle against the reference itself and write tests/golden/*.npz  (TEST INFRASTRUCTURE).

Runs ONLY in the build container (needs /root/reference).  It
  1. imports the reference's own Unet.py / AutoencoderKL.py / layers/*.py with a stand-in for the
     missing third-party ``xformers`` op (softmax(q k^T/sqrt(d)) v, see oracle/sd_oracle.py header),
  2. executes the reference's own ``sample`` / ``diffusion_param_schedule`` / ``noise_like`` function
     objects, compiled at run time from /root/reference/LatentDiffusionInpainting.py (that file cannot
     be imported: it needs open_clip + gradio and launches a UI at import) in a namespace where
     ``torch.device('cuda')`` resolves to the CPU and ``noise_like`` replays injected noise,
  3. asserts oracle == reference on every case, and
  4. stores inputs / expected outputs (data only, no reference source) as small fixtures.

Usage:  python oracle/make_golden.py            (about 2-3 minutes, ~16 GB RAM for the full UNet)
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import sd_oracle as O  # noqa: E402
import stable_diffusion_amd  # noqa: E402,F401
from stable_diffusion_amd.synth import fill_state_dict, make_inputs  # noqa: E402


def _install_xformers_standin():
    xf, ops = types.ModuleType("xformers"), types.ModuleType("xformers.ops")
    ops.memory_efficient_attention = lambda q, k, v, attn_bias=None, op=None: O.attention(q, k, v)
    xf.ops = ops
    sys.modules["xformers"], sys.modules["xformers.ops"] = xf, ops


def _load_reference_sampler():
    """Compile the reference's sampler pieces from its source at run time (nothing is copied)."""
    src = open(os.path.join(REF, "LatentDiffusionInpainting.py")).read()
    tree = ast.parse(src)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("to_device", "noise_like", "pad_image", "make_batch_sd"):
            keep.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "GPUManager":
            keep.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "LatentDiffusionInpainting":
            node.body = [n for n in node.body if isinstance(n, ast.FunctionDef)
                         and n.name in ("diffusion_param_schedule", "sample")]
            keep.append(node)
    mod = ast.Module(body=keep, type_ignores=[])

    class TorchOnCpu:
        """``torch`` as the sampler sees it, with device('cuda') -> cpu (no GPU in this container)."""
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")

    import functools
    from PIL import Image
    ns = {"torch": TorchOnCpu(), "nn": torch.nn, "np": np, "partial": functools.partial,
          "tqdm": lambda it, **kw: it, "Image": Image, "repeat": __import__("einops").repeat, "time": __import__("time"), "print": lambda *a, **k: None}
    exec(compile(mod, "<reference sampler>", "exec"), ns)
    return ns


def reference_schedules(ns, step_counts):
    """G8: the schedule tables of the step counts the configs use, read from the REFERENCE object itself: its own
    sample() (compiled from its source) runs on a UNet stub that returns zeros and records the timesteps it is called
    with, and a decoder stub; the ddim_* attributes it leaves behind are the reference's tables.  The oracle's
    ddim_schedule() is asserted identical to them."""
    LDI = ns["LatentDiffusionInpainting"]

    class ZeroUnet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.seen = []

        def forward(self, xc, timesteps=None, context=None):
            self.seen.append(int(timesteps[0]))
            return torch.zeros((xc.shape[0], 4) + tuple(xc.shape[2:]))

    class ZeroVae(torch.nn.Module):
        def decode(self, z):
            return torch.zeros((z.shape[0], 3, 8, 8))

    saved = ns["noise_like"]
    ns["noise_like"] = lambda shape, device, repeat=False: torch.zeros(shape)
    g8 = {}
    for n in step_counts:
        model = LDI.__new__(LDI)
        torch.nn.Module.__init__(model)
        model.diffusion_model, model.first_stage_model = ZeroUnet(), ZeroVae()
        model.dpm_num_timesteps, model.scale_factor, model.channels = 1000, 0.18215, 4
        model.diffusion_param_schedule()
        c = {"c_concat": [torch.zeros(1, 5, 1, 1)], "c_crossattn": [torch.zeros(1, 77, 1024)]}
        model.sample(n, [1, 4, 1, 1], c, eta=1.0, unconditional_guidance_scale=9.0, unconditional_conditioning=c,
                     x_T=torch.zeros(1, 4, 1, 1))
        seen = model.diffusion_model.seen if hasattr(model.diffusion_model, "seen") else None
        ts = np.flip(np.asarray(seen))                       # sample() walks the timesteps from the last to the first
        sch = O.ddim_schedule(n)
        assert np.array_equal(ts, np.asarray(sch["timesteps"])), "timesteps n=%d" % n
        g8["n%d_timesteps" % n] = ts.astype(np.asarray(sch["timesteps"]).dtype)
        for k, attr in (("alphas", "ddim_alphas"), ("alphas_prev", "ddim_alphas_prev"), ("sigmas", "ddim_sigmas"),
                        ("sqrt_one_minus_alphas", "ddim_sqrt_one_minus_alphas")):
            ref_v = np.asarray(getattr(model, attr), dtype=np.float64)
            assert np.array_equal(ref_v, np.asarray(sch[k], dtype=np.float64)), "schedule %s n=%d" % (k, n)
            g8["n%d_%s" % (n, k)] = ref_v
        print("  n=%d: %d timesteps and 4 coefficient tables read from the reference object; oracle identical" % (n, len(ts)))
    ns["noise_like"] = saved
    return g8


def reference_preprocess(ns):
    """G9: the reference's own pad_image() and make_batch_sd() (compiled from its source) on a small odd-sized RGB image
    and an RGB mask with greys around the 0.5 threshold and coloured pixels (PIL's L conversion); plus the mask branch of
    inpaint()'s c_cat loop (F.interpolate to H/8 x W/8, LatentDiffusionInpainting.py:283-285).  The oracle's restatements
    are asserted identical."""
    from PIL import Image
    rs = np.random.RandomState(9)
    w0, h0 = 150, 97
    img = rs.randint(0, 256, size=(h0, w0, 3)).astype(np.uint8)
    msk = np.zeros((h0, w0, 3), dtype=np.uint8)
    msk[20:70, 30:120] = 255
    msk[5:15, 5:60] = rs.randint(120, 136, size=(10, 55, 1)).astype(np.uint8)        # greys around 127.5
    msk[75:95, 100:150] = rs.randint(0, 256, size=(20, 50, 3)).astype(np.uint8)       # coloured: exercises RGB -> L
    pim, pmk = ns["pad_image"](Image.fromarray(img)), ns["pad_image"](Image.fromarray(msk))
    batch = ns["make_batch_sd"](pim, pmk, txt="a prompt", num_samples=2)
    w, h = pim.size
    mask_lat = torch.nn.functional.interpolate(batch["mask"].float(), size=(h // 8, w // 8))
    opim, opmk = O.pad_image(Image.fromarray(img)), O.pad_image(Image.fromarray(msk))
    assert np.array_equal(np.array(opim), np.array(pim)) and np.array_equal(np.array(opmk), np.array(pmk))
    ob = O.make_batch_sd(opim, opmk, txt="a prompt", num_samples=2)
    for k in ("image", "mask", "masked_image"):
        assert torch.equal(ob[k], batch[k]), k
        assert torch.equal(batch[k][0], batch[k][1])
    assert ob["txt"] == batch["txt"]
    print("  pad_image %dx%d -> %dx%d, make_batch_sd: oracle identical to the reference" % (w0, h0, w, h))
    return {"image_u8": img, "mask_u8": msk, "padded_image_u8": np.array(pim), "padded_mask_u8": np.array(pmk),
            "image": _np(batch["image"][:1]), "mask": _np(batch["mask"][:1]), "masked_image": _np(batch["masked_image"][:1]),
            "mask_latent": _np(mask_lat[:1]), "num_samples": np.array(2)}


def _close(a, b, what, tol=2e-5):
    d = (a - b).abs().max().item()
    s = b.abs().max().item()
    print("  %-34s max|oracle-ref| = %.3e   (max|ref| = %.3e)" % (what, d, s))
    assert d <= tol * max(1.0, s), what
    return d


def _np(t):
    return t.detach().numpy().astype(np.float32)


def _csum(sd):
    return np.array([float(sum(v.double().sum() for v in sd.values())),
                     float(sum(v.double().abs().sum() for v in sd.values()))])


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    if "--only-g9" in sys.argv:       # regenerate the pre-processing fixture alone (seconds)
        np.savez_compressed(os.path.join(OUT, "g9_preprocess.npz"), **reference_preprocess(_load_reference_sampler()))
        return
    if "--only-g8" in sys.argv:       # regenerate the schedule fixture alone (seconds)
        np.savez(os.path.join(OUT, "g8_schedule.npz"), **reference_schedules(_load_reference_sampler(), (10, 45, 50)))
        return
    _install_xformers_standin()
    sys.path.insert(0, REF)
    from layers.ResBlock import ResBlock, Upsample, Downsample
    from layers.SpatialTransformer import SpatialTransformer, MemoryEfficientAttnBlock
    from Unet import Unet, timestep_embedding
    from AutoencoderKL import AutoencoderKL
    os.makedirs(OUT, exist_ok=True)
    rs = np.random.RandomState(1234)
    f32 = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32))

    # ---- G1 ResBlocks -------------------------------------------------------------------------
    print("G1 ResBlock")
    g = {}
    for name, (cin, cout, temb) in {"rb_a": (64, 128, 256), "rb_b": (128, 128, 0)}.items():
        m = ResBlock(cin, cout, temb).eval()
        sd = fill_state_dict(m.state_dict(), seed=11)
        x = f32(2, cin, 16, 16)
        t = f32(2, temb) if temb else None
        y = m(x, t_emb=t)
        sdp = {"b." + k: v for k, v in sd.items()}
        _close(O.resblock(sdp, "b", x, t), y, name)
        g[name + "_x"], g[name + "_y"], g[name + "_psum"] = _np(x), _np(y), _csum(sd)
        if t is not None:
            g[name + "_t"] = _np(t)
    np.savez(os.path.join(OUT, "g1_resblock.npz"), **g)

    # ---- G2 SpatialTransformer ----------------------------------------------------------------
    print("G2 SpatialTransformer")
    m = SpatialTransformer(128, 2).eval()
    sd = fill_state_dict(m.state_dict(), seed=12)
    x, ctx = f32(2, 128, 8, 8), f32(2, 77, 1024)
    y = m(x, context=ctx)
    blk = m.transformer_blocks[0]
    from einops import rearrange
    t0 = m.proj_in(rearrange(m.norm(x), "b c h w -> b (h w) c"))
    t1 = blk.attn1(blk.norm1(t0), context=None) + t0
    t2 = blk.attn2(blk.norm2(t1), context=ctx) + t1
    t3 = blk.ff(blk.norm3(t2)) + t2
    taps = {}
    sdp = {"b." + k: v for k, v in sd.items()}
    _close(O.spatial_transformer(sdp, "b", x, ctx, taps), y, "spatial_transformer")
    for k, r in (("attn1", t1), ("attn2", t2), ("ff", t3)):
        _close(taps[k], r, "  tap " + k)
    np.savez(os.path.join(OUT, "g2_spatial_transformer.npz"), x=_np(x), ctx=_np(ctx), y=_np(y), attn1=_np(t1),
             attn2=_np(t2), ff=_np(t3), psum=_csum(sd))

    # ---- G3 VAE attention block ---------------------------------------------------------------
    print("G3 MemoryEfficientAttnBlock")
    m = MemoryEfficientAttnBlock(64).eval()
    sd = fill_state_dict(m.state_dict(), seed=13)
    x = f32(1, 64, 8, 8)
    y = m(x)
    _close(O.vae_attn_block({"b." + k: v for k, v in sd.items()}, "b", x), y, "vae_attn_block")
    np.savez(os.path.join(OUT, "g3_vae_attn.npz"), x=_np(x), y=_np(y), psum=_csum(sd))

    # ---- G4 Up / Down -------------------------------------------------------------------------
    print("G4 Upsample / Downsample")
    g = {}
    x = f32(2, 64, 10, 12)
    g["x"] = _np(x)
    m = Upsample(64).eval(); sd = fill_state_dict(m.state_dict(), seed=14)
    y = m(x); _close(O.upsample({"b." + k: v for k, v in sd.items()}, "b", x), y, "upsample"); g["up_y"] = _np(y)
    m = Downsample(64).eval(); sd = fill_state_dict(m.state_dict(), seed=15)
    y = m(x); _close(O.downsample({"b." + k: v for k, v in sd.items()}, "b", x), y, "downsample(1,1,1,1)"); g["down_y"] = _np(y)
    m = Downsample(64, padding=(0, 1, 0, 1)).eval(); sd = fill_state_dict(m.state_dict(), seed=16)
    y = m(x); _close(O.downsample({"b." + k: v for k, v in sd.items()}, "b", x, (0, 1, 0, 1)), y, "downsample(0,1,0,1)")
    g["down_enc_y"] = _np(y)
    np.savez(os.path.join(OUT, "g4_updown.npz"), **g)

    # ---- G5 attention known-answer tests (fp64 definition; the op is parity-unpinned upstream) --
    print("G5 attention KATs (fp64 mathematical definition)")
    g = {}
    for name, (bh, nq, nk, d) in {"cross": (4, 200, 77, 64), "self": (3, 130, 130, 64), "vae": (1, 130, 130, 512)}.items():
        q, k, v = f32(bh, nq, d), f32(bh, nk, d), f32(bh, nk, d)
        o64 = O.attention(q.double(), k.double(), v.double())
        _close(O.attention(q, k, v), o64.float(), "attention " + name)
        g[name + "_q"], g[name + "_k"], g[name + "_v"], g[name + "_o"] = _np(q), _np(k), _np(v), _np(o64.float())
    np.savez(os.path.join(OUT, "g5_attention.npz"), **g)

    # ---- timestep embedding -------------------------------------------------------------------
    t = torch.tensor([1, 21, 501, 981], dtype=torch.long)
    _close(O.timestep_embedding(t, 320), timestep_embedding(t, 320), "timestep_embedding", tol=0)
    np.savez(os.path.join(OUT, "g0_timestep_embedding.npz"), t=t.numpy(), emb=_np(timestep_embedding(t, 320)))

    # ---- G6 whole UNet and whole VAE at L=8 ---------------------------------------------------
    print("G6 full Unet() / AutoencoderKL() at L=8 (checkpoint-shaped weights from fill_state_dict)")
    unet = Unet().eval()
    sd_u = fill_state_dict(unet.state_dict(), seed=0)
    vae = AutoencoderKL().eval()
    sd_v = fill_state_dict(vae.state_dict(), seed=1)
    keys = {"unet_keys": np.array(list(sd_u.keys())), "vae_keys": np.array(list(sd_v.keys())),
            "unet_shapes": np.array([str(tuple(v.shape)) for v in sd_u.values()]),
            "vae_shapes": np.array([str(tuple(v.shape)) for v in sd_v.values()])}
    np.savez_compressed(os.path.join(OUT, "g6_keys.npz"), **keys)
    x, ctx = f32(2, 9, 8, 8), f32(2, 77, 1024)
    t = torch.tensor([981, 981], dtype=torch.long)
    y = unet(x, timesteps=t, context=ctx)
    taps = {}
    _close(O.unet_forward(sd_u, x, t, ctx, taps), y, "unet_forward", tol=5e-5)
    g = {"x": _np(x), "ctx": _np(ctx), "t": t.numpy(), "y": _np(y), "psum": _csum(sd_u)}
    for k, v in taps.items():
        g["tap_" + k] = np.array([v.double().mean().item(), v.double().abs().mean().item()])
    np.savez(os.path.join(OUT, "g6_unet.npz"), **g)
    z = f32(1, 4, 8, 8)
    y = vae.decode(z)
    _close(O.vae_decode(sd_v, z), y, "vae_decode", tol=5e-5)
    img = f32(1, 3, 32, 32)
    mom = vae.encode(img).parameters
    _close(O.vae_encode_moments(sd_v, img), mom, "vae_encode_moments", tol=5e-5)
    np.savez(os.path.join(OUT, "g6_vae.npz"), z=_np(z), y=_np(y), img=_np(img), moments=_np(mom), psum=_csum(sd_v))

    # ---- G7/G8 the reference's own sampler ------------------------------------------------------
    print("G7/G8 reference sample() (compiled from the reference source at run time)")
    ns = _load_reference_sampler()
    LDI = ns["LatentDiffusionInpainting"]
    model = LDI.__new__(LDI)
    torch.nn.Module.__init__(model)
    model.diffusion_model, model.first_stage_model = unet, vae
    model.dpm_num_timesteps, model.scale_factor, model.channels = 1000, 0.18215, 4
    model.diffusion_param_schedule()
    g8 = reference_schedules(ns, (10, 45, 50))
    steps, L, b, scale = 4, 8, 1, 9.0  # n=3 hits the t=1000 index bug noted at LatentDiffusionInpainting.py:156
    inp = make_inputs(b, L, steps, seed=100)
    it = iter(inp["noises"])
    ns["noise_like"] = lambda shape, device, repeat=False: next(it)
    xs, zs = [], []
    orig_forward, orig_decode = unet.forward, vae.decode
    unet.forward = lambda xc, timesteps=None, context=None: (xs.append(xc[:b, :4].clone()), orig_forward(xc, timesteps=timesteps, context=context))[1]
    vae.decode = lambda z: (zs.append(z.clone()), orig_decode(z))[1]
    cond = {"c_concat": [inp["c_concat"]], "c_crossattn": [inp["cond"]]}
    uc = {"c_concat": [inp["c_concat"]], "c_crossattn": [inp["uncond"]]}
    imgs = model.sample(steps, [b, 4, L, L], cond, eta=1.0, unconditional_guidance_scale=scale,
                        unconditional_conditioning=uc, x_T=inp["x_T"])
    unet.forward, vae.decode = orig_forward, orig_decode
    ref_img = np.stack([np.array(im) for im in imgs])
    ref_traj = xs[1:] + [zs[0] * 0.18215]
    # the reference's schedule attributes for this n, straight from the object sample() ran on
    for k, attr in (("alphas", "ddim_alphas"), ("alphas_prev", "ddim_alphas_prev"), ("sigmas", "ddim_sigmas"),
                    ("sqrt_one_minus_alphas", "ddim_sqrt_one_minus_alphas")):
        ref_v = np.asarray(getattr(model, attr), dtype=np.float64)
        ora_v = np.asarray(O.ddim_schedule(steps)[k], dtype=np.float64)
        assert ref_v.shape == ora_v.shape and np.array_equal(ref_v, ora_v), k
        print("  schedule %-24s identical (%s)" % (k, type(getattr(model, attr)).__name__))
    trace = []
    x_fin, img = O.sample(sd_u, sd_v, steps, inp["cond"], inp["uncond"], inp["c_concat"], inp["x_T"], inp["noises"],
                          scale, trace=trace)
    for i in range(steps):
        _close(trace[i], ref_traj[i], "DDIM latent after step %d" % i, tol=1e-4)
    dimg = np.abs(img.astype(np.int32) - ref_img.astype(np.int32)).max()
    print("  decoded image max |oracle-ref| (uint8) = %d" % dimg)
    assert dimg <= 1
    np.savez(os.path.join(OUT, "g7_ddim.npz"), steps=steps, L=L, b=b, scale=scale,
             traj=np.stack([_np(t) for t in ref_traj]), image=ref_img)
    # schedules for the step counts the configs use: same code path as the one just proven identical
    np.savez(os.path.join(OUT, "g8_schedule.npz"), **g8)
    np.savez_compressed(os.path.join(OUT, "g9_preprocess.npz"), **reference_preprocess(ns))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()

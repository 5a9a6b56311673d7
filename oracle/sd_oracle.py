"""CPU oracle for the SD-2.x inpainting hot path (TEST INFRASTRUCTURE ONLY).

This file is a checker, not a product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``stable-diffusion_amd/`` does (the product fails loudly when
the HIP library is missing instead of falling back to this).

It restates, in plain fp32 torch *functions over a state_dict*, the arithmetic
of the reference modules (paths relative to /root/reference):

  * ``timestep_embedding``            Unet.py:7-25
  * ``unet_forward``                  Unet.py:109-146 (structure :57-107)
  * ``resblock``                      layers/ResBlock.py:64-76
  * ``upsample`` / ``downsample``     layers/ResBlock.py:16-20, :35-38
  * ``spatial_transformer``           layers/SpatialTransformer.py:206-220
  * ``transformer_block``             layers/SpatialTransformer.py:181-185
  * ``cross_attention``               layers/SpatialTransformer.py:69-96
  * ``feed_forward`` (GEGLU)          layers/SpatialTransformer.py:24-26, :45-46
  * ``vae_attn_block``                layers/SpatialTransformer.py:132-161
  * ``vae_decode`` / ``vae_encode_moments``  AutoencoderKL.py:177-196, :212-215,
                                      :100-120, :206-210
  * ``ddim_schedule``                 LatentDiffusionInpainting.py:125-135, :150-159
  * ``cfg_ddim_update`` / ``sample``  LatentDiffusionInpainting.py:174-230

The network topology is *derived from the state_dict keys* (the key names are
the drop-in contract), so the same functions run the full
checkpoint-shaped UNet/VAE and any reduced one used in unit tests.

Third-party op: ``xformers.ops.memory_efficient_attention`` (facebookresearch/
xformers, version not pinned anywhere in the reference, source not under
/root/reference).  Call sites layers/SpatialTransformer.py:86 and :151.  It is
restated from its documented semantics ``softmax(q k^T / sqrt(d)) v`` (no bias,
no dropout).  The reference holds no test or fixture at that boundary, so the
attention op itself is "parity unpinned"; everything around it is pinned by
running the reference's own modules (oracle/make_golden.py) with this same
definition plugged in as the stand-in.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

GROUPS = 32
HEAD_DIM = 64


# --------------------------------------------------------------------------- primitives
def _gn(x, sd, p, eps):
    return F.group_norm(x, GROUPS, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _linear(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def attention(q, k, v):
    """softmax(q k^T / sqrt(d)) v on (B*H, N, d) tensors (xformers semantics)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    return torch.matmul(torch.softmax(s, dim=-1), v)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# --------------------------------------------------------------------------- blocks
def resblock(sd, p, x, t_emb=None):
    h = _conv(F.silu(_gn(x, sd, p + ".norm1", 1e-5)), sd, p + ".conv1", padding=1)
    if t_emb is not None:
        h = h + _linear(F.silu(t_emb), sd, p + ".emb_proj")[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, p + ".norm2", 1e-5)), sd, p + ".conv2", padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = _conv(x, sd, p + ".skip_connection")
    return x + h


def upsample(sd, p, x, padding=(1, 1, 1, 1)):
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    x = F.pad(x, padding, mode="constant", value=0)
    return _conv(x, sd, p + ".conv")


def downsample(sd, p, x, padding=(1, 1, 1, 1)):
    x = F.pad(x, padding, mode="constant", value=0)
    return _conv(x, sd, p + ".conv", stride=2)


def _split_heads(t, heads):
    b, n, _ = t.shape
    return t.reshape(b, n, heads, HEAD_DIM).permute(0, 2, 1, 3).reshape(b * heads, n, HEAD_DIM)


def cross_attention(sd, p, x, context=None):
    heads = sd[p + ".to_q.weight"].shape[0] // HEAD_DIM
    ctx = x if context is None else context
    q = _split_heads(F.linear(x, sd[p + ".to_q.weight"]), heads)
    k = _split_heads(F.linear(ctx, sd[p + ".to_k.weight"]), heads)
    v = _split_heads(F.linear(ctx, sd[p + ".to_v.weight"]), heads)
    o = attention(q, k, v)
    b = x.shape[0]
    o = o.reshape(b, heads, o.shape[1], HEAD_DIM).permute(0, 2, 1, 3).reshape(b, o.shape[1], heads * HEAD_DIM)
    return _linear(o, sd, p + ".to_out.0")


def feed_forward(sd, p, x):
    a, gate = _linear(x, sd, p + ".net.0.proj").chunk(2, dim=-1)
    return _linear(a * F.gelu(gate), sd, p + ".net.2")


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def transformer_block(sd, p, x, context=None, taps=None):
    x = cross_attention(sd, p + ".attn1", _ln(x, sd, p + ".norm1")) + x
    if taps is not None:
        taps["attn1"] = x
    x = cross_attention(sd, p + ".attn2", _ln(x, sd, p + ".norm2"), context) + x
    if taps is not None:
        taps["attn2"] = x
    x = feed_forward(sd, p + ".ff", _ln(x, sd, p + ".norm3")) + x
    if taps is not None:
        taps["ff"] = x
    return x


def spatial_transformer(sd, p, x, context=None, taps=None):
    b, c, h, w = x.shape
    t = _gn(x, sd, p + ".norm", 1e-6).permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = _linear(t, sd, p + ".proj_in")
    t = transformer_block(sd, p + ".transformer_blocks.0", t, context, taps)
    t = _linear(t, sd, p + ".proj_out")
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def vae_attn_block(sd, p, x):
    b, c, h, w = x.shape
    hn = _gn(x, sd, p + ".norm", 1e-6)
    q, k, v = (_conv(hn, sd, p + "." + n).reshape(b, c, h * w).transpose(1, 2) for n in "qkv")
    o = attention(q, k, v).transpose(1, 2).reshape(b, c, h, w)
    return x + _conv(o, sd, p + ".proj_out")


# --------------------------------------------------------------------------- UNet
def _unet_stage(sd, p, h, t_emb, context):
    """One entry of input_blocks / output_blocks / middle_block: dispatch on its keys."""
    if (p + ".weight") in sd:                      # input_blocks.0 : plain conv (Unet.py:49)
        return _conv(h, sd, p, padding=1)
    if (p + ".conv.weight") in sd:                 # Downsample directly in the list (Unet.py:75)
        return downsample(sd, p, h)
    j = 0
    while True:
        q = "%s.%d" % (p, j)
        if (q + ".norm1.weight") in sd:
            h = resblock(sd, q, h, t_emb)
        elif (q + ".norm.weight") in sd:
            h = spatial_transformer(sd, q, h, context)
        elif (q + ".conv.weight") in sd:
            h = upsample(sd, q, h)
        else:
            return h
        j += 1


def _count(sd, prefix):
    n = 0
    while any(k.startswith("%s.%d." % (prefix, n)) for k in sd):
        n += 1
    return n


def unet_forward(sd, x, timesteps, context, taps=None):
    """Unet.forward (Unet.py:109-146). sd keys as Unet().state_dict()."""
    emb_ch = sd["time_embed.0.weight"].shape[1]
    t_emb = timestep_embedding(timesteps, emb_ch)
    t_emb = _linear(F.silu(_linear(t_emb, sd, "time_embed.0")), sd, "time_embed.2")
    hs, h = [], x
    for i in range(_count(sd, "input_blocks")):
        h = _unet_stage(sd, "input_blocks.%d" % i, h, t_emb, context)
        hs.append(h)
        if taps is not None:
            taps["input_blocks.%d" % i] = h
    for j in range(_count(sd, "middle_block")):
        q = "middle_block.%d" % j
        h = resblock(sd, q, h, t_emb) if (q + ".norm1.weight") in sd else spatial_transformer(sd, q, h, context)
    if taps is not None:
        taps["middle_block"] = h
    for i in range(_count(sd, "output_blocks")):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _unet_stage(sd, "output_blocks.%d" % i, h, t_emb, context)
        if taps is not None:
            taps["output_blocks.%d" % i] = h
    return _conv(F.silu(_gn(h, sd, "out.0", 1e-5)), sd, "out.2", padding=1)


# --------------------------------------------------------------------------- VAE
def vae_decode(sd, z, taps=None):
    """AutoencoderKL.decode (AutoencoderKL.py:212-215, Decoder.forward :177-196)."""
    z = _conv(z, sd, "post_quant_conv")
    d = "decoder"
    h = _conv(z, sd, d + ".conv_in", padding=1)
    h = resblock(sd, d + ".mid.block_1", h)
    h = vae_attn_block(sd, d + ".mid.attn_1", h)
    h = resblock(sd, d + ".mid.block_2", h)
    if taps is not None:
        taps["mid"] = h
    for lvl in reversed(range(_count(sd, d + ".up"))):
        for j in range(_count(sd, "%s.up.%d.block" % (d, lvl))):
            h = resblock(sd, "%s.up.%d.block.%d" % (d, lvl, j), h)
        if ("%s.up.%d.upsample.conv.weight" % (d, lvl)) in sd:
            h = upsample(sd, "%s.up.%d.upsample" % (d, lvl), h)
        if taps is not None:
            taps["up.%d" % lvl] = h
    return _conv(F.silu(_gn(h, sd, d + ".norm_out", 1e-6)), sd, d + ".conv_out", padding=1)


def vae_encode_moments(sd, x):
    """AutoencoderKL.encode up to the moments tensor (Encoder.forward AutoencoderKL.py:100-120, quant_conv :208)."""
    e = "encoder"
    h = _conv(x, sd, e + ".conv_in", padding=1)
    nlev = _count(sd, e + ".down")
    for lvl in range(nlev):
        for j in range(_count(sd, "%s.down.%d.block" % (e, lvl))):
            h = resblock(sd, "%s.down.%d.block.%d" % (e, lvl, j), h)
        if ("%s.down.%d.downsample.conv.weight" % (e, lvl)) in sd:
            h = downsample(sd, "%s.down.%d.downsample" % (e, lvl), h, padding=(0, 1, 0, 1))
    h = resblock(sd, e + ".mid.block_1", h)
    h = vae_attn_block(sd, e + ".mid.attn_1", h)
    h = resblock(sd, e + ".mid.block_2", h)
    h = _conv(F.silu(_gn(h, sd, e + ".norm_out", 1e-6)), sd, e + ".conv_out", padding=1)
    return _conv(h, sd, "quant_conv")


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution(moments).sample() with the randn injected (AutoencoderKL.py:9-21)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


# --------------------------------------------------------------------------- pre-processing
def pad_image(input_image):
    """pad_image (LatentDiffusionInpainting.py:58-63): PIL image -> PIL image padded on the bottom / right by edge
    replication to the next multiple of 64, at least 128 on each side."""
    from PIL import Image
    pad_w, pad_h = np.max(((2, 2), np.ceil(np.array(input_image.size) / 64).astype(int)), axis=0) * 64 - input_image.size
    return Image.fromarray(np.pad(np.array(input_image), ((0, pad_h), (0, pad_w), (0, 0)), mode='edge'))


def make_batch_sd(image, mask, txt, num_samples=1):
    """make_batch_sd (LatentDiffusionInpainting.py:65-89) on PIL images; `repeat(x, "1 ... -> n ...")` restated with
    torch.repeat (einops is not needed for a leading-axis repeat)."""
    image = np.array(image.convert("RGB"))
    image = image[None].transpose(0, 3, 1, 2)
    image = torch.from_numpy(image).to(dtype=torch.float32) / 127.5 - 1.0
    mask = np.array(mask.convert("L"))
    mask = mask.astype(np.float32) / 255.0
    mask = mask[None, None]
    mask[mask < 0.5] = 0
    mask[mask >= 0.5] = 1
    mask = torch.from_numpy(mask)
    masked_image = image * (mask < 0.5)
    rep = lambda t: t.repeat(num_samples, 1, 1, 1)
    return {"image": rep(image), "txt": num_samples * [txt], "mask": rep(mask), "masked_image": rep(masked_image)}


def image_conditioning(sd_vae, batch, noise, scale_factor=0.18215):
    """The c_cat loop of inpaint() (LatentDiffusionInpainting.py:279-293) with the posterior's randn injected:
    [nearest-resized mask | scale_factor * encode(masked_image).sample()]."""
    h, w = batch["mask"].shape[-2:]
    mask_lat = F.interpolate(batch["mask"].float(), size=(h // 8, w // 8))
    lat = scale_factor * gaussian_sample(vae_encode_moments(sd_vae, batch["masked_image"].float()), noise)
    return torch.cat([mask_lat, lat], dim=1)


# --------------------------------------------------------------------------- text encoder
def clip_text_encode(sd, tokens, heads=16):
    """LatentDiffusionInpainting.get_conditional (:232-245) on token ids, with open_clip's text tower restated from its
    published definition (mlfoundations/open_clip `CLIP.encode_text` / `ResidualAttentionBlock`; the package is a
    dependency the reference does not pin and that is absent here, so this restatement is PARITY UNPINNED):
        x = token_embedding[tokens] + positional_embedding
        for every block but the last:  x = x + MHA(ln_1(x), causal -inf mask);  x = x + c_proj(GELU(c_fc(ln_2(x))))
        x = ln_final(x)
    MHA is nn.MultiheadAttention's arithmetic: in_proj -> (q, k, v), heads of width/heads, softmax(q k^T / sqrt(d) + mask) v,
    out_proj.  sd: open_clip's state_dict names."""
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    b, t, w = x.shape
    mask = torch.full((t, t), float("-inf")).triu_(1)
    n = _count(sd, "transformer.resblocks")
    for i in range(n - 1):
        p = "transformer.resblocks.%d" % i
        h = F.layer_norm(x, (w,), sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"], 1e-5)
        qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
        q, k, v = (z.reshape(b, t, heads, w // heads).permute(0, 2, 1, 3) for z in qkv.chunk(3, dim=-1))
        a = torch.softmax(q @ k.transpose(-1, -2) * (w // heads) ** -0.5 + mask, dim=-1) @ v
        a = a.permute(0, 2, 1, 3).reshape(b, t, w)
        x = x + F.linear(a, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        h = F.layer_norm(x, (w,), sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"], 1e-5)
        h = F.gelu(F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]))
        x = x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    return F.layer_norm(x, (w,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)


# --------------------------------------------------------------------------- sampler
def ddim_schedule(num_ddim_timesteps, eta=1.0, num_train=1000, linear_start=1e-4, linear_end=2e-2):
    """Timesteps and per-step coefficients exactly as LatentDiffusionInpainting.py:125-135, :150-159
    builds them, including its dtypes (fp32 alphas_cumprod; float64 alphas_prev / sigmas)."""
    alphas = 1.0 - (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_train, dtype=torch.float64) ** 2).numpy()
    alphas_cumprod = torch.tensor(np.cumprod(alphas, axis=0), dtype=torch.float32)
    c = num_train // num_ddim_timesteps
    ts = np.asarray(list(range(0, num_train, c))) + 1
    a = alphas_cumprod[ts]
    a_prev = np.asarray([alphas_cumprod[0]] + alphas_cumprod[ts[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    sqrt_1ma = np.sqrt(1.0 - a)
    return {"timesteps": ts, "alphas": a, "alphas_prev": a_prev, "sigmas": sigmas, "sqrt_one_minus_alphas": sqrt_1ma}


def cfg_ddim_update(x, e_uncond, e_cond, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise, temperature=1.0):
    """One CFG combine + DDIM latent update (LatentDiffusionInpainting.py:198-212); coefficients enter
    through torch.full(...) exactly as there, i.e. rounded to fp32."""
    b = x.shape[0]
    full = lambda v: torch.full((b, 1, 1, 1), v)
    e = e_uncond + scale * (e_cond - e_uncond)
    a_t, a_prev, sigma_t, s1m = full(a_t), full(a_prev), full(sigma_t), full(sqrt_one_minus_at)
    pred_x0 = (x - s1m * e) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e
    return a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * noise * temperature, pred_x0


def cfg_ddim_update_np(x, e_uncond, e_cond, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise, temperature=1.0):
    """The same update as IEEE fp32 arithmetic, one rounding per operation, in numpy (no vectorised-FMA or
    host-dependent kernel choices): the bit-exact target of csrc/cfg_ddim.hip.  Equals cfg_ddim_update() bit for
    bit in the build container; torch CPU builds on other hosts may differ from it in the last ulp."""
    F = np.float32
    r = lambda v: F(torch.full((1,), v).item())
    x, eu, ec, nz = (np.asarray(t, dtype=np.float32) for t in (x, e_uncond, e_cond, noise))
    a_t, a_prev, sg, s1m = r(a_t), r(a_prev), r(sigma_t), r(sqrt_one_minus_at)
    e = eu + F(scale) * (ec - eu)
    pred_x0 = (x - s1m * e) / np.sqrt(a_t)
    dir_xt = np.sqrt(F(1.0) - a_prev - sg * sg) * e
    return np.sqrt(a_prev) * pred_x0 + dir_xt + sg * nz * F(temperature), pred_x0


def sample(sd_unet, sd_vae, num_ddim_timesteps, cond, uncond, c_concat, x_T, noises, scale, eta=1.0,
           decode=True, trace=None):
    """LatentDiffusionInpainting.sample (:137-230) with the per-step randn injected (``noises[i]`` is the
    draw of loop iteration i).  cond/uncond: (b,77,1024); c_concat: (b,5,h,w) = [mask | masked latent]."""
    sch = ddim_schedule(num_ddim_timesteps, eta)
    ts = sch["timesteps"]
    total = ts.shape[0]
    x = x_T
    b = x.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t_in = torch.full((2 * b,), int(step), dtype=torch.long)
        xc = torch.cat([torch.cat([x] * 2), torch.cat([c_concat] * 2)], dim=1)
        cc = torch.cat([uncond, cond])
        e_u, e_c = unet_forward(sd_unet, xc, t_in, cc).chunk(2)
        x, _ = cfg_ddim_update(x, e_u, e_c, scale, sch["alphas"][index], sch["alphas_prev"][index],
                               sch["sigmas"][index], sch["sqrt_one_minus_alphas"][index], noises[i])
        if trace is not None:
            trace.append(x.clone())
    if not decode:
        return x
    out = vae_decode(sd_vae, 1.0 / 0.18215 * x)
    img = torch.clamp((out + 1.0) / 2.0, min=0.0, max=1.0)
    return x, (img.numpy().transpose(0, 2, 3, 1) * 255).astype(np.uint8)

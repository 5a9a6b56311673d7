"""Drop-in for the reference's AutoencoderKL.py: same classes, constructor signatures and state_dict keys
(`encoder.*`, `decoder.*`, `quant_conv.*`, `post_quant_conv.*`); decode()/encode() take and return NCHW fp32 tensors,
the arithmetic runs on libsdmi355.so."""
import torch
from torch import nn

from layers.ResBlock import ResBlock, Downsample, Upsample
from layers.SpatialTransformer import MemoryEfficientAttnBlock
from stable_diffusion_amd import engine as E

CH, CH_MULT, NUM_RES, Z_CH = 128, (1, 2, 4, 4), 2, 4


class DiagonalGaussianDistribution(object):
    """Posterior over latents (reference AutoencoderKL.py:8-31, :45-46: what `encode()` returns and `inpaint()` samples); the
    training-only `kl` / `nll` (:33-43) are out of scope and not carried."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape).to(device=self.parameters.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


def _mid(block_in):
    mid = nn.Module()
    mid.block_1 = ResBlock(block_in, block_in, 0)
    mid.attn_1 = MemoryEfficientAttnBlock(block_in)
    mid.block_2 = ResBlock(block_in, block_in, 0)
    return mid


def _conv_in(x, conv):
    x = x.contiguous().float()
    n, c, h, w = x.shape
    a = E.new_act(n, h, w, conv.out_channels, x.device)
    cs = None
    if E.USE_COLSTATS and E.CONV_IN_COLSTATS and (h * w) % 64 == 0 and h * w >= 2304:
        cs = torch.empty((n * h * w // 32, conv.out_channels, 2), dtype=torch.float32, device=x.device)   # for the GroupNorm that follows
        a.stats[a.off] = (cs, h * w // 32, conv.out_channels)
    E.L.call("sdmi_conv3x3_in_ex", x.data_ptr(), n, c, h, w, conv.weight.data_ptr(), conv.bias.data_ptr(), conv.out_channels,
             a.ptr, a.ld, E._ptr(cs), E._stream())
    E._audit("conv_in %d->%d @%dx%d" % (c, conv.out_channels, h, w), a)
    return a


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(CH_MULT), NUM_RES
        self.conv_in = nn.Conv2d(3, CH, kernel_size=3, stride=1, padding=1)
        self.down = nn.ModuleList()
        block_in = CH
        for lvl, mult in enumerate(CH_MULT):
            down = nn.Module()
            down.block = nn.ModuleList()
            for _ in range(NUM_RES):
                down.block.append(ResBlock(block_in, CH * mult, 0))
                block_in = CH * mult
            if lvl != len(CH_MULT) - 1:
                down.downsample = Downsample(block_in, padding=(0, 1, 0, 1))
            self.down.append(down)
        self.mid = _mid(block_in)
        self.norm_out = nn.GroupNorm(num_groups=32, num_channels=block_in, eps=1e-6, affine=True)
        self.conv_out = nn.Conv2d(block_in, 2 * Z_CH, kernel_size=3, stride=1, padding=1)

    def run(self, x):
        h = _conv_in(x, self.conv_in)
        for lvl, down in enumerate(self.down):
            for blk in down.block:
                h = blk.run(h)
            if lvl != self.num_resolutions - 1:
                h = down.downsample.run(h)
        h = self.mid.block_2.run(self.mid.attn_1.run(self.mid.block_1.run(h)))
        h = E.groupnorm(h, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps, silu=True)
        return E.conv3x3(h, self.conv_out.weight, self.conv_out.bias)

    @torch.no_grad()
    def forward(self, x):
        return E.to_nchw(self.run(x))


class Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(CH_MULT), NUM_RES
        block_in = CH * CH_MULT[-1]
        self.conv_in = nn.Conv2d(Z_CH, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = _mid(block_in)
        self.up = nn.ModuleList()
        for lvl in reversed(range(len(CH_MULT))):
            up = nn.Module()
            up.block = nn.ModuleList()
            for _ in range(NUM_RES + 1):
                up.block.append(ResBlock(block_in, CH * CH_MULT[lvl], 0))
                block_in = CH * CH_MULT[lvl]
            if lvl != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out = nn.GroupNorm(num_groups=32, num_channels=block_in, eps=1e-6, affine=True)
        self.conv_out = nn.Conv2d(block_in, 3, kernel_size=3, stride=1, padding=1)

    def run(self, z, taps=None, image_u8=False):
        h = _conv_in(z, self.conv_in)
        h = self.mid.block_2.run(self.mid.attn_1.run(self.mid.block_1.run(h)))
        if taps is not None:
            taps["mid"] = h
        for lvl in reversed(range(self.num_resolutions)):
            for blk in self.up[lvl].block:
                h = blk.run(h)
            if lvl != 0:
                h = self.up[lvl].upsample.run(h)
            if taps is not None:
                taps["up.%d" % lvl] = h
        h = E.groupnorm(h, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps, silu=True)
        if image_u8:  # (n, H, W, 3) uint8 straight from conv_out's epilogue (LatentDiffusionInpainting.py:228-230)
            return E.conv3x3(h, self.conv_out.weight, self.conv_out.bias, out_u8_image=True)
        return E.conv3x3(h, self.conv_out.weight, self.conv_out.bias, out_nchw=True)

    @torch.no_grad()
    def forward(self, z):
        return self.run(z)


def _conv1x1_nchw(x, conv):
    """Tiny 1x1 conv on an NCHW fp32 tensor with <= 8 channels (quant_conv 8->8, post_quant_conv 4->4): through the
    GEMM kernel on a channel-padded NHWC copy."""
    n, c, h, w = x.shape
    a = E.from_nchw(x)                      # channel-padded to 8 with zeros
    a.c = a.ld
    co = conv.out_channels
    wp = E.PACK.get((conv.weight,), "1x1pad", lambda: _padded_1x1(conv.weight, a.ld))
    o = E.new_act(n, h, w, 8, x.device)
    o.buf.zero_()
    E.L.call("sdmi_gemm", a.ptr, a.ld, 0, wp.data_ptr(), wp.shape[1], 0, o.ptr, o.ld, 0, n * h * w, co, a.ld, 1,
             conv.bias.data_ptr(), 0, 0, h * w, 0, 0, 1.0, E.L.F_BIAS, 0, 0, E._stream())
    o.c = co
    return E.to_nchw(o)


def _padded_1x1(weight, kpad):
    co, ci = weight.shape[0], weight.shape[1]
    w = torch.zeros((co, kpad), dtype=torch.float32, device=weight.device)
    w[:, :ci] = weight.reshape(co, ci)
    out = torch.empty((co, kpad), dtype=E.act_dtype(), device=weight.device)
    E.L.call("sdmi_pack_linear", w.data_ptr(), out.data_ptr(), co, kpad, 0, E._stream())
    return out


class AutoencoderKL(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = Decoder()
        self.quant_conv = nn.Conv2d(2 * Z_CH, 2 * Z_CH, kernel_size=1)
        self.post_quant_conv = nn.Conv2d(Z_CH, Z_CH, kernel_size=1)

    @torch.no_grad()
    def encode(self, x):
        h = self.encoder(x)
        return DiagonalGaussianDistribution(_conv1x1_nchw(h, self.quant_conv))

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(_conv1x1_nchw(z.contiguous().float(), self.post_quant_conv))

    @torch.no_grad()
    def decode_image_u8(self, z):
        """decode(z) followed by the reference's image post-processing `clamp((x + 1) / 2, 0, 1) * 255 -> uint8`,
        (n, H, W, 3) as PIL takes it, with the post-processing done in the last conv's epilogue (same bits as
        `(torch.clamp((decode(z) + 1.0) / 2.0, 0.0, 1.0).permute(0, 2, 3, 1) * 255).to(torch.uint8)`)."""
        return self.decoder.run(_conv1x1_nchw(z.contiguous().float(), self.post_quant_conv), image_u8=True)

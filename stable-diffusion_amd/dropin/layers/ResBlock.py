"""Drop-in for the reference's layers/ResBlock.py (same class names, constructor signatures, parameter names and
forward signatures - SURVEY.md 8(b)); the arithmetic runs in libsdmi355.so through stable_diffusion_amd.engine.

forward() accepts the reference's NCHW fp32 tensors (module boundary).  run() is the internal fast path on NHWC bf16
activation views (engine.Act) that Unet / AutoencoderKL use so nothing is converted between blocks."""
import torch
from torch import nn

from stable_diffusion_amd import engine as E


class Upsample(nn.Module):
    """nearest-2x -> zero-pad -> conv3x3 (reference layers/ResBlock.py:4-20) as ONE conv launch with an upsample gather."""

    def __init__(self, in_channels, out_channels=None, padding=(1, 1, 1, 1)):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(in_channels, out_channels or in_channels, kernel_size=3)

    def run(self, a, out=None):
        return E.conv3x3(a, self.conv.weight, self.conv.bias, stride=1, pad=self.padding, upsample=True, out=out)

    def forward(self, x):
        return E.to_nchw(self.run(E.from_nchw(x)))


class Downsample(nn.Module):
    """zero-pad -> conv3x3 stride 2, C -> C (reference layers/ResBlock.py:22-38; out_channels is ignored there too)."""

    def __init__(self, in_channels, out_channels=None, padding=(1, 1, 1, 1)):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2)

    def run(self, a, out=None):
        return E.conv3x3(a, self.conv.weight, self.conv.bias, stride=2, pad=self.padding, out=out)

    def forward(self, x):
        return E.to_nchw(self.run(E.from_nchw(x)))


class ResBlock(nn.Module):
    """GN+SiLU -> conv3x3 (+bias +timestep-embedding) -> GN+SiLU -> conv3x3 (+bias +skip)  (reference :40-76)."""

    def __init__(self, in_channels, out_channels, time_emb_channels):
        super().__init__()
        self.norm1 = nn.GroupNorm(num_groups=32, num_channels=in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.emb_proj = nn.Linear(time_emb_channels, out_channels) if time_emb_channels > 0 else nn.Identity()
        self.norm2 = nn.GroupNorm(num_groups=32, num_channels=out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.skip_connection = nn.Identity() if out_channels == in_channels else nn.Conv2d(in_channels, out_channels, 1)

    def run(self, a, emb=None, ldrb=0, out=None):
        """emb: device pointer (int) or fp32 tensor holding emb_proj(SiLU(t_emb)) rows of stride ldrb, or None."""
        h = E.groupnorm(a, self.norm1.weight, self.norm1.bias, self.norm1.eps, silu=True, conv_cout=self.conv1.out_channels)
        h = E.conv3x3(h, self.conv1.weight, self.conv1.bias, rowbias=emb, ldrb=ldrb,
                      gn_next=(self.norm2.weight, self.norm2.bias, self.norm2.eps, True), keep_raw=False)
        h = E.groupnorm(h, self.norm2.weight, self.norm2.bias, self.norm2.eps, silu=True, conv_cout=self.conv2.out_channels)
        if isinstance(self.skip_connection, nn.Conv2d):
            sk = self.skip_connection
            skip = E.gemm(a, E.packed_linear((sk.weight,)), sk.weight.shape[0], bias=sk.bias)
        else:
            skip = a
        return E.conv3x3(h, self.conv2.weight, self.conv2.bias, residual=skip, out=out)

    def forward(self, x, t_emb=None, context=None):
        emb, ldrb = None, 0
        if t_emb is not None:
            p = self.emb_proj
            emb = E.linear_small_m(t_emb.contiguous().float(), E.packed_linear((p.weight,)), p.bias, silu_in=True)
            ldrb = emb.shape[1]
        return E.to_nchw(self.run(E.from_nchw(x), emb, ldrb))

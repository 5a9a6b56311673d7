"""Drop-in for the reference's Unet.py: `Unet()` with the same module tree / state_dict keys and
`forward(x, timesteps=None, context=None)` on NCHW fp32 tensors, executed on libsdmi355.so.

Differences in HOW (not WHAT): activations are NHWC bf16 views between blocks; the first conv reads the caller's
NCHW fp32 input directly and the last conv writes NCHW fp32 directly; the 12 skip `torch.cat`s of Unet.py:132 are
gone - every producer writes straight into its half of a pre-sized concat buffer; the time-embedding MLP and all 22
`emb_proj` projections run as three skinny-GEMM launches whose result the conv epilogues add per sample."""
import torch
from torch import nn

from layers.ResBlock import ResBlock, Downsample, Upsample
from layers.SpatialTransformer import SpatialTransformer, ContextKV
from stable_diffusion_amd import engine as E


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding [cos | sin] (reference Unet.py:7-25), computed on the device."""
    return E.timestep_embedding(timesteps, dim, max_period)


# SD-2 inpainting UNet hyper-parameters (reference Unet.py:30-37)
IN_CH, MODEL_CH, OUT_CH = 9, 320, 4
RES_BLOCKS, ATTN_DS, CH_MULT, HEAD_CH = (2, 2, 2, 2), (1, 2, 4), (1, 2, 4, 4), 64


class TimeEmb(object):
    """Precomputed timestep-embedding table of one UNet evaluation (Unet.precompute_time_embedding)."""
    __slots__ = ("table",)

    def __init__(self, table):
        self.table = table


class Unet(nn.Module):
    def __init__(self):
        super().__init__()
        self.emb_channels = MODEL_CH
        temb = MODEL_CH * 4
        self.time_embed = nn.Sequential(nn.Linear(MODEL_CH, temb), nn.SiLU(), nn.Linear(temb, temb))

        def stage(cin, cout, ds, up=False):
            mods = [ResBlock(cin, cout, temb)]
            if ds in ATTN_DS:
                mods.append(SpatialTransformer(cout, cout // HEAD_CH))
            if up:
                mods.append(Upsample(cout, out_channels=cout))
            return nn.ModuleList(mods)

        self.input_blocks = nn.ModuleList([nn.Conv2d(IN_CH, MODEL_CH, 3, padding=1)])
        skip_ch, ch, ds = [MODEL_CH], MODEL_CH, 1
        for level, mult in enumerate(CH_MULT):
            for _ in range(RES_BLOCKS[level]):
                self.input_blocks.append(stage(ch, mult * MODEL_CH, ds))
                ch = mult * MODEL_CH
                skip_ch.append(ch)
            if level != len(CH_MULT) - 1:
                self.input_blocks.append(Downsample(ch, out_channels=ch))
                skip_ch.append(ch)
                ds *= 2
        self.middle_block = nn.ModuleList([ResBlock(ch, ch, temb), SpatialTransformer(ch, ch // HEAD_CH), ResBlock(ch, ch, temb)])
        self.output_blocks = nn.ModuleList([])
        for level in reversed(range(len(CH_MULT))):
            for i in range(RES_BLOCKS[level] + 1):
                last = bool(level) and i == RES_BLOCKS[level]
                self.output_blocks.append(stage(ch + skip_ch.pop(), CH_MULT[level] * MODEL_CH, ds, up=last))
                ch = CH_MULT[level] * MODEL_CH
                if last:
                    ds //= 2
        self.out = nn.Sequential(nn.GroupNorm(num_groups=32, num_channels=ch), nn.SiLU(), nn.Conv2d(MODEL_CH, OUT_CH, 3, padding=1))

    # ------------------------------------------------------------------ helpers
    def _resblocks(self):
        for blk in list(self.input_blocks) + [self.middle_block] + list(self.output_blocks):
            if isinstance(blk, nn.ModuleList):
                for m in blk:
                    if isinstance(m, ResBlock):
                        yield m

    def _cross_attn_modules(self):
        for blk in list(self.input_blocks) + [self.middle_block] + list(self.output_blocks):
            if isinstance(blk, nn.ModuleList):
                for m in blk:
                    if isinstance(m, SpatialTransformer):
                        yield m.transformer_blocks[0].attn2

    @torch.no_grad()
    def precompute_context(self, context):
        """Project the (2b, 77, 1024) text context through every cross-attention to_k / to_v once.  The result can be
        passed as `context=` to forward() for every DDIM step of the same prompt (step-invariant hoisting)."""
        if isinstance(context, list):
            context = context[0]
        return ContextKV(list(self._cross_attn_modules()), context)

    def _embeddings(self, timesteps):
        """t_emb MLP (Unet.py:112-114) + every ResBlock's emb_proj(SiLU(t_emb)) (layers/ResBlock.py:70), batched."""
        t = E.timestep_embedding(timesteps, self.emb_channels)
        l0, l2 = self.time_embed[0], self.time_embed[2]
        h = E.linear_small_m(t, E.packed_linear((l0.weight,)), l0.bias)
        t_emb = E.linear_small_m(h, E.packed_linear((l2.weight,)), l2.bias, silu_in=True)
        rbs = list(self._resblocks())
        w = E.packed_linear(tuple(rb.emb_proj.weight for rb in rbs))
        b = E.stacked_f32(tuple(rb.emb_proj.bias for rb in rbs))
        return E.linear_small_m(t_emb, w, b, silu_in=True)          # (Bc, sum C_out)

    @torch.no_grad()
    def precompute_time_embedding(self, timesteps):
        """time_embed MLP + every ResBlock's emb_proj for these timesteps, as one (Bc, sum C_out) fp32 table.  It depends on
        t only, so a sampler computes it once per schedule step and passes it as `timesteps=` (step-invariant hoisting,
        SURVEY 8(f)-3); forward() accepts either the LongTensor of the reference or this object."""
        return TimeEmb(self._embeddings(timesteps))

    def _emb_pointers(self, allp):
        offs, o = {}, 0
        for rb in self._resblocks():
            offs[id(rb)] = allp.data_ptr() + 4 * o
            o += rb.emb_proj.weight.shape[0]
        assert o == allp.shape[1], "time-embedding table does not match this UNet"
        return allp, offs, allp.shape[1]

    @staticmethod
    def _run_stage(mods, h, emb, ldrb, ctx, out):
        n = len(mods)
        for i, m in enumerate(mods):
            dst = out if i == n - 1 else None
            if isinstance(m, ResBlock):
                h = m.run(h, emb[id(m)], ldrb, out=dst)
            elif isinstance(m, SpatialTransformer):
                h = m.run(h, ctx, out=dst)
            else:
                h = m.run(h, out=dst)
        return h

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None):
        return self.forward_into(x, timesteps, context, None)

    @torch.no_grad()
    def forward_into(self, x, timesteps, context, eps_out):
        """forward() whose last conv writes into `eps_out`, a contiguous fp32 (n, 4, h, w) tensor (e.g. one half of a
        sampler's CFG noise-prediction buffer), instead of a new tensor."""
        if isinstance(context, list):
            context = context[0]
        st = self._begin(x, timesteps)
        return self._finish(st, context, eps_out)

    @torch.no_grad()
    def cfg_prefix(self, x, timesteps):
        """The part of forward() that does not see the text context: conv_in, the first ResBlock and the first
        SpatialTransformer up to and including its self-attention.  The two halves of a classifier-free-guidance batch
        share latent, timestep and image conditioning (LatentDiffusionInpainting.py:179-193) and differ only in the
        prompt, so a sampler computes this once and continues it twice: `forward_from_prefix(p, ctx_uncond)` and
        `forward_from_prefix(p.branch(), ctx_cond)`.  Same kernels on the same shapes: the bits of forward()."""
        st = self._begin(x, timesteps)
        first = self.input_blocks[1]
        assert isinstance(first[0], ResBlock) and isinstance(first[1], SpatialTransformer), "unexpected first stage"
        st.a1 = first[0].run(st.h, st.emb[id(first[0])], st.ldrb)
        st.t1 = first[1].run_prefix(st.a1)
        return st

    @torch.no_grad()
    def forward_from_prefix(self, st, context, eps_out=None):
        """eps_out: optional contiguous fp32 (n, 4, h, w) tensor (e.g. one half of a CFG eps buffer) the last conv
        writes into directly."""
        if isinstance(context, list):
            context = context[0]
        return self._finish(st, context, eps_out)

    def _begin(self, x, timesteps):
        """Geometry, the concat buffers of the skip connections, the timestep-embedding table and conv_in."""
        x = x.contiguous().float()
        n, _, hh, ww = x.shape
        dev = x.device
        st = _FwdState()
        st.keep, st.emb, st.ldrb = self._emb_pointers(timesteps.table if isinstance(timesteps, TimeEmb) else self._embeddings(timesteps))
        if st.keep.shape[0] != n:
            raise E.L.SdmiError("Unet.forward: %d timesteps for a batch of %d" % (st.keep.shape[0], n))
        st.n, st.dev = n, dev

        # ---- geometry + concat plan: skip k is consumed by output block (K-1-k) next to an `h` of cat_c1[k] channels
        nin = len(self.input_blocks)
        skip_c, skip_hw = [], []
        ch_h, ch_w = hh, ww
        for blk in self.input_blocks:
            if isinstance(blk, nn.Conv2d):
                c = blk.out_channels
            elif isinstance(blk, Downsample):
                c = blk.conv.out_channels
                ch_h, ch_w = (ch_h + 2 - 3) // 2 + 1, (ch_w + 2 - 3) // 2 + 1
            else:
                c = blk[0].conv1.out_channels
            skip_c.append(c)
            skip_hw.append((ch_h, ch_w))
        st.skip_c, st.skip_hw = skip_c, skip_hw
        st.cat = self._new_cat(st)
        blk = self.input_blocks[0]
        assert isinstance(blk, nn.Conv2d)
        dst = st.cat[0].cols(st.cat[0].c - skip_c[0], skip_c[0])
        cs = None
        if E.USE_COLSTATS and E.CONV_IN_COLSTATS and (hh * ww) % 64 == 0 and hh * ww >= 2304:
            # per-channel statistics for the two GroupNorms that read this tensor (first ResBlock; last output block via the skip)
            cs = torch.empty((n * hh * ww // 32, blk.out_channels, 2), dtype=torch.float32, device=dev)
            dst.stats[dst.off] = (cs, hh * ww // 32, blk.out_channels)
        else:
            dst.stats.pop(dst.off, None)
        E.L.call("sdmi_conv3x3_in_ex", x.data_ptr(), n, x.shape[1], hh, ww, blk.weight.data_ptr(), blk.bias.data_ptr(),
                 blk.out_channels, dst.ptr, dst.ld, E._ptr(cs), E._stream())
        st.h = dst
        st.a1 = st.t1 = None
        return st

    def _new_cat(self, st):
        nin = len(self.input_blocks)
        cat = [None] * nin
        for j, blk in enumerate(self.output_blocks):
            k = nin - 1 - j
            c1 = blk[0].norm1.num_channels - st.skip_c[k]
            sh, sw = st.skip_hw[k]
            cat[k] = E.new_act(st.n, sh, sw, c1 + st.skip_c[k], st.dev)
        return cat

    def _finish(self, st, context, eps_out=None):
        if context is None:
            raise E.L.SdmiError("Unet.forward: context=None - every cross-attention's to_k / to_v is Linear(1024, inner) and "
                                "cannot project the tokens themselves (the reference fails in to_k for the same call)")
        ctx = context if isinstance(context, ContextKV) else self.precompute_context(context)
        if ctx._shape[0] != st.n:
            raise E.L.SdmiError("Unet.forward: context holds %d samples, x holds %d" % (ctx._shape[0], st.n))
        emb, ldrb, cat, skip_c = st.emb, st.ldrb, st.cat, st.skip_c
        nin = len(self.input_blocks)
        c1_of = lambda k: cat[k].c - skip_c[k]
        # ---- down path: block i writes its output into the upper half of cat[i]
        h = st.h
        for i, blk in enumerate(self.input_blocks):
            if i == 0:
                continue                                   # conv_in: done by _begin
            dst = cat[i].cols(c1_of(i), skip_c[i])
            if isinstance(blk, Downsample):
                h = blk.run(h, out=dst)
            elif i == 1 and st.t1 is not None:             # continue the shared context-free prefix
                h = blk[1].run(st.a1, ctx, out=dst, prefix=st.t1)
            else:
                h = self._run_stage(blk, h, emb, ldrb, ctx, dst)
        # ---- middle: result goes into the lower half of the deepest concat buffer
        h = self.middle_block[0].run(h, emb[id(self.middle_block[0])], ldrb)
        h = self.middle_block[1].run(h, ctx)
        h = self.middle_block[2].run(h, emb[id(self.middle_block[2])], ldrb, out=cat[nin - 1].cols(0, c1_of(nin - 1)))
        # ---- up path
        nout = len(self.output_blocks)
        for j, blk in enumerate(self.output_blocks):
            k = nin - 1 - j
            dst = cat[k - 1].cols(0, c1_of(k - 1)) if j + 1 < nout else None
            h = self._run_stage(blk, cat[k], emb, ldrb, ctx, dst)
        gn, conv = self.out[0], self.out[2]
        h = E.groupnorm(h, gn.weight, gn.bias, gn.eps, silu=True)
        return E.conv3x3(h, conv.weight, conv.bias, out_nchw=eps_out if eps_out is not None else True)


class _FwdState(object):
    """What forward() has built before the text context is needed (see Unet.cfg_prefix)."""

    def branch(self, unet):
        """A second continuation of the same prefix: its own concat buffers (the up path writes into them) holding a copy
        of the one skip tensor computed so far; the first ResBlock's output and the token stream are only read."""
        o = _FwdState()
        o.__dict__.update(self.__dict__)
        o.cat = unet._new_cat(self)
        c0, k0 = self.cat[0], self.skip_c[0]
        o.cat[0].buf[..., o.cat[0].c - k0:o.cat[0].c].copy_(c0.buf[..., c0.c - k0:c0.c])
        o.h = o.cat[0].cols(o.cat[0].c - k0, k0)
        return o

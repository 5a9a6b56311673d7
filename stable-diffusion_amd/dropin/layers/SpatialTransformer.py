"""Drop-in for the reference's layers/SpatialTransformer.py: same module tree and parameter names
(SURVEY.md 8(b)); tokens stay (B, HW, C) bf16 views end to end, q/k/v projections are fused into one GEMM,
attention is the flash kernel of csrc/attention.hip reading heads in place, residual adds / biases / GEGLU live in
GEMM epilogues.  No xformers, no einops, no permutes."""
import torch
from torch import nn

from stable_diffusion_amd import engine as E


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class ContextKV(object):
    """Pre-projected text context: the to_k|to_v outputs of every cross-attention layer, computed by ONE stacked GEMM
    (they depend only on the prompt, not on the latent or the timestep: 32 Linears per UNet step in the reference,
    layers/SpatialTransformer.py:73-74).  `kv[id(attn_module)]` is that layer's (B, 77, 2*inner) view."""

    def __init__(self, attn_modules, context):
        self._modules = list(attn_modules)
        self._ws = tuple(w for m in self._modules for w in (m.to_k.weight, m.to_v.weight))
        self._ntot = sum(w.shape[0] for w in self._ws)
        self._allkv = None
        self.update(context)
        self.kv, off = {}, 0
        for m in self._modules:
            n2 = m.to_k.weight.shape[0] + m.to_v.weight.shape[0]
            self.kv[id(m)] = self._allkv.cols(off, n2)
            off += n2

    def update(self, context):
        """Project a new prompt INTO THE SAME BUFFERS (same batch and token count): a HIP graph captured over a forward
        that read this object stays valid, which is how the sampler reuses one captured step across images."""
        ctx = context if isinstance(context, E.Act) else E.tokens_from_f32(context)
        if self._allkv is not None:
            assert (ctx.n, ctx.h * ctx.w) == self._shape, "ContextKV.update: batch / token count changed"
        self._shape = (ctx.n, ctx.h * ctx.w)
        self._allkv = E.gemm(ctx, E.packed_linear(self._ws), self._ntot, out=self._allkv)
        return self


def _as_ctx(context):
    if context is None or isinstance(context, (E.Act, ContextKV)):
        return context
    return E.tokens_from_f32(context)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def run(self, a, ln=None):
        """ln: the LayerNorm module in front of this layer, folded into the GEMM when `a` carries row statistics."""
        w = self.proj.weight
        if ln is not None:
            return E.gemm_ln(a, ln, (w,), bias=self.proj.bias, geglu=True)
        return E.gemm(a, E.packed_linear((w,), geglu=True), w.shape[0], bias=self.proj.bias, geglu=True)

    def forward(self, x):
        return _tokens_out(self.run(_tokens_in(x)), x)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, int(dim * mult)), nn.Dropout(dropout), nn.Linear(int(dim * mult), dim_out or dim))

    def run(self, a, residual=None, ln=None):
        g = self.net[0].run(a, ln)
        lin = self.net[2]
        return E.gemm(g, E.packed_linear((lin.weight,)), lin.weight.shape[0], bias=lin.bias, residual=residual)

    def forward(self, x):
        return _tokens_out(self.run(_tokens_in(x)), x)


def _tokens_in(x):
    return x if isinstance(x, E.Act) else E.tokens_from_f32(x)


def _tokens_out(a, like):
    if isinstance(like, E.Act):
        return a
    out = E.to_nchw(E.Act(a.buf, a.n, a.hw, 1, a.c, a.off))     # (B, C, T, 1) fp32
    return out[:, :, :, 0].transpose(1, 2).contiguous()


class MemoryEfficientCrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("head_dim 64 only (the reference hard-codes inner_dim = 64*heads)")
        self.heads, self.dim_head = heads, dim_head
        inner = 64 * heads
        context_dim = context_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.attention_op = None

    def run(self, a, context=None, residual=None, ln=None, rowstats=False):
        """ln: the LayerNorm module in front of this layer (BasicTransformerBlock.norm1/2): `a` is then the UN-normalised
        token stream and the LayerNorm is folded into the q (|k|v) projection.  rowstats: leave per-row statistics of the
        result for the LayerNorm that follows."""
        inner = 64 * self.heads
        # the query projection carries 1/8 * log2(e), folded into to_q before its rounding to 16 bits: scores leave the
        # QK^T MFMAs in the log2 domain (sdmi_attention_log2q)
        qs = lambda ws: (E.Q_LOG2_SCALE,) + (None,) * (len(ws) - 1)
        proj = (lambda ws: E.gemm_ln(a, ln, ws, scales=qs(ws))) if ln is not None else \
               (lambda ws: E.gemm(a, E.packed_linear(ws, scales=qs(ws)), sum(w.shape[0] for w in ws)))
        if context is None:          # self-attention: one fused q|k|v GEMM
            if self.to_k.weight.shape[1] != a.c:
                raise E.L.SdmiError("attention without context: to_k expects %d channels, the tokens have %d (the reference "
                                    "fails in to_k as well)" % (self.to_k.weight.shape[1], a.c))
            qkv = proj((self.to_q.weight, self.to_k.weight, self.to_v.weight))
            q, k, v = qkv.cols(0, inner), qkv.cols(inner, inner), qkv.cols(2 * inner, inner)
            o = None
        else:
            if isinstance(context, ContextKV):
                kv = context.kv[id(self)]
            else:
                kv = E.gemm(context, E.packed_linear((self.to_k.weight, self.to_v.weight)), 2 * inner)
            k, v = kv.cols(0, inner), kv.cols(inner, inner)
            # LayerNorm + to_q inside the attention kernel where the engine offers it (one launch less)
            o = E.cross_attention_lnq(a, ln, self.to_q.weight, k, v, self.heads) if ln is not None else None
            q = proj((self.to_q.weight,)) if o is None else None
        if o is None:
            o = E.attention(q, k, v, self.heads, log2q=True)
        lin = self.to_out[0]
        return E.gemm(o, E.packed_linear((lin.weight,)), lin.weight.shape[0], bias=lin.bias, residual=residual, rowstats=rowstats)

    def forward(self, x, context=None, mask=None):
        if mask is not None:
            raise NotImplementedError
        return _tokens_out(self.run(_tokens_in(x), _as_ctx(context)), x)


class MemoryEfficientAttnBlock(nn.Module):
    """VAE mid-block single-head attention, d = C (reference :98-161)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.attention_op = None

    def run(self, a, out=None):
        c = self.in_channels
        xn = E.groupnorm(a, self.norm.weight, self.norm.bias, self.norm.eps, silu=False)
        if c == 512 or c == 64:
            # flash path: one fused q|k|v projection with the softmax scale * log2(e) folded into q's weight and bias, then
            # the single-head kernel (d = 512: sdmi_attention_d512; d = 64: the UNet's kernel with one head)
            sc = (float(c) ** -0.5 * 1.4426950408889634, None, None)
            ws, bs = (self.q.weight, self.k.weight, self.v.weight), (self.q.bias, self.k.bias, self.v.bias)
            qkv = E.gemm(xn, E.packed_linear(ws, scales=sc), 3 * c, bias=E.stacked_f32(bs, scales=sc))
            q, k, v = qkv.cols(0, c), qkv.cols(c, c), qkv.cols(2 * c, c)
            o = E.attention_d512(q, k, v, log2q=True) if c == 512 else E.attention(q, k, v, 1, log2q=True)
            return E.gemm(o, E.packed_linear((self.proj_out.weight,)), c, bias=self.proj_out.bias, residual=a, out=out)
        raise E.L.SdmiError("MemoryEfficientAttnBlock: one head of %d channels has no kernel (512, the reference's only "
                            "configuration, and 64 are built)" % c)

    def forward(self, x):
        return E.to_nchw(self.run(E.from_nchw(x)))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, context_dim=None):
        super().__init__()
        self.dim, self.n_heads, self.context_dim = dim, n_heads, context_dim
        self.attn1 = MemoryEfficientCrossAttention(query_dim=dim, heads=n_heads, context_dim=None)
        self.ff = FeedForward(dim)
        self.attn2 = MemoryEfficientCrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    # Each LayerNorm is folded into the GEMM that consumes it (E.gemm_ln): the GEMM that produced the token stream left
    # per-row (sum, sumsq), the consumer multiplies the un-normalised rows by gamma-scaled weights and corrects per row in
    # its epilogue.  No LayerNorm launch, no normalised copy of the stream.
    def run_attn1(self, t):
        """The context-free head of the block: x + attn1(LN(x))."""
        return self.attn1.run(t, None, residual=t, ln=self.norm1, rowstats=True)

    def run(self, t, context=None, taps=None, after_attn1=False):
        if not after_attn1:
            t = self.run_attn1(t)
        if taps is not None:
            taps["attn1"] = t
        if context is None:     # cross-attention degenerates to self-attention over LN(x) (reference :70-72)
            t = self.attn2.run(E.layernorm(t, self.norm2.weight, self.norm2.bias, self.norm2.eps), None, residual=t, rowstats=True)
        else:
            t = self.attn2.run(t, context, residual=t, ln=self.norm2, rowstats=True)
        if taps is not None:
            taps["attn2"] = t
        t = self.ff.run(t, residual=t, ln=self.norm3)
        if taps is not None:
            taps["ff"] = t
        return t

    def forward(self, x, context=None):
        return _tokens_out(self.run(_tokens_in(x), _as_ctx(context)), x)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads):
        super().__init__()
        self.in_channels, self.n_heads = in_channels, n_heads
        inner = n_heads * 64
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, context_dim=1024)])
        self.proj_out = zero_module(nn.Linear(in_channels, inner))

    def run_prefix(self, a):
        """Everything before the text context enters (GroupNorm, proj_in, LayerNorm, self-attention + residual): the token
        stream that `run(a, context, prefix=...)` continues from.  Depends on `a` only, so the two halves of a CFG batch
        (same latent, same timestep, different prompt) can share it."""
        xn = E.groupnorm(a, self.norm.weight, self.norm.bias, self.norm.eps, silu=False)
        t = E.gemm(xn, E.packed_linear((self.proj_in.weight,)), self.proj_in.weight.shape[0], bias=self.proj_in.bias, rowstats=True)
        return self.transformer_blocks[0].run_attn1(t)

    def run(self, a, context=None, out=None, taps=None, prefix=None):
        if prefix is None:
            xn = E.groupnorm(a, self.norm.weight, self.norm.bias, self.norm.eps, silu=False)
            t = E.gemm(xn, E.packed_linear((self.proj_in.weight,)), self.proj_in.weight.shape[0], bias=self.proj_in.bias, rowstats=True)
            t = self.transformer_blocks[0].run(t, context, taps)
        else:
            t = self.transformer_blocks[0].run(prefix, context, taps, after_attn1=True)
        return E.gemm(t, E.packed_linear((self.proj_out.weight,)), self.proj_out.weight.shape[0], bias=self.proj_out.bias,
                      residual=a, out=out, colstats=True)

    def forward(self, x, t_emb=None, context=None):
        if isinstance(context, list):
            context = context[0]
        return E.to_nchw(self.run(E.from_nchw(x), _as_ctx(context)))

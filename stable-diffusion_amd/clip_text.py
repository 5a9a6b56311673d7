"""The text encoder of the reference's conditioning path on libsdmi355.so: `LatentDiffusionInpainting.get_conditional`
(LatentDiffusionInpainting.py:232-245) over the text tower that open_clip's "ViT-H-14" builds (:121-123): token +
positional embedding, all but the last of the 24 pre-LN transformer blocks (width 1024, 16 heads of 64, MLP 4096 with
exact GELU, additive causal mask), then ln_final.  Output (B, 77, 1024) fp32: the `context` of Unet.forward.

Parameter names are open_clip's (`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.{i}.ln_1 /
attn.in_proj_weight / attn.in_proj_bias / attn.out_proj / ln_2 / mlp.c_fc / mlp.c_proj`, `ln_final`, `text_projection`,
`logit_scale`), so the `cond_stage_model.*` entries of the reference's checkpoint remap (:257) load with
load_state_dict().  open_clip itself (mlfoundations/open_clip, version not pinned by the reference) and its BPE vocabulary
are not in this image: callers pass token ids (what `open_clip.tokenize` returns) or text plus the vocabulary file
(stable_diffusion_amd/tokenizer.py restates the BPE tokenizer), and parity is against the oracle's
restatement of the published block (clip_text_encode in the test oracle) - parity unpinned, see DESIGN.md."""
import torch
from torch import nn

from . import _lib as L
from . import engine as E


class _MultiheadAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight = [q; k; v] stacked)."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * width))
        self.out_proj = nn.Linear(width, width)


class _Mlp(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.c_fc = nn.Linear(width, hidden)
        self.gelu = nn.GELU()
        self.c_proj = nn.Linear(hidden, width)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _MultiheadAttention(width)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = _Mlp(width, int(width * mlp_ratio))

    def run(self, x):
        """x: Act (B, 1, T, width) residual stream; x + attn(ln_1(x)) then x + mlp(ln_2(x)), causal."""
        w = x.c
        h = E.layernorm(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        qkv = E.gemm(h, E.packed_linear((self.attn.in_proj_weight,)), 3 * w, bias=self.attn.in_proj_bias)
        o = E.new_act(x.n, x.h, x.w, w, x.device)
        q, k, v = qkv.cols(0, w), qkv.cols(w, w), qkv.cols(2 * w, w)
        L.call("sdmi_attention_causal", q.ptr, q.ld, q.hw * q.ld, k.ptr, k.ld, k.hw * k.ld, v.ptr, v.ld, v.hw * v.ld,
               o.ptr, o.ld, o.hw * o.ld, x.n, self.heads, x.hw, 0.125, E._stream())
        op = self.attn.out_proj
        x = E.gemm(o, E.packed_linear((op.weight,)), w, bias=op.bias, residual=x)
        h = E.layernorm(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        f = E.gemm(h, E.packed_linear((self.mlp.c_fc.weight,)), self.mlp.c_fc.weight.shape[0], bias=self.mlp.c_fc.bias)
        L.call("sdmi_gelu_rows", f.ptr, f.ld, f.n * f.hw, f.c, E._stream())
        return E.gemm(f, E.packed_linear((self.mlp.c_proj.weight,)), w, bias=self.mlp.c_proj.bias, residual=x)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])


class ClipTextEncoder(nn.Module):
    """`cond_stage_model` of the reference after `del .visual` (LatentDiffusionInpainting.py:121-123)."""

    def __init__(self, vocab_size=49408, context_length=77, width=1024, heads=16, layers=24):
        super().__init__()
        if width != 64 * heads:
            raise NotImplementedError("head_dim 64 only (ViT-H-14 text tower: width 1024, 16 heads)")
        self.context_length, self.vocab_size, self.width = context_length, vocab_size, width
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, width))      # unused by get_conditional, kept for the keys
        self.logit_scale = nn.Parameter(torch.ones([]))
        mask = torch.empty(context_length, context_length).fill_(float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)           # the kernel applies it as "keys 0..i"

    @torch.no_grad()
    def get_conditional(self, tokens, tokenizer=None):
        """tokens: LongTensor (B, context_length) of BPE ids (open_clip.tokenize's output), or - as the reference's
        get_conditional(text) takes it (:232-233) - a string / list of strings, tokenized on the host by
        stable_diffusion_amd.tokenizer (needs the CLIP BPE vocabulary file: `tokenizer=` or env SDMI_CLIP_BPE).  Returns
        (B, T, width) fp32: ln_final of the penultimate block's output, as the reference computes it (:236-245)."""
        dev = self.positional_embedding.device
        if isinstance(tokens, str) or (isinstance(tokens, (list, tuple)) and tokens and isinstance(tokens[0], str)):
            from . import tokenizer as TK
            tokens = TK.tokenize(tokens, self.context_length, tokenizer)
        tokens = torch.as_tensor(tokens, dtype=torch.long, device=dev).contiguous()
        if tokens.dim() != 2 or tokens.shape[1] != self.context_length:
            raise L.SdmiError("get_conditional: tokens must be (B, %d), got %s" % (self.context_length, tuple(tokens.shape)))
        if int(tokens.min()) < 0 or int(tokens.max()) >= self.vocab_size:
            raise L.SdmiError("get_conditional: token id outside [0, %d)" % self.vocab_size)
        b, t = tokens.shape
        x = E.new_act(b, 1, t, self.width, dev)
        L.call("sdmi_embed_tokens", tokens.data_ptr(), E._f32c(self.token_embedding.weight, "token_embedding").data_ptr(),
               E._f32c(self.positional_embedding, "positional_embedding").data_ptr(), x.ptr, b * t, t, self.width,
               self.vocab_size, x.ld, E._stream())
        blocks = self.transformer.resblocks
        for i, r in enumerate(blocks):
            if i == len(blocks) - 1:
                break                                   # the reference stops before the last block (penultimate layer)
            x = r.run(x)
        x = E.layernorm(x, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps)
        return x.buf.float().reshape(b, t, self.width)

    forward = get_conditional

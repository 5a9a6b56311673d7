"""Host-side BPE tokenizer of the reference's conditioning path: `open_clip.tokenize(text)` at
LatentDiffusionInpainting.py:233 (the CLIP byte-pair encoder open_clip ships as `SimpleTokenizer`: lower-cased text, a byte
-> printable-unicode map, merges read from `bpe_simple_vocab_16e6.txt.gz`, `<start_of_text>` / `<end_of_text>` around every
prompt, zero padding to 77 ids, truncation that keeps `<end_of_text>` last).

The vocabulary file is NOT in this repository or image (open_clip is not installed and there is no network): pass its path
(`SimpleTokenizer(bpe_path=...)`, or env SDMI_CLIP_BPE).  Without it this module raises: there is no built-in fallback
vocabulary.  Parity unpinned: the algorithm is restated from the published tokenizer and tested on a synthetic merge table
(tests/test_tokenizer.py); with the real file the ids are the ones `ClipTextEncoder.get_conditional` expects.  `ftfy` (used by
open_clip to repair mojibake before tokenizing) is not installed here: text is HTML-unescaped and whitespace-cleaned only, which
is what `ftfy.fix_text` leaves unchanged for well-formed input."""
import gzip
import html
import os
from functools import lru_cache

import regex as re

CONTEXT_LENGTH = 77


@lru_cache()
def bytes_to_unicode():
    """Every byte as a printable unicode character (the BPE works on strings and must not see whitespace / control bytes)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return set(zip(word[:-1], word[1:]))


def _clean(text):
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer(object):
    def __init__(self, bpe_path=None, vocab_size=49408):
        bpe_path = bpe_path or os.environ.get("SDMI_CLIP_BPE")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("CLIP BPE vocabulary not found (%r): pass bpe_path= or set SDMI_CLIP_BPE to open_clip's "
                                    "bpe_simple_vocab_16e6.txt.gz; it is not shipped with this repository" % (bpe_path,))
        self.byte_encoder = bytes_to_unicode()
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        n_merges = vocab_size - 256 - 256 - 2              # 48894 for the 49408-entry CLIP vocabulary
        merges = [tuple(m.split()) for m in lines[1:1 + n_merges] if m.strip()]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<start_of_text>", "<end_of_text>"]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        self.pat = re.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        ids = []
        for tok in re.findall(self.pat, _clean(text).lower()):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, ids):
        inv = {v: k for k, v in self.byte_encoder.items()}
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts, context_length=CONTEXT_LENGTH):
        return tokenize(texts, context_length, self)


def tokenize(texts, context_length=CONTEXT_LENGTH, tokenizer=None):
    """`open_clip.tokenize`: (len(texts), context_length) int64 ids, `<start_of_text>` ids `<end_of_text>` then zeros; a prompt
    that does not fit is cut and its last id set to `<end_of_text>`."""
    import torch
    if isinstance(texts, str):
        texts = [texts]
    tok = tokenizer or SimpleTokenizer()
    out = torch.zeros((len(texts), context_length), dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [tok.sot] + tok.encode(t) + [tok.eot]
        if len(ids) > context_length:
            ids = ids[:context_length]
            ids[-1] = tok.eot
        out[i, :len(ids)] = torch.tensor(ids)
    return out

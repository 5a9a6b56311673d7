"""CPU: the host-side CLIP BPE tokenizer (stable_diffusion_amd/tokenizer.py, `open_clip.tokenize` at
LatentDiffusionInpainting.py:233) on a synthetic merge table - the real vocabulary file is not in the image (parity unpinned)."""
import gzip

import pytest
import torch

from stable_diffusion_amd import tokenizer as TK


@pytest.fixture
def tok(tmp_path):
    merges = ["l o", "lo w</w>", "e r</w>", "a n", "an d</w>"]
    p = tmp_path / "bpe.txt.gz"
    with gzip.open(p, "wb") as f:
        f.write(("#version: synthetic\n" + "\n".join(merges) + "\n").encode("utf-8"))
    return TK.SimpleTokenizer(str(p), vocab_size=512 + len(merges) + 2)


def test_missing_vocabulary_raises(monkeypatch):
    monkeypatch.delenv("SDMI_CLIP_BPE", raising=False)
    with pytest.raises(FileNotFoundError, match="bpe_simple_vocab"):
        TK.SimpleTokenizer()


def test_bpe_merges_and_layout(tok):
    enc = tok.encoder
    b2u = TK.bytes_to_unicode()
    assert len(b2u) == 256 and len(set(b2u.values())) == 256
    assert tok.sot == 512 + 5 and tok.eot == 512 + 6                       # the specials close the vocabulary
    assert tok.encode("low") == [512 + 1]                                  # l o w</w> -> lo w</w> -> low</w>
    assert tok.encode("lower") == [512 + 0, enc["w"], 512 + 2]             # lo | w | er</w>  ("lo w</w>" needs the word end)
    assert tok.encode("LOW  and\nlower") == [512 + 1, 512 + 4, 512 + 0, enc["w"], 512 + 2]   # lower-cased, whitespace collapsed
    assert tok.encode("a&amp;b") == [enc["a</w>"], enc["&</w>"], enc["b</w>"]]                # HTML unescaped, punctuation split
    assert tok.decode(tok.encode("low and lower")).strip() == "low and lower"


def test_tokenize_shape_padding_truncation(tok):
    ids = TK.tokenize(["low", "and " * 100], tokenizer=tok)
    assert ids.shape == (2, 77) and ids.dtype == torch.long
    assert ids[0].tolist()[:3] == [tok.sot, 512 + 1, tok.eot] and int(ids[0, 3:].abs().sum()) == 0
    assert int(ids[1, 0]) == tok.sot and int(ids[1, -1]) == tok.eot and bool((ids[1, 1:-1] == 512 + 4).all())
    assert TK.tokenize("low", tokenizer=tok).shape == (1, 77)
    assert torch.equal(tok(["low"]), TK.tokenize(["low"], tokenizer=tok))

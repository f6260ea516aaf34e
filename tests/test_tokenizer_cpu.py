"""f4 host side: ``turbodiffusion_amd.tokenizer.HuggingfaceTokenizer`` against the reference's class of the same name
(rcm/utils/umt5.py:58-98, imported live through oracle/ref_harness.py where /root/reference exists) on a T5-style
vocabulary trained here (Unigram + Metaspace + ``$A </s>``; ``<pad>`` = 0, ``</s>`` = 1, ``<unk>`` = 2 — the layout of
``google/umt5-xxl``, which cannot be downloaded on this box), and against committed expectations everywhere."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402
from turbodiffusion_amd import tokenizer as T  # noqa: E402

CORPUS = [
    "a stylish woman walks down a tokyo street filled with warm glowing neon and animated city signage",
    "she wears a black leather jacket, a long red dress, and black boots, and carries a black purse",
    "the street is damp and reflective, creating a mirror effect of the colorful lights",
    "many pedestrians walk about", "a cat surfing a wave at sunset, cinematic lighting, 4k",
    "an astronaut riding a horse on mars", "drone shot over a snowy mountain range at dawn",
    "close-up of a hummingbird drinking nectar from a red flower in slow motion",
] * 4

PROMPTS = [
    "A stylish woman walks down a Tokyo street",
    "  a cat   surfing &amp;amp; a wave\n at sunset  ",
    "",
    "drone_shot over a snowy mountain range at dawn!!! " * 12,     # longer than seq_len: truncated, ends with </s>
    "naïve café — unknown glyphs ☃",
]


@pytest.fixture(scope="module")
def vocab_dir(tmp_path_factory):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    tok = Tokenizer(models.Unigram())
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.decoder = decoders.Metaspace()
    tok.train_from_iterator(CORPUS, trainers.UnigramTrainer(vocab_size=120, special_tokens=["<pad>", "</s>", "<unk>"],
                                                           unk_token="<unk>", show_progress=False))
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", pair="$A </s> $B </s>", special_tokens=[("</s>", 1)])
    d = tmp_path_factory.mktemp("vocab")
    tok.save(str(d / "tokenizer.json"))
    return str(d)


def test_cleaning_functions():
    assert T.basic_clean("  a &amp;lt;b&amp;gt; c ") == "a <b> c"            # html.unescape twice, umt5.py:35
    assert T.whitespace_clean(" a \n\t b  c ") == "a b c"
    assert T.canonicalize("Hello_World, it's  ME!") == "hello world its me"
    assert T.canonicalize("a.b, c.d", keep_punctuation_exact_string=", ") == "ab, cd"
    if rh.available():
        u = rh.load_aux("utils.umt5")
        for s in PROMPTS + ["x &quot;y&quot; _z_", "A.B, C.D"]:
            assert T.basic_clean(s) == u.basic_clean(s) and T.whitespace_clean(s) == u.whitespace_clean(s)
            assert T.canonicalize(s) == u.canonicalize(s)
            assert T.canonicalize(s, ", ") == u.canonicalize(s, ", ")


@pytest.mark.parametrize("clean", [None, "whitespace", "lower", "canonicalize"])
def test_ids_and_mask_layout(vocab_dir, clean):
    tok = T.HuggingfaceTokenizer(vocab_dir, seq_len=24, clean=clean)
    ids, mask = tok(PROMPTS, return_mask=True, add_special_tokens=True)
    assert ids.shape == mask.shape == (len(PROMPTS), 24) and ids.dtype == mask.dtype == torch.long
    for row, m in zip(ids, mask):
        n = int(m.sum())
        assert 1 <= n <= 24 and bool(m[:n].all()) and not bool(m[n:].any())      # right padded
        assert int(row[n - 1]) == 1 and bool((row[n:] == 0).all())               # ... </s> <pad> <pad>
    assert int(mask[2].sum()) == 1                                               # empty prompt: </s> only
    assert int(mask[3].sum()) == 24                                              # truncated to seq_len, </s> kept
    assert torch.equal(tok(PROMPTS[0]), ids[:1])                                 # a str is one prompt; ids only by default
    assert T.prompt_lengths(mask) == mask.sum(1).tolist()
    free = T.HuggingfaceTokenizer(vocab_dir)                                     # no seq_len: padded to the longest, no cut
    fi, fm = free(PROMPTS, return_mask=True)
    assert fi.shape[1] == int(fm.sum(1).max()) > 24
    no_eos = tok(PROMPTS[0], add_special_tokens=False)
    assert int((no_eos == 1).sum()) == 0
    with pytest.raises(AssertionError):
        T.HuggingfaceTokenizer(vocab_dir, clean="upper")
    with pytest.raises(FileNotFoundError):
        T.HuggingfaceTokenizer("google/umt5-xxl")                                # hub ids are not resolved: no network
    with pytest.raises(TypeError):
        tok(PROMPTS[0], padding="longest")


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("clean,seq_len", [("whitespace", 24), ("whitespace", 512), ("lower", 16), ("canonicalize", 24), (None, 24)])
def test_against_the_reference_class(vocab_dir, clean, seq_len):
    """The reference's HuggingfaceTokenizer reading the same vocabulary through AutoTokenizer (a saved
    PreTrainedTokenizerFast directory), called exactly as UMT5EncoderModel calls it (umt5.py:499, :504)."""
    from transformers import PreTrainedTokenizerFast
    fast = PreTrainedTokenizerFast(tokenizer_file=os.path.join(vocab_dir, "tokenizer.json"), pad_token="<pad>", eos_token="</s>",
                                   unk_token="<unk>")
    hf_dir = os.path.join(vocab_dir, "hf")
    fast.save_pretrained(hf_dir)
    u = rh.load_aux("utils.umt5")
    ref = u.HuggingfaceTokenizer(name=hf_dir, seq_len=seq_len, clean=clean)
    mine = T.HuggingfaceTokenizer(name=hf_dir, seq_len=seq_len, clean=clean)
    assert mine.vocab_size == ref.vocab_size
    r_ids, r_mask = ref(PROMPTS, return_mask=True, add_special_tokens=True)
    m_ids, m_mask = mine(PROMPTS, return_mask=True, add_special_tokens=True)
    assert torch.equal(m_ids, r_ids) and torch.equal(m_mask, r_mask)
    assert torch.equal(mine(PROMPTS[1]), ref(PROMPTS[1]))

"""f4, host side: prompt text -> the [B, seq_len] token ids and attention mask the umT5 encoder takes.

Reference: ``rcm/utils/umt5.py`` — ``basic_clean`` / ``whitespace_clean`` / ``canonicalize`` (:33-55) and ``HuggingfaceTokenizer``
(:58-98: ``AutoTokenizer.from_pretrained(name)``; with ``seq_len`` every prompt is truncated and right-padded to it; returns
``input_ids`` or ``(input_ids, attention_mask)``), driven by ``UMT5EncoderModel`` as ``HuggingfaceTokenizer(name=tokenizer_path,
seq_len=512, clean="whitespace")`` and ``tokenizer(texts, return_mask=True, add_special_tokens=True)`` (:499, :504).

Here the same class name, arguments and return values on the ``tokenizers`` library directly (the Rust core the reference's
``AutoTokenizer`` ends up in for ``google/umt5-xxl``: a ``tokenizer.json`` — Unigram model, Metaspace pre-tokeniser,
``$A </s>`` template): no ``transformers`` import, no hub access.  ``name`` is a local ``tokenizer.json`` or a directory that
holds one (there is no network on the serving box; the reference's default, the hub id ``google/umt5-xxl``, resolves to
exactly that file in the hub cache).  Truncation and padding are the library's own (`enable_truncation(max_length)` counts
the end-of-sequence token the template adds, `enable_padding(length=...)` pads on the right with the pad token), which is what
``padding="max_length", truncation=True, max_length=seq_len`` selects in the reference's call.  ``ftfy.fix_text`` (mojibake
repair, the first step of ``basic_clean``) is used when the package is importable; it is not part of this image, and without it
``basic_clean`` is ``html.unescape`` twice + ``strip`` — identical on text that is not mojibake.

CPU host code: pinned by ``tests/test_tokenizer_cpu.py`` to the reference's own class (imported live) on a vocabulary trained
in the test, including truncation at ``seq_len``, empty prompts, HTML entities and the three cleaning modes."""
from __future__ import annotations

import html
import os
import re
import string
from typing import List, Sequence, Union

import torch

try:                                     # optional, umt5.py:22
    import ftfy as _ftfy
except ImportError:                      # not in this image
    _ftfy = None


def basic_clean(text: str) -> str:
    """umt5.py:33-36"""
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text: str) -> str:
    """umt5.py:39-42"""
    return re.sub(r"\s+", " ", text).strip()


_PUNCT = str.maketrans("", "", string.punctuation)


def canonicalize(text: str, keep_punctuation_exact_string=None) -> str:
    """umt5.py:45-55"""
    text = text.replace("_", " ")
    if keep_punctuation_exact_string:
        text = keep_punctuation_exact_string.join(p.translate(_PUNCT) for p in text.split(keep_punctuation_exact_string))
    else:
        text = text.translate(_PUNCT)
    return re.sub(r"\s+", " ", text.lower()).strip()


def _tokenizer_file(name: str) -> str:
    if os.path.isdir(name):
        f = os.path.join(name, "tokenizer.json")
        if os.path.isfile(f):
            return f
        raise FileNotFoundError(f"{name}: no tokenizer.json in this directory")
    if os.path.isfile(name):
        return name
    raise FileNotFoundError(f"{name!r} is neither a tokenizer.json nor a directory holding one (hub ids are not resolved: "
                            "no network; point at the hub cache's snapshot directory)")


class HuggingfaceTokenizer:
    """Same constructor and call as the reference's (umt5.py:58-98).  ``pad_token`` names the padding token of the vocabulary
    (T5 family: ``<pad>``, id 0)."""

    def __init__(self, name, seq_len=None, clean=None, pad_token="<pad>", **kwargs):
        if clean not in (None, "whitespace", "lower", "canonicalize"):
            raise AssertionError(f"clean = {clean!r}")       # the reference asserts (umt5.py:60)
        from tokenizers import Tokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = Tokenizer.from_file(_tokenizer_file(name))
        self.vocab_size = self.tokenizer.get_vocab_size(with_added_tokens=False)
        self.pad_token = pad_token
        self.pad_id = self.tokenizer.token_to_id(pad_token)
        if self.pad_id is None:
            raise ValueError(f"{name}: the vocabulary has no {pad_token!r} token")
        if seq_len is not None:
            self.tokenizer.enable_truncation(max_length=int(seq_len))
            self.tokenizer.enable_padding(length=int(seq_len), pad_id=self.pad_id, pad_token=pad_token, direction="right")
        else:
            self.tokenizer.no_truncation()
            self.tokenizer.enable_padding(pad_id=self.pad_id, pad_token=pad_token, direction="right")   # to the longest

    def __call__(self, sequence: Union[str, Sequence[str]], **kwargs):
        return_mask = kwargs.pop("return_mask", False)
        add_special_tokens = kwargs.pop("add_special_tokens", True)
        if kwargs:
            raise TypeError(f"unsupported tokenizer arguments {sorted(kwargs)} (the reference's call sites pass return_mask and "
                            "add_special_tokens only, umt5.py:504)")
        if isinstance(sequence, str):
            sequence = [sequence]
        if self.clean:
            sequence = [self._clean(u) for u in sequence]
        enc = self.tokenizer.encode_batch(list(sequence), add_special_tokens=add_special_tokens)
        ids = torch.tensor([e.ids for e in enc], dtype=torch.long)
        if return_mask:
            return ids, torch.tensor([e.attention_mask for e in enc], dtype=torch.long)
        return ids

    def _clean(self, text: str) -> str:
        if self.clean == "whitespace":
            return whitespace_clean(basic_clean(text))
        if self.clean == "lower":
            return whitespace_clean(basic_clean(text)).lower()
        if self.clean == "canonicalize":
            return canonicalize(basic_clean(text))
        return text


def prompt_lengths(mask: torch.Tensor) -> List[int]:
    """``mask.gt(0).sum(dim=1)`` of umt5.py:507 as a host list"""
    return mask.gt(0).sum(dim=1).tolist()

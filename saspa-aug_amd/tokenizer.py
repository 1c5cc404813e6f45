"""CLIP tokenizer front-end.

The reference tokenises inside diffusers (`CLIPTokenizer`, 77 tokens, BOS 49406, EOS = PAD =
49407).  The BPE vocabulary (`vocab.json` + `merges.txt`) ships with the HF
checkpoint, which is not available offline, so:
  * `CLIPBPETokenizer(vocab_dir)` implements the published CLIP byte-pair encoding and is
    used whenever a checkpoint directory provides the two files;
  * `HashTokenizer` is a deterministic stand-in for synthetic-weight runs: every lower-cased
    word piece maps to a stable id in [0, vocab-2).  It exists so that the run_aug entrypoint
    is runnable end to end without a checkpoint; it is NOT a CLIP vocabulary."""
import json
import os
import re
import unicodedata
import zlib
from functools import lru_cache

import numpy as np


class HashTokenizer:
    def __init__(self, vocab=49408, max_len=77, pad_id=None):
        self.vocab, self.max_len = vocab, max_len
        self.bos, self.eos = vocab - 2, vocab - 1
        self.pad = self.eos if pad_id is None else pad_id

    def __call__(self, text, max_len=None):
        max_len = max_len or self.max_len
        words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", (text or "").lower())
        ids = [self.bos] + [zlib.crc32(w.encode()) % (self.vocab - 2) for w in words][: max_len - 2] + [self.eos]
        ids += [self.pad] * (max_len - len(ids))
        return np.asarray(ids, np.int64)[None]


@lru_cache()
def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class CLIPBPETokenizer:
    def __init__(self, vocab_dir, max_len=77, pad_id=None):
        with open(os.path.join(vocab_dir, "vocab.json"), encoding="utf-8") as f:
            self.encoder = json.load(f)
        with open(os.path.join(vocab_dir, "merges.txt"), encoding="utf-8") as f:
            merges = f.read().strip().split("\n")[1:]
        self.ranks = {tuple(m.split()): i for i, m in enumerate(merges)}
        self.byte_enc = _bytes_to_unicode()
        self.max_len = max_len
        self.bos, self.eos = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self.pad = self.eos if pad_id is None else pad_id      # HF CLIPTokenizer pads with EOS, OpenAI's clip.tokenize with 0
        # CLIP's pre-tokenisation pattern (openai/CLIP simple_tokenizer.py, transformers CLIPTokenizer): unicode letter /
        # number classes need the `regex` module -- `[a-z]+` would split accented sub-class names into punctuation tokens
        import regex
        self.pat = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                 regex.IGNORECASE)
        self.cache = {}

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        self.cache[token] = word
        return word

    def __call__(self, text, max_len=None):
        max_len = max_len or self.max_len
        # CLIPTokenizer's normaliser (transformers: NFC, whitespace runs -> one space, lower-case; accents kept).  The slow
        # tokenizer's optional ftfy.fix_text pass (mojibake / HTML-entity repair) is NOT reproduced: it is the identity on
        # the ASCII prompts of the reference's prompt files.
        text = unicodedata.normalize("NFC", text or "")
        text = re.sub(r"\s+", " ", text).strip().lower()
        unk = self.encoder.get("<|endoftext|>")
        ids = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_enc[b] for b in tok.encode("utf-8"))
            ids += [self.encoder.get(p, unk) for p in self._bpe(tok)]
        ids = [self.bos] + ids[: max_len - 2] + [self.eos]
        ids += [self.pad] * (max_len - len(ids))
        return np.asarray(ids, np.int64)[None]


def make_tokenizer(vocab_dir=None, vocab=49408, pad_id=None):
    if vocab_dir and os.path.exists(os.path.join(vocab_dir, "vocab.json")):
        return CLIPBPETokenizer(vocab_dir, pad_id=pad_id)
    return HashTokenizer(vocab, pad_id=pad_id)


# ---- BERT tokenizer of the BLIP-Diffusion Q-Former (subject category text, e.g. "bird") ----------------------
class BertHashTokenizer:
    """Stand-in for synthetic-weight runs ([CLS] words [SEP], no padding); NOT the bert-base-uncased vocabulary."""

    def __init__(self, vocab=30523):
        self.vocab = vocab
        self.cls, self.sep = (101, 102) if vocab > 1000 else (1, 2)

    def __call__(self, text):
        words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", (text or "").lower())
        lo = 1000 if self.vocab > 2000 else 3
        return np.asarray([self.cls] + [lo + zlib.crc32(w.encode()) % (self.vocab - lo) for w in words] + [self.sep], np.int64)[None]


class BertWordPieceTokenizer:
    """bert-base-uncased WordPiece from the checkpoint's vocab.txt (lower-case, punctuation split, greedy
    longest-match-first with ## continuation pieces)."""

    def __init__(self, vocab_file):
        with open(vocab_file, encoding="utf-8") as f:
            self.vocab = {tok.rstrip("\n"): i for i, tok in enumerate(f)}
        self.cls, self.sep, self.unk = self.vocab["[CLS]"], self.vocab["[SEP]"], self.vocab["[UNK]"]

    def _pieces(self, word):
        out, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = ("##" if start else "") + word[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            out.append(cur)
            start = end
        return out

    def __call__(self, text):
        ids = [self.cls]
        for w in re.findall(r"[a-z0-9]+|[^\sa-z0-9]", (text or "").lower()):
            ids += self._pieces(w)
        return np.asarray(ids + [self.sep], np.int64)[None]


def make_bert_tokenizer(vocab_dir=None, vocab=30523):
    if vocab_dir and os.path.exists(os.path.join(vocab_dir, "vocab.txt")):
        return BertWordPieceTokenizer(os.path.join(vocab_dir, "vocab.txt"))
    return BertHashTokenizer(vocab)

"""Byte-level BPE tokenizer producing the id rows `CLIP.encode_text` consumes.

Boundary input of the hot path, not a kernel: the reference's
`clip/simple_tokenizer.py:63-133` defines the vocabulary layout (256 byte symbols, 256
end-of-word byte symbols, 48,894 merges, <|startoftext|>=49406, <|endoftext|>=49407); this
is an independent implementation of that published algorithm.  Pinned bit-exactly by
tests/golden/tokens.npz (18 prompts run through the reference tokenizer).

The merges table (`bpe_simple_vocab_16e6.txt.gz`, OpenAI CLIP) is data that is not shipped
in this repo; it is looked up at $DBMM_BPE_PATH, next to this file, or in ~/.cache/clip.
`ftfy` is optional: without it only the (identity-on-ASCII) unicode fix-up is skipped.
"""
import gzip
import html
import os
import re as _stdre
from functools import lru_cache

try:
    import regex as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       _re.IGNORECASE)
except ImportError:  # pragma: no cover
    _re = None
    _PAT = None

try:
    import ftfy as _ftfy
except ImportError:
    _ftfy = None

N_MERGES = 49152 - 256 - 2
_BPE_NAME = "bpe_simple_vocab_16e6.txt.gz"


def default_bpe():
    cands = [os.environ.get("DBMM_BPE_PATH"), os.path.join(os.path.dirname(os.path.abspath(__file__)), _BPE_NAME),
             os.path.expanduser(os.path.join("~/.cache/clip", _BPE_NAME))]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError(
        f"{_BPE_NAME} not found (set DBMM_BPE_PATH or place it in ~/.cache/clip); it is the OpenAI CLIP merges "
        "table and is not shipped with this repository")


@lru_cache()
def byte_symbols():
    """byte value -> printable unicode symbol (GPT-2 convention: printable latin-1 bytes map
    to themselves, the other 68 bytes to code points 256, 257, ... in byte order)."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _vocab_order():
    """symbols in vocabulary order: kept bytes ascending, then the remapped ones (the order of
    the reference's bytes_to_unicode().values())."""
    t = byte_symbols()
    kept = [b for b in range(256) if ord(t[b]) < 256]
    rest = [b for b in range(256) if ord(t[b]) >= 256]
    return [t[b] for b in kept + rest]


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None):
        bpe_path = bpe_path or default_bpe()
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        base = _vocab_order()
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {s: i for i, s in enumerate(vocab)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.byte_encoder = byte_symbols()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self._cache = {"<|startoftext|>": ["<|startoftext|>"], "<|endoftext|>": ["<|endoftext|>"]}

    def _merge_word(self, token):
        """greedy lowest-rank pair merging of one pre-token (symbols + '</w>' on the last)."""
        if token in self._cache:
            return self._cache[token]
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (parts[i], parts[i + 1]), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    merged.append(parts[i] + parts[i + 1]); i += 2
                else:
                    merged.append(parts[i]); i += 1
            parts = merged
        self._cache[token] = parts
        return parts

    @staticmethod
    def _clean(text):
        if _ftfy is not None:
            text = _ftfy.fix_text(text)
        text = html.unescape(html.unescape(text)).strip()
        return _stdre.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text):
        if _PAT is None:
            raise RuntimeError("the `regex` package is required for tokenisation")
        ids = []
        for tok in _PAT.findall(self._clean(text)):
            sym = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(sym))
        return ids

    def decode(self, tokens):
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

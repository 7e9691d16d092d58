"""Caption -> CLIP token ids (host side of scope row 8f-1).

Mirrors sound_synthesis/modeling/codecs/text_codec/tokenize.py:Tokenize (get_tokens :59-69) and the
reference's clip.tokenize (modules/clip/clip.py:164-215) on top of a byte-level BPE with CLIP's
merge table (modules/clip/simple_tokenizer.py:61-135): lower-cased text, <|startoftext|> = 49406,
<|endoftext|> = 49407, padded with `pad_value` to `context_length`, truncated keeping the EOT.

The merge table (`bpe_simple_vocab_16e6.txt.gz`, OpenAI CLIP's public vocabulary) is not shipped with
this package: point `DIFFSOUND_BPE_PATH` (or the `bpe_path` argument) at the copy inside a checkout
of the reference (`…/sound_synthesis/modeling/modules/clip/`).
"""
import gzip
import html
import os
from functools import lru_cache

import torch

try:
    import regex as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       _re.IGNORECASE)
except ImportError:  # pragma: no cover
    import re as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+",
                       _re.IGNORECASE | _re.UNICODE)


# The part of CLIP's merge table that the synthetic benchmark / test captions exercise (synth._WORDS), with the original
# ranks and token ids: written by oracle/make_golden.py from the reference's bpe_simple_vocab_16e6.txt.gz and checked there
# against the reference's tokenizer on 3000 captions.  Package data: bench.py tokenises with it inside the timed region.
CLOSED_VOCAB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bpe_closed_vocab.json")


def find_bpe_file(bpe_path=None):
    for c in (bpe_path, os.environ.get("DIFFSOUND_BPE_PATH"),
              os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "bpe_simple_vocab_16e6.txt.gz")):
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError(
        "CLIP BPE merge table not found: set DIFFSOUND_BPE_PATH to bpe_simple_vocab_16e6.txt.gz "
        "(shipped with the reference under sound_synthesis/modeling/modules/clip/)")


@lru_cache()
def _byte_alphabet():
    """Printable stand-ins for all 256 byte values (GPT-2 / CLIP byte-level BPE convention)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    def __init__(self, end_idx=49152, bpe_path=None):
        if bpe_path is not None and str(bpe_path).endswith(".json"):
            self._init_closed(bpe_path)
            return
        self.closed_words = None
        path = find_bpe_file(bpe_path)
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in lines[1:end_idx - 256 - 2 + 1]]
        alphabet = _byte_alphabet()
        # vocabulary order: the 188 printable bytes first, then the remapped ones (the reference's
        # bytes_to_unicode() ordering), then the same with '</w>', then merges, then the two specials
        printable = [b for b in range(256) if alphabet[b] == chr(b)]
        ordered = [alphabet[b] for b in printable] + [alphabet[b] for b in range(256) if b not in printable]
        vocab = ordered + [c + "</w>" for c in ordered] + ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.bytes = alphabet
        self._cache = {}

    def _init_closed(self, path):
        """Closed-vocabulary table (data/bpe_closed_vocab.json = CLOSED_VOCAB_PATH, written by oracle/make_golden.py from the full
        CLIP table): for a fixed word list it holds exactly the merges the full table applies to those words, with
        their ORIGINAL ranks, and the ids of the resulting tokens -- so encoding any text over that word list runs the
        same greedy algorithm and yields the same ids as the full table.  Any other word raises (never a silent
        wrong id).  The GPU box has no copy of the 1.3 MB merge table; bench.py tokenises its synthetic captions with
        this file inside the timed region."""
        import json
        with open(path) as f:
            t = json.load(f)
        self.encoder = dict(t["encoder"])
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {(a, b): r for a, b, r in t["merges"]}
        self.closed_words = frozenset(t["words"])
        self.bytes = _byte_alphabet()
        self._cache = {}

    def _bpe(self, word):
        """Greedy lowest-rank pair merging of one pre-token (symbols: chars, last one tagged '</w>')."""
        if word in self._cache:
            return self._cache[word]
        if self.closed_words is not None and word not in self.closed_words:
            raise KeyError("%r is outside the closed vocabulary of this merge table (%d words): use the full CLIP table "
                           "(DIFFSOUND_BPE_PATH)" % (word, len(self.closed_words)))
        syms = list(word[:-1]) + [word[-1] + "</w>"]
        while len(syms) > 1:
            best, best_rank = None, None
            for a, b in zip(syms, syms[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(syms):
                if i + 1 < len(syms) and syms[i] == best[0] and syms[i + 1] == best[1]:
                    merged.append(best[0] + best[1])
                    i += 2
                else:
                    merged.append(syms[i])
                    i += 1
            syms = merged
        self._cache[word] = syms
        return syms

    def encode(self, text):
        text = html.unescape(html.unescape(text)).strip()          # basic_clean (ftfy is a no-op on ASCII)
        text = " ".join(text.split()).lower()                       # whitespace_clean + lower
        ids = []
        for piece in _PAT.findall(text):
            if piece in ("<|startoftext|>", "<|endoftext|>"):
                ids.append(self.encoder[piece])
                continue
            word = "".join(self.bytes[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._bpe(word))
        return ids


def tokenize(texts, context_length=77, add_start_and_end=True, with_mask=True, pad_value=0, tokenizer=None,
             just_token=False):
    """clip.py:164-215."""
    if isinstance(texts, str):
        texts = [texts]
    sot = [tokenizer.encoder["<|startoftext|>"]] if add_start_and_end else []
    eot = [tokenizer.encoder["<|endoftext|>"]] if add_start_and_end else []
    all_tokens = [sot + tokenizer.encode(t.lower()) + eot for t in texts]
    if just_token:
        return all_tokens
    result = torch.full((len(all_tokens), context_length), pad_value, dtype=torch.long)
    mask = torch.zeros(len(all_tokens), context_length, dtype=torch.bool)
    for i, toks in enumerate(all_tokens):
        if len(toks) > context_length:
            toks = toks[:context_length - 1] + [toks[-1]]           # truncate, keep the final token
        result[i, :len(toks)] = torch.tensor(toks, dtype=torch.long)
        mask[i, :len(toks)] = True
    out = {"token": result}
    if with_mask:
        out["mask"] = mask
    return out


class Tokenize:
    """tokenize.py:Tokenize with the Diffsound settings (context 77, SOT/EOT, pad 0, no clip_embedding)."""

    def __init__(self, context_length=256, add_start_and_end=False, just_token=False, with_mask=True, pad_value=0,
                 clip_embedding=False, condition_emb_config=None, tokenizer_config=None, bpe_path=None):
        assert not clip_embedding, "Diffsound embeds inside the diffusion model (clip_embedding: False)"
        self.context_length = context_length
        self.add_start_and_end = add_start_and_end
        self.with_mask = with_mask
        self.pad_value = pad_value
        self.just_token = just_token
        end_idx = 49152
        if tokenizer_config and "params" in tokenizer_config:
            end_idx = tokenizer_config["params"].get("end_idx", end_idx)
        self._end_idx, self._bpe_path, self._tok = end_idx, bpe_path, None

    @property
    def tokenizer(self):
        if self._tok is None:   # the merge table is only needed once text actually arrives
            self._tok = SimpleTokenizer(self._end_idx, self._bpe_path)
        return self._tok

    def get_tokens(self, text, **kwargs):
        return tokenize(text, context_length=self.context_length, add_start_and_end=self.add_start_and_end,
                        with_mask=self.with_mask, pad_value=self.pad_value, tokenizer=self.tokenizer,
                        just_token=self.just_token)

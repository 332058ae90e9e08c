"""CLIP byte-pair tokenizer -- host side of the text tower, mirror of
ape/modeling/text/eva02_clip/tokenizer.py (`SimpleTokenizer`, `tokenize(texts, context_length=77)`, which is OpenAI CLIP's
tokenizer): lower-cased, whitespace-collapsed text is split by CLIP's pattern, every piece goes to UTF-8 bytes, bytes to a
printable alphabet, and adjacent symbols are merged in the order of the merge table until no listed pair is left.

The merge table is DATA (`bpe_simple_vocab_16e6.txt.gz`, 1.3 MB, shipped with the reference and with every CLIP
distribution); it is not part of this repository.  `find_bpe_vocab()` looks for it (argument, $APE_BPE_VOCAB, the reference
checkout, an installed open_clip / clip package) and raises with that list if nothing is found.
"""
import gzip
import html
import os
from functools import lru_cache

import torch

try:
    import regex as _re
except ImportError:                                     # pragma: no cover
    _re = None

SOT, EOT = "<|startoftext|>", "<|endoftext|>"
VOCAB_SIZE = 49408
_FILE = "bpe_simple_vocab_16e6.txt.gz"


def find_bpe_vocab(path=None):
    cands = [path, os.environ.get("APE_BPE_VOCAB")]
    for root in (os.environ.get("APE_REFERENCE"), "/root/reference"):
        if root:
            cands.append(os.path.join(root, "ape", "modeling", "text", "eva02_clip", _FILE))
            cands.append(os.path.join(root, "ape", "modeling", "text", "eva01_clip", _FILE))
    for mod in ("open_clip", "clip"):
        try:
            m = __import__(mod)
            cands.append(os.path.join(os.path.dirname(m.__file__), _FILE))
        except Exception:
            pass
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError(f"CLIP merge table {_FILE} not found; looked at {[c for c in cands if c]} -- pass bpe_path or set "
                            "APE_BPE_VOCAB")


@lru_cache()
def byte_alphabet():
    """256 byte values -> 256 distinct printable code points (printable latin-1 bytes map to themselves, the others to
    256, 257, ... in byte order): the alphabet the merge table is written in"""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


_FTFY_WARNED = [False]


def _clean(text):
    try:
        import ftfy                                    # the reference repairs mojibake first; identity for clean text
        text = ftfy.fix_text(text)
    except ImportError:
        if not _FTFY_WARNED[0] and not text.isascii():
            # plain ASCII passes through ftfy unchanged; anything else may tokenize differently from the reference
            import warnings
            warnings.warn("ape_amd tokenizer: `ftfy` is not installed -- the reference applies ftfy.fix_text before BPE; non-ASCII "
                          "prompts may tokenize differently (class-name vocabularies are unaffected)", stacklevel=2)
            _FTFY_WARNED[0] = True
    text = html.unescape(html.unescape(text)).strip()
    return " ".join(text.split()).strip() if _re is None else _re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path=None):
        if _re is None:
            raise ImportError("the CLIP tokenizer needs the `regex` package (unicode classes \\p{L} / \\p{N})")
        alphabet = byte_alphabet()
        with gzip.open(find_bpe_vocab(bpe_path), "rt", encoding="utf-8") as fh:
            lines = fh.read().split("\n")
        merges = [tuple(ln.split()) for ln in lines[1: 49152 - 256 - 2 + 1]]        # header line, then 48 894 merges
        symbols = [alphabet[b] for b in sorted(alphabet, key=lambda b: ord(alphabet[b]))]
        symbols = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + [SOT, EOT]
        self.encoder = {s: i for i, s in enumerate(symbols)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.alphabet = alphabet
        self.inverse_alphabet = {c: b for b, c in alphabet.items()}
        self.memo = {SOT: (SOT,), EOT: (EOT,)}
        self.splitter = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                    _re.IGNORECASE)
        self.sot_token, self.eot_token = self.encoder[SOT], self.encoder[EOT]

    def _merge(self, piece):
        """symbols of one pre-token after all applicable merges (lowest rank first, every occurrence of the pair at once)"""
        hit = self.memo.get(piece)
        if hit is not None:
            return hit
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word, word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(best[0] + best[1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = tuple(word)
        self.memo[piece] = out
        return out

    def encode(self, text):
        ids = []
        for piece in self.splitter.findall(_clean(text).lower()):
            mapped = "".join(self.alphabet[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge(mapped))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.inverse_alphabet[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts, context_length=77):
        """-> int64 [len(texts), context_length]: <sot> ids <eot>, zero padded; too long texts are cut and end with <eot>"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros((len(texts), context_length), dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot_token] + self.encode(t) + [self.eot_token]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot_token
            out[i, : len(ids)] = torch.tensor(ids)
        return out


_DEFAULT = []


def tokenize(texts, context_length=77):
    if not _DEFAULT:
        _DEFAULT.append(SimpleTokenizer())
    return _DEFAULT[0](texts, context_length)


def get_tokenizer(model_name=None):
    """every EVA-CLIP configuration of the reference uses this tokenizer (eva02_clip/factory.py get_tokenizer)"""
    return tokenize

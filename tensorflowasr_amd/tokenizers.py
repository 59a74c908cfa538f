"""Text <-> token-id drop-in for the transducer / CTC path: the caller-side data format on the
label side of the hot path.  Mirrors tensorflow_asr/tokenizers.py (same class names, `make()`, `tokenize`, `detokenize`,
`prepand_blank`, `normalize_indices`, `detokenize_unicode_points`, `normalize_text`) on plain Python / NumPy objects instead of
tf string tensors; SentencePiece models are read with the `sentencepiece` package (the reference wraps the same model file in
tensorflow_text's FastSentencepieceTokenizer, tokenizers.py:267-342).

Reference behaviour kept on purpose:
  * characters: an out-of-vocabulary character maps to the BLANK index (StaticHashTable default, tokenizers.py:205-208);
  * `detokenize` of the character tokenizer does not drop blanks explicitly - the blank token is the empty string
    ("<blank>" -> "", :196-197), so they vanish in the join (:238-241);
  * `normalize_text` (:136-146): drop U+2047, NFKC, control / format characters -> space, drop the unknown / pad token
    strings, collapse spaces, lower-case, strip.
"""
import codecs
import re
import unicodedata
from dataclasses import dataclass, field

import numpy as np

ENGLISH_CHARACTERS = ["<blank>", " "] + [chr(c) for c in range(ord("a"), ord("z") + 1)] + ["'"]


@dataclass
class DecoderConfig:
    """The fields of configs.DecoderConfig (configs.py:24-58) that the tokenizers read; `from_dict` accepts the YAML mapping."""

    type: str = "wordpiece"
    blank_index: int = 0
    pad_token: str = "<pad>"
    pad_index: int = -1
    unknown_token: str = "<unk>"
    unknown_index: int = 0
    bos_token: str = "<s>"
    bos_index: int = -1
    eos_token: str = "</s>"
    eos_index: int = -1
    beam_width: int = 0
    norm_score: bool = True
    vocabulary: str = None
    vocab_size: int = 1000
    normalization_form: str = "NFKC"
    keep_whitespace: bool = False
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_dict(cls, config=None):
        config = dict(config or {})
        known = {k: config.pop(k) for k in list(config) if k in cls.__dataclass_fields__ and k != "extra"}
        return cls(**known, extra=config)


def _is_cc_cf(ch):
    return unicodedata.category(ch) in ("Cc", "Cf")


def normalize_text(text, decoder_config=None):
    """Tokenizer.normalize_text (tokenizers.py:136-146) on a Python str (bytes are decoded as UTF-8)."""
    c = decoder_config or DecoderConfig()
    if isinstance(text, bytes):
        text = text.decode("utf-8")
    text = text.replace("⁇", "")
    text = unicodedata.normalize(c.normalization_form, text)
    text = "".join(" " if _is_cc_cf(ch) else ch for ch in text)
    # the reference passes the token strings to regex_replace as PATTERNS; "<unk>" / "<pad>" have no metacharacters
    text = re.sub(c.unknown_token, "", text)
    text = re.sub(c.pad_token, "", text)
    text = re.sub(r" +", " ", text)
    return text.lower().strip()


class Tokenizer:
    def __init__(self, decoder_config=None):
        if isinstance(decoder_config, dict) or decoder_config is None:
            decoder_config = DecoderConfig.from_dict(decoder_config)
        self.decoder_config = decoder_config
        self.scorer = None
        self.tokens, self.num_classes, self.max_length = [], None, 0
        self.blank = decoder_config.blank_index
        self.initialized = False

    # --- shapes / bookkeeping (tokenizers.py:118-133)
    @property
    def shape(self):
        return [self.max_length if self.max_length > 0 else None]

    @property
    def prepand_shape(self):
        return [self.max_length + 1 if self.max_length > 0 else None]

    def update_length(self, length):
        self.max_length = max(self.max_length, int(length))

    def reset_length(self):
        self.max_length = 0

    def add_scorer(self, scorer=None):
        self.scorer = scorer

    def normalize_text(self, text, decoder_config=None):
        return normalize_text(text, decoder_config or self.decoder_config)

    def normalize_indices(self, indices):
        """-1 (padding of sparse->dense decodes) -> blank (tokenizers.py:152-165)."""
        a = np.asarray(indices, dtype=np.int32)
        return np.where(a == -1, np.int32(self.blank), a)

    def prepand_blank(self, ids):
        """Prediction-network input of a transducer: [blank] + labels (tokenizers.py:167-169)."""
        return np.concatenate([np.asarray([self.blank], np.int32), np.asarray(ids, np.int32)])

    def tokenize(self, text):
        raise NotImplementedError

    def detokenize(self, indices):
        raise NotImplementedError

    def detokenize_unicode_points(self, indices):
        raise NotImplementedError

    def _rows(self, indices):
        a = np.asarray(indices, dtype=np.int64)
        return a[None] if a.ndim == 1 else a


class CharTokenizer(Tokenizer):
    """Character vocabulary (tokenizers.py:182-262); vocabulary file = one token per line, '#' comments, '<blank>' = ''."""

    def make(self):
        c = self.decoder_config
        if c.vocabulary is not None:
            with codecs.open(c.vocabulary, "r", "utf-8") as fin:
                lines = fin.readlines()
        else:
            lines = ENGLISH_CHARACTERS
        self.tokens = []
        for line in lines:
            line = unicodedata.normalize(c.normalization_form, line.lower()).strip("\n")
            if line.startswith("#") or not line:
                continue
            self.tokens.append("" if line == "<blank>" else line)
        if self.blank is None:
            self.blank = len(self.tokens)
        self.num_classes = len(self.tokens)
        self.tokens2indices = {}
        for i, t in enumerate(self.tokens):
            self.tokens2indices.setdefault(t, i)
        self.initialized = True
        return self

    def tokenize(self, text):
        text = self.normalize_text(text)
        return np.asarray([self.tokens2indices.get(ch, self.blank) for ch in text], dtype=np.int32)

    def detokenize(self, indices):
        """[B, U] (or [U]) ids -> list of B transcripts."""
        rows = self._rows(self.normalize_indices(indices))
        blank_tok = self.tokens[self.blank] if 0 <= self.blank < len(self.tokens) else ""
        out = []
        for r in rows:
            s = "".join(self.tokens[i] if 0 <= i < len(self.tokens) else blank_tok for i in r)
            out.append(self.normalize_text(s))
        return out

    def detokenize_unicode_points(self, indices):
        ids = self.normalize_indices(np.asarray(indices).reshape(-1))
        pts = [ord(self.tokens[i][0]) for i in ids if 0 <= i < len(self.tokens) and self.tokens[i]]
        return np.asarray(pts, dtype=np.int32)


class SentencePieceTokenizer(Tokenizer):
    """SentencePiece model file (tokenizers.py:267-342: no BOS/EOS, ids as int32)."""

    def make(self):
        import sentencepiece as sp

        c = self.decoder_config
        self.blank = c.blank_index
        self.tokenizer = sp.SentencePieceProcessor()
        with open(c.vocabulary, "rb") as f:
            self.tokenizer.LoadFromSerializedProto(f.read())
        self.num_classes = int(self.tokenizer.GetPieceSize())
        self.initialized = True
        return self

    def tokenize(self, text):
        return np.asarray(self.tokenizer.EncodeAsIds(self.normalize_text(text)), dtype=np.int32)

    def detokenize(self, indices):
        rows = self._rows(self.normalize_indices(indices))
        n = self.num_classes
        return [self.normalize_text(self.tokenizer.DecodeIds([int(i) for i in r if 0 <= i < n])) for r in rows]

    def detokenize_unicode_points(self, indices):
        s = self.detokenize(np.asarray(indices).reshape(1, -1))[0]
        return np.asarray([ord(ch) for ch in s], dtype=np.int32)


class WordPieceTokenizer(Tokenizer):
    """Greedy longest-match-first WordPiece over a vocabulary file, '##' continuation pieces (what tensorflow_text's
    FastWordpieceTokenizer computes, tokenizers.py:345-420); words that cannot be segmented map to the unknown token."""

    suffix = "##"

    def make(self):
        with open(self.decoder_config.vocabulary, "r", encoding="utf-8") as f:
            self.vocab = f.read().splitlines()
        if not self.vocab:
            raise ValueError("Unable to read vocabulary")
        self.tokens2indices = {}
        for i, t in enumerate(self.vocab):
            self.tokens2indices.setdefault(t, i)
        self.tokens = self.vocab
        self.num_classes = len(self.vocab)
        self.unk = self.tokens2indices.get(self.decoder_config.unknown_token, self.decoder_config.unknown_index)
        self.initialized = True
        return self

    def _word(self, w):
        ids, start = [], 0
        while start < len(w):
            end, cur = len(w), None
            while start < end:
                piece = (self.suffix if start > 0 else "") + w[start:end]
                if piece in self.tokens2indices:
                    cur = self.tokens2indices[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def tokenize(self, text):
        text = self.normalize_text(text)
        if self.decoder_config.keep_whitespace:
            words = [w for w in text.replace(" ", "| |").split("|") if w != ""]
        else:
            words = text.split()
        ids = [i for w in words for i in self._word(w)]
        return np.asarray(ids, dtype=np.int32)

    def detokenize(self, indices):
        rows = self._rows(self.normalize_indices(indices))
        out = []
        for r in rows:
            words = []
            for i in r:
                if not 0 <= i < self.num_classes:
                    continue
                t = self.vocab[i]
                if t.startswith(self.suffix) and words:
                    words[-1] += t[len(self.suffix):]
                else:
                    words.append(t)
            out.append(self.normalize_text(" ".join(words)))
        return out

    def detokenize_unicode_points(self, indices):
        s = self.detokenize(np.asarray(indices).reshape(1, -1))[0]
        return np.asarray([ord(ch) for ch in s], dtype=np.int32)


TOKENIZER_TYPES = {"characters": CharTokenizer, "wordpiece": WordPieceTokenizer, "sentencepiece": SentencePieceTokenizer}


def get(decoder_config):
    """tokenizers.get (tokenizers.py:40-50); accepts a DecoderConfig, a mapping, or an object with `.decoder_config`."""
    if hasattr(decoder_config, "decoder_config"):
        decoder_config = decoder_config.decoder_config
    if isinstance(decoder_config, dict):
        decoder_config = DecoderConfig.from_dict(decoder_config)
    if decoder_config.type not in TOKENIZER_TYPES:
        raise ValueError(f"type must be in {list(TOKENIZER_TYPES)}, received {decoder_config.type}")
    return TOKENIZER_TYPES[decoder_config.type](decoder_config).make()

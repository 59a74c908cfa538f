"""Label-side data format: the tokenizer drop-in against the reference's documented behaviour
(tensorflow_asr/tokenizers.py) and, when /root/reference is present (build container), its shipped vocabularies."""
import os

import numpy as np
import pytest

from tensorflowasr_amd import tokenizers as tk

REF = "/root/reference/examples/datasets/librispeech"


def test_normalize_text_rules():
    n = tk.normalize_text
    assert n("  Hello   WORLD \n") == "hello world"
    assert n("a⁇b") == "ab"                       # U+2047 removed (sentencepiece's unk surface)
    assert n("a\x07b​c") == "a b c"               # Cc / Cf -> space
    assert n("x <unk> y<pad>") == "x y"
    assert n("ﬁne Ⅳ") == "fine iv"           # NFKC then lower
    assert n(b"Caf\xc3\xa9") == "café"


def test_character_tokenizer_default_vocabulary():
    t = tk.get({"type": "characters", "blank_index": 0})
    assert t.num_classes == 29 and t.tokens[0] == "" and t.tokens[1] == " " and t.tokens[28] == "'"
    ids = t.tokenize("Hello, it's ME")
    # ',' is out of vocabulary -> blank (hash-table default), letters a=2..z=27, space=1, apostrophe=28
    assert ids.tolist() == [9, 6, 13, 13, 16, 0, 1, 10, 21, 28, 20, 1, 14, 6]
    assert t.prepand_blank(ids)[0] == 0 and len(t.prepand_blank(ids)) == len(ids) + 1
    assert t.detokenize(ids) == ["hello it's me"]   # blank = '' disappears in the join
    assert t.detokenize(np.array([[9, 10, -1, -1], [2, 1, 3, 0]])) == ["hi", "a b"]
    assert t.detokenize_unicode_points(np.array([9, 0, 10, -1])).tolist() == [ord("h"), ord("i")]
    t.update_length(7)
    assert t.shape == [7] and t.prepand_shape == [8]


def test_wordpiece_greedy_longest_match(tmp_path):
    vocab = ["<unk>", "<pad>", "un", "##aff", "##able", "run", "##ning", "a", "##b"]
    p = tmp_path / "wp.vocab"
    p.write_text("\n".join(vocab))
    t = tk.get({"type": "wordpiece", "vocabulary": str(p), "unknown_index": 0})
    assert t.tokenize("Unaffable running").tolist() == [2, 3, 4, 5, 6]
    assert t.tokenize("xyz ab").tolist() == [0, 7, 8]
    assert t.detokenize(np.array([2, 3, 4, 5, 6])) == ["unaffable running"]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference vocabularies only exist in the build container")
def test_reference_vocabularies_round_trip():
    chars = tk.get({"type": "characters", "blank_index": 0, "vocabulary": f"{REF}/characters/english.vocab"})
    assert chars.num_classes == 29
    s = "the quick brown fox's tail"
    assert chars.detokenize(chars.tokenize(s)) == [s]
    import sentencepiece as sp

    for name, V in (("train_bpe_1000", 1000), ("train_bpe_256", 256)):
        path = f"{REF}/sentencepiece/{name}.model"
        t = tk.get({"type": "sentencepiece", "blank_index": 0, "vocabulary": path})
        assert t.num_classes == V
        raw = sp.SentencePieceProcessor(model_file=path)
        for text in ("HE HOPED there would be stew for dinner", "turnips and carrots  and bruised potatoes", "it's"):
            ids = t.tokenize(text)
            assert ids.dtype == np.int32 and ids.tolist() == raw.encode(tk.normalize_text(text))
            assert (ids > 0).all()  # id 0 (<unk> / blank) never appears for in-vocabulary text
            assert t.detokenize(ids) == [tk.normalize_text(text)]
            assert t.detokenize(np.concatenate([ids, [-1, -1]])) == [tk.normalize_text(text)]  # -1 padding -> blank -> dropped

"""Caller-side data format: `.tsv` transcripts -> padded TrainData batches, against the
contract of tensorflow_asr/datasets.py (entry parsing, metadata JSON, padding values, drop_remainder, per-replica slices)."""
import json
import wave

import numpy as np
import pytest

from tensorflowasr_amd import datasets as ds
from tensorflowasr_amd import tokenizers as tk


def _write_wav(path, x16, rate=16000, nch=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.asarray(x16, "<i2").tobytes())


@pytest.fixture()
def corpus(tmp_path):
    rng = np.random.default_rng(0)
    texts = ["hello world", "a\tb tab inside", "it's", "the quick brown fox", "zz"]
    lens = [4000, 1601, 800, 6400, 160]
    lines = ["PATH\tDURATION\tTRANSCRIPT"]
    waves = []
    for i, (t, n) in enumerate(zip(texts, lens)):
        x = rng.integers(-32768, 32767, n).astype(np.int16)
        p = tmp_path / f"u{i}.wav"
        _write_wav(p, x)
        waves.append(x)
        lines.append(f"{p}\t{n / 16000:.6f}\t{t}")
    tsv = tmp_path / "transcripts.tsv"
    tsv.write_text("\n".join(lines) + "\n")
    return tmp_path, tsv, texts, lens, waves


def test_read_wav_matches_16bit_decode(corpus, tmp_path):
    root, tsv, texts, lens, waves = corpus
    x = ds.read_wav(str(root / "u0.wav"))
    assert x.dtype == np.float32 and np.array_equal(x, waves[0].astype(np.float32) / 32768.0)   # tf.audio.decode_wav scaling
    st = np.stack([waves[0][:800], waves[1][:800]], 1)                                           # stereo -> channel mean, requantised
    _write_wav(tmp_path / "st.wav", st.reshape(-1), nch=2)
    y = ds.read_wav(str(tmp_path / "st.wav"))
    want = np.clip(np.rint(st.astype(np.float64).mean(1) / 32768.0 * 32768.0), -32768, 32767) / 32768.0
    assert np.allclose(y, want, atol=0)
    _write_wav(tmp_path / "r8k.wav", waves[2], rate=8000)
    with pytest.raises(ValueError, match="sample rate"):
        ds.read_wav(str(tmp_path / "r8k.wav"))
    (tmp_path / "x.flac").write_bytes(b"fLaC\0\0\0\0")
    with pytest.raises(ValueError, match="PCM WAV"):
        ds.read_wav(str(tmp_path / "x.flac"))


def test_entries_metadata_and_batches(corpus):
    root, tsv, texts, lens, waves = corpus
    tok = tk.get({"type": "characters", "blank_index": 0})
    meta = root / "meta.json"
    d = ds.get(tok, {"stage": "train", "data_paths": [str(tsv)], "metadata": str(meta), "drop_remainder": True}, "slice")
    d.update_metadata()
    assert d.entries[1] == [str(root / "u1.wav"), f"{1601 / 16000:.6f}", "a\tb tab inside"]   # split on the first two tabs only
    assert d.num_entries == 5 and d.max_input_length == 6400 and d.max_label_length == len("the quick brown fox")
    saved = json.loads(meta.read_text())["train"]
    assert saved == dict(max_input_length=6400, max_label_length=19, num_entries=5)
    d2 = ds.ASRSliceDataset("train", tok, [str(tsv)], metadata=str(meta))                         # a second reader picks the metadata up
    assert d2.max_input_length == 6400 and d2.total_steps == 5

    got = list(d.batches(batch_size=2))
    assert len(got) == 2 and d.total_steps == 2                                                   # drop_remainder: 5 // 2
    b = got[0]
    assert b["sig"].shape == (2, 6400) and b["labels"].shape == (2, 19) and b["preds"].shape == (2, 20)
    assert b["nsamp"].tolist() == [4000, 1601] and b["ulen"].tolist() == [11, len(tok.tokenize(texts[1]))]
    assert np.array_equal(b["sig"][0, :4000], waves[0] / np.float32(32768)) and not b["sig"][0, 4000:].any()
    assert b["preds"][0, 0] == 0 and np.array_equal(b["preds"][0, 1:12], b["labels"][0, :11]) and not b["labels"][0, 11:].any()
    assert b["plen"].tolist() == (b["ulen"] + 1).tolist()

    d.drop_remainder = False
    got = list(d.batches(batch_size=2))
    assert len(got) == 3 and got[2]["sig"].shape[0] == 1 and d.total_steps == 3

    # without metadata: padded to the batch maxima
    d3 = ds.ASRSliceDataset("eval", tok, [str(tsv)], drop_remainder=False)
    g = list(d3.batches(batch_size=3))
    assert g[0]["sig"].shape == (3, 4000) and g[1]["sig"].shape == (2, 6400) and g[1]["labels"].shape[1] == 19

    td = next(d.create(batch_size=2))
    assert td.inputs.inputs.shape == (2, 6400) and td.inputs.predictions_length.tolist() == (b["ulen"] + 1).tolist()
    assert td.labels.labels_length.tolist() == b["ulen"].tolist()


def test_replica_slices_share_padded_shapes(corpus):
    root, tsv, texts, lens, waves = corpus
    tok = tk.get({"type": "characters", "blank_index": 0})
    mk = lambda: ds.ASRSliceDataset("train", tok, [str(tsv)], shuffle=True, seed=7, drop_remainder=True)
    full = list(mk().batches(batch_size=4, world=1))
    r0 = list(mk().batches(batch_size=2, rank=0, world=2))
    r1 = list(mk().batches(batch_size=2, rank=1, world=2))
    assert len(full) == len(r0) == len(r1) == 1
    assert r0[0]["sig"].shape == r1[0]["sig"].shape == (2, full[0]["sig"].shape[1])               # common lengths (DP contract)
    assert np.array_equal(np.concatenate([r0[0]["sig"], r1[0]["sig"]]), full[0]["sig"])
    assert np.array_equal(np.concatenate([r0[0]["labels"], r1[0]["labels"]]), full[0]["labels"])
    shapes, gb, padded = ds.get_global_shape(2, 2, mk())
    assert gb == 4 and shapes["batch_size"] == 4 and padded == (None, None)


def test_indefinite_repeats_and_ga_steps(corpus):
    root, tsv, texts, lens, waves = corpus
    tok = tk.get({"type": "characters", "blank_index": 0})
    d = ds.ASRSliceDataset("train", tok, [str(tsv)], indefinite=True, drop_remainder=True)
    it = d.batches(batch_size=2, ga_steps=2)
    seen = [next(it)["nsamp"].tolist() for _ in range(5)]
    assert seen[0] == seen[2] == seen[4] and seen[1] == seen[3]                                    # 2 batches per epoch, repeated
    assert d.use_ga and d.total_steps == 1                                                        # ceil(2 / ga_steps)


@pytest.mark.gpu
def test_tsv_corpus_trains_and_decodes_end_to_end(corpus):
    """`.tsv` + WAV files -> tokenizer -> padded TrainData -> HIP train steps -> greedy decode -> detokenize (the whole caller
    contract around the hot path on a toy corpus)."""
    import torch

    from tensorflowasr_amd import configs
    from tensorflowasr_amd.conformer import ConformerTransducer

    root, tsv, texts, lens, waves = corpus
    tok = tk.get({"type": "characters", "blank_index": 0})
    d = ds.ASRSliceDataset("train", tok, [str(tsv)], indefinite=True, drop_remainder=True)
    d.update_metadata()
    dev = torch.device("cuda", 0)
    model = ConformerTransducer(configs.conformer_tiny(vocab_size=tok.num_classes), dev, dtype=torch.float32, seed=0)
    model.optimizer["schedule"] = 2e-3
    it = d.create(batch_size=4, device=dev)
    first = next(it)
    assert first.inputs.inputs.shape == (4, 6400) and first.inputs.inputs.device.type == "cuda"
    losses = [float(model.train_step(first)["loss"].mean()) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    from tensorflowasr_amd.schemas import PredictInput

    out = model.recognize(PredictInput(first.inputs.inputs, first.inputs.inputs_length))
    hyp = tok.detokenize(out.tokens.cpu().numpy())
    assert len(hyp) == 4 and all(isinstance(h, str) for h in hyp)

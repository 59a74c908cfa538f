"""`.tsv` transcript reader and the padded-batch contract on the caller side of the hot path.
Mirrors the slice dataset of tensorflow_asr/datasets.py on NumPy / torch objects (no tf.data):

  * entries:  `PATH \\t DURATION \\t TRANSCRIPT`, first line = header, split on the first two tabs only (datasets.py:268-283);
  * metadata: JSON {stage: {max_input_length, max_label_length, num_entries}} (datasets.py:205-262), input length =
    ceil(duration * sample_rate) (math_util.py:308-312), label length = len(tokenize(transcript));
  * item:     waveform f32 in [-1, 1) -> (inputs, inputs_length, labels, labels_length, predictions = [blank] + labels,
    predictions_length) (datasets.py:298-328);
  * batches:  padded to the metadata maxima when known, else to the batch maxima; padding 0.0 for audio and `blank` for
    labels / predictions; `drop_remainder`; optional shuffle; `indefinite` repetition (datasets.py:332-388);
  * data parallel: the global batch is `batch_size * world` utterances and rank r trains on its contiguous slice
    (datasets.py:100-108, base_model.py:86); every rank's slice is padded to the SAME lengths (the DP contract of dp.py).

Audio: the reference converts every file to 16-bit WAV bytes (librosa.load -> tf.audio.encode_wav) and decodes them with
tf.audio.decode_wav (data_util.py:25-35), i.e. samples are int16 / 32768.  Here PCM WAV files are read with the standard
library (`wave`): 16-bit samples / 32768 reproduce those floats exactly at the native rate; other widths are converted
with the same 16-bit quantisation (round(x * 32768) clamped).  A file whose rate differs from `sample_rate`, or a
non-WAV container (FLAC/MP3: librosa is not available here), raises - no silent resampling.
"""
import json
import math
import os
import wave

import numpy as np
import torch

from .schemas import TrainData, TrainInput, TrainLabel


def get_nsamples(duration, sample_rate=16000):
    """math_util.get_nsamples (math_util.py:308-312)."""
    return math.ceil(float(duration) * sample_rate)


def get_num_batches(nsamples, batch_size, drop_remainders=True):
    """math_util.get_num_batches (math_util.py:31-40)."""
    if nsamples is None or batch_size is None:
        return None
    if drop_remainders:
        return math.floor(float(nsamples) / float(batch_size))
    return math.ceil(float(nsamples) / float(batch_size))


def read_wav(path, sample_rate=16000):
    """PCM WAV -> mono f32 waveform as the reference's load_and_convert_to_wav + read_raw_audio produce it."""
    path = os.path.realpath(os.path.expanduser(path))
    try:
        with wave.open(path, "rb") as w:
            nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
    except (wave.Error, EOFError) as e:
        raise ValueError(f"{path}: only PCM WAV files can be read here ({e})") from e
    if rate != sample_rate:
        raise ValueError(f"{path}: sample rate {rate} != {sample_rate} (resampling is not available; convert the corpus first)")
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float64) / 8388608.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)  # librosa mono=True: channel mean
    if width != 2 or nch > 1:
        # the reference re-encodes to 16-bit WAV: round(x * 2^15) clamped to int16, decoded as / 2^15
        x = np.clip(np.rint(np.asarray(x, np.float64) * 32768.0), -32768, 32767) / 32768.0
    return np.ascontiguousarray(x, dtype=np.float32)


class ASRSliceDataset:
    """datasets.ASRSliceDataset (datasets.py:143-388, 473-494) without tf.data: iterate it for TrainData batches."""

    def __init__(self, stage, tokenizer, data_paths, cache=False, shuffle=False, indefinite=False, drop_remainder=True,
                 enabled=True, metadata=None, buffer_size=None, sample_rate=16000, name="", seed=0, reader=read_wav, **kwargs):
        if not isinstance(data_paths, (list, tuple)):
            raise ValueError("data_paths must be a list of string paths")
        self.stage, self.tokenizer, self.data_paths = stage, tokenizer, list(data_paths or [])
        self.cache, self.shuffle, self.indefinite, self.drop_remainder = cache, shuffle, indefinite, drop_remainder
        self.enabled, self.metadata, self.buffer_size, self.sample_rate = enabled, metadata, buffer_size, sample_rate
        self.name = name or stage
        self.use_ga = False
        self.reader = reader
        self._rng = np.random.default_rng(seed)
        self._cache = {}
        for k, v in kwargs.items():
            setattr(self, k, v)
        self.entries = []
        self.max_input_length = None
        self.max_label_length = None
        self.total_steps = None
        self.num_entries = None
        self.load_metadata()

    # ------------------------------------------------------------------ metadata (datasets.py:205-262)
    def compute_metadata(self):
        if not self.tokenizer.initialized:
            raise ValueError("Tokenizer must be initialized before computing metadata")
        self.max_input_length = self.max_input_length or 0
        self.max_label_length = self.max_label_length or 0
        self.read_entries()
        for _, duration, transcript in self.entries:
            self.max_input_length = max(self.max_input_length, get_nsamples(duration, self.sample_rate))
            self.max_label_length = max(self.max_label_length, len(self.tokenizer.tokenize(transcript)))

    def save_metadata(self):
        if self.metadata is None:
            return
        path = os.path.realpath(os.path.expanduser(self.metadata))
        content = {}
        if os.path.exists(path):
            with open(path, "r", encoding="utf-8") as f:
                try:
                    content = json.loads(f.read())
                except json.JSONDecodeError as e:
                    raise ValueError(f"File {path} is currently not in json format. Please update the file") from e
        content[self.stage] = dict(max_input_length=self.max_input_length, max_label_length=self.max_label_length,
                                   num_entries=self.total_steps)
        with open(path, "w", encoding="utf-8") as f:
            f.write(json.dumps(content, indent=2))

    def load_metadata(self):
        if self.metadata is None or not self.enabled:
            return
        path = os.path.realpath(os.path.expanduser(self.metadata))
        if not os.path.exists(path):
            return
        with open(path, "r", encoding="utf-8") as f:
            try:
                content = json.loads(f.read()).get(self.stage, {})
            except json.JSONDecodeError as e:
                raise ValueError(f"File {path} must be in json format") from e
        if not content:
            return
        self.max_input_length = content.get("max_input_length")
        self.max_label_length = content.get("max_label_length")
        self.total_steps = int(content.get("num_entries", 0))
        self.num_entries = self.total_steps

    def update_metadata(self):
        self.load_metadata()
        self.compute_metadata()
        self.save_metadata()

    # ------------------------------------------------------------------ entries (datasets.py:268-287)
    def read_entries(self):
        if len(self.entries) > 0:
            return
        if not self.enabled:
            return
        for p in self.data_paths:
            p = os.path.realpath(os.path.expanduser(p))
            if not os.path.exists(p):
                raise FileNotFoundError(p)
            with open(p, "r", encoding="utf-8") as f:
                for line in f.read().splitlines()[1:]:  # header
                    parts = line.split("\t", 2)
                    if len(parts) != 3:
                        raise ValueError(f"{p}: expected PATH<TAB>DURATION<TAB>TRANSCRIPT, got {line!r}")
                    self.entries.append(parts)
        if self.shuffle:
            self._rng.shuffle(self.entries)
        self.total_steps = len(self.entries)
        self.num_entries = self.total_steps

    def vocab_generator(self):
        for *_, transcript in self.entries:
            yield transcript

    # ------------------------------------------------------------------ one item (datasets.py:298-328)
    def parse(self, path, transcript):
        if self.cache and path in self._cache:
            return self._cache[path]
        inputs = self.reader(path, self.sample_rate)
        labels = np.asarray(self.tokenizer.tokenize(transcript), np.int32)
        item = (inputs, labels, np.asarray(self.tokenizer.prepand_blank(labels), np.int32))
        if self.cache:
            self._cache[path] = item
        return item

    # ------------------------------------------------------------------ batches (datasets.py:332-388)
    def padded_batch(self, items, input_length=None, label_length=None):
        """List of parsed items -> NumPy dict; padded to (input_length, label_length) when given, else to the batch maxima."""
        B = len(items)
        nsamp = np.asarray([len(it[0]) for it in items], np.int32)
        ulen = np.asarray([len(it[1]) for it in items], np.int32)
        N = int(input_length if input_length else (nsamp.max() if B else 0))
        U = int(label_length if label_length else (ulen.max() if B else 0))
        if B and (nsamp.max() > N or ulen.max() > U):
            raise ValueError(f"item longer than the padded shape ({int(nsamp.max())} > {N} samples or {int(ulen.max())} > {U} labels); "
                             "recompute the metadata")
        blank = self.tokenizer.blank
        sig = np.zeros((B, N), np.float32)
        labels = np.full((B, U), blank, np.int32)
        preds = np.full((B, U + 1), blank, np.int32)
        for i, (x, lab, pred) in enumerate(items):
            sig[i, : len(x)] = x
            labels[i, : len(lab)] = lab
            preds[i, : len(pred)] = pred
        return dict(sig=sig, nsamp=nsamp, labels=labels, ulen=ulen, preds=preds, plen=ulen + 1)

    def batches(self, batch_size, ga_steps=1, rank=0, world=1, padded_shapes=None):
        """Generator of per-rank NumPy batches.  The global batch is batch_size * world consecutive entries; rank r gets the
        slice [r*batch_size, (r+1)*batch_size) of it, padded to the global batch's lengths (or the metadata maxima)."""
        self.read_entries()
        if ga_steps > 1 and self.stage == "train":
            self.use_ga = True
        G = batch_size * world
        if self.num_entries:
            self.total_steps = get_num_batches(self.num_entries, G, drop_remainders=self.drop_remainder)
            if self.use_ga:
                self.total_steps = get_num_batches(self.total_steps, ga_steps, drop_remainders=False)
        in_len, lab_len = padded_shapes if padded_shapes is not None else (self.max_input_length, self.max_label_length)
        while True:
            order = np.arange(len(self.entries))
            if self.shuffle:
                order = self._rng.permutation(len(self.entries))  # reshuffle_each_iteration (every rank: same seed, same order)
            for s in range(0, len(order), G):
                idx = order[s : s + G]
                if len(idx) < G and self.drop_remainder:
                    break
                if in_len and lab_len:
                    N, U = in_len, lab_len
                    mine = [self.parse(self.entries[i][0], self.entries[i][2]) for i in idx[rank * batch_size : (rank + 1) * batch_size]]
                else:
                    # common padded lengths need every item's length: durations / token counts give them without decoding audio
                    allit = [self.parse(self.entries[i][0], self.entries[i][2]) for i in idx]
                    N = in_len or max(len(it[0]) for it in allit)
                    U = lab_len or max(len(it[1]) for it in allit)
                    mine = allit[rank * batch_size : (rank + 1) * batch_size]
                yield self.padded_batch(mine, N, U)
            if not self.indefinite:
                return

    def create(self, batch_size, ga_steps=1, rank=0, world=1, padded_shapes=None, device=None):
        """datasets.ASRSliceDataset.create: generator of schemas.TrainData (signals on `device`, lengths on the host)."""
        if not self.enabled:
            return None
        dev = device if device is not None else torch.device("cpu")

        def gen():
            for b in self.batches(batch_size, ga_steps, rank, world, padded_shapes):
                yield to_train_data(b, dev)

        return gen()


def to_train_data(batch, device):
    """NumPy batch dict -> schemas.TrainData in the layout ConformerTransducer.train_step takes (schemas.py:20-45)."""
    return TrainData(
        TrainInput(torch.from_numpy(batch["sig"]).to(device), torch.from_numpy(batch["nsamp"]),
                   torch.from_numpy(batch["preds"]).to(device), torch.from_numpy(batch["plen"]).to(device)),
        TrainLabel(torch.from_numpy(batch["labels"]).to(device), torch.from_numpy(batch["ulen"])))


def get_global_shape(batch_size, world, *datasets):
    """datasets.get_global_shape (datasets.py:100-137): model shapes, global batch and the padded (input, label) lengths."""
    max_in = max([d.max_input_length or 0 for d in datasets] + [0]) or None
    max_lab = max([d.max_label_length or 0 for d in datasets] + [0]) or None
    shapes = dict(batch_size=batch_size * world, input_shape=[max_in], prediction_shape=[max_lab + 1] if max_lab else [None])
    return shapes, batch_size * world, (max_in, max_lab)


class DatasetConfig:
    """configs.DatasetConfig (configs.py:61-79): the YAML mapping of one dataset; unknown keys become attributes."""

    def __init__(self, config=None):
        config = dict(config or {})
        self.name = config.pop("name", "")
        self.enabled = config.pop("enabled", True)
        self.stage = config.pop("stage", None)
        self.data_paths = config.pop("data_paths", None)
        self.tfrecords_dir = config.pop("tfrecords_dir", None)
        self.tfrecords_shards = config.pop("tfrecords_shards", 16)
        self.tfrecords_buffer_size = config.pop("tfrecords_buffer_size", 32 * 1024 * 1024)
        self.shuffle = config.pop("shuffle", False)
        self.cache = config.pop("cache", False)
        self.drop_remainder = config.pop("drop_remainder", True)
        self.buffer_size = config.pop("buffer_size", 1000)
        self.metadata = config.pop("metadata", None)
        self.sample_rate = config.pop("sample_rate", 16000)
        for k, v in config.items():
            setattr(self, k, v)


def get(tokenizer, dataset_config, dataset_type="slice", dataset_cache=False):
    """datasets.get (datasets.py:82-96) for the slice / generator types; TFRecords are a TensorFlow container and the
    HuggingFace type needs network access - both raise here."""
    if isinstance(dataset_config, dict):
        dataset_config = DatasetConfig(dataset_config)
    if dataset_type not in ("slice", "generator"):
        raise ValueError(f"dataset_type must be 'slice' or 'generator' here, received {dataset_type!r}")
    cfg = dict(vars(dataset_config))
    cfg["cache"] = dataset_cache or cfg.get("cache", False)
    for k in ("tfrecords_dir", "tfrecords_shards", "tfrecords_buffer_size", "tfrecords_compression_type", "item_mapping"):
        cfg.pop(k, None)
    return ASRSliceDataset(tokenizer=tokenizer, **cfg)

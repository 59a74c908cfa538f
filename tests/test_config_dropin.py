"""CPU: the reference's own model config (examples/models/transducer/conformer/small.yml.j2) maps onto ConformerConfig
unchanged (drop-in surface).  Skipped where /root/reference is absent (the GPU box)."""
import os

import pytest

from tensorflowasr_amd import configs

REF = "/root/reference/examples/models/transducer/conformer/small.yml.j2"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_small_yml_maps_onto_config():
    import jinja2
    import yaml

    txt = jinja2.Template(open(REF).read()).render(decoder_config={"vocabsize": 1000}, modeldir="/tmp/m", kaggle_model_handle="x")
    doc = yaml.safe_load(txt)
    assert doc["model_config"]["class_name"] == "tensorflow_asr.models.transducer.conformer>Conformer"
    cfg = configs.ConformerConfig.from_reference(doc["model_config"]["config"])
    s = configs.conformer_s()
    for k in ("dmodel", "num_blocks", "head_size", "num_heads", "kernel_size", "filters", "embed_dim", "rnn_units", "joint_dim",
              "vocab_size", "dropout", "ffm_residual", "l2", "num_feature_bins", "nfft"):
        assert getattr(cfg, k) == getattr(s, k), k
    assert cfg.time_masking["num_masks"] == 10 and cfg.time_masking["p_upperbound"] == 0.05
    assert cfg.freq_masking["mask_factor"] == 27
    lr = doc["learning_config"]["optimizer_config"]["config"]["learning_rate"]["config"]
    assert configs.transformer_schedule(1, lr["dmodel"], lr["warmup_steps"], lr["scale"], eval(lr["max_lr"])) > 0


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_small_streaming_yml_maps_onto_config():
    """examples/models/transducer/conformer/small-streaming.yml.j2 (the reference's other published Conformer result): chunked
    attention mask (encoder_chunk_size 16, encoder_history_size 64) and LayerNormalization after the depthwise conv."""
    import jinja2
    import yaml

    path = REF.replace("small.yml.j2", "small-streaming.yml.j2")
    txt = jinja2.Template(open(path).read()).render(decoder_config={"vocabsize": 1000}, modeldir="/tmp/m", kaggle_model_handle="x")
    cfg = configs.ConformerConfig.from_reference(yaml.safe_load(txt)["model_config"]["config"])
    assert (cfg.chunk_size, cfg.history_size, cfg.convm_dw_norm) == (16, 64, "layer")
    assert cfg.sub_norm == "layer"  # `norms: [layer, layer]` of the yml's encoder_subsampling (subsampling.py:205-213)
    assert (cfg.dmodel, cfg.head_size, cfg.num_blocks, cfg.kernel_size) == (144, 36, 16, 31)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_ctc_conformer_yml_maps_onto_config():
    """examples/models/ctc/conformer/small.yml.j2: d=176 / head 44, per-layer attention biases, Dense(vocab) decoder."""
    import jinja2
    import yaml

    path = "/root/reference/examples/models/ctc/conformer/small.yml.j2"
    txt = jinja2.Template(open(path).read()).render(decoder_config={"vocabsize": 1000}, modeldir="/tmp/m", kaggle_model_handle="x")
    mc = yaml.safe_load(txt)["model_config"]
    cfg = configs.ConformerConfig.from_reference(mc["config"], class_name=mc["class_name"])
    ref = configs.conformer_ctc_s()
    for k in ("head", "dmodel", "head_size", "num_heads", "num_blocks", "filters", "mhsam_use_attention_bias", "kernel_size", "vocab_size"):
        assert getattr(cfg, k) == getattr(ref, k), k
    assert cfg.freq_masking["num_masks"] == 2 and cfg.freq_masking["prob"] == 0.5
    from tensorflowasr_amd import params

    names = [n for n, *_ in params.param_specs(cfg)]
    assert "dec/logits/w" in names and "enc/block3/mhsa/u" in names and "enc/u" not in names and "pred/emb" not in names


def test_unsupported_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        configs.ConformerConfig.from_reference({"encoder_mha_type": "mha", "vocab_size": 10})
    with pytest.raises(NotImplementedError):
        configs.ConformerConfig.from_reference({"prediction_rnn_type": "gru", "vocab_size": 10})
    # every constructor option that would change the network but is not built (models/transducer/conformer.py:23-77)
    for k, v in (("encoder_interleave_relpe", False), ("encoder_mhsam_causal", True), ("encoder_convm_scale_factor", 4),
                 ("encoder_convm_use_group_conv", True), ("encoder_module_norm_position", "post"), ("encoder_block_norm_position", "pre"),
                 ("encoder_trainable", False), ("prediction_projection_units", 256), ("prejoint_encoder_linear", False),
                 ("postjoint_linear", True), ("prediction_layer_norm", False), ("bias_regularizer", {"class_name": "L2"})):
        with pytest.raises(NotImplementedError):
            configs.ConformerConfig.from_reference({k: v, "vocab_size": 10})
    # options that only choose an implementation of the same mathematics are accepted
    configs.ConformerConfig.from_reference({"encoder_mhsam_flash_attention": True, "prediction_rnn_implementation": 1, "prediction_rnn_unroll": True,
                                            "vocab_size": 10})


def test_speech_config_options_are_mapped_or_rejected():
    """FeatureExtraction.__init__ (feature_extraction.py:32-130): epsilon and the mel edges reach the front end; the options the MI355X
    front end does not implement (signal / feature normalisation, padding, librosa-like framing, log10, other feature types, other
    transform sizes) raise instead of being ignored; the epsilon range check is the reference's assertion."""
    base = {"vocab_size": 10}
    cfg = configs.ConformerConfig.from_reference(dict(base, speech_config={"epsilon": 1e-5, "lower_edge_hertz": 125.0, "upper_edge_hertz": 7600.0,
                                                                           "preemphasis": 0.0, "num_feature_bins": 40}))
    assert (cfg.epsilon, cfg.lower_edge_hertz, cfg.upper_edge_hertz, cfg.preemphasis, cfg.num_feature_bins) == (1e-5, 125.0, 7600.0, 0.0, 40)
    dflt = configs.ConformerConfig.from_reference(dict(base, speech_config={}))
    assert (dflt.epsilon, dflt.lower_edge_hertz, dflt.upper_edge_hertz, dflt.preemphasis, dflt.nfft) == (1e-6, 0.0, 8000.0, 0.97, 512)
    for bad in ({"normalize_signal": True}, {"normalize_zscore": True}, {"normalize_min_max": True}, {"padding": 160}, {"log_base": "10"},
                {"use_librosa_like_stft": True}, {"pad_end": False}, {"feature_type": "mfcc"}, {"nfft": 1024}, {"frame_ms": 40}):
        with pytest.raises(NotImplementedError):
            configs.ConformerConfig.from_reference(dict(base, speech_config=bad))
    with pytest.raises(AssertionError):
        configs.ConformerConfig.from_reference(dict(base, speech_config={"epsilon": 1e-9}))
    # explicitly stating the supported values is accepted
    configs.ConformerConfig.from_reference(dict(base, speech_config={"normalize_signal": False, "padding": 0, "log_base": "e", "pad_end": True,
                                                                       "feature_type": "log_mel_spectrogram"}))


def test_product_mel_matrix_equals_the_oracles_for_other_edges():
    import numpy as np

    from oracle import conformer_ref as R
    from tensorflowasr_amd.conformer import _mel_weight_matrix

    for nb, lo, hi in ((80, 0.0, 8000.0), (80, 125.0, 7600.0), (40, 20.0, 4000.0)):
        assert np.array_equal(_mel_weight_matrix(nb, 257, 16000, lo, hi), R.mel_weight_matrix(nb, 257, 16000, lo, hi))


def test_m_config_is_the_paper_shape():
    m = configs.conformer_m()
    assert (m.dmodel, m.num_heads, m.head_size, m.num_blocks, m.rnn_units, m.joint_dim) == (256, 4, 64, 16, 640, 640)


def test_schedule_at_keras_first_update_is_zero():
    """keras evaluates a LearningRateSchedule at `iterations` BEFORE the increment, i.e. at 0 for the first update: TransformerSchedule gives
    min(inf, 0) = 0 there (schedules.py:28-37), then max_lr / min_lr as usual; the oracle and the host restatement agree."""
    from oracle import conformer_ref as R

    for fn in (configs.transformer_schedule, R.transformer_schedule):
        assert float(fn(0, 144, 10000, 2.0, 0.05 / 12)) == 0.0
        assert float(fn(0, 144, 10000, 2.0, 0.05 / 12, 1e-6)) == pytest.approx(1e-6)
        assert float(fn(1, 144, 10000, 2.0, 0.05 / 12)) == pytest.approx(2.0 * 144 ** -0.5 * 10000 ** -1.5, rel=1e-5)


def _frozen():
    import json

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_configs.json")))


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_frozen_config_fixture_is_the_reference_rendering():
    """tests/golden/reference_configs.json (what the GPU box builds models from) regenerates from /root/reference."""
    import jinja2
    import yaml

    for key, v in _frozen().items():
        txt = jinja2.Template(open("/root/reference/" + v["source"]).read()).render(decoder_config={"vocabsize": 1000}, modeldir="/tmp/m",
                                                                                  kaggle_model_handle="x", repodir="/root/reference", datadir="/tmp/d")
        doc = yaml.safe_load(txt)
        assert doc["model_config"] == v["model_config"], key
        assert doc["learning_config"]["optimizer_config"] == v["learning_config"]["optimizer_config"], key


def test_optimizer_and_noise_configs_of_the_shipped_ymls():
    """learning_config.optimizer_config / gwn_config of the four shipped model configs map onto the train step's optimizer
    (small.yml.j2:73-91, contextnet/small.yml.j2:217-240)."""
    from tensorflowasr_amd import base_model

    fz = _frozen()
    o = base_model.optimizer_from_config(fz["transducer/conformer/small"]["learning_config"]["optimizer_config"], 144)
    assert (o["beta1"], o["beta2"], o["eps"], o["weight_decay"]) == (0.9, 0.98, 1e-9, 1e-6)
    assert o["schedule"] == dict(dmodel=144, warmup_steps=10000, scale=2.0, max_lr=pytest.approx(0.05 / 12))
    o = base_model.optimizer_from_config(fz["transducer/contextnet/small"]["learning_config"]["optimizer_config"], 320)
    assert o["schedule"]["min_lr"] == 1e-6 and o["schedule"]["max_lr"] == 0.0025 and o["schedule"]["warmup_steps"] == 15000
    assert configs.transformer_schedule(0, **o["schedule"]) == 1e-6
    assert fz["transducer/contextnet/small"]["learning_config"]["gwn_config"] == {"predict_net_step": 20000, "predict_net_stddev": 0.075}
    assert base_model.optimizer_from_config({"class_name": "Adam", "config": {"learning_rate": 1e-3}}, 144)["schedule"] == 1e-3
    with pytest.raises(NotImplementedError):
        base_model.optimizer_from_config({"class_name": "SGD", "config": {}}, 144)
    # ContextNet kwargs -> config: identical to the hand-written configs.contextnet()
    import dataclasses

    c = configs.contextnet_from_reference(fz["transducer/contextnet/small"]["model_config"]["config"])
    r = configs.contextnet()
    assert [f.name for f in dataclasses.fields(c) if getattr(c, f.name) != getattr(r, f.name)] == []


def test_model_from_config_rejects_models_off_the_path():
    from tensorflowasr_amd import base_model

    with pytest.raises(NotImplementedError):
        base_model.model_from_config({"class_name": "tensorflow_asr.models.ctc.jasper>Jasper", "config": {}})


def test_yaml_number_expressions_are_parsed_not_evaluated():
    """`max_lr: 0.05/(144**0.5)` (small.yml.j2:80) is arithmetic; anything that is not arithmetic raises (no eval)."""
    from tensorflowasr_amd.base_model import _eval_number

    assert _eval_number("0.05/(144**0.5)") == pytest.approx(0.05 / 12)
    assert _eval_number("0.05 / sqrt(256)") == pytest.approx(0.05 / 16)
    assert _eval_number("-1e-3") == -1e-3 and _eval_number(0.25) == 0.25
    for bad in ("().__class__.__base__.__subclasses__()", "__import__('os').getcwd()", "abs(1)", "sqrt", "1 if 1 else 2", "[1][0]"):
        with pytest.raises((ValueError, SyntaxError)):
            _eval_number(bad)

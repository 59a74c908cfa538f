"""Conformer-CTC on the same HIP kernels and C ABI as the transducer path.

Mirrors  tensorflow_asr.models.ctc.conformer.Conformer      (models/ctc/conformer.py:57-143: ConformerEncoder + ConformerDecoder)
         ConformerDecoder.call                              (models/ctc/conformer.py:21-47: one Dense(vocab_size) named "logits")
         CtcModel.call / recognize / recognize_beam         (models/ctc/base_ctc.py:74-149)
         CtcLoss.call                                       (losses/ctc_loss.py:47-66: tf.nn.ctc_loss, blank 0, mean over the batch)

The log-mel frontend, SpecAugment, subsampling, Conformer blocks (native block executor), optimizer, gradient accumulation
and the data-parallel hooks are inherited from ConformerTransducer; this class replaces the prediction / joint networks and the
RNN-T loss by the Dense decoder and tfasr_ctc_loss, with a hand-written backward.  Decoding: tfasr_ctc_greedy_decode (device)
and tfasr_ctc_beam_search_host (host routine, as tf.nn.ctc_beam_search_decoder is).  No CPU fallback.
"""
import torch

from . import kernels as K
from .conformer import ConformerTransducer
from .schemas import PredictInput, PredictOutput, TrainData, TrainInput, TrainOutput


class ConformerCTC(ConformerTransducer):
    def __init__(self, cfg, device=None, dtype=torch.bfloat16, seed=0, dp=None):
        if cfg.head != "ctc":
            raise ValueError("ConformerCTC needs a config with head='ctc' (configs.conformer_ctc_s())")
        super().__init__(cfg, device, dtype, seed, dp)
        self.time_reduction_factor = cfg.time_reduction_factor
        self.use_pred_stream = False  # there is no prediction network to overlap

    # ------------------------------------------------------------------------------------------- forward
    def _forward_ctc(self, inputs: TrainInput, training, ctx, masks=None):
        dev = self.device
        self._drop_epoch += 1
        sig = inputs.inputs.to(dev, non_blocking=True)
        slen = [int(v) for v in inputs.inputs_length.tolist()]
        feats, flen = self.frontend(sig, slen, training, masks)
        enc, T, elen, elen_dev = self.encoder_fwd(feats, flen, training, ctx)
        ps = self.ps
        B = sig.shape[0]
        logits = K.matmul(enc, ps.w2d("dec/logits/w"), bias=ps.p("dec/logits/b")).view(B, T, self.cfg.vocab_size)
        if ctx is not None:
            ctx["dec"] = dict(enc=enc, B=B, T=T)
        return logits, elen, elen_dev

    def __call__(self, inputs: TrainInput, training=False):
        """CtcModel.call (base_ctc.py:74-81)."""
        logits, elen, _ = self._forward_ctc(inputs, training, None)
        return TrainOutput(logits=logits, logits_length=torch.tensor(elen, dtype=torch.int32))

    # ------------------------------------------------------------------------------------------- loss + backward
    def loss_and_backward(self, data: TrainData, training=True, masks=None, want_backward=True, packed=True, reduce=True):
        """forward, CtcLoss (mean over the batch) and the full backward into the flat gradient buffer (accumulating)."""
        self.dp.set_reduce(bool(reduce))
        ctx = {} if want_backward else None
        dev = self.device
        logits, elen, elen_dev = self._forward_ctc(data.inputs, training, ctx, masks)
        B, T, V = logits.shape
        labels = data.labels.labels.to(dev, non_blocking=True).to(torch.int32).contiguous()
        llen = [int(v) for v in data.labels.labels_length.tolist()]
        # BaseLoss.call: logit_length = max(logit_length, label_length)  (losses/base_loss.py:36), bounded by the padded length
        tl = [min(max(int(a), b), T) for a, b in zip(elen, llen)]
        tl_dev, ul_dev = self._h2d(tl), self._h2d(llen)
        gscale = torch.full((B,), 1.0 / (B * self.dp.world), dtype=torch.float32, device=dev)
        costs, dlogits = K.ctc_loss_fwd_bwd(logits, labels, ul_dev, tl_dev, grad_scale=gscale, grads=logits, want_grads=want_backward,
                                            blank=self.blank)
        if not want_backward:
            return costs
        s = ctx["dec"]
        denc = self._dense_bwd(dlogits.view(B * T, V), s["enc"], "dec/logits/w", "dec/logits/b")
        self.dp.grads_ready(self.ps.offsets["dec/logits/w"], self.ps.n_reg)
        self.encoder_bwd(denc, ctx)
        self.dp.finish_grads()
        return costs

    # ------------------------------------------------------------------------------------------- decoding
    @torch.no_grad()
    def _infer_logits(self, inputs: PredictInput):
        enc, elen = self.encode(inputs.inputs, inputs.inputs_length)
        B, T, d = enc.shape
        ps = self.ps
        # decision arithmetic in f32 on the f32 master weights (the reference's greedy / beam decoders read f32 logits)
        enc32 = enc.reshape(B * T, d)
        if enc32.dtype != torch.float32:
            enc32 = K.cast(enc32.contiguous(), torch.empty(B * T, d, dtype=torch.float32, device=self.device))
        logits = K.matmul(enc32, ps.p2d("dec/logits/w"), bias=ps.p("dec/logits/b")).view(B, T, self.cfg.vocab_size)
        return logits, elen

    @torch.no_grad()
    def recognize(self, inputs: PredictInput, **kwargs):
        """CtcModel.recognize (base_ctc.py:102-124): tf.nn.ctc_greedy_decoder(merge_repeated=True, blank_index=blank), dense, 0 padded."""
        logits, elen = self._infer_logits(inputs)
        tokens, tlen = K.ctc_greedy_decode(logits, self._h2d(elen), blank=self.blank)
        width = max(int(tlen.max().item()), 1)  # tf.sparse.to_dense: as wide as the longest decoded sequence
        return PredictOutput(tokens=tokens[:, :width], next_tokens=None, next_encoder_states=None, next_decoder_states=None)

    @torch.no_grad()
    def recognize_beam(self, inputs: PredictInput, beam_width=10, **kwargs):
        """CtcModel.recognize_beam (base_ctc.py:128-149): tf.nn.ctc_beam_search_decoder(beam_width) - top path, dense.  TF's beam
        search decoder treats the LAST class as blank (the reference does not pass blank_index there): reproduced."""
        logits, elen = self._infer_logits(inputs)
        toks, n, _ = K.ctc_beam_search(logits, torch.tensor(elen, dtype=torch.int32), beam_width=beam_width, blank_index=None)
        width = max(int(n.max().item()), 1)
        return PredictOutput(tokens=toks[:, :width].to(self.device), next_tokens=None, next_encoder_states=None, next_decoder_states=None)

    def recognize_encoded(self, *a, **k):
        raise NotImplementedError("transducer greedy search does not apply to a CTC model")

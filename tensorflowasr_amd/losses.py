"""Loss objects with the reference's interface (tensorflow_asr/losses/{base_loss,rnnt_loss,ctc_loss}.py), running on the HIP
kernels: `RnntLoss(blank, reduction)(y_true: TrainLabel, y_pred: TrainOutput)` and `CtcLoss(...)`.

`__call__` returns the reduced loss (Keras `sum_over_batch_size` = mean over the batch, rnnt_loss.py:34);
`.call()` returns the per-sample losses [B] like the reference's `call`; `.call_with_grad()` additionally returns
dLoss/dlogits (scaled by the reduction) for a caller that owns the backward pass.
"""
import torch

from . import kernels as K
from .schemas import TrainLabel, TrainOutput


def rnnt_loss(logits, logits_length, labels, labels_length, blank=0, name=None, use_cpu=False, output_shapes=None):
    """Signature of tensorflow_asr.losses.impl.rnnt.rnnt_loss (impl/rnnt.py:11-35) -> per-sample loss [B].
    `use_cpu` / `output_shapes` are accepted for drop-in compatibility and ignored (there is no CPU path here)."""
    costs, _ = K.rnnt_loss_fwd_bwd(logits.contiguous(), labels.to(torch.int32).contiguous(), labels_length.to(torch.int32),
                                   logits_length.to(torch.int32), want_grads=False, blank=blank)
    return costs


class BaseLoss:
    def __init__(self, blank=0, reduction="sum_over_batch_size", name=None):
        assert blank == 0, "Only support blank=0"  # losses/base_loss.py:24
        assert reduction in ("sum_over_batch_size", "sum", "none", None)
        self.blank, self.reduction, self.name = blank, reduction, name

    def _prepare(self, y_true: TrainLabel, y_pred: TrainOutput):
        """BaseLoss.call (losses/base_loss.py:28-37): casts + logit_length = max(logit_length, label_length)."""
        dev = y_pred.logits.device
        logit_length = y_pred.logits_length.to(dev).to(torch.int32)
        labels = y_true.labels.to(dev).to(torch.int32).contiguous()
        label_length = y_true.labels_length.to(dev).to(torch.int32)
        logit_length = torch.maximum(logit_length, label_length)
        return y_pred.logits.contiguous(), logit_length, labels, label_length

    def _reduce(self, costs):
        if self.reduction == "sum_over_batch_size":
            return costs.mean()
        if self.reduction == "sum":
            return costs.sum()
        return costs

    def _grad_scale(self, B, device):
        s = 1.0 / B if self.reduction == "sum_over_batch_size" else 1.0
        return torch.full((B,), s, dtype=torch.float32, device=device)

    def __call__(self, y_true, y_pred):
        return self._reduce(self.call(y_true, y_pred))

    def get_config(self):
        return {"blank": self.blank, "reduction": self.reduction, "name": self.name}


class RnntLoss(BaseLoss):
    """tensorflow_asr.losses.rnnt_loss.RnntLoss (rnnt_loss.py:30-61)."""

    def __init__(self, blank=0, reduction="sum_over_batch_size", output_shapes=None, name=None):
        super().__init__(blank=blank, reduction=reduction, name=name)
        self.output_shapes = output_shapes

    def call(self, y_true, y_pred):
        logits, logit_length, labels, label_length = self._prepare(y_true, y_pred)
        costs, _ = K.rnnt_loss_fwd_bwd(logits, labels, label_length, logit_length, want_grads=False, blank=self.blank)
        return costs

    def call_with_grad(self, y_true, y_pred, inplace=False):
        logits, logit_length, labels, label_length = self._prepare(y_true, y_pred)
        costs, grads = K.rnnt_loss_fwd_bwd(logits, labels, label_length, logit_length,
                                           grad_scale=self._grad_scale(logits.shape[0], logits.device),
                                           grads=logits if inplace else None, blank=self.blank)
        return costs, grads


class CtcLoss(BaseLoss):
    """tensorflow_asr.losses.ctc_loss.CtcLoss (ctc_loss.py:42-66): tf.nn.ctc_loss(logits_time_major=False, blank_index=0)."""

    def call(self, y_true, y_pred):
        logits, logit_length, labels, label_length = self._prepare(y_true, y_pred)
        costs, _ = K.ctc_loss_fwd_bwd(logits, labels, label_length, logit_length, want_grads=False, blank=self.blank)
        return costs

    def call_with_grad(self, y_true, y_pred, inplace=False):
        logits, logit_length, labels, label_length = self._prepare(y_true, y_pred)
        costs, grads = K.ctc_loss_fwd_bwd(logits, labels, label_length, logit_length,
                                          grad_scale=self._grad_scale(logits.shape[0], logits.device),
                                          grads=logits if inplace else None, blank=self.blank)
        return costs, grads

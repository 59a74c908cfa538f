"""Data contract of the hot path — same NamedTuples (names, field order, meaning) as tensorflow_asr/schemas.py:20-62,
holding torch tensors instead of tf tensors."""
import typing

import torch


class TrainInput(typing.NamedTuple):
    inputs: torch.Tensor  # [B, N] float32 PCM
    inputs_length: torch.Tensor  # [B] int32 samples
    predictions: torch.Tensor  # [B, U+1] int32, blank-prepended (tokenizers.py:165-167)
    predictions_length: torch.Tensor  # [B] int32


class TrainOutput(typing.NamedTuple):
    logits: torch.Tensor  # [B, T', U+1, V]
    logits_length: torch.Tensor  # [B] int32


class TrainLabel(typing.NamedTuple):
    labels: torch.Tensor  # [B, U] int32
    labels_length: torch.Tensor  # [B] int32


class TrainData(typing.NamedTuple):
    inputs: TrainInput
    labels: TrainLabel


class PredictInput(typing.NamedTuple):
    inputs: torch.Tensor
    inputs_length: torch.Tensor
    previous_tokens: typing.Optional[torch.Tensor] = None
    previous_encoder_states: typing.Optional[torch.Tensor] = None
    previous_decoder_states: typing.Optional[torch.Tensor] = None


class PredictOutput(typing.NamedTuple):
    tokens: torch.Tensor
    next_tokens: torch.Tensor
    next_encoder_states: typing.Optional[torch.Tensor] = None
    next_decoder_states: typing.Optional[torch.Tensor] = None

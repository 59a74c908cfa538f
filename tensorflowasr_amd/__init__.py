"""MI355X-native hot path of TensorFlowASR's Conformer-Transducer."""
import os as _os

# HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A data-parallel rank has the main stream, the prediction
# network's stream and RCCL's streams: with 4 queues the prediction network ended up SHARING a hardware queue with the main stream and
# its ~4 ms of small kernels ran in line instead of beside the encoder (27.5 instead of 24.2 ms/step, measured with a one-rank RCCL
# group: `bench.py --dp-hooks`).  The HIP runtime reads the variable when it initialises (the first HIP call of the process), so it is
# set here, at import, and only if the user has not chosen a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

"""TEST INFRASTRUCTURE ONLY (never imported by the product path): torch-CPU restatement of the reference's ContextNet encoder,
tensorflow_asr/models/encoders/contextnet.py, used as the parity oracle of tensorflowasr_amd/contextnet.py.

Parity status: PINNED to the reference's own classes - tests/golden/wiring_contextnet.npz holds per-block outputs, the masks that
reached the squeeze-excite pool / the BatchNorms, the moving statistics and the logits of
`tensorflow_asr.models.transducer.contextnet.ContextNet` CONSTRUCTED and RUN from /root/reference over oracle/tf_shim + oracle/keras_shim
(oracle/gen_wiring_from_reference.py); tests/test_reference_wiring.py checks this file against it block by block and
tests/test_reference_wiring_gpu.py checks the HIP path.  What the Keras library layers themselves compute (SeparableConv1D,
BatchNormalization, GlobalAveragePooling1D, Dense) is [ext], restated twice independently (here with torch, in keras_shim with NumPy):
causal padding = left pad (K-1), VALID, stride s (convolution.py:25-37 semantics for 1-D); BatchNormalization momentum 0.99 /
epsilon 1e-3, training-mode batch statistics over every frame (observed: no mask reaches them); GlobalAveragePooling1D honours the
propagated sequence mask (observed: masked mean); swish = x*sigmoid(x).

Weights: dict name -> tensor with the product's names (params.contextnet_specs):
  {m}/dw [K, Cin] (keras depthwise kernel [K, Cin, 1]), {m}/pw/w [Cin, Cout], {m}/pw/b, {m}/bn/g, {m}/bn/b,
  {blk}se/fc1/w [C, C/8], fc1/b, fc2/w [C/8, C], fc2/b.
"""
import torch
import torch.nn.functional as F


def swish(x):
    return x * torch.sigmoid(x)


def conv_module(x, W, m, K, stride, act, stats=None):
    """ConvModule.call (contextnet.py:76-90): x [B, T, Cin] -> [B, ceil(T/stride), Cout]."""
    B, T, ci = x.shape
    dw = W[m + "/dw"]                                              # [K, Cin]
    xp = F.pad(x.transpose(1, 2), (K - 1, 0))                      # causal: left pad K-1
    y = F.conv1d(xp, dw.t().reshape(ci, 1, K), stride=stride, groups=ci)   # depthwise, VALID
    y = y.transpose(1, 2) @ W[m + "/pw/w"] + W[m + "/pw/b"]        # pointwise 1x1
    mean = y.mean((0, 1))
    var = y.var((0, 1), unbiased=False)
    if stats is not None:
        stats[m + "/bn"] = (mean.detach(), var.detach())
    y = (y - mean) / torch.sqrt(var + 1e-3) * W[m + "/bn/g"] + W[m + "/bn/b"]
    return swish(y) if act == "swish" else y


def se_module(x, lens, W, p):
    """SEModule.call after its conv module (contextnet.py:159-170): masked average pool -> fc1 -> swish -> fc2 -> sigmoid."""
    B, T, C = x.shape
    mask = (torch.arange(T)[None, :] < torch.as_tensor(lens)[:, None]).to(x.dtype)[..., None]
    pool = (x * mask).sum(1) / mask.sum(1).clamp(min=1.0)
    h = swish(pool @ W[p + "fc1/w"] + W[p + "fc1/b"])
    s = torch.sigmoid(h @ W[p + "fc2/w"] + W[p + "fc2/b"])
    return x * s[:, None, :]


def encoder_forward(feats, flen, W, blocks, stats=None):
    """ContextNetEncoder.call (contextnet.py:313-318).  feats [B, T, F]; blocks = params.contextnet_modules(cfg)."""
    x = feats
    lens = [int(n) for n in flen]
    for blk in blocks:
        x0 = x
        for (m, ci, co, K, s, act) in blk["convs"]:
            x = conv_module(x, W, m, K, s, act, stats)
        lens = [-(-n // blk["stride"]) for n in lens]              # conv_output_length(..., "causal") = ceil(L / stride)
        x = se_module(x, lens, W, blk["prefix"] + "se/")
        if blk["res"] is not None:
            m, ci, co, K, s, act = blk["res"]
            x = x + conv_module(x0, W, m, K, s, act, stats)
        x = swish(x)
    return x, lens


def init_weights(specs, seed=0):
    """Random weights with the product's names/shapes (non-trivial affine parameters and biases)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape, _reg, _init, _f in specs:
        if not name.startswith("enc/"):
            continue
        if name.endswith("/bn/g"):
            W[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("/b"):
            W[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            fan = shape[0] if len(shape) == 2 else 1
            W[name] = torch.randn(shape, generator=g) / max(1.0, fan) ** 0.5
    return W

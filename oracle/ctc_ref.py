"""ORACLE (test infrastructure only).  CTC loss / greedy decoding restatement.

Reference: CtcLoss.call -> tf.nn.ctc_loss(labels, logits, label_length, logit_length, logits_time_major=False,
blank_index=0) (tensorflow_asr/losses/ctc_loss.py:47-66) and tf.nn.ctc_greedy_decoder(merge_repeated=True)
(models/ctc/base_ctc.py:102-124).  TensorFlow's kernels are not available here, so the definition is restated twice,
independently: (a) `torch.nn.functional.ctc_loss` (a separate implementation of the same Graves-2006 definition) with
autograd through log_softmax for the gradient, (b) brute-force enumeration of all alignments on tiny cases.
Pinning: **parity unpinned** against TF itself (the reference ships no CTC golden vector); pinned to (a)+(b).
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F


def ctc_loss_and_grad(logits, labels, label_len, logit_len, blank=0, dtype=torch.float64):
    """logits [B,T,V] -> (loss [B], dloss/dlogits [B,T,V])."""
    x = torch.as_tensor(np.asarray(logits)).to(dtype).clone().requires_grad_(True)
    lp = F.log_softmax(x, dim=-1).transpose(0, 1)  # [T,B,V]
    loss = F.ctc_loss(lp, torch.as_tensor(np.asarray(labels)).long(), torch.as_tensor(np.asarray(logit_len)).long(),
                      torch.as_tensor(np.asarray(label_len)).long(), blank=blank, reduction="none", zero_infinity=False)
    loss.sum().backward()
    return loss.detach().numpy(), x.grad.numpy()


def ctc_loss_bruteforce(logits, labels, blank=0):
    """-log sum over all frame-level paths that collapse to `labels` (single sample, tiny shapes)."""
    lp = F.log_softmax(torch.as_tensor(np.asarray(logits), dtype=torch.float64), -1).numpy()
    T, V = lp.shape
    tot = -np.inf
    for path in itertools.product(range(V), repeat=T):
        col, prev = [], None
        for c in path:
            if c != prev and c != blank:
                col.append(c)
            prev = c
        if col == list(labels):
            tot = np.logaddexp(tot, sum(lp[t, c] for t, c in enumerate(path)))
    return -tot


def ctc_greedy_decode(logits, logit_len, blank=0):
    """argmax per frame, merge repeats, drop blanks; dense [B,T] padded with blank + lengths."""
    x = np.asarray(logits)
    B, T, _ = x.shape
    out = np.full((B, T), blank, np.int32)
    lens = np.zeros(B, np.int32)
    am = x.argmax(-1)
    for b in range(B):
        prev, n = -1, 0
        for t in range(int(logit_len[b])):
            c = int(am[b, t])
            if c != prev and c != blank:
                out[b, n] = c
                n += 1
            prev = c
        lens[b] = n
    return out, lens


def ctc_beam_search(logits, T_len, beam_width, blank):
    """Textbook CTC prefix beam search in the log domain (the algorithm of tf.nn.ctc_beam_search_decoder, top path, no
    repeated-label merging beyond CTC's collapse).  logits [T,V] of one utterance -> (labels, log prob)."""
    x = np.asarray(logits, np.float64)[:T_len]
    lp = x - np.logaddexp.reduce(x, axis=1, keepdims=True)
    NEG = -np.inf
    beams = {(): (0.0, NEG)}
    for t in range(T_len):
        nxt = {}

        def add(k, pb=NEG, pnb=NEG):
            a, b_ = nxt.get(k, (NEG, NEG))
            nxt[k] = (np.logaddexp(a, pb), np.logaddexp(b_, pnb))

        for pre, (pb, pnb) in beams.items():
            tot = np.logaddexp(pb, pnb)
            add(pre, pb=tot + lp[t, blank])
            if pre:
                add(pre, pnb=pnb + lp[t, pre[-1]])
            for c in range(lp.shape[1]):
                if c == blank:
                    continue
                src = pb if (pre and pre[-1] == c) else tot
                if src > NEG:
                    add(pre + (c,), pnb=src + lp[t, c])
        ranked = sorted(nxt.items(), key=lambda kv: (-np.logaddexp(*kv[1]), kv[0]))[:beam_width]
        beams = dict(ranked)
    best = min(beams.items(), key=lambda kv: (-np.logaddexp(*kv[1]), kv[0]))
    return list(best[0]), float(np.logaddexp(*best[1]))


def ctc_best_labelling_bruteforce(logits, blank):
    """argmax over LABELLINGS of the total path probability, by enumerating every alignment (tiny T, V only)."""
    import itertools

    x = np.asarray(logits, np.float64)
    lp = x - np.logaddexp.reduce(x, axis=1, keepdims=True)
    T, V = lp.shape
    tot = {}
    for path in itertools.product(range(V), repeat=T):
        lab, prev = [], None
        for c in path:
            if c != blank and c != prev:
                lab.append(c)
            prev = c
        p = sum(lp[t, c] for t, c in enumerate(path))
        k = tuple(lab)
        tot[k] = np.logaddexp(tot.get(k, -np.inf), p)
    best = min(tot.items(), key=lambda kv: (-kv[1], kv[0]))
    return list(best[0]), float(best[1])

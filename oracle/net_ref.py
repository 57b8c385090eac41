"""CPU ORACLE (test infrastructure): float64 NumPy restatement of the reference
network's forward pass (training_pipeline.create_nn, training_pipeline.py:59-114)
on the weights of a checkers_mcts_amd.net.PolicyValueNet state_dict.

The reference's own network arithmetic lives in TensorFlow/Keras, which is not
available here: NN parity versus Keras is UNPINNED.  This module
pins the build's fp32 GPU path against an independent float64 evaluation of
the same architecture and weights (tolerance 1e-5 on pi and v).
"""
import numpy as np


def _conv(x, w, b):
    """x [B,H,W,Cin] float64, w [Cout,Cin,kh,kw] (torch layout), 'same' zero padding."""
    B, H, W, Cin = x.shape
    Cout, _, kh, kw = w.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((B, H + 2 * ph, W + 2 * pw, Cin))
    xp[:, ph:ph + H, pw:pw + W] = x
    out = np.zeros((B, H, W, Cout))
    for i in range(kh):
        for j in range(kw):
            out += xp[:, i:i + H, j:j + W] @ w[:, :, i, j].T
    return out + b


def _bn(x, sd, prefix, eps=1e-3):
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - m) / np.sqrt(v + eps) * g + b


def forward(state_dict, x_nhwc):
    """state_dict: name -> array; x_nhwc [B,8,8,14].  Returns (p [B,512], v [B]) float64."""
    sd = {k: np.asarray(v, np.float64) for k, v in state_dict.items()}
    x = np.asarray(x_nhwc, np.float64)

    def block(x, name):
        y = np.maximum(_conv(x, sd[name + ".conv.weight"], sd[name + ".conv.bias"]), 0.0)
        return _bn(y, sd, name + ".bn")

    for i in range(7):
        x = block(x, "body.%d" % i)
    p = block(block(x, "pol1"), "pol2").reshape(x.shape[0], 512)            # (H, W, C) flatten
    logits = p @ sd["pol_fc.weight"].T + sd["pol_fc.bias"]
    logits -= logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    p = e / e.sum(axis=1, keepdims=True)
    v = block(x, "val1").reshape(x.shape[0], 64)
    v = np.maximum(v @ sd["val_fc1.weight"].T + sd["val_fc1.bias"], 0.0)
    v = _bn(v, sd, "val_bn")
    v = np.tanh(v @ sd["val_fc2.weight"].T + sd["val_fc2.bias"]).reshape(-1)
    return p, v

#!/usr/bin/env python3
"""Calibrates and reports the trained-like weight statistics of fixture g22 (BUILD CONTAINER; TEST INFRASTRUCTURE).

``python oracle/calibrate_trained_like.py roberta-base|recformer-base [--report]``

Calibration: one sequential pass over a 16-sequence synthetic batch with ``qk_layer_gain = 1``; at each layer the std of the valid
pre-softmax logits is measured and the layer's query / key gain set so that it becomes TARGET (q and k both carry the gain, so the
logits scale with its square); the layer is then evaluated WITH the new gain before moving on.  The printed tuple is frozen into
``oracle.ref_cpu.TRAINED_LIKE_QK_GAIN``: weights are a function of (shapes, seed, constants) only.

``--report`` (after the constants are in place): logit sigma / |max| / kurtosis / mean largest probability per layer, hidden-state
outliers, and the spread of user x item cosines on 64 users x 256 items -- the numbers quoted in DESIGN.md §4 for g22.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_cpu as O  # noqa: E402
from mergerec_amd.synthetic import make_domain  # noqa: E402

TARGET = 4.0
SEED = {"roberta-base": 2200, "recformer-base": 2300}


def family(name):
    if name == "roberta-base":
        cfg = O.EncoderConfig()
        return cfg, O.roberta_param_shapes(cfg), "roberta"
    cfg = O.EncoderConfig(max_pos=4098, token_type_size=4, max_item_embeddings=51, one_sided_window=32)
    return cfg, O.recformer_param_shapes(cfg), "recformer"


def embed(sd, e, cfg, kind):
    if kind == "roberta":
        return O.roberta_embeddings(sd, e["input_ids"], cfg, "model."), e["attention_mask"]
    x = O.recformer_embeddings(sd, e["input_ids"], e["token_type_ids"], e["item_position_ids"], cfg, "model.")
    return x, e["attention_mask"] * (e["global_attention_mask"] + 1)


def layer_fn(sd, lp, x, mask, cfg, kind):
    return O.roberta_layer(sd, lp, x, mask, cfg) if kind == "roberta" else O.longformer_layer(sd, lp, x, mask, cfg)


def logits_of(sd, lp, x, mask, cfg):
    B, L, d = x.shape
    H, dh = cfg.heads, d // cfg.heads
    q = F.linear(x, sd[lp + "attention.self.query.weight"], sd[lp + "attention.self.query.bias"]).view(B, L, H, dh).transpose(1, 2)
    k = F.linear(x, sd[lp + "attention.self.key.weight"], sd[lp + "attention.self.key.bias"]).view(B, L, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    m = mask > 0
    valid = m[:, None, :, None] & m[:, None, None, :]
    if cfg.one_sided_window:
        idx = torch.arange(L)
        valid = valid & ((idx[:, None] - idx[None, :]).abs() <= cfg.one_sided_window)[None, None]
    return s, valid.expand_as(s)


def calibrate(name):
    cfg, shapes, kind = family(name)
    sd = O.trained_like_state_dict(shapes, SEED[name], cfg, (1.0,) * cfg.layers)
    cal = make_domain("Cal", 8, 16, 16, cfg.vocab, 777, kind=kind)
    e = cal.sequence_batches[0].sequence
    gains = []
    with torch.no_grad():
        x, mask = embed(sd, e, cfg, kind)
        for l in range(cfg.layers):
            lp = f"model.encoder.layer.{l}."
            s, valid = logits_of(sd, lp, x, mask, cfg)
            gains.append(round(float((TARGET / s[valid].std()).sqrt()), 3))
            for n in ("query", "key", "query_global", "key_global"):
                if lp + f"attention.self.{n}.weight" in sd:
                    sd[lp + f"attention.self.{n}.weight"] *= gains[-1]
                    sd[lp + f"attention.self.{n}.bias"] *= gains[-1]
            x = layer_fn(sd, lp, x, mask, cfg, kind)
    print(f'    "{name}": {tuple(gains)},')


def report(name):
    cfg, shapes, kind = family(name)
    t0 = time.time()
    sd = O.trained_like_state_dict(shapes, SEED[name], cfg, O.TRAINED_LIKE_QK_GAIN[name])
    dom = make_domain("Pantry", 256, 64, 32, cfg.vocab, 20262, kind=kind)

    def enc(batches, key):
        outs, keep = [], None
        with torch.no_grad():
            for b in batches:
                e = getattr(b, key)
                x, mask = embed(sd, e, cfg, kind)
                hidden = [x]
                for l in range(cfg.layers):
                    x = layer_fn(sd, f"model.encoder.layer.{l}.", x, mask, cfg, kind)
                    hidden.append(x)
                outs.append(x[:, 0])
                keep = (hidden, mask)
        return O.maybe_normalize(torch.cat(outs)), keep

    E, _ = enc(dom.item_batches, "items")
    U, (hidden, mask) = enc(dom.sequence_batches, "sequence")
    S = U @ E.T
    q = lambda t, p: float(t.flatten().kthvalue(max(1, int(p * t.numel()))).values)
    print(f"[{name}] cosines: min {float(S.min()):.3f}  p5 {q(S, .05):.3f}  median {q(S, .5):.3f}  p95 {q(S, .95):.3f}  max {float(S.max()):.3f}")
    top = torch.topk(S, 51, dim=1).values
    gap = top[:, :-1] - top[:, 1:]
    print(f"  top-50 gaps: min {float(gap.min()):.2e}  median {float(gap.median()):.2e}  share below 4e-6: {float((gap < 4e-6).float().mean()):.4f}")
    for l in range(cfg.layers):
        s, valid = logits_of(sd, f"model.encoder.layer.{l}.", hidden[l], mask, cfg)
        sv = s[valid]
        p = torch.softmax(s.masked_fill(~valid, float("-inf")), -1)
        rows = (mask > 0)[:, None, :].expand(s.shape[:3])
        pmax = p.max(-1).values[rows]
        x = hidden[l]
        print(f"  layer {l:2d}: logits sigma {float(sv.std()):.2f}  |max| {float(sv.abs().max()):5.1f}  kurtosis {float(((sv - sv.mean()) ** 4).mean() / sv.var() ** 2):5.1f}  "
              f"largest p: mean {float(pmax.mean()):.3f} p95 {q(pmax, .95):.3f}   hidden |max| {float(x.abs().max()):6.1f} rms {float(x.pow(2).mean().sqrt()):.2f}")
    cls = hidden[-1][:, 0]
    v, i = cls.abs().mean(0).topk(6)
    print(f"  final CLS: rms {float(cls.pow(2).mean().sqrt()):.2f}, largest mean-|x| dims {i.tolist()} = {[round(float(t), 2) for t in v]}; "
          f"common-mode share of its energy {float(cls.mean(0).pow(2).sum() / cls.pow(2).sum(1).mean()):.3f}   ({time.time() - t0:.0f}s)")


if __name__ == "__main__":
    torch.set_num_threads(8)
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["roberta-base", "recformer-base"]
    for n in names:
        (report if "--report" in sys.argv else calibrate)(n)

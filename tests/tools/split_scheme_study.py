#!/usr/bin/env python3
"""Precision study (CPU, test infrastructure): how far do candidate split-precision GEMM arithmetics move the final cosine logits?

Every ``F.linear`` of the oracle's BLaIR-base forward is replaced by an emulation of the candidate scheme whose partial products are
summed in float64 (so only the scheme's own operand rounding shows), and the normalised CLS embeddings / logits of a sample of the
g12 real-scale domain are compared with an all-float64 run.
  bf16x3   : x = hi + lo (bf16 pieces), hi*hi + hi*lo + lo*hi                       (the library's three-product arithmetic)
  f16i8    : x = fp16(x) + xl;  fp16*fp16  +  int8(x)*int8(wl) + int8(xl)*int8(w)   per-row scales over the whole K, xl / wl scale = 2^-11 of it
  f16i8c256: the same with scales per (row, 256-wide K chunk)
  f16mx8   : the same split with the cross-term operands in MX-fp8: e4m3 elements, one power-of-two scale per (row, 32 k) block
             (what v_mfma_scale_f32_32x32x64_f8f6f4 applies in hardware, so ONE f32 accumulator serves all three products)
Usage: python tests/tools/split_scheme_study.py [n_users] [n_items] [modes...]"""
import sys
from collections import OrderedDict
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import ref_cpu as O  # noqa: E402
from mergerec_amd.synthetic import make_domain  # noqa: E402
from tests.conftest import load_golden  # noqa: E402

_orig_linear = F.linear


def _bf16_split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi.double(), lo.double()


def _i8(t, scale):
    return torch.clamp(torch.round(t / scale), -127, 127)


def _mx8(t):
    """MX-fp8 round trip: blocks of 32 along the last dim share a power-of-two scale chosen so the block maximum lands in e4m3's top
    binade (<= 448); elements are rounded to e4m3 (torch.float8_e4m3fn, round-to-nearest-even, saturating)."""
    shp = t.shape
    b = t.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    e = torch.floor(torch.log2(amax)) - 8.0          # e4m3 max 448 = 1.75 * 2^8: block max maps into [2^8, 2^9)
    scale = torch.exp2(e)
    q = (b / scale).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).reshape(shp)


def make_linear(mode):
    def linear(x, w, b=None):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float()
        w2 = w.float()
        if mode == "f64":
            y = x2.double() @ w2.double().T
        elif mode == "bf16x3":
            xh, xl = _bf16_split(x2)
            wh, wl = _bf16_split(w2)
            y = xh @ wh.T + xh @ wl.T + xl @ wh.T
        elif mode == "f16mx8":
            xh = x2.to(torch.float16).float(); xl = x2 - xh
            wh = w2.to(torch.float16).float(); wl = w2 - wh
            y = xh.double() @ wh.double().T + _mx8(x2).double() @ _mx8(wl).double().T + _mx8(xl).double() @ _mx8(w2).double().T
        else:
            chunk = 256 if mode.endswith("c256") else x2.shape[1]
            xh = x2.to(torch.float16).float(); xl = x2 - xh
            wh = w2.to(torch.float16).float(); wl = w2 - wh
            y = xh.double() @ wh.double().T
            for k0 in range(0, x2.shape[1], chunk):
                xs, xls = x2[:, k0:k0 + chunk], xl[:, k0:k0 + chunk]
                ws, wls = w2[:, k0:k0 + chunk], wl[:, k0:k0 + chunk]
                sx = xs.abs().amax(1, keepdim=True).clamp_min(1e-30) / 127
                sw = ws.abs().amax(1, keepdim=True).clamp_min(1e-30) / 127
                xq, xlq = _i8(xs, sx), _i8(xls, sx * 2.0 ** -11)
                wq, wlq = _i8(ws, sw), _i8(wls, sw * 2.0 ** -11)
                acc = xq.double() @ wlq.double().T + xlq.double() @ wq.double().T   # exact integers
                y = y + acc * (sx.double() * sw.double().T * 2.0 ** -11)
        y = y.float()
        if b is not None:
            y = y + b
        return y.reshape(*shp[:-1], w.shape[0])
    return linear


def main():
    n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    n_items = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    torch.set_num_threads(8)
    fx = load_golden("g12_realscale_blair_base.pt")
    cfg = O.EncoderConfig()
    pre0 = O.random_state_dict(O.roberta_param_shapes(cfg), seed=fx["seed_pre"], std=0.02)
    pre = OrderedDict((k, pre0[k]) for k in fx["key_order"])
    fts = [O.perturbed_state_dict(pre, seed=s, std=fx["ft_std"]) for s in fx["seed_ft"]]
    sd = OrderedDict((k, pre[k] + 0.5 * ((fts[0][k] - pre[k]) + (fts[1][k] - pre[k]))) for k in pre)  # alpha = 0.5 (rounding detail irrelevant here)
    dom = make_domain("Pantry", fx["n_items"], fx["n_users"], 32, cfg.vocab, fx["seed_domain"])
    ub = [b.sequence for b in dom.sequence_batches[: (n_users + 31) // 32]]
    ib = [b.items for b in dom.item_batches[: (n_items + 31) // 32]]
    res = {}
    modes = sys.argv[3:] or ["bf16x3", "f16i8", "f16i8c256", "f16mx8"]
    for mode in ["f64"] + modes:
        O.F.linear = make_linear(mode)
        try:
            with torch.no_grad():
                U = torch.cat([O.maybe_normalize(O.roberta_encode(sd, b["input_ids"], b["attention_mask"], cfg, "model.")) for b in ub])[:n_users]
                E = torch.cat([O.maybe_normalize(O.roberta_encode(sd, b["input_ids"], b["attention_mask"], cfg, "model.")) for b in ib])[:n_items]
        finally:
            O.F.linear = _orig_linear
        res[mode] = (U.double(), E.double())
        if mode != "f64":
            U0, E0 = res["f64"]
            du, de = (U.double() - U0).abs().max().item(), (E.double() - E0).abs().max().item()
            dl = (U.double() @ E.double().T - U0 @ E0.T).abs()
            print(f"{mode:10s} user emb {du:.2e}  item emb {de:.2e}  logits max {dl.max().item():.2e} rms {dl.pow(2).mean().sqrt().item():.2e}", flush=True)
        else:
            print("f64 reference done", flush=True)


if __name__ == "__main__":
    main()

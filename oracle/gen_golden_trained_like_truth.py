#!/usr/bin/env python3
"""fp64 ground truth for a slice of fixture g22 (BUILD CONTAINER; TEST INFRASTRUCTURE): ``python oracle/gen_golden_trained_like_truth.py blair|recformer``.

On trained-like weights two fp32 implementations of the same encoder (transformers' and oracle/ref_cpu.py's) already differ by several
1e-5 in the logits -- summation order amplified by peaky softmax rows and 100x hidden-state outliers.  To tell how much of a GPU
arithmetic's distance from the REFERENCE is the reference's own rounding, the restatement is evaluated in float64 on the first four user
batches and eight item batches of the fixture's domain (same merged weights: the oracle's merge is bit-equal to the reference's, checked
by checksum) and stored beside the reference's outputs as ``truth64`` = dict(U, E, users, items).  The test then reports, per arithmetic,
the distance to the reference AND to the truth, next to the reference's own distance to the truth."""
import sys
import time
from collections import OrderedDict
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ref_cpu as O  # noqa: E402
from mergerec_amd.synthetic import make_domain  # noqa: E402

NB_U, NB_I = 4, 8


def main():
    family = sys.argv[1]
    torch.set_num_threads(8)
    path = ROOT / "tests" / "golden" / f"g22_trained_like_{family}_base.pt"
    fx = torch.load(path, weights_only=False)
    rec = family == "recformer"
    cfg = O.EncoderConfig(max_pos=4098, token_type_size=4, max_item_embeddings=51, one_sided_window=32) if rec else O.EncoderConfig()
    shapes = O.recformer_param_shapes(cfg) if rec else O.roberta_param_shapes(cfg)
    t0 = time.time()
    pre0 = O.trained_like_state_dict(shapes, fx["seed_pre"], cfg, O.TRAINED_LIKE_QK_GAIN[fx["gain_key"]])
    pre = OrderedDict((k, pre0[k]) for k in fx["key_order"])
    fts = [O.perturbed_state_dict(pre, seed=s, std=fx["ft_std"]) for s in fx["seed_ft"]]
    base, shape_dict = O.flatten_model(pre)
    tv = O.get_task_vectors(base, [O.flatten_model(ft)[0] for ft in fts])
    merged = O.get_state_dict(O.merge_task_wise(base, tv, torch.tensor(fx["alphas"])), shape_dict)
    fsum = float(sum(v.double().sum() for v in merged.values() if v.is_floating_point()))  # (Recformer's position_ids is fp32 in the flat vector)
    assert abs(fsum - fx["merged_checksum"]) < 1e-9 * max(1.0, abs(fx["merged_checksum"])) + 1e-5, (fsum, fx["merged_checksum"])
    m64 = OrderedDict((k, v.double() if not k.endswith("position_ids") else v) for k, v in merged.items())
    dom = make_domain("Pantry", fx["n_items"], fx["n_users"], 32, cfg.vocab, fx["seed_domain"], kind="recformer" if rec else "roberta", max_seq_len=fx["max_seq_len"])

    def enc(b):
        if rec:
            return O.recformer_encode(m64, b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"], b["item_position_ids"], cfg, "model.")
        return O.roberta_encode(m64, b["input_ids"], b["attention_mask"], cfg, "model.")

    with torch.no_grad():
        U = O.maybe_normalize(torch.cat([enc(b.sequence) for b in dom.sequence_batches[:NB_U]]))
        E = O.maybe_normalize(torch.cat([enc(b.items) for b in dom.item_batches[:NB_I]]))
    nu, ni = U.shape[0], E.shape[0]
    fx["truth64"] = dict(U=U, E=E, users=nu, items=ni)
    assert nu <= fx["U"].shape[0]
    ref_u = float((fx["U"][:nu].double() - U).abs().max())
    print(f"[{family}] fp64 truth on {nu} users x {ni} items in {time.time() - t0:.0f}s; reference (fp32, transformers) |dU| vs truth {ref_u:.2e}")
    fx["reference_vs_truth64"] = dict(user_max_abs_diff=ref_u)
    torch.save(fx, path)


if __name__ == "__main__":
    main()

"""Error attribution on fixture g22 (BLaIR-base, trained-like weights): the encoder on the first 128 users / 256 items in every combination
of linear arithmetic x attention arithmetic, distance of the logits to the float64 truth stored in the fixture (and to the reference on the
rows the fixture keeps).   PYTHONPATH=. python tests/tools/trained_like_attribution.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_trained_like_gpu import _build_state_dicts  # noqa: E402

DEV = "cuda:0"


def main():
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.module import ModelType
    from mergerec_amd.synthetic import make_domain

    fx, cfg, rec, pre, fts = _build_state_dicts("blair")
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "device": DEV})
    model.load_state_dict(pre)
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, pre, fts, set(), disable_softmax=True)
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": list(fx["alphas"])}})
    sd = {k: v.detach().clone() for k, v in mm.get_state_dict().items()}
    del mm, model
    dom = make_domain("Pantry", fx["n_items"], fx["n_users"], 32, cfg.vocab, fx["seed_domain"], max_seq_len=fx["max_seq_len"])
    t = fx["truth64"]
    truth = t["U"] @ t["E"].T
    rows = fx["E_rows"].long()
    tr = rows[rows < t["items"]]
    ref = fx["U"][: t["users"]] @ fx["E_sample"][: tr.numel()].T
    print(f"reference (fp32 transformers) vs float64 on {t['users']} x {tr.numel()}: {float((ref.double() - truth[:, tr]).abs().max()):.2e}")
    modes = sys.argv[1:] or ["f32", "bf16x6", "bf16x3"]
    for lin in modes:
        for att in ("0", "6", "3"):
            os.environ["MERGEREC_ATTN_PRODUCTS"] = att
            m = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "device": DEV, "gemm_mode": lin})
            m.load_state_dict(sd)
            U = torch.cat([m.encode_normalized(b.sequence, True) for b in dom.sequence_batches[:4]]).cpu()
            E = torch.cat([m.encode_normalized(b.items, True) for b in dom.item_batches[:8]]).cpu()
            S = (U.double() @ E.double().T)
            print(f"linears {lin:7s} attention {'f32' if att == '0' else 'x' + att:4s}: logits vs float64 max {float((S - truth).abs().max()):.2e}  rms {float((S - truth).pow(2).mean().sqrt()):.2e}; "
                  f"|dU| {float((U.double() - t['U']).abs().max()):.2e} |dE| {float((E.double() - t['E']).abs().max()):.2e}; vs reference {float((S[:, tr] - ref.double()).abs().max()):.2e}", flush=True)
            del m
    os.environ.pop("MERGEREC_ATTN_PRODUCTS", None)


if __name__ == "__main__":
    main()

"""Run BASELINE.json configs[1] and configs[2] end to end through the drop-in API on one MI355X with synthetic weights/data
at the TRUE architecture dims, time every stage, and cross-check a sample against the CPU oracle.

  cfg2: 2-domain merge BLaIR-base, fixed alpha = 0.5 (TASK_VECTOR / TASK_WISE)
  cfg3: 3-domain merge Recformer-base, per-layer-group alpha (TIES / LAYER_WISE, the published recipe of scripts/3_mergerec)

python tests/tools/run_configs.py [--users 2048] [--items 4968] [--out profiles/r01_configs.json]
"""
import argparse, json, os, sys, time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mergerec_amd.data import load_domain
from mergerec_amd.evaluator import Evaluator
from mergerec_amd.merger import LearnType, MergeType, load_merging_module
from mergerec_amd.module import ModelType, RecModule
from mergerec_amd.utils import test_model_on_dataloaders as test_model
from oracle import ref_cpu as O


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def run(name, model_type, merge_type, learn_type, n_dom, alpha_fn, args):
    dev = "cuda:0"
    t0 = sync()
    mk = {"init_seed": 1000, "device": dev}
    model = ModelType[model_type].value(model_kwargs=dict(mk))
    pre = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())
    fts = []
    for i in range(n_dom):
        g = torch.Generator().manual_seed(1001 + i)
        fts.append(OrderedDict((k, v if k.endswith("position_ids") else v + 1e-3 * torch.randn(v.shape, generator=g)) for k, v in pre.items()))
    t_ckpt = sync() - t0
    t0 = sync()
    mm = load_merging_module(MergeType[merge_type], LearnType[learn_type], model, pre, fts, set(), ties_density=0.2, disable_softmax=True)
    t_init = sync() - t0
    groups = list(mm.per_weights.keys())
    weights = {"global_weights": {k: [1.0] for k in groups}, "global_biases": {k: [0.0] for k in groups},
               "per_weights": {k: alpha_fn(k) for k in groups}}
    mm.load_weights_from_dict(weights)
    t0 = sync()
    sd = {k: v.detach() for k, v in mm.get_state_dict().items()}
    t_merge = sync() - t0
    model2 = ModelType[model_type].value(model_kwargs=dict(mk))
    model2.load_state_dict(sd)
    module = RecModule(model=model2, evaluator=Evaluator(["NDCG", "RECALL"], [1, 5, 10, 50]), similarity="cosine")
    kind = "recformer" if model_type.startswith("RECFORMER") else "roberta"
    dom = load_domain(f"synthetic:Cfg:{args.items}:{args.users}", kind=kind, vocab=model2.spec.vocab, seed=77)
    t0 = sync()
    _, metrics, _, labels = test_model(module, [dom.item_dataloader(32)], [dom.sequence_dataloader(32)], [name])
    t_eval = sync() - t0
    # ---- oracle cross-check on a sample (merge bit-exactness + embeddings)
    base, shape_dict = O.flatten_model(pre)
    models = [O.flatten_model(OrderedDict((k, ft[k]) for k in pre))[0] for ft in fts]
    tvs = O.ties_vectors(base, models, 0.2) if merge_type == "TIES" else O.get_task_vectors(base, models)
    if learn_type == "LAYER_WISE":
        grp = O.group_parameters_by_layer(shape_dict)
        merged = O.merge_layer_wise(base, tvs, grp, {k: torch.tensor(weights["per_weights"][k]) for k in grp})
    else:
        merged = O.merge_task_wise(base, tvs, torch.tensor(weights["per_weights"]["all"]))
    bit_exact = bool(torch.equal(torch.cat([v.reshape(-1) for v in sd.values()]).cpu(), merged))
    osd = O.get_state_dict(merged, shape_dict)
    spec = model2.spec
    cfg = O.EncoderConfig(hidden=spec.hidden, heads=spec.heads, layers=spec.layers, intermediate=spec.intermediate, vocab=spec.vocab,
                          max_pos=spec.max_pos, token_type_size=spec.token_type_size, max_item_embeddings=spec.max_item_embeddings,
                          one_sided_window=max(spec.one_sided_window, 0))
    nb = args.oracle_seqs
    b = {k: v[:nb] for k, v in dom.sequences.items()}
    L = int(b["attention_mask"].sum(1).max())
    b = {k: v[:, :L] for k, v in b.items()}
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    with torch.no_grad():
        if kind == "recformer":
            u = O.recformer_encode(osd, b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"], b["item_position_ids"], cfg, "model.")
        else:
            u = O.roberta_encode(osd, b["input_ids"], b["attention_mask"], cfg, "model.")
    u = O.maybe_normalize(u)
    diff = float((module.eval_user_embeddings[:nb] - u).abs().max())
    E = module.item_embeddings.detach().cpu()
    logit_diff = float((module.eval_user_embeddings[:nb] @ E.T - u @ E.T).abs().max())
    out = dict(config=name, model_type=model_type, merge_type=merge_type, learn_type=learn_type, domains=n_dom, params=mm.layout.numel,
               items=dom.n_items, users=dom.n_users, seconds=dict(synthesize_checkpoints=t_ckpt, load_merging_module=t_init, merge=t_merge, test_model=t_eval),
               sequences_per_s_test_model=dom.n_users / t_eval, metrics={k: round(v, 5) for k, v in metrics[0].items()},
               parity=dict(merged_params_bit_exact=bit_exact, user_embedding_max_abs_diff=diff, logit_max_abs_diff=logit_diff, tolerance=1e-4, oracle_sequences=nb))
    del mm, model, model2, module
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=2048)
    ap.add_argument("--items", type=int, default=4968)
    ap.add_argument("--oracle-seqs", type=int, default=4)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(7)
    res = [
        run("cfg2: 2-domain BLaIR-base, alpha=0.5", "BLAIR_BASE", "TASK_VECTOR", "TASK_WISE", 2, lambda k: [0.5, 0.5], args),
        run("cfg3: 3-domain Recformer-base, TIES + per-layer-group alpha", "RECFORMER_BASE", "TIES", "LAYER_WISE", 3,
            lambda k: (0.1 + 0.5 * torch.rand(3, generator=g)).tolist(), args),
    ]
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt)


if __name__ == "__main__":
    main()

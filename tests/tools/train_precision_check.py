"""Gradient accuracy of the two training-graph arithmetics ("f32": exact-fp32 products, "bf16x3": split-precision MFMA products) against
float64 autograd through the CPU oracle, on a tiny BLaIR (2 x 128, 2 heads) in-batch fine-tuning step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_cpu as O
from mergerec_amd.configs import NegativeSampleConfig
from mergerec_amd.evaluator import Evaluator
from mergerec_amd.model_batch import BatchSequenceWithNegative
from mergerec_amd.module import ModelType, RecModule

DEV = "cuda:0"
over = dict(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=130)
model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 5, "spec_overrides": over, "device": DEV, "gemm_mode": "f32"})
mod = RecModule(model=model, evaluator=Evaluator(["NDCG"], [10]), negative_sample=NegativeSampleConfig(in_batch=True), similarity="cosine")
mod.train()
g = torch.Generator().manual_seed(8)


def toks(B, L):
    lens = torch.randint(3, L + 1, (B,), generator=g)
    ids = torch.randint(3, 300, (B, L), generator=g)
    ids[:, 0] = 0
    mask = (torch.arange(L).view(1, L) < lens.view(B, 1)).long()
    return {"input_ids": ids * mask + (1 - mask), "attention_mask": mask}


seq, tgt = toks(24, 100), toks(24, 20)
cfg = O.EncoderConfig(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=130)
p = {k: v.detach().cpu().double().requires_grad_(not k.endswith("position_ids")) for k, v in model.state_dict().items()}
u = O.maybe_normalize(O.roberta_encode(p, seq["input_ids"], seq["attention_mask"], cfg, prefix="model."))
t = O.maybe_normalize(O.roberta_encode(p, tgt["input_ids"], tgt["attention_mask"], cfg, prefix="model."))
s, l = O.negative_sample_scores(u, t, None, "IN_BATCH", None)
ref = O.finetune_loss(s, l, 0.05)
ref.backward()
layout = model._weights.layout
want = torch.zeros(layout.padded_numel, dtype=torch.float64)
for k, v in p.items():
    if v.grad is not None:
        o = layout.offsets[k]
        want[o:o + v.numel()] = v.grad.reshape(-1)
batch = BatchSequenceWithNegative(sequence=seq, target=tgt).to(DEV)
leaf = model.train_leaf()
for mode in ("f32", "bf16x3"):
    model.train_mode = mode
    leaf.grad = None
    loss = mod.training_step(batch, 0)
    loss.backward()
    got = leaf.grad.cpu().double()
    print(f"{mode:7s} loss {float(loss.detach()):.7f} (float64 oracle {float(ref):.7f}); gradient: max |err| / max |g| = "
          f"{float((got - want).abs().max() / want.abs().max()):.2e}, ||err|| / ||g|| = {float((got - want).norm() / want.norm()):.2e}")

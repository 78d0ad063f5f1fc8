"""One fine-tuning step of scripts/1_finetune/blair_base.sh's shape (batch 64 sequences + 64 target items, in-batch negatives,
gradient accumulation 4) at BLaIR-base / Recformer scale on synthetic Amazon-shaped tokens: ms per micro-step (forward + backward),
ms per optimizer step (fused clip + AdamW), the per-kernel device-time table, and with FB_CPU=1 the same micro-step through the CPU
oracle + torch autograd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mergerec_amd import ops
from mergerec_amd.configs import NegativeSampleConfig
from mergerec_amd.evaluator import Evaluator
from mergerec_amd.model_batch import BatchSequenceWithNegative
from mergerec_amd.module import ModelType, RecModule
from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths, _ids_from_lengths

DEV = "cuda:0"
MODEL = os.environ.get("FB_MODEL", "BLAIR_BASE")
B, ACC, K = int(os.environ.get("FB_B", 64)), int(os.environ.get("FB_ACC", 4)), int(os.environ.get("FB_STEPS", 8))
torch.manual_seed(0)
model = ModelType[MODEL].value(model_kwargs={"init_seed": 7})
model.train_mode = os.environ.get("FB_MODE", "bf16x3")  # the reference recipe's precision is bf16-mixed
REC = MODEL.startswith("RECFORMER")


def recformer_fields(enc, item_len=38):
    """token types (0 <s>, 1 attribute-name tokens = the first 3 of every item, 2 value tokens, 3 padding), item positions 1.. (50 max) and
    the global mask on <s> for a batch that came out of _ids_from_lengths (sequence = <s> + items, recformer_utils.py:45-68)"""
    ids, mask = enc["input_ids"], enc["attention_mask"]
    L = ids.shape[1]
    pos = torch.arange(L).view(1, L).expand_as(ids)
    within = (pos - 1) % item_len
    tt = torch.where(within < 3, torch.ones_like(ids), torch.full_like(ids, 2))
    tt[:, 0] = 0
    tt = torch.where(mask.bool(), tt, torch.full_like(ids, 3))
    ip = (1 + (pos - 1) // item_len).clamp(max=50)
    ip[:, 0] = 0
    ip = ip * mask
    ga = torch.zeros_like(ids)
    ga[:, 0] = 1
    return dict(enc, token_type_ids=tt, item_position_ids=ip, global_attention_mask=ga)

mod = RecModule(model=model, evaluator=Evaluator(["NDCG"], [10]), negative_sample=NegativeSampleConfig(in_batch=True), similarity="cosine",
                temperature=0.05, learning_rate=5e-5, warmup_steps=100, weight_decay=0.0)
mod.trainer = type("Tr", (), {"estimated_stepping_batches": 10000, "gradient_clip_val": float(os.environ.get("FB_CLIP", 1.0))})()
opt = mod.configure_optimizers()
leaf = model.train_leaf()
mod.train()
g = torch.Generator().manual_seed(1)
batches = []
for _ in range(ACC):
    ul, il = blair_sequence_lengths(B, g), blair_item_lengths(B, g)
    seq_e, tgt_e = _ids_from_lengths(ul, model.spec.vocab, g), _ids_from_lengths(il, model.spec.vocab, g)
    if REC:
        seq_e, tgt_e = recformer_fields(seq_e), recformer_fields(tgt_e)
    batches.append((BatchSequenceWithNegative(sequence=seq_e, target=tgt_e).to(DEV), int(ul.sum() + il.sum())))
tokens = sum(t for _, t in batches) / ACC


def micro(i):
    loss = mod.training_step(batches[i][0], i)
    (loss / ACC).backward()
    return loss


def step():
    for i in range(ACC):
        loss = micro(i)
    opt.step(leaf.grad)
    leaf.grad = None
    model.arena_changed()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / K
P = model._weights.layout.padded_numel
# optimizer alone
torch.cuda.synchronize()
gfake = torch.randn_like(leaf.detach()) * 1e-3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    opt.step(gfake)
e1.record()
torch.cuda.synchronize()
opt_ms = e0.elapsed_time(e1) / 10
sp = model.spec
per_tok = 2.0 * sp.layers * (4 * sp.hidden * sp.hidden + 2 * sp.hidden * sp.intermediate + (2 * sp.hidden * sp.hidden if REC else 0))  # + key/value_global
print(f"{MODEL} [{model.train_mode}]: batch {B} sequences + {B} targets ({tokens:.0f} tokens / micro-step), accumulation {ACC}, P = {P/1e6:.1f} M: "
      f"{ms:.1f} ms / optimizer step = {ms/ACC:.1f} ms / micro-step ({B*ACC*1e3/ms:.0f} sequences/s, {3*per_tok*tokens*ACC/ms/1e9:.0f} TFLOP/s of linear-map math); loss {float(loss.detach()):.4f}")
print(f"  clip + AdamW over the arena: {opt_ms:.3f} ms (sum of squares 4 B + step 28 B per parameter -> {32*P/opt_ms/1e9:.2f} TB/s)")
if os.environ.get("FB_CPU", "0") == "1":
    from oracle import ref_cpu as O
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    b = batches[0][0].to("cpu")
    t0 = time.perf_counter()
    assert not REC, "FB_CPU: BLaIR only"
    u = O.maybe_normalize(O.roberta_encode(p, b.sequence["input_ids"], b.sequence["attention_mask"], O.EncoderConfig(), prefix="model."))
    t = O.maybe_normalize(O.roberta_encode(p, b.target["input_ids"], b.target["attention_mask"], O.EncoderConfig(), prefix="model."))
    s, l = O.negative_sample_scores(u, t, None, "IN_BATCH", None)
    ref = O.finetune_loss(s, l, 0.05)
    ref.backward()
    cpu_s = time.perf_counter() - t0
    print(f"  CPU oracle + torch autograd ({torch.get_num_threads()} threads): {cpu_s*1e3:.0f} ms / micro-step -> x{cpu_s*1e3/(ms/ACC):.0f}; loss {float(ref):.4f}")
if os.environ.get("FB_PROFILE", "1") == "1":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    rows = sorted(((e.key, e.device_time_total / 1e3, e.count) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[1])
    tot = sum(r[1] for r in rows)
    print(f"  device time per optimizer step {tot:.2f} ms over {sum(r[2] for r in rows)} launches:")
    for k, t, c in rows[:18]:
        print(f"    {t:8.3f} ms  x{c:<5d} {k[:110]}")

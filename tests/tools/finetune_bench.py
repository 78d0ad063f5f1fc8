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
if MODEL.startswith("RECFORMER"):
    raise SystemExit("synthetic Recformer batches: use tests/tools/train_bench.py's generator (not wired here)")
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
    batches.append((BatchSequenceWithNegative(sequence=_ids_from_lengths(ul, model.spec.vocab, g), target=_ids_from_lengths(il, model.spec.vocab, g)).to(DEV),
                    int(ul.sum() + il.sum())))
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
flops = 3 * 2 * 7.08e6 * 12 * tokens  # forward + 2x backward of the 12 layers' linear maps (attention extra)
print(f"{MODEL} [{model.train_mode}]: batch {B} sequences + {B} targets ({tokens:.0f} tokens / micro-step), accumulation {ACC}, P = {P/1e6:.1f} M: "
      f"{ms:.1f} ms / optimizer step = {ms/ACC:.1f} ms / micro-step ({B*ACC*1e3/ms:.0f} sequences/s, {3*2*7.08e6*12*tokens*ACC/ms/1e9:.0f} TFLOP/s of linear-map math); loss {float(loss.detach()):.4f}")
print(f"  clip + AdamW over the arena: {opt_ms:.3f} ms (sum of squares 4 B + step 28 B per parameter -> {32*P/opt_ms/1e9:.2f} TB/s)")
if os.environ.get("FB_CPU", "0") == "1":
    from oracle import ref_cpu as O
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    b = batches[0][0].to("cpu")
    t0 = time.perf_counter()
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

"""One collaborative-merging optimisation step at BLaIR-base scale (scripts/3_mergerec/blair_base_taskvector_taskwise.sh shape):
8 fine-tuned checkpoints, batch of 16 pseudo-user sequences (item texts, ~40 tokens), 8 catalogs of M items, SINGLE_PSEUDO_LABEL_KD.
Reports ms/step with a per-stage breakdown (HIP events) and, with TB_CPU=1, the same step through the CPU oracle + torch autograd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collections import OrderedDict
import torch
from mergerec_amd import ops
from mergerec_amd.merger import LearnType, MergeType, load_merging_module
from mergerec_amd.model_batch import BatchDistillationSequence
from mergerec_amd.module import DistillSequenceModule, ModelType
from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss
from mergerec_amd.synthetic import blair_item_lengths

DEV = "cuda:0"
N, B, M = int(os.environ.get("TB_N", 8)), int(os.environ.get("TB_B", 16)), int(os.environ.get("TB_M", 22855))
learn = os.environ.get("TB_LEARN", "TASK_WISE")
torch.manual_seed(0)
MODEL = os.environ.get("TB_MODEL", "BLAIR_BASE")
model = ModelType[MODEL].value(model_kwargs={"init_seed": 7})
REC = MODEL.startswith("RECFORMER")
pre = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())
fts = []
for i in range(N):
    g = torch.Generator().manual_seed(100 + i)
    fts.append(OrderedDict((k, v if k.endswith("position_ids") else v + 1e-3 * torch.randn(v.shape, generator=g)) for k, v in pre.items()))
mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[learn], model, pre, fts, set(), disable_softmax=True, initial_per_weight=0.2)
mm.train_mode = os.environ.get("TB_MODE", "f32")
del fts
g = torch.Generator().manual_seed(1)
D = model.spec.hidden
items = [torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1) for _ in range(N)]
teachers = [torch.randn(64, M, generator=g).clamp(-1, 1) for _ in range(N)]  # 64 teacher rows per domain are enough for the step
mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(0.05, 1000.0), "cosine",
                            trainable_args_kwargs={"freeze_global_weight": True, "freeze_global_bias": True})
mod.item_embeddings = items
lens = blair_item_lengths(B, g)
L = int(lens.max())
ids = torch.full((B, L), 1, dtype=torch.int64)
mask = torch.zeros(B, L, dtype=torch.int64)
for b in range(B):
    n = int(lens[b])
    ids[b, :n] = torch.randint(4, 50000, (n,), generator=g)
    ids[b, 0] = 0
    mask[b, :n] = 1
enc = {"input_ids": ids, "attention_mask": mask}
if REC:  # key/value tokens of one item after <s>: types 1 (attribute name) / 2 (value), item position 1, global attention on <s>
    tt = torch.where(mask.bool(), torch.full_like(ids, 2), torch.full_like(ids, 3)); tt[:, 0] = 0; tt[:, 1:4] = torch.where(mask[:, 1:4].bool(), 1, 3)
    ip = mask.clone(); ip[:, 0] = 0
    ga = torch.zeros_like(ids); ga[:, 0] = 1
    enc.update(token_type_ids=tt, item_position_ids=ip, global_attention_mask=ga)
batch = BatchDistillationSequence(dataset_indexes=[b % N for b in range(B)], sequence_ids=torch.arange(B) % 64, sequence=enc).to(DEV)
opt = mod.configure_optimizers()
mod.train()


def step():
    opt.zero_grad(set_to_none=True)
    loss = mod.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
K = int(os.environ.get("TB_STEPS", 20))
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
host_ms = (time.perf_counter() - t0) * 1e3 / K   # the host thread's own time per step (launches, no wait): level with `ms` = host-bound
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / K
P = mm.layout.numel
print(f"{MODEL} {learn} [{mm.train_mode}]: N={N} domains, B={B} sequences ({int(lens.sum())} tokens), P={P/1e6:.1f} M parameters, M={M}: {ms:.2f} ms/step "
      f"({1e3/ms:.1f} steps/s, {B*1e3/ms:.0f} sequences/s); loss {loss.item():.4f}")
print(f"  host thread: {host_ms:.2f} ms per step issuing the launches (device-bound when well below the step time)")
print(f"  parameter-sized streams per step: merge fwd {(N+2)*P*4/1e9:.2f} GB + alpha-gradient {(N+1)*P*4/1e9:.2f} GB "
      f"-> {((2*N+3)*P*4/1e9)/(ms/1e3)/1e3:.2f} TB/s of the step if nothing else moved")
if os.environ.get("TB_HOSTPROF", "0") == "1":  # where the host thread spends a step (cProfile inflates Python frames; read the shares)
    import cProfile, pstats

    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):  # the backward's launches in THIS thread, so that the profile sees them
        pr.enable()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("TB_HOSTROWS", 28)))

if os.environ.get("TB_CPU", "0") == "1" and not REC:
    # the same step through the CPU oracle (merge + encoder + loss restatements) with torch autograd, 16 threads
    from oracle import ref_cpu as O
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    base = mm.compact_base().cpu()
    tv = mm.compact_task_vectors().cpu()
    shapes = OrderedDict((k, torch.Size(v)) for k, v in mm.layout.shapes.items())
    cfg = O.EncoderConfig()
    per = torch.full((N,), 0.2, requires_grad=True)
    items_c = [e.cpu() for e in items]
    teach_c = [t.cpu() for t in teachers]
    t0 = time.perf_counter()
    merged = O.merge_task_wise(base, tv, 1.0 * per + 0.0)
    sd = O.get_state_dict(merged, shapes)
    reps = O.maybe_normalize(O.roberta_encode(sd, ids, mask, cfg, prefix="model."))
    ref = O.forward_distill(reps, items_c, teach_c, batch.dataset_indexes, (torch.arange(B) % 64).tolist(),
                            lambda z, t: O.distill_loss("SINGLE_PSEUDO_LABEL_KD", z, t, 0.05, 1000.0))
    ref.backward()
    cpu_s = time.perf_counter() - t0
    print(f"  CPU oracle + torch autograd ({torch.get_num_threads()} threads): {cpu_s*1e3:.0f} ms/step -> x{cpu_s*1e3/ms:.0f}; loss {ref.item():.4f}; "
          f"d alpha max rel diff vs GPU {float(((mm.per_weights['all'].grad.cpu() - per.grad).abs() / per.grad.abs().clamp_min(1e-12)).max()):.2e}")
if os.environ.get("TB_PROFILE", "1") == "1":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = sorted(((e.key, e.device_time_total / 3e3, e.count // 3) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[1])
    tot = sum(r[1] for r in rows)
    print(f"  device time per step {tot:.2f} ms over {sum(r[2] for r in rows)} launches:")
    for k, t, c in rows[: int(os.environ.get("TB_ROWS", 14))]:
        print(f"    {t:7.3f} ms  x{c:<4d} {k[:110]}")

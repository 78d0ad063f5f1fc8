#!/usr/bin/env python3
"""bench.py -- sequences/sec of MergeRec's merged-model inference path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "sequences/sec full-catalog scoring, 8-domain merged BLaIR-base"):
  8 synthetic fine-tuned BLaIR-base checkpoints (theta_pre ~ N(0, 0.02^2), tau_i ~ N(0, 1e-3^2)) merged with
  fixed alpha = 1/8 ("average", merge_test.py:47-55); catalog of an Arts-sized domain (M = 22,855);
  Amazon-shaped synthetic sequences; fp32 end to end (the parity configuration).
A "step" is one pass of the WHOLE hot path over one batch, per rank:
  (1) N-way alpha-weighted merge of this rank's arena slice (+ all-gather of slices when N > 1),
  (2) encode `items_per_step` catalog items and refresh those rows of the item-embedding matrix
      (+ all-gather of the refreshed rows when N > 1) -- users_per_step / items_per_step = 2 matches the
      measured users:items ratio of the Amazon domains, so U/users_per_step steps re-encode one full catalog,
  (3) encode `users_per_step` user sequences (CLS pooled, L2-normalised),
  (4) score them against the FULL catalog and take the canonical top-50 (+ CE terms, label ranks).
Nothing is cached across steps; inputs are resident in HBM before the timed region.
value = users_per_step * N * K / max-over-ranks wall time (weak scaling: per-GPU work is fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak
SUSTAINED_CLOCK_GHZ = {"bf16x3": 1.1, "bf16x6": 1.44}  # measured inside the encoder GEMM kernels (profiles/r01_inkernel_clock.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--domains", type=int, default=8, help="number of fine-tuned checkpoints merged")
    ap.add_argument("--catalog", type=int, default=22855, help="catalog size M (Arts-sized)")
    ap.add_argument("--users-per-step", type=int, default=256)
    ap.add_argument("--items-per-step", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seqs", type=int, default=64)
    ap.add_argument("--cpu-items", type=int, default=128)
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP-event timing")
    ap.add_argument("--gemm-mode", choices=["bf16x6", "bf16x3", "f32"], default=None,
                    help="encoder GEMM arithmetic: bf16x3 (bench default) / bf16x6 = 3 / 6 bf16 MFMA products per fp32 product, "
                         "f32 = exact fp32 MFMA.  The in-run `parity` object reports the distance to the CPU oracle for the chosen mode.")
    ap.add_argument("--merge-placement", choices=["replicated", "sliced"], default="replicated",
                    help="N > 1: 'replicated' = every rank holds all task vectors and merges the whole arena locally (no collective: HBM streams "
                         "the 5 GB of a merge in under 1 ms, xGMI would need longer for the 0.5 GB all-gather alone); 'sliced' = each rank "
                         "merges 1/N of the arena and one all-gather assembles it (task vectors could then be sharded too: memory / N)")
    return ap.parse_args()


def synth_arena(layout, padded, n_dom, device, seed=1000):
    """Seeded synthetic weights generated directly in arena layout on the device (plumbing, untimed)."""
    g = torch.Generator(device=device).manual_seed(seed)
    base = torch.zeros(padded, dtype=torch.float32, device=device)
    for k in layout.shapes:
        v = layout.view(base, k)
        if "LayerNorm.weight" in k:
            v.fill_(1.0)
        elif "LayerNorm.bias" in k:
            v.zero_()
        else:
            v.copy_(torch.randn(v.shape, generator=g, device=device) * 0.02)
    tv = torch.zeros(n_dom, padded, dtype=torch.float32, device=device)
    for i in range(n_dom):
        gi = torch.Generator(device=device).manual_seed(seed + 1 + i)
        for k in layout.shapes:
            layout.view(tv[i], k).copy_(torch.randn(layout.shapes[k], generator=gi, device=device) * 1e-3)
    return base, tv


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist

    # one rank per GPU; MERGEREC_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals of the N > 1 path
    backend = os.environ.get("MERGEREC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks need {world} GPUs, {ndev} visible")
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mergerec_amd import ops, parallel
    from mergerec_amd.engine import ArenaLayout, EncoderRunner, EncoderSpec, WeightSet
    from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths, _ids_from_lengths

    spec = EncoderSpec.blair_base()
    layout = ArenaLayout(spec.param_shapes("model."))
    sliced = world > 1 and args.merge_placement == "sliced"
    plan = parallel.SlicePlan(layout.padded_numel, world if sliced else 1)
    runner = EncoderRunner(spec)
    n_dom, M, d = args.domains, args.catalog, spec.hidden
    U_step, I_step = args.users_per_step, args.items_per_step

    # ---------------- resident state (untimed setup) ----------------
    base, tv = synth_arena(layout, plan.padded, n_dom, dev)
    alpha = torch.full((1, n_dom), 1.0 / n_dom, dtype=torch.float32, device=dev)  # "average" weights
    arena = torch.zeros(plan.padded, dtype=torch.float32, device=dev)
    # bench default: the fastest arithmetic that meets the path's 1e-4 logit tolerance with >= 50x margin on these dims
    # (bf16x3: measured 1.1e-6 on embeddings, 6e-7 on logits); MERGEREC_GEMM_MODE / --gemm-mode select the others
    gemm_mode = args.gemm_mode or os.environ.get("MERGEREC_GEMM_MODE") or "bf16x3"
    W = WeightSet(layout, arena[: layout.padded_numel], gemm_mode)
    lo, hi = plan.bounds(rank if sliced else 0)
    scratch = torch.empty(hi - lo, dtype=torch.float32, device=dev) if sliced else None

    def merge_slice(p_begin, p_count, out_slice):
        ops.merge_nway(base, tv, alpha, None, out=out_slice, p_begin=p_begin, p_count=p_count, out_is_slice=True)

    def merge_arena():
        if sliced:  # this rank's slice, then ONE all-gather of the merged slices
            parallel.sharded_merge(merge_slice, arena, plan, scratch)
        else:  # whole arena from the rank's own copy of the task vectors (the N = 1 path on every rank)
            merge_slice(0, plan.padded, arena)

    n_total = args.steps + args.warmup
    g = torch.Generator().manual_seed(1234 + rank)
    # N > 1: token-balanced sharding (parallel.balanced_share).  Every rank draws the SAME global pool of lengths (N x the per-GPU batch, one
    # shared seed) and takes its snake-dealt share: equal sequence counts and near-equal token totals per rank, so no rank waits for another
    # at the step's all-gather.  (Independent random shards differ by a few % in tokens: the slowest of 8 sets the pace.)
    g_pool = torch.Generator().manual_seed(4321)

    def my_share(lengths_fn, per_rank):
        if world == 1:
            return lengths_fn(per_rank, g)
        pool = lengths_fn(per_rank * world, g_pool)
        mine = pool[parallel.balanced_share(pool, world, rank)]
        return mine[torch.randperm(per_rank, generator=g)]       # batch order is not length order

    user_batches, item_batches, label_batches = [], [], []
    for s in range(n_total):
        ul = my_share(blair_sequence_lengths, U_step)
        il = my_share(blair_item_lengths, I_step)
        ub = _ids_from_lengths(ul, spec.vocab, g)
        ib = _ids_from_lengths(il, spec.vocab, g)
        user_batches.append(({k: v.to(dev) for k, v in ub.items()}, ul))
        item_batches.append(({k: v.to(dev) for k, v in ib.items()}, il))
        label_batches.append(torch.randint(0, M, (U_step,), generator=g).to(dev))
    # items first, then users, padded to a common length on the host (padding never reaches a kernel)
    mixed_batches = []
    for (ib, il), (ub, ul) in zip(item_batches, user_batches):
        L = max(ib["input_ids"].shape[1], ub["input_ids"].shape[1])
        pad = lambda t, v: torch.nn.functional.pad(t, (0, L - t.shape[1]), value=v)
        mixed_batches.append(({"input_ids": torch.cat([pad(ib["input_ids"], spec.pad_id), pad(ub["input_ids"], spec.pad_id)]),
                               "attention_mask": torch.cat([pad(ib["attention_mask"], 0), pad(ub["attention_mask"], 0)])},
                              torch.cat([il, ul])))
    avg_user_tokens = float(torch.cat([l for _, l in user_batches]).float().mean())
    avg_item_tokens = float(torch.cat([l for _, l in item_batches]).float().mean())

    # full catalog encoded once with the merged model (setup): E (M, d), row == item id
    merge_arena()
    W.refresh()
    E = torch.empty(M, d, dtype=torch.float32, device=dev)
    gi = torch.Generator().manual_seed(99)
    for s0 in range(0, M, 512):
        n = min(512, M - s0)
        il = blair_item_lengths(n, gi)
        ib = {k: v.to(dev) for k, v in _ids_from_lengths(il, spec.vocab, gi).items()}
        E[s0 : s0 + n] = runner.encode(W, ib, dev, normalize=True, lens=il, validate=False)
    torch.cuda.synchronize()

    item_blocks = [(r * I_step, (r + 1) * I_step) for r in range(world)]
    state = {"cursor": 0}

    def step(i):
        # (1) merge (replicated: the whole arena locally; sliced: this rank's slice + all-gather)
        merge_arena()
        W.refresh()  # bf16x6 mode: re-split the freshly merged arena into its three bf16 piece arenas
        # (2)+(3) ONE packed encoder pass over [catalog slice ; user sequences] (varlen: no padding is computed)
        mb, ml = mixed_batches[i]
        emb = runner.encode(W, mb, dev, normalize=True, lens=ml, validate=False)
        e_new, u = emb[:I_step], emb[I_step:]
        e_all = parallel.all_gather_rows(e_new.contiguous(), item_blocks) if world > 1 else e_new
        c = state["cursor"]
        n = e_all.shape[0]
        if c + n > M:
            c = 0
        E[c : c + n] = e_all  # refresh those catalog rows before scoring
        state["cursor"] = c + n
        # (4) full-catalog scoring + canonical top-50 + CE terms
        return ops.score_topk(u.contiguous(), E, 50, label_batches[i], 1.0 / 0.05)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.PROF.enabled = not args.no_profile
    ops.PROF.records.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(args.warmup, n_total):
        last = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.PROF.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    value = U_step * world * args.steps / elapsed

    # ---------------- per-kernel rooflines from HIP events over the timed region ----------------
    kernels = {}
    roofline = None
    if not args.no_profile:
        for name, r in ops.PROF.summary().items():
            sec = r["ms"] / 1e3
            ent = dict(launches=r["launches"], avg_ms=r["ms"] / max(r["launches"], 1), share_of_step=r["ms"] / (elapsed * 1e3))
            if r["flops"] > 0:
                if name.endswith(("bf16x6", "bf16x3")):  # 6 (3) bf16 MFMA flops are executed per algorithmic flop; peak = dense bf16 MFMA
                    np_ = 6 if name.endswith("x6") else 3
                    ent.update(bound="mfma", achieved=r["flops"] / sec / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                               mfma_flops_per_algorithmic_flop=np_, mfma_utilization=np_ * r["flops"] / sec / 1e12 / MFMA_BF16_PEAK_TF)
                else:
                    ent.update(bound="mfma", achieved=r["flops"] / sec / 1e12, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s")
            else:
                ent.update(bound="hbm", achieved=r["bytes"] / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            ent["frac"] = ent["achieved"] / ent["peak"]
            kernels[name] = ent
        dom = max(kernels.items(), key=lambda kv: kv[1]["share_of_step"])
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 on gfx950; tools/pmc_summary.py) -- PMC cannot be
        # collected from inside the timed run.
        traffic, traffic_src, traffic_raw = None, None, None
        tpath = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{gemm_mode}.json")
        if world == 1 and os.path.exists(tpath):
            t = json.load(open(tpath)).get("gemm_nt_bf16x" if dom[0].startswith("gemm_nt_bf16x") else dom[0])
            if t:
                traffic = t["fetch_bytes_x2_per_launch"] + t["write_bytes_per_launch"]
                traffic_raw = t["fetch_bytes_raw_per_launch"] + t["write_bytes_per_launch"]  # uncorrected FETCH_SIZE: exact if 64-B row fragments are tallied exactly
                traffic_src = f"profiles/r01_pmc_traffic_{gemm_mode}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
        roofline = dict(kernel=dom[0], **{k: dom[1][k] for k in ("mfma_flops_per_algorithmic_flop", "mfma_utilization") if k in dom[1]}, bound=dom[1]["bound"], achieved=dom[1]["achieved"], peak=dom[1]["peak"], unit=dom[1]["unit"],
                        frac=dom[1]["frac"], traffic=traffic, traffic_uncorrected=traffic_raw, traffic_unit="HBM bytes per launch", traffic_source=traffic_src,
                        algorithmic_bytes_per_launch=ops.PROF.summary()[dom[0]]["bytes"] / max(dom[1]["launches"], 1),
                        launches=dom[1]["launches"], avg_launch_ms=dom[1]["avg_ms"])
        if "mfma_utilization" in roofline:
            # the dense peak assumes 2.4 GHz; under this kernel the chip sustains ~1.1 GHz (in-kernel s_memtime / s_memrealtime of the same
            # kernel, profiles/r01_inkernel_clock.txt: 1.04-1.16 GHz on two devices) -- the matrix pipe's busy fraction at THAT clock:
            roofline["sustained_clock_ghz"] = SUSTAINED_CLOCK_GHZ.get(gemm_mode)
            roofline["sustained_clock_source"] = "profiles/r01_inkernel_clock.txt (exp/gemm_phases.hip)"
            if roofline["sustained_clock_ghz"]:
                roofline["mfma_busy_at_sustained_clock"] = roofline["mfma_utilization"] * 2.4 / roofline["sustained_clock_ghz"]

    # ---------------- CPU baseline (oracle port, rank 0, N == 1 only) ----------------
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity = run_cpu_baseline(args, spec, layout, base, tv, alpha, W, runner, dev, M, E)

    if rank == 0:
        out = OrderedDict(
            metric="sequences/sec full-catalog scoring, 8-domain merged BLaIR-base; NDCG@10 parity",
            value=value, unit="sequences/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="f32" if gemm_mode == "f32" else f"f32 via {gemm_mode} split MFMA (encoder GEMMs: {gemm_mode[-1]} bf16 products per fp32 product; merge, attention, scoring in f32)",
            data="synthetic",
            config=dict(
                workload=f"{n_dom}-domain merged BLaIR-base (alpha=1/{n_dom}), full-catalog scoring, Arts-sized catalog",
                domains_merged=n_dom, catalog_items=M, users_per_step_per_gpu=U_step, items_per_step_per_gpu=I_step,
                avg_user_tokens=avg_user_tokens, avg_item_tokens=avg_item_tokens, topk=50, params=layout.numel,
                parallelism=(f"dp{world}: " + ("arena-slice merge + all-gather" if sliced else "task vectors replicated, whole-arena merge per rank (no collective)")
                             + ", catalog rows sharded + all-gather, users data-parallel (token-balanced shards)"),
            ),
            roofline=roofline, cpu_baseline=cpu_baseline, kernels=kernels, parity=parity,
        )
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run_cpu_baseline(args, spec, layout, base, tv, alpha, W, runner, dev, M, E):
    """Times the oracle (oracle/ref_cpu.py, fp32 torch on the host cores) on a bounded sample of the same
    step and scales it to the step's composition.  Also cross-checks the GPU embeddings of the sample."""
    from oracle import ref_cpu as O
    from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths, _ids_from_lengths

    # threads actually used: this process's CPU share (the GPU box exposes 256 logical CPUs but grants a
    # 16-CPU share per GPU; 256 torch threads oversubscribe it by 16x)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(avail, 16)
    torch.set_num_threads(cores)
    n_dom = tv.shape[0]
    base_c, tv_c, a_c = base.cpu(), tv.cpu(), alpha.cpu().reshape(-1)
    t0 = time.perf_counter()
    merged = O.merge_task_wise(base_c, tv_c, a_c)  # the reference's 3-op expression (task_wise.py:43-47)
    t_merge = time.perf_counter() - t0
    sd = layout.views(merged)
    cfg = O.EncoderConfig()
    g = torch.Generator().manual_seed(4321)
    ul = blair_sequence_lengths(args.cpu_seqs, g)
    il = blair_item_lengths(args.cpu_items, g)
    ub = _ids_from_lengths(ul, spec.vocab, g)
    ib = _ids_from_lengths(il, spec.vocab, g)
    with torch.no_grad():
        t0 = time.perf_counter()
        u_c = O.maybe_normalize(O.roberta_encode(sd, ub["input_ids"], ub["attention_mask"], cfg, "model."))
        t_users = time.perf_counter() - t0
        t0 = time.perf_counter()
        e_c = O.maybe_normalize(O.roberta_encode(sd, ib["input_ids"], ib["attention_mask"], cfg, "model."))
        t_items = time.perf_counter() - t0
        E_c = E.cpu()
        nU = 32  # the reference's per-batch scoring shape (module.py:137 at --batch_size 32)
        Uc = O.maybe_normalize(torch.randn(nU, spec.hidden, generator=g))
        t0 = time.perf_counter()
        sc = O.score(Uc, E_c)
        O.topk_canonical(sc, 50)
        t_score = time.perf_counter() - t0
    U_step, I_step = args.users_per_step, args.items_per_step
    t_step = t_merge + I_step * (t_items / args.cpu_items) + U_step * (t_users / args.cpu_seqs) + (U_step / nU) * t_score
    # parity of the GPU path on the very same sample
    u_g = runner.encode(W, {k: v.to(dev) for k, v in ub.items()}, dev, normalize=True).cpu()
    e_g = runner.encode(W, {k: v.to(dev) for k, v in ib.items()}, dev, normalize=True).cpu()
    merged_g = torch.cat([v.reshape(-1) for v in W.views.values()]).cpu()
    merged_cc = torch.cat([v.reshape(-1) for v in sd.values()])
    parity = dict(
        merged_params_bit_exact=bool(torch.equal(merged_g, merged_cc)),
        user_embedding_max_abs_diff=float((u_g - u_c).abs().max()), item_embedding_max_abs_diff=float((e_g - e_c).abs().max()),
        logit_max_abs_diff=float((u_g @ e_g.T - u_c @ e_c.T).abs().max()), tolerance=1e-4,
    )
    base_out = dict(
        value=U_step / t_step, unit="sequences/s", cores=cores, kind="port",
        sample=(f"oracle/ref_cpu.py fp32 torch, {cores} threads: {n_dom}-way merge of all {layout.numel} params ({t_merge:.2f}s), "
                f"{args.cpu_seqs} user sequences ({t_users:.2f}s), {args.cpu_items} items ({t_items:.2f}s), 32x{M} scoring+top-50 "
                f"({t_score * 1e3:.1f}ms); scaled to one step = merge + {I_step} items + {U_step} users + scoring"),
        cpu_s_per_step=t_step,
    )
    return base_out, parity


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- sequences/sec of MergeRec's merged-model inference path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver may launch it under
torch.distributed.run (one rank per GPU, RCCL); called plainly with `--gpus N > 1` (no WORLD_SIZE in the environment) it starts
`python -m torch.distributed.run --nproc-per-node N bench.py <same args>` itself as a CHILD process -- before anything in this
process touches the GPU -- relays rank 0's line and exits with the child's return code.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "sequences/sec full-catalog scoring, 8-domain merged BLaIR-base"):
  8 synthetic fine-tuned BLaIR-base checkpoints (theta_pre ~ N(0, 0.02^2), tau_i ~ N(0, 1e-3^2)) merged with
  fixed alpha = 1/8 ("average", merge_test.py:47-55); catalog of an Arts-sized domain (M = 22,855);
  Amazon-shaped synthetic sequences; fp32-grade arithmetic end to end (the parity configuration).
The step runs through the PRODUCT objects with their defaults -- `load_merging_module(...)` (sliced over the ranks when
N > 1: each rank holds 1/N of the base vector and of every task vector), `RecModule.forward(BatchItem)`,
`RecModule.test_step(BatchSequence)`, `on_test_epoch_end()` -- i.e. the calls merge_test.py / utils.test_model make.
Input checks run inside the packing kernel; sequence lengths ride along from the collator side of `.to(device)`.
A "step" is one pass of the WHOLE hot path over one batch, per rank:
  (1) N-way alpha-weighted merge of the parameter arena (this rank's slice + ONE all-gather when N > 1),
  (2) encode `items_per_step` catalog items and refresh those rows of the item-embedding matrix
      (+ all-gather of the refreshed rows when N > 1) -- users_per_step / items_per_step = 2 matches the
      measured users:items ratio of the Amazon domains, so U/users_per_step steps re-encode one full catalog,
  (3) encode `users_per_step` user sequences (CLS pooled, L2-normalised),
  (4) score them against the FULL catalog and take the canonical top-50 (+ CE terms, label ranks).
The timed region is one evaluation epoch over K steps in the product's own order (utils.Trainer.test): the K steps' catalog rows go
through `ItemEncoderMixin.encode_items` first (the callback's code: token-coalesced passes, sharded over the ranks + all-gather),
then the K user steps (merge + encode + score each), then the epoch end (metric gather + Recall / NDCG) -- all inside the timing.
Nothing is cached across steps; inputs are resident in HBM before the timed region.
value = users_per_step * N * K / max-over-ranks wall time (weak scaling: per-GPU work is fixed).
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import threading
import time
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak
PROFILE_ROUNDS = ("r04", "r03", "r02")  # counter reductions under profiles/: newest first


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
    ap.add_argument("--gemm-mode", choices=["f16x3", "bf16x6", "bf16x3", "f32"], default=None,
                    help="encoder arithmetic: f16x3 (bench default: 3 fp16 MFMA products per fp32 product, ~2^-21 each) / bf16x6 / bf16x3 = "
                         "6 / 3 bf16 products (~2^-24 / ~2^-16), f32 = exact fp32 MFMA.  The in-run `parity` object reports the distance to "
                         "the CPU oracle for the chosen mode.")
    ap.add_argument("--pipelined-merge", type=int, choices=[0, 1], default=None,
                    help="1: the per-step merge (+ arena all-gather under the sliced placement) + weight split run on a second HIP stream into a "
                         "second arena, under the previous step's encoder kernels (TaskVectorMergingModuleBase.pipeline_merges); 0: on the step's "
                         "own stream.  Default: 1 only where there is a collective to hide (N > 1 with --merge-placement sliced) -- on one GPU the "
                         "A/B is a wash (38.36 / 38.38 ms: the 5 TB/s merge stream takes from the GEMMs what it saves)")
    ap.add_argument("--merge-placement", choices=["sliced", "replicated"], default=None,
                    help="N > 1: 'replicated' (bench default: the merge runs every step) = every rank holds everything and merges alone, no "
                         "collective; 'sliced' (north_star's split, the library default for merge-once runs) = each rank holds 1/N of the base "
                         "vector and of every task vector, merges that arena slice, ONE all-gather assembles the arena")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ power / clock sampling
class DeviceSampler(threading.Thread):
    """Samples the GPU's shader clock and socket power from the amdgpu hwmon files while the timed region runs (a host thread
    reading sysfs every few ms: no GPU work, no sync).  Evidence for / against a power-limited clock in THIS run."""

    def __init__(self, pci_bus_id: str | None, period_s: float = 0.004):
        super().__init__(daemon=True)
        self.period = period_s
        self.stop_flag = threading.Event()
        self.sclk, self.power, self.t = [], [], []
        self.dir, self.cap_w, self.note = None, None, None
        cands = []
        if pci_bus_id:
            cands += glob.glob(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/hwmon/hwmon*")
        if not cands:  # a box may expose the hwmon files of every GPU of its host: without a PCI match take the card that is drawing power
            allc = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            if len(allc) == 1:
                cands = allc
            elif allc:
                draw = [(self._read(os.path.join(c, "power1_input")) or self._read(os.path.join(c, "power1_average")) or 0.0, c) for c in allc]
                cands = [max(draw)[1]]
                self.note = "no PCI match: busiest card at start of the timed region"
        for c in cands:
            if os.path.exists(os.path.join(c, "power1_average")) or os.path.exists(os.path.join(c, "power1_input")) or os.path.exists(os.path.join(c, "freq1_input")):
                self.dir = c
                break
        if self.dir is None:
            self.note = "no readable amdgpu hwmon directory"
            return
        self.f_power = next((os.path.join(self.dir, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(self.dir, n))), None)
        self.f_sclk = os.path.join(self.dir, "freq1_input") if os.path.exists(os.path.join(self.dir, "freq1_input")) else None
        cap = self._read(os.path.join(self.dir, "power1_cap"))
        self.cap_w = cap / 1e6 if cap else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:  # noqa: BLE001
            return None

    def run(self):
        if self.dir is None:
            return
        while not self.stop_flag.is_set():
            self.t.append(time.perf_counter())
            if self.f_sclk:
                v = self._read(self.f_sclk)
                if v:
                    self.sclk.append(v / 1e6)  # Hz -> MHz
            if self.f_power:
                v = self._read(self.f_power)
                if v:
                    self.power.append(v / 1e6)  # uW -> W
            time.sleep(self.period)

    def summary(self):
        mean = lambda xs: (sum(xs) / len(xs)) if xs else None
        series = None
        if self.t and self.t[-1] - self.t[0] >= 2.0 and len(self.sclk) == len(self.t) == len(self.power):
            # runs of two seconds or more: the clock / power series per second (does the rate hold once the burst is over?)
            t0, buckets = self.t[0], {}
            for ti, c, w in zip(self.t, self.sclk, self.power):
                buckets.setdefault(int(ti - t0), []).append((c, w))
            series = [dict(second=k, sclk_mhz=round(mean([c for c, _ in v])), power_w=round(mean([w for _, w in v]))) for k, v in sorted(buckets.items())]
        return dict(per_second=series, sclk_mhz_mean=mean(self.sclk), sclk_mhz_min=min(self.sclk) if self.sclk else None, power_w_mean=mean(self.power),
                    power_w_max=max(self.power) if self.power else None, power_cap_w=self.cap_w, samples=max(len(self.sclk), len(self.power)),
                    source=(self.dir or self.note))


def _sha16(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def synth_state_dicts(model, n_dom, device, seed=1001):
    """Seeded synthetic fine-tuned checkpoints, generated on the device (plumbing, untimed): theta_i = theta_pre + N(0, 1e-3^2)."""
    pre = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    fts = []
    for i in range(n_dom):
        g = torch.Generator(device=device).manual_seed(seed + i)
        fts.append(OrderedDict((k, v if k.endswith("position_ids") else v + 1e-3 * torch.randn(v.shape, generator=g, device=device)) for k, v in pre.items()))
    return pre, fts


def main():
    if "WORLD_SIZE" not in os.environ:
        pre = parse()
        if pre.gpus > 1:
            raise SystemExit(spawn_ranks(pre))
    # rank 0 prints ONE JSON line on stdout: everything else that writes to file descriptor 1 while the bench runs -- the product
    # objects' progress prints ("Calculating task vectors..."), gloo's connection banner from C++ -- is sent to stderr instead
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    real_stdout = os.fdopen(saved_fd, "w")
    try:
        _main(real_stdout)
    finally:
        sys.stdout.flush()
        real_stdout.flush()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as a child process group and relay rank 0's
    JSON line.  Runs BEFORE this process initialises the GPU (no HIP call, no torch.cuda.is_available()): the parent only waits.  No
    retry; the return code is the launcher's (non-zero as soon as any rank failed)."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface (the container hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    n_json = 0
    for line in proc.stdout:  # rank 0 writes exactly one JSON line to fd 1; anything else a launcher might print goes to stderr
        if line.startswith("{") and n_json == 0:
            sys.stdout.write(line)
            sys.stdout.flush()
            n_json += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and n_json == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no result line\n")
        rc = 1
    return rc


def _main(real_stdout):
    args = parse()
    import torch.distributed as dist

    from mergerec_amd import ops, parallel

    rank, world = parallel.init_from_env(verbose=False)  # one rank per GPU; MERGEREC_DIST_BACKEND=gloo rehearses several ranks on one GPU
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    backend = dist.get_backend() if world > 1 else None
    dev = torch.device("cuda", torch.cuda.current_device())

    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchItem, BatchSequence
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths, _ids_from_lengths

    n_dom, M = args.domains, args.catalog
    U_step, I_step = args.users_per_step, args.items_per_step
    # bench default: the fastest arithmetic that stays at the REFERENCE's own fp32 rounding level on trained-like weights (fixture g22,
    # tests/test_trained_like_gpu.py: peaky attention, LayerNorm outliers, massive activations) -- r04: f16x3.  bf16x3, the r01-r03 default,
    # is 10x the fp32 noise there (1.3e-3 on logits) and is opt-in only.  MERGEREC_GEMM_MODE / --gemm-mode select the others
    gemm_mode = args.gemm_mode or os.environ.get("MERGEREC_GEMM_MODE") or "f16x3"

    # ---------------- resident state (untimed setup), through the drop-in API ----------------
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 1000, "gemm_mode": gemm_mode})
    spec, d = model.spec, model.spec.hidden
    pre, fts = synth_state_dicts(model, n_dom, dev)
    # N > 1 default: "replicated".  The bench re-merges EVERY step (as the reference re-merges on every forward, _base.py:78-81); at that
    # cadence the sliced placement's arena all-gather (499 MB x (N - 1) / N received per rank and step) costs more than the local merge it
    # saves at N = 2 and 4 and about the same at N = 8 (predicted 2.6 / 1.6 / 0.9 ms against 1.1 ms), and 8 task vectors
    # are 4 GB of a 288 GB part.  "sliced" stays the library default (merge_test.py merges ONCE: 1 / N of the upload and of the memory) and is
    # north_star's wording; --merge-placement sliced measures it.
    placement = args.merge_placement or ("replicated" if world > 1 else None)
    merged_model = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, pre, fts, set(), disable_softmax=True,
                                       placement=placement)
    sliced = merged_model.slice_plan is not None
    del fts
    torch.cuda.empty_cache()
    merged_model.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]},
                                         "per_weights": {"all": [1.0 / n_dom] * n_dom}})  # "average" (merge_test.py:47-55)
    module = RecModule(model=model, evaluator=Evaluator(["NDCG", "RECALL"], [1, 5, 10, 50]), similarity="cosine")
    module.eval()
    if args.pipelined_merge if args.pipelined_merge is not None else sliced:
        merged_model.pipeline_merges(True)

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

    # the collator side: CPU batches in the reference's dataclasses; `.to(dev)` (what the Trainer does per batch) happens HERE, before
    # the timed region, so inputs are HBM-resident; the per-row lengths travel with the moved batch (model_batch.Encoding.host_lens)
    user_batches = []
    tok_u = 0
    for s in range(n_total):
        ul = my_share(blair_sequence_lengths, U_step)
        tok_u += int(ul.sum())
        labels = torch.randint(0, M, (U_step,), generator=g)
        user_batches.append(BatchSequence(sequence=_ids_from_lengths(ul, spec.vocab, g), labels=labels).to(dev))
    # catalog rows: the SAME global list of item batches on every rank (shared seed).  As in the product, the collator's 128-item batches
    # are coalesced ON THE HOST into token-sized encoder passes (data.coalesce_batches: utils.Trainer's 65,536-token budget) before they
    # are moved; encode_items then deals the passes over the ranks (parallel.ShardedLoader: contiguous blocks = this rank's steps).
    from mergerec_amd.data import coalesce_batches

    g_items = torch.Generator().manual_seed(777)
    per_pass = max(1, 65536 // (I_step * 40))  # collator batches per pass: fixed count, so every rank builds the same list

    def item_passes(n_steps):
        out = []
        for r in range(world):
            blk = []
            for s in range(n_steps):
                il = blair_item_lengths(I_step, g_items)
                blk.append((BatchItem(items=_ids_from_lengths(il, spec.vocab, g_items)), int(il.sum())))
            for c0 in range(0, n_steps, per_pass):
                out.extend(b.to(dev) for b in coalesce_batches([b for b, _ in blk[c0:c0 + per_pass]], 1 << 30))
        return out, sum(t for _, t in blk) if n_steps else 0

    items_warm, _ = item_passes(args.warmup)
    items_timed, tok_i = item_passes(args.steps)
    avg_user_tokens, avg_item_tokens = tok_u / (n_total * U_step), tok_i / max(args.steps * I_step, 1)

    # full catalog encoded once with the merged model (setup): E (M, d), row == item id
    merged_model.load_weights()
    E = torch.empty(M, d, dtype=torch.float32, device=dev)
    gi = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for s0 in range(0, M, 1024):
            n = min(1024, M - s0)
            E[s0 : s0 + n] = module.forward(BatchItem(items=_ids_from_lengths(blair_item_lengths(n, gi), spec.vocab, gi)).to(dev))
    module.item_embeddings = torch.nn.Parameter(E, requires_grad=False)
    model.check_inputs()
    torch.cuda.synchronize()

    from mergerec_amd.module.callbacks import ItemEncoderMixin

    module.trainer = type("T", (), {"coalesce_tokens": 0})()  # the passes were coalesced on the host above (HBM-resident inputs)
    state = {"cursor": 0}

    @torch.no_grad()
    def step(i):
        # (1) the merge, as TaskVectorMergingModuleBase.forward does it on every call (_base.py:78-81): into the arena the model reads
        merged_model.load_weights(force=True)
        # (3) + (4) RecModule.test_step: encode users, full-catalog scoring, canonical top-50, CE terms
        module.test_step(user_batches[i], i)

    def epoch(lo, hi, passes):
        # (2) the catalog rows of these steps: the callback's encode (callbacks.py:18-38) -- coalesced passes, rank shards, all-gather
        if not passes:
            return None
        merged_model.load_weights(force=True)
        e_all = ItemEncoderMixin.encode_items(passes, module)
        c, n, off = state["cursor"], e_all.shape[0], 0
        while off < n:  # refresh those rows before scoring (a long run re-encodes more rows than the catalog holds: wrap around)
            c = 0 if c >= M else c
            take = min(n - off, M - c)
            module.item_embeddings.data[c : c + take] = e_all[off : off + take]
            c, off = c + take, off + take
        state["cursor"] = c
        module.on_test_epoch_start()
        for i in range(lo, hi):
            step(i)
        if world > 1:  # per-user label ranks meet on every rank (a few bytes per user), as utils.Trainer.test gathers them
            parallel.all_gather_vector(torch.cat(module._ranks))
        return module.on_test_epoch_end()

    epoch(0, args.warmup, items_warm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sampler = None
    if rank == 0:
        bus_id = None
        try:
            p = torch.cuda.get_device_properties(dev)
            bus_id = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            bus_id = None
        sampler = DeviceSampler(bus_id)
        sampler.start()
    ops.PROF.enabled = not args.no_profile
    ops.PROF.records.clear()
    parallel.COLL.enabled = world > 1
    parallel.COLL.records.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    metrics = epoch(args.warmup, n_total, items_timed)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0   # this rank's own time to the end of its work (before the closing barrier)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.PROF.enabled = False
    parallel.COLL.enabled = False
    if sampler is not None:
        sampler.stop_flag.set()
        sampler.join(timeout=1.0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        # per-rank view: every rank's own time to finish its work, and the time it spent inside the data-path collectives
        coll = parallel.COLL.summary()
        mine = torch.tensor([own_elapsed, coll["ms"] / 1e3, float(coll["bytes_received_per_rank"])], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = torch.stack(every).cpu()
    value = U_step * world * args.steps / elapsed

    # ---------------- per-kernel rooflines from HIP events over the timed region ----------------
    kernels = {}
    roofline = None
    if not args.no_profile:
        summ = ops.PROF.summary()
        for name, r in summ.items():
            sec = r["ms"] / 1e3
            ent = dict(launches=r["launches"], avg_ms=r["ms"] / max(r["launches"], 1), share_of_step=r["ms"] / (elapsed * 1e3))
            if r["flops"] > 0:
                if "bf16x" in name or "f16x3" in name:  # 6 (3) bf16 / fp16 MFMA flops are executed per algorithmic flop; peak = dense 16-bit MFMA
                    np_ = 6 if "bf16x6" in name else 3
                    ent.update(bound="mfma", achieved=r["flops"] / sec / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                               mfma_flops_per_algorithmic_flop=np_, mfma_utilization=np_ * r["flops"] / sec / 1e12 / MFMA_BF16_PEAK_TF)
                else:
                    ent.update(bound="mfma", achieved=r["flops"] / sec / 1e12, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s")
            else:
                ent.update(bound="hbm", achieved=r["bytes"] / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            ent["frac"] = ent["achieved"] / ent["peak"]
            kernels[name] = ent
        if "score_gemm" in kernels and "topk_rows" in kernels:
            # full-catalog scoring = scoring GEMM + row select TOGETHER against the fp32-MFMA peak (north_star's ">= 40 % on full-catalog
            # scoring" is about the pair, not the GEMM half)
            sg, tk = summ["score_gemm"], summ["topk_rows"]
            sec = (sg["ms"] + tk["ms"]) / 1e3
            kernels["score_plus_topk"] = dict(launches=sg["launches"], avg_ms=(sg["ms"] + tk["ms"]) / max(sg["launches"], 1),
                                              share_of_step=(sg["ms"] + tk["ms"]) / (elapsed * 1e3), bound="mfma", achieved=sg["flops"] / sec / 1e12,
                                              peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=sg["flops"] / sec / 1e12 / MFMA_F32_PEAK_TF,
                                              note="score_gemm + topk_rows as one unit of work")
        dom = max((kv for kv in kernels.items() if kv[0] != "score_plus_topk"), key=lambda kv: kv[1]["share_of_step"])
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command (separate FETCH_SIZE /
        # WRITE_SIZE passes, FETCH_SIZE x2 on gfx950; tools/pmc_summary.py) -- PMC cannot be collected from inside the timed run.  The pass
        # records the sha of the kernel sources it profiled: a different sha here means the figure is STALE and is reported as such.
        traffic = traffic_raw = traffic_src = None
        traffic_stale = None
        src_sha = _sha16(sorted(glob.glob(os.path.join(ROOT, "mergerec_amd", "csrc", "*.hip"))))

        def committed(kind):
            """newest reduction of that kind for this arithmetic: from MERGEREC_COUNTER_DIR when the caller names one (tools/refresh_profiles.sh
            points it at the passes it has just made -- the tracked profiles/ directory is never written by a run), else from the committed
            profiles/ (this round's file if it exists, else the newest earlier round's); the path read is what ``*_source`` records"""
            override = os.environ.get("MERGEREC_COUNTER_DIR")
            if override and os.path.abspath(override).startswith(ROOT + os.sep):
                override = os.path.relpath(os.path.abspath(override), ROOT)   # recorded as a path inside the repository
            for base, shown in ((os.path.join(ROOT, override) if override and not os.path.isabs(override) else override, override),
                                (os.path.join(ROOT, "profiles"), "profiles")) if override else ((os.path.join(ROOT, "profiles"), "profiles"),):
                for rnd in PROFILE_ROUNDS:
                    path = os.path.join(base, f"{rnd}_{kind}_{gemm_mode}.json")
                    if os.path.exists(path):
                        return json.load(open(path)), f"{shown}/{rnd}_{kind}_{gemm_mode}.json"
            return None, None

        tj, tname = committed("pmc_traffic") if world == 1 else (None, None)
        if tj:
            t = tj.get(dom[0]) or tj.get(dom[0].split("/")[0])
            if t:
                traffic = t["fetch_bytes_x2_per_launch"] + t["write_bytes_per_launch"]
                traffic_raw = t["fetch_bytes_raw_per_launch"] + t["write_bytes_per_launch"]
                traffic_src = f"{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
                traffic_stale = tj.get("_csrc_sha16") != src_sha
        # matrix-pipe occupancy of the same kernel from the committed SQ passes (tools/sq_pass.sh: SQ_VALU_MFMA_BUSY_CYCLES over
        # GRBM_GUI_ACTIVE / 8 x 1024 SIMD-cycles, the wave-cycle split, the profiled clock) -- also only collectable outside the timed run
        mfma_busy = None
        sj, sname = committed("pmc_sq") if world == 1 else (None, None)
        if sj:
            q = sj.get(dom[0]) or sj.get(dom[0].split("/")[0])
            if q and "mfma_busy" in q:
                mfma_busy = dict(value=q["mfma_busy"], clock_ghz_under_profiler=q.get("clock_ghz"), mfma_tflops_executed=q.get("mfma_tflops_executed"),
                                 wave_parked=q.get("wave_parked"), wave_issue_stall=q.get("wave_issue_stall"), wave_active=q.get("wave_active"),
                                 source=f"{sname} (rocprofv3 --pmc SQ passes of bench.py)", stale=sj.get("_csrc_sha16") != src_sha)
        roofline = dict(kernel=dom[0], **{k: dom[1][k] for k in ("mfma_flops_per_algorithmic_flop", "mfma_utilization") if k in dom[1]},
                        bound=dom[1]["bound"], achieved=dom[1]["achieved"], peak=dom[1]["peak"], unit=dom[1]["unit"], frac=dom[1]["frac"],
                        traffic=traffic, traffic_uncorrected=traffic_raw, traffic_unit="HBM bytes per launch", traffic_source=traffic_src,
                        traffic_stale=traffic_stale, mfma_busy=mfma_busy, csrc_sha16=src_sha,
                        algorithmic_bytes_per_launch=summ[dom[0]]["bytes"] / max(dom[1]["launches"], 1),
                        algorithmic_flops_per_launch=summ[dom[0]]["flops"] / max(dom[1]["launches"], 1),
                        launches=dom[1]["launches"], avg_launch_ms=dom[1]["avg_ms"])
        if sampler is not None:  # measured in THIS process over the timed region (amdgpu hwmon); not a model
            roofline.update(sampler.summary())

    # ---------------- CPU baseline (oracle port, rank 0, N == 1 only) ----------------
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity = run_cpu_baseline(args, spec, merged_model, module, dev, M)

    if rank == 0:
        arith = {"f32": "f32 (exact fp32 MFMA in every kernel)",
                 "bf16x6": "f32 via bf16x6 split MFMA (encoder GEMMs and attention: 6 bf16 products per fp32 product, fp32 accumulation; merge, embedding, LayerNorm and scoring in f32)",
                 "f16x3": "f32 via f16x3 split MFMA (encoder GEMMs and attention: 3 fp16 products per fp32 product -- two fp16 pieces per operand, ~2^-21 per product --, fp32 accumulation; merge, embedding, LayerNorm and scoring in f32)",
                 "bf16x3": "f32 via bf16x3 split MFMA (encoder GEMMs and attention: 3 bf16 products per fp32 product, fp32 accumulation; merge, embedding, LayerNorm and scoring in f32)"}
        out = OrderedDict(
            metric="sequences/sec full-catalog scoring, 8-domain merged BLaIR-base; NDCG@10 parity",
            value=value, unit="sequences/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype=arith[gemm_mode], data="synthetic",
            config=dict(
                workload=f"{n_dom}-domain merged BLaIR-base (alpha=1/{n_dom}), full-catalog scoring, Arts-sized catalog",
                domains_merged=n_dom, catalog_items=M, users_per_step_per_gpu=U_step, items_per_step_per_gpu=I_step,
                avg_user_tokens=avg_user_tokens, avg_item_tokens=avg_item_tokens, topk=50, params=merged_model.layout.numel,
                scoring_route=("fused (selection inside the scoring kernel, no score block)"
                               if ops._lib.load().mr_score_topk_ws_bytes_ex(U_step, M, d, 50) < 4 * U_step * M else
                               f"staged (scoring GEMM into a {4 * U_step * M / 1e6:.0f} MB cache-resident block + row select; mr_score_fused_mode auto)"),
                path="product objects with defaults: load_merging_module -> RecModule.forward(BatchItem) / test_step(BatchSequence) / on_test_epoch_end; "
                     "input checks in the packing kernel, no per-step host sync",
                parallelism=(f"dp{world}: " + ("task vectors + base sliced 1/N per rank, arena-slice merge + all-gather" if sliced else
                                               "task vectors replicated, whole-arena merge per rank (no collective)")
                             + ", catalog rows sharded + all-gather, users data-parallel (token-balanced shards), label ranks gathered"),
            ),
            # multi-GPU record: ranks of the RCCL communicator (backend "nccl" IS RCCL on ROCm; 0 = no communicator, e.g. N = 1 or a gloo rehearsal)
            dist_backend=backend, rccl_ranks=(dist.get_world_size() if backend == "nccl" else 0),
            merge_placement=("sliced" if sliced else "replicated") if world > 1 else "single",
            # per-rank view (N > 1): each rank's own time to the end of its work, and what it spent inside the data-path collectives
            per_rank=(None if world == 1 else dict(
                ms_per_step_min=float(per_rank[:, 0].min()) / args.steps * 1e3, ms_per_step_max=float(per_rank[:, 0].max()) / args.steps * 1e3,
                collective_ms_per_step_max=float(per_rank[:, 1].max()) / args.steps * 1e3,
                collective_bytes_received_per_rank_and_step=float(per_rank[:, 2].max()) / args.steps,
                note="collective time = event-bracketed all-gathers on each rank's stream (includes waiting for the slowest rank)")),
            roofline=roofline, cpu_baseline=cpu_baseline, kernels=kernels, parity=parity,
            epoch_metrics_sample={k: round(v, 6) for k, v in list(metrics.items())[:3]},
        )
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_cpu_baseline(args, spec, merged_model, module, dev, M):
    """Times the oracle (oracle/ref_cpu.py, fp32 torch on the host cores) on a bounded sample of the same step and scales it to the
    step's composition.  Also the in-run parity record: the GPU path against the oracle on the very same sample -- merged parameters,
    embeddings, logits, ranked indices and NDCG@10 (sample users scored against the sample items; the committed real-scale fixture
    tests/golden/g12 is the full-size proof)."""
    from oracle import ref_cpu as O
    from mergerec_amd.model_batch import BatchItem, BatchSequence
    from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths, _ids_from_lengths

    # threads actually used: this process's CPU share (the GPU box exposes 256 logical CPUs but grants a
    # 16-CPU share per GPU; 256 torch threads oversubscribe it by 16x)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(avail, 16)
    torch.set_num_threads(cores)
    layout = merged_model.layout
    n_dom = merged_model.task_vectors_tensor.shape[0]
    base_c, tv_c = merged_model.base_model_tensor.data.cpu(), merged_model.task_vectors_tensor.data.cpu()
    a_c = merged_model.effective_alpha().detach().cpu().reshape(-1)
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
        E_c = module.item_embeddings.detach().cpu()
        nU = 32  # the reference's per-batch scoring shape (module.py:137 at --batch_size 32)
        Uc = O.maybe_normalize(torch.randn(nU, spec.hidden, generator=g))
        t0 = time.perf_counter()
        sc = O.score(Uc, E_c)
        O.topk_canonical(sc, 50)
        t_score = time.perf_counter() - t0
    U_step, I_step = args.users_per_step, args.items_per_step
    t_step = t_merge + I_step * (t_items / args.cpu_items) + U_step * (t_users / args.cpu_seqs) + (U_step / nU) * t_score

    # ---- parity of the GPU product path on the very same sample
    with torch.no_grad():
        e_g = module.forward(BatchItem(items=ib).to(dev))
        ref_scores = O.score(u_c, e_c)
        k = min(50, args.cpu_items)
        ref_top = torch.topk(ref_scores, k, dim=1)
        pos = (torch.exp(torch.rand(args.cpu_seqs, generator=g) * torch.log(torch.tensor(float(k)))).floor().long() - 1).clamp(0, k - 1)
        labels = ref_top.indices[torch.arange(args.cpu_seqs), pos]  # labels the oracle ranks log-uniformly inside the list (NDCG@10 ~ 0.5)
        probe = type(module)(model=module.model, evaluator=module.evaluator, similarity="cosine")
        probe.eval()
        probe.item_embeddings = torch.nn.Parameter(e_g, requires_grad=False)
        probe.on_test_epoch_start()
        probe.test_step(BatchSequence(sequence=ub, labels=labels).to(dev), 0)
        got = probe.on_test_epoch_end()
    u_g, idx_g = probe.eval_user_embeddings, probe.eval_topk_indices
    _, ref_idx = O.topk_canonical(ref_scores, k)
    want = O.evaluate(ref_scores, labels, ["NDCG", "RECALL"], [1, 5, 10, 50], "test/")
    sa, sb = torch.gather(ref_scores, 1, idx_g), torch.gather(ref_scores, 1, ref_idx)
    merged_g = torch.cat([v.reshape(-1) for v in module.model.state_dict().values()]).cpu()
    merged_cc = torch.cat([v.reshape(-1) for v in sd.values()])
    parity = dict(
        merged_params_bit_exact=bool(torch.equal(merged_g, merged_cc)),
        user_embedding_max_abs_diff=float((u_g - u_c).abs().max()), item_embedding_max_abs_diff=float((e_g.cpu() - e_c).abs().max()),
        logit_max_abs_diff=float((u_g @ e_g.cpu().T - ref_scores).abs().max()), tolerance=1e-4,
        ndcg10_abs_diff=abs(got["test/NDCG@10"] - want["test/NDCG@10"]), ndcg10_oracle=want["test/NDCG@10"], ndcg10_tolerance=1e-3,
        topk_rows_mismatched=int((idx_g != ref_idx).any(1).sum()),
        topk_rows_mismatched_beyond_near_ties=int((((idx_g != ref_idx) & ((sa - sb).abs() > 2e-6)).any(1)).sum()), near_tie=2e-6,
        sample=f"{args.cpu_seqs} users x {args.cpu_items} items of this run, top-{k}; full-size proof: tests/test_realscale_gpu.py (tests/golden/g12)",
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

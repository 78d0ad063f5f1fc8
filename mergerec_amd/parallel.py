"""One-process-per-GPU sharding of the path over the 8 GPUs of a node (RCCL over xGMI via
torch.distributed backend "nccl").  The reference is single-GPU; this partitioning is new.

  merge      : two placements.  "sliced" (``sharded_merge``; the default of ``load_merging_module`` and of bench.py with several ranks:
               north_star's split): rank r holds elements [lo_r, hi_r) of the base vector and of every task vector (64-float aligned
               slices), merges that arena slice with the same kernel, then ONE all-gather of the merged slices fills every rank's arena
               (P_pad/8 * 4 B = 62 MB per rank for BLaIR-base).  "replicated" (alpha learning, ``merge_train.py``; ``--merge-placement
               replicated`` in the bench): every rank keeps all task vectors (8 x 0.5 GB of 288 GB) and merges the whole arena locally
               with the N = 1 kernel call -- no collective.  No other collective touches parameters.
  catalog    : item rows are split in contiguous blocks; each rank encodes its block, one all-gather of the
               (M/world, d) embedding blocks gives every rank the full E (row index == item id is kept).
  users      : data-parallel over test sequences, token-balanced across ranks (``balanced_share``); scoring is local against the full E.
  metrics    : label ranks / lse / label logits are all-gathered (a few bytes per user); the metric sums
               are then evaluated identically on every rank.

The compute callables are passed in, so the partition/collective logic is testable on CPU with gloo."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import time

import torch
import torch.distributed as dist

ALIGN = 64


def init_from_env(verbose: bool = True) -> Tuple[int, int]:
    """Entry-point helper of the CLIs: under ``python -m torch.distributed.run --nproc-per-node N <script>`` (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment) bind this process to its GPU and create the process group -- backend "nccl" (= RCCL
    over xGMI), or MERGEREC_DIST_BACKEND=gloo to rehearse several ranks on fewer GPUs (collectives staged through the host).
    Without those variables: a no-op, (0, 1).  Must run before any model is built (models take ``torch.cuda.current_device()``)."""
    import os

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or is_dist():
        return world()
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    backend = os.environ.get("MERGEREC_DIST_BACKEND", "nccl")
    # RCCL's (and torch's) cross-process buffer sharing needs dmabuf IPC on this driver stack; the runtime reads the variable when it
    # initialises, which nothing has done yet (device_count() below does not): a launcher that did not export it still gets a working group
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ws > ndev and int(os.environ.get("LOCAL_WORLD_SIZE", ws)) > ndev:
        raise SystemExit(f"{ws} ranks need {ws} GPUs, {ndev} visible (set MERGEREC_DIST_BACKEND=gloo to share GPUs in a rehearsal)")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    if verbose and dist.get_rank() == 0:
        print(f"[dist] {ws} ranks, backend {backend}")
    return world()


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


@dataclass(frozen=True)
class SlicePlan:
    """Equal 64-aligned slices of a length-`padded` vector; `padded` is a multiple of world * 64."""

    total: int
    world: int

    @property
    def slice_len(self) -> int:
        per = (self.total + self.world - 1) // self.world
        return (per + ALIGN - 1) // ALIGN * ALIGN

    @property
    def padded(self) -> int:
        return self.slice_len * self.world

    def bounds(self, rank: int) -> Tuple[int, int]:
        lo = rank * self.slice_len
        return lo, lo + self.slice_len


def row_blocks(n_rows: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal [lo, hi) row blocks (first `rem` ranks get one more row)."""
    q, r = divmod(n_rows, world_size)
    out, lo = [], 0
    for i in range(world_size):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def balanced_share(pool_lengths: torch.Tensor, world_size: int, rank: int) -> torch.Tensor:
    """Token-balanced data-parallel sharding: indices into a global pool of ``world_size * n`` sequence lengths (the same on every rank)
    for this rank's ``n`` sequences.  The length-sorted pool is dealt in rounds of ``world_size``, alternating direction (snake order):
    equal counts, token totals within a fraction of a per cent of each other -- so no rank waits for the slowest at the step's
    all-gather (independent random shards of Amazon-shaped lengths differ by up to ~13 % in tokens at 8 ranks)."""
    total = pool_lengths.numel()
    if total % world_size:
        raise ValueError("pool size must be a multiple of the world size")
    n = total // world_size
    order = torch.argsort(pool_lengths, descending=True, stable=True)
    rounds = order.view(n, world_size)
    rounds = torch.where((torch.arange(n) % 2 == 1).view(-1, 1), rounds.flip(1), rounds)
    return rounds[:, rank]


class CollectiveLog:
    """Time and bytes of the data-path collectives (bench.py reports them beside the step time).  Device collectives ("nccl" = RCCL) are
    bracketed by two events on the stream they are enqueued on -- no synchronisation, the durations are read after the run; host-staged
    ones (gloo rehearsals) by the host clock.  Off unless ``enabled``."""

    def __init__(self):
        self.enabled = False
        self.records = []   # (name, bytes received per rank, (start event, end event) | host seconds)

    def summary(self):
        ms, nbytes, n = 0.0, 0, 0
        for _, b, t in self.records:
            ms += (t[0].elapsed_time(t[1]) if isinstance(t, tuple) else t * 1e3)
            nbytes += b
            n += 1
        return dict(calls=n, ms=ms, bytes_received_per_rank=nbytes)


COLL = CollectiveLog()


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None):
    """all_gather_into_tensor; with the gloo backend (CPU rehearsals, or several ranks sharing one GPU in tests)
    device tensors are staged through host memory.  The production backend is "nccl" (= RCCL over xGMI)."""
    rec = COLL.enabled
    nbytes = out.numel() * out.element_size() - inp.numel() * inp.element_size()
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        if rec:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(o, i, group=group)
        out.copy_(o)
        if rec:
            torch.cuda.synchronize()
            COLL.records.append(("all_gather(host-staged)", nbytes, time.perf_counter() - t0))
    else:
        ev = None
        if rec and inp.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_gather_into_tensor(out, inp, group=group)
        if ev is not None:
            ev[1].record()
            COLL.records.append(("all_gather", nbytes, ev))


def sharded_merge(merge_slice: Callable[[int, int, torch.Tensor], None], arena: torch.Tensor, plan: SlicePlan,
                  scratch: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """merge_slice(p_begin, p_count, out_slice) writes this rank's merged slice; all-gather into `arena`
    (length plan.padded).  world == 1 degenerates to a plain in-place merge."""
    rank, ws = world()
    assert arena.numel() == plan.padded, "arena must be allocated with SlicePlan.padded elements"
    lo, hi = plan.bounds(rank)
    if ws == 1:
        merge_slice(lo, hi - lo, arena[lo:hi])
        return arena
    if scratch is None:
        scratch = torch.empty(hi - lo, dtype=arena.dtype, device=arena.device)
    merge_slice(lo, hi - lo, scratch)
    _all_gather_into(arena, scratch, group=group)
    return arena


def all_gather_rows(local: torch.Tensor, blocks: Sequence[Tuple[int, int]], group=None) -> torch.Tensor:
    """Concatenate per-rank row blocks (possibly unequal) in rank order -> (sum rows, d)."""
    rank, ws = world()
    if ws == 1:
        return local
    d = local.shape[1:]
    mx = max(hi - lo for lo, hi in blocks)
    pad = torch.zeros((mx, *d), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((ws * mx, *d), dtype=local.dtype, device=local.device)
    _all_gather_into(out, pad, group=group)
    return torch.cat([out[i * mx : i * mx + (hi - lo)] for i, (lo, hi) in enumerate(blocks)], dim=0)


def all_gather_vector(local: torch.Tensor, group=None) -> torch.Tensor:
    """Variable-length 1-D gather (per-user ranks / lse), rank order preserved."""
    rank, ws = world()
    if ws == 1:
        return local
    n = torch.tensor([local.numel()], dtype=torch.int64)
    ns = torch.zeros(ws, dtype=torch.int64)
    if dist.get_backend(group) == "gloo":
        dist.all_gather_into_tensor(ns, n, group=group)
    else:
        nd = ns.to(local.device)
        dist.all_gather_into_tensor(nd, n.to(local.device), group=group)
        ns = nd.cpu()
    sizes = [int(x) for x in ns]
    blocks, lo = [], 0
    for s in sizes:
        blocks.append((lo, lo + s))
        lo += s
    return all_gather_rows(local.reshape(-1, 1), blocks, group=group).reshape(-1)


# ------------------------------------------------------------------------------------------------ dataloader shards
def deal_rows(weights: Optional[torch.Tensor], n_rows: int, world_size: int) -> List[torch.Tensor]:
    """Row ids of every rank, each list ascending.  ``weights`` None: contiguous near-equal blocks (``row_blocks``).  Otherwise the
    rows are sorted by weight (tokens, or a cheap proxy such as the number of items of a sequence) and dealt in snake order --
    counts differ by at most one and the weight totals by a fraction of a per cent at Amazon-shaped length distributions."""
    if weights is None:
        return [torch.arange(lo, hi) for lo, hi in row_blocks(n_rows, world_size)]
    if weights.numel() != n_rows:
        raise ValueError("one weight per row expected")
    order = torch.argsort(weights.to(torch.float64), descending=True, stable=True)
    pos = torch.arange(n_rows)
    rnd, col = pos // world_size, pos % world_size
    owner = torch.where(rnd % 2 == 1, world_size - 1 - col, col)
    return [torch.sort(order[owner == r]).values for r in range(world_size)]


def _batch_rows(batch) -> int:
    enc = getattr(batch, "items", None)
    if enc is None:
        enc = batch.sequence
    return int(enc["input_ids"].shape[0])


class ShardedLoader:
    """This rank's share of a dataloader that yields BatchItem / BatchSequence, plus what is needed to put per-row results back
    into the loader's row order on EVERY rank (``gather_rows``).  With one rank (or no process group) it is the loader itself.

    Accepted loaders, cheapest split first:
      * ``data.TokenizedBatches`` (pre-tokenised tensors): row-level split, exact token balance;
      * ``torch.utils.data.DataLoader`` over a map-style dataset in sequential order: the same dataset / collator / batch size over
        this rank's rows only, so each rank tokenises just its share; balance proxy = the length of a sample's item list;
      * a list / tuple of batches, or any other iterable (materialised): contiguous blocks of whole batches.
    ``balance`` False keeps contiguous blocks (the catalog: item lengths are near-uniform)."""

    def __init__(self, loader, balance: bool = True, group=None):
        self.group = group
        self.rank, self.world = world()
        self.loader = loader
        self.index_lists: Optional[List[torch.Tensor]] = None
        if self.world == 1:
            self.local = loader
            return
        from .data import TokenizedBatches

        W, r = self.world, self.rank
        if isinstance(loader, TokenizedBatches) and loader.rows is None:
            n = loader.hi - loader.lo
            wts = loader.enc["attention_mask"][loader.lo:loader.hi].ne(0).sum(1) if balance else None
            self.index_lists = [ix + loader.lo for ix in deal_rows(wts, n, W)]
            self.local = TokenizedBatches(loader.enc, loader.bs, loader.labels, rows=self.index_lists[r])
            self.index_lists = [ix - loader.lo for ix in self.index_lists]
            return
        try:
            from torch.utils.data import DataLoader, SequentialSampler, Subset
        except Exception:  # pragma: no cover
            DataLoader = None
        if DataLoader is not None and isinstance(loader, DataLoader) and isinstance(getattr(loader, "sampler", None), SequentialSampler) \
                and hasattr(loader.dataset, "__getitem__") and loader.batch_size is not None:
            ds = loader.dataset
            n = len(ds)
            wts = self._balance_weights(ds, n) if balance and n else None
            self.index_lists = deal_rows(wts, n, W)
            # the same loader over this rank's rows: every DataLoader option that does not depend on the sampler is kept
            extra = dict(pin_memory=loader.pin_memory, worker_init_fn=loader.worker_init_fn, timeout=loader.timeout)
            if loader.num_workers > 0:
                extra.update(persistent_workers=loader.persistent_workers, prefetch_factor=loader.prefetch_factor,
                             multiprocessing_context=loader.multiprocessing_context)
            self.local = DataLoader(Subset(ds, self.index_lists[r].tolist()), batch_size=loader.batch_size, collate_fn=loader.collate_fn,
                                    shuffle=False, num_workers=loader.num_workers, drop_last=False, **extra)
            return
        batches = loader if isinstance(loader, (list, tuple)) else list(loader)
        sizes = [_batch_rows(b) for b in batches]
        starts = [0]
        for n in sizes:
            starts.append(starts[-1] + n)
        blocks = row_blocks(len(batches), W)
        self.index_lists = [torch.arange(starts[lo], starts[hi]) for lo, hi in blocks]
        lo, hi = blocks[r]
        self.local = batches[lo:hi]

    def _balance_weights(self, ds, n: int) -> Optional[torch.Tensor]:
        """One weight per sample for the token-balanced deal, computed ONCE: from the dataset's own length table when it has one
        (``balance_weights()``: no sample is materialised), else by rank 0 walking the dataset and one broadcast -- not by every rank
        calling ``ds[i]`` n times before its first batch.  None: samples carry no length proxy -> contiguous blocks."""
        if hasattr(ds, "balance_weights"):
            w = ds.balance_weights()
            return None if w is None else torch.as_tensor(w, dtype=torch.int64)
        probe = ds[0]
        if not (isinstance(probe, (tuple, list)) and len(probe) == 2 and hasattr(probe[1], "__len__")):
            return None
        wts = torch.zeros(n, dtype=torch.int64)
        if self.rank == 0:
            wts = torch.tensor([len(ds[i][1]) for i in range(n)], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            buf = wts.to(dev)
            dist.broadcast(buf, src=0, group=self.group)
            wts = buf.cpu()
        else:
            dist.broadcast(wts, src=0, group=self.group)
        return wts

    def __iter__(self):
        return iter(self.local)

    def __len__(self):
        return len(self.local)

    @property
    def n_rows(self) -> Optional[int]:
        return None if self.index_lists is None else sum(ix.numel() for ix in self.index_lists)

    def gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """(n_local, ...) per-row results in this rank's row order -> (n_rows, ...) in the loader's row order, on every rank.
        ONE all-gather of max-count-padded blocks (RCCL over xGMI under the "nccl" backend)."""
        if self.world == 1:
            return local
        counts = [ix.numel() for ix in self.index_lists]
        if local.shape[0] != counts[self.rank]:
            raise ValueError(f"rank {self.rank} produced {local.shape[0]} rows for a shard of {counts[self.rank]}")
        mx = max(counts)
        tail = local.shape[1:]
        send = local.new_zeros((mx, *tail))
        send[: local.shape[0]] = local
        recv = local.new_empty((self.world * mx, *tail))
        _all_gather_into(recv, send.contiguous(), group=self.group)
        out = local.new_empty((sum(counts), *tail))
        for r, ix in enumerate(self.index_lists):
            if ix.numel():
                out[ix.to(local.device)] = recv[r * mx : r * mx + ix.numel()]
        return out


# ------------------------------------------------------------------------------------------------ data-parallel alpha training
def allreduce_mean_grads(params, group=None) -> None:
    """Data-parallel collaborative merging (BASELINE config 5: pseudo-user batches split over the GPUs of a node): every rank runs
    the step on its own batch shard; the only exchange is the mean of d loss / d alpha -- a few dozen floats -- in ONE all-reduce
    (RCCL on GPUs; staged through the host for the gloo rehearsal).  A no-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    ws = dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo" and grads[0].is_cuda

    def reduce_(t):
        if staged:
            host = t.cpu()
            dist.all_reduce(host, group=group)
            t.copy_(host)
        else:
            dist.all_reduce(t, group=group)
        t /= ws

    if len(grads) == 1 and grads[0].is_contiguous():  # the fine-tuning arena: one contiguous gradient, reduced in place
        reduce_(grads[0].view(-1))
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    reduce_(flat)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def shard_indices(n: int, rank: int, world: int, epoch_seed: int, shuffle: bool = True):
    """The sample indices of one rank for one epoch: a seeded permutation shared by all ranks, padded by wrap-around to a multiple
    of the world size and dealt round-robin (torch's DistributedSampler rule), so every rank sees the same number of batches."""
    if shuffle:
        g = torch.Generator().manual_seed(epoch_seed)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    total = (n + world - 1) // world * world
    order = order + order[: total - n]
    return order[rank:total:world]

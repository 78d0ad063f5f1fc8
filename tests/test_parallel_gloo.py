"""World-size-2 gloo tests (CPU) of the multi-GPU partition + collective logic.  The compute callables are
the oracle's CPU restatements (the HIP kernels cannot run here); what is under test is slicing, alignment,
all-gather assembly and that the sharded result equals the single-process result bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        ret[rank] = fn(rank, world_size)
    finally:
        dist.destroy_process_group()


def _run(fn, world_size=2):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world_size, _free_port(), fn, ret), nprocs=world_size, join=True)
    return [ret[r] for r in range(world_size)]


def _merge_job(rank, world_size):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mergerec_amd.parallel import SlicePlan, sharded_merge
    from oracle import ref_cpu as O

    g = torch.Generator().manual_seed(0)
    P = 64 * 37 + 64 * 5  # not a multiple of world * 64 * anything nice
    plan = SlicePlan(P, world_size)
    base = torch.zeros(plan.padded); base[:P] = torch.randn(P, generator=g)
    tv = torch.zeros(3, plan.padded); tv[:, :P] = torch.randn(3, P, generator=g)
    seg_off = [0, 64 * 10, 64 * 11, plan.padded]
    alpha = torch.rand(3, 3, generator=g)

    def merge_slice(p_begin, p_count, out):
        for p0 in range(p_begin, p_begin + p_count, 64):
            s = max(i for i in range(3) if seg_off[i] <= p0)
            out[p0 - p_begin : p0 - p_begin + 64] = O.merge_task_wise(base[p0 : p0 + 64], tv[:, p0 : p0 + 64], alpha[s])

    arena = torch.full((plan.padded,), float("nan"))
    sharded_merge(merge_slice, arena, plan)
    full = torch.empty(plan.padded)
    merge_slice_all = lambda: [merge_slice(0, plan.padded, full)]
    merge_slice_all()
    return bool(torch.equal(arena, full)), plan.bounds(rank)


def test_sharded_merge_equals_single_process():
    res = _run(_merge_job, 2)
    assert all(ok for ok, _ in res)
    (lo0, hi0), (lo1, hi1) = res[0][1], res[1][1]
    assert lo0 == 0 and hi0 == lo1 and lo0 % 64 == 0 and lo1 % 64 == 0 and hi1 % 64 == 0


def _catalog_job(rank, world_size):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mergerec_amd.parallel import all_gather_rows, all_gather_vector, row_blocks

    M, d = 101, 8
    full = torch.arange(M * d, dtype=torch.float32).view(M, d)
    blocks = row_blocks(M, world_size)
    lo, hi = blocks[rank]
    E = all_gather_rows(full[lo:hi].clone(), blocks)
    ranks_local = torch.arange(rank * 10, rank * 10 + 3 + rank, dtype=torch.int32)
    allr = all_gather_vector(ranks_local)
    return bool(torch.equal(E, full)), allr.tolist(), blocks


def test_catalog_blocks_and_metric_gather():
    res = _run(_catalog_job, 2)
    assert all(r[0] for r in res)
    assert res[0][1] == res[1][1] == [0, 1, 2, 10, 11, 12, 13]
    assert res[0][2] == [(0, 51), (51, 101)]


def test_slice_plan_covers_everything():
    from mergerec_amd.parallel import SlicePlan, row_blocks

    for total in (64, 124_645_632 + 64 * 7, 1_000_000 - 64 * 3 + 64):
        for w in (1, 2, 4, 8):
            p = SlicePlan(total, w)
            assert p.padded >= total and p.padded % (64 * w) == 0
            assert [p.bounds(r) for r in range(w)][-1][1] == p.padded
    assert row_blocks(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert row_blocks(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


def _dp_grad_job(rank, world_size):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mergerec_amd.parallel import allreduce_mean_grads, shard_indices

    ps = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, 2)), torch.nn.Parameter(torch.zeros(1))]
    ps[0].grad = torch.full((3,), float(rank + 1))
    ps[1].grad = torch.arange(4.0).view(2, 2) * (rank + 1)
    allreduce_mean_grads(ps)  # ps[2] has no gradient: skipped
    idx = shard_indices(11, rank, world_size, epoch_seed=7)
    return ps[0].grad.tolist(), ps[1].grad.tolist(), ps[2].grad, idx


def test_data_parallel_alpha_gradients_and_sharding():
    out = _run(_dp_grad_job)
    for g0, g1, g2, _ in out:
        assert g0 == [1.5, 1.5, 1.5] and g1 == [[0.0, 1.5], [3.0, 4.5]] and g2 is None
    a, b = out[0][3], out[1][3]
    assert len(a) == len(b) == 6 and set(a) | set(b) == set(range(11))  # every sample seen, equal batch counts (one wrap-around)


def test_balanced_share_equalises_tokens():
    """token-balanced sharding of Amazon-shaped lengths: a partition of the pool, equal counts, totals within 0.5 % (independent draws: > 5 %)"""
    from mergerec_amd.parallel import balanced_share
    from mergerec_amd.synthetic import blair_sequence_lengths

    world, per = 8, 256
    pool = blair_sequence_lengths(world * per, torch.Generator().manual_seed(4321))
    shares = [balanced_share(pool, world, r) for r in range(world)]
    assert all(s.numel() == per for s in shares)
    assert sorted(torch.cat(shares).tolist()) == list(range(world * per))
    tokens = [int(pool[s].sum()) for s in shares]
    assert max(tokens) <= 1.005 * min(tokens), tokens
    squares = [float((pool[s].double() ** 2).sum()) for s in shares]  # attention work
    assert max(squares) <= 1.01 * min(squares), squares
    with pytest.raises(ValueError):
        balanced_share(pool[:-1], world, 0)



def _sharded_loader_job(rank, world_size):
    """Every loader form ShardedLoader accepts: the per-row 'result' (here: a row signature computed from the batch tensors)
    gathered over the ranks must equal the one a single process computes, in the loader's row order."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from torch.utils.data import DataLoader, Dataset

    from mergerec_amd.data import TokenizedBatches, coalesce_batches, load_domain
    from mergerec_amd.model_batch import BatchItem, BatchSequence
    from mergerec_amd.parallel import ShardedLoader

    def signature(batches):  # per row: (sum of attended ids, label or -1) -- depends on the row only, not on its batch
        rows = []
        for b in batches:
            enc = b.items if isinstance(b, BatchItem) else b.sequence
            s = (enc["input_ids"] * enc["attention_mask"]).sum(1).double()
            lab = b.labels.double() if isinstance(b, BatchSequence) else torch.full_like(s, -1.0)
            rows.append(torch.stack([s, lab], dim=1))
        return torch.cat(rows) if rows else torch.empty(0, 2, dtype=torch.float64)

    dom = load_domain("synthetic:Toy:53:41", vocab=300)
    out = {}
    # (a) pre-tokenised tensors: row-level, token-balanced (users) / contiguous (items)
    for name, loader, balance in (("tok_users", dom.sequence_dataloader(8), True), ("tok_items", dom.item_dataloader(8), False)):
        sh = ShardedLoader(loader, balance=balance)
        got = sh.gather_rows(signature(coalesce_batches(sh, 4096)))
        out[name] = (bool(torch.equal(got, signature(loader))), [ix.numel() for ix in sh.index_lists],
                     [int(dom.sequences["attention_mask"][ix].sum()) for ix in sh.index_lists] if balance else None)
    # (b) a torch DataLoader over a map-style dataset whose samples are (index, item list): Subset route, item-count balance
    class DS(Dataset):
        def __init__(self):
            g = torch.Generator().manual_seed(5)
            self.seqs = [list(range(int(n))) for n in torch.randint(1, 30, (37,), generator=g)]
        def __len__(self):
            return len(self.seqs)
        def __getitem__(self, i):
            return i, self.seqs[i]
    def collate(samples):
        L = max(len(s) for _, s in samples)
        ids = torch.tensor([[3 + x for x in s] + [1] * (L - len(s)) for _, s in samples])
        mask = torch.tensor([[1] * len(s) + [0] * (L - len(s)) for _, s in samples])
        return BatchSequence(sequence={"input_ids": ids, "attention_mask": mask}, labels=torch.tensor([i for i, _ in samples]))
    dl = DataLoader(DS(), batch_size=5, collate_fn=collate)
    sh = ShardedLoader(dl, balance=True)
    got = sh.gather_rows(signature(sh))
    out["dataloader"] = (bool(torch.equal(got, signature(dl))), [ix.numel() for ix in sh.index_lists], None)
    # (c) a plain list of batches: whole-batch blocks
    batches = list(dom.sequence_dataloader(7))
    sh = ShardedLoader(batches, balance=True)
    got = sh.gather_rows(signature(sh))
    out["list"] = (bool(torch.equal(got, signature(batches))), [ix.numel() for ix in sh.index_lists], None)
    # (d) a generator (materialised)
    sh = ShardedLoader((b for b in dom.item_dataloader(6)), balance=False)
    got = sh.gather_rows(signature(sh))
    out["generator"] = (bool(torch.equal(got, signature(dom.item_dataloader(6)))), [ix.numel() for ix in sh.index_lists], None)
    return out


def test_sharded_loader_reassembles_rows_for_every_loader_form():
    res = _run(_sharded_loader_job, 2)
    assert res[0] == res[1]
    for name, (ok, counts, tokens) in res[0].items():
        assert ok, name
        assert sum(counts) in (41, 53, 37) and abs(counts[0] - counts[1]) <= 7, (name, counts)
    counts, tokens = res[0]["tok_users"][1], res[0]["tok_users"][2]
    assert abs(counts[0] - counts[1]) <= 1 and max(tokens) <= 1.08 * min(tokens), (counts, tokens)  # snake dealing: near-equal token totals


def test_deal_rows_partitions_and_single_rank_is_identity():
    from mergerec_amd.data import load_domain
    from mergerec_amd.parallel import ShardedLoader, deal_rows

    w = torch.tensor([5, 1, 9, 3, 7, 2, 8])
    shares = deal_rows(w, 7, 3)
    assert sorted(torch.cat(shares).tolist()) == list(range(7)) and all(s.tolist() == sorted(s.tolist()) for s in shares)
    assert [s.tolist() for s in deal_rows(None, 7, 3)] == [[0, 1, 2], [3, 4], [5, 6]]
    loader = load_domain("synthetic:Toy:20:10", vocab=300).item_dataloader(4)
    sh = ShardedLoader(loader)  # no process group: the loader itself
    assert sh.local is loader and sh.gather_rows(torch.ones(3)).tolist() == [1, 1, 1]


def test_bench_gpus_n_spawns_children_and_returns_their_exit_code():
    """`python bench.py --gpus 2` outside torch.distributed.run starts its own ranks as a child process (bench.spawn_ranks) and must not
    paper over a rank that dies: with an unusable backend every rank fails at process-group creation (or, on a box without a GPU, at the
    device binding before it), the parent prints no result line and exits with the launcher's non-zero code."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PYTHONPATH=str(root), MERGEREC_DIST_BACKEND="no-such-backend")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip(), (r.returncode, r.stdout[-500:])
    assert "torch.distributed.run" in r.stderr or "ChildFailedError" in r.stderr or "elastic" in r.stderr

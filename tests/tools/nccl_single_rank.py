#!/usr/bin/env python3
"""RCCL ("nccl" backend) smoke on ONE GPU with ONE rank: the process group the CLIs create (parallel.init_from_env's nccl branch) and every
collective call shape / dtype the product issues under it -- all_gather_into_tensor on fp32 / int64 / int32 row blocks and on a 1-D arena
slice, all_reduce (sum, in place; max on float64), barrier.  With one rank a collective is the identity, so results are checked exactly; what this
covers is that RCCL initialises on the device and accepts these calls (several ranks need several GPUs: tests/test_dist_gpu.py rehearses those
over gloo with the real kernels).  Prints OK."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from mergerec_amd import parallel  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    g = torch.Generator().manual_seed(0)
    for dtype, shape in ((torch.float32, (37, 768)), (torch.int64, (37, 52)), (torch.int32, (5, 3)), (torch.float32, (64 * 1000,))):
        src = (torch.randn(shape, generator=g) * 100).to(dtype).to(dev)
        out = torch.empty_like(src)
        parallel._all_gather_into(out, src)
        assert torch.equal(out, src), (dtype, shape)
    # the arena path: merge a slice into scratch, all-gather into the arena
    plan = parallel.SlicePlan(total=1000, world=1)
    arena = torch.zeros(plan.padded, device=dev)
    scratch = torch.arange(plan.padded, dtype=torch.float32, device=dev)
    parallel._all_gather_into(arena, scratch)
    assert torch.equal(arena, scratch)
    a = torch.arange(8, dtype=torch.float32, device=dev)
    dist.all_reduce(a)
    assert torch.equal(a.cpu(), torch.arange(8, dtype=torch.float32))
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 1.25
    n = torch.tensor([5], dtype=torch.int64, device=dev)
    ns = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, n)
    assert int(ns[0]) == 5
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()

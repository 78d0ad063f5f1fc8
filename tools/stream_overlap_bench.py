"""EXPERIMENT: does running two independent halves of the packed batch on two HIP streams (one half's attention beside the other's
GEMMs) beat one pass over the whole batch?  BLaIR-base, bf16x3, 128 items + 256 users."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd.engine import ArenaLayout, EncoderRunner, EncoderSpec, WeightSet
from mergerec_amd.synthetic import blair_item_lengths, blair_sequence_lengths, _ids_from_lengths

dev = torch.device("cuda:0")
spec = EncoderSpec.blair_base()
layout = ArenaLayout(spec.param_shapes("model."))
g = torch.Generator().manual_seed(0)
arena = (torch.randn(layout.padded_numel, generator=g) * 0.02).to(dev)
W = WeightSet(layout, arena, os.environ.get("SO_MODE", "bf16x3")).refresh()
runner = EncoderRunner(spec)
lens = torch.cat([blair_item_lengths(128, g), blair_sequence_lengths(256, g)])
batch = _ids_from_lengths(lens, spec.vocab, g)


def sub(idx):
    return {k: v[idx] for k, v in batch.items()}, lens[idx]


order = torch.argsort(lens, descending=True)
halves = [order[0::2], order[1::2]]  # length-balanced halves
pb_all = runner.pack(batch, dev, lens=lens, validate=False)
pbs = [runner.pack(*sub(h), validate=False, device=dev) if False else runner.pack(sub(h)[0], dev, lens=sub(h)[1], validate=False) for h in halves]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def one():
    return runner.forward_packed(W, pb_all, normalize=True)


def two_seq():
    return [runner.forward_packed(W, pb, normalize=True) for pb in pbs]


def two_par():
    outs = []
    cur = torch.cuda.current_stream(dev)
    for st, pb in zip(streams, pbs):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(runner.forward_packed(W, pb, normalize=True))
    for st in streams:
        cur.wait_stream(st)
    return outs


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


a = one()
b = two_par()
full = torch.empty_like(a)
full[halves[0].to(dev)] = b[0]
full[halves[1].to(dev)] = b[1]
print("max |one pass - two streams| =", float((a - full).abs().max()))
for r in range(2):
    print(f"round {r}: one pass {timeit(one):.2f} ms | two halves, one stream {timeit(two_seq):.2f} ms | two halves, two streams {timeit(two_par):.2f} ms")

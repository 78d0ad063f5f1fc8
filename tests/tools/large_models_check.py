"""BLaIR-large and Recformer-large (24 x 1024, 16 heads) through the HIP encoder against the CPU oracle on a few short sequences,
all three GEMM modes; random weights at the true dims.  Prints max |difference| of the normalised CLS embeddings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collections import OrderedDict
import torch
from mergerec_amd.engine import ArenaLayout, EncoderRunner, EncoderSpec, WeightSet
from oracle import ref_cpu as O

dev = "cuda:0"


def check(kind, spec, verbose=True):
    """-> {mode: (max |embedding diff|, max |cosine-logit diff|)}"""
    g = torch.Generator().manual_seed(3)
    shapes = spec.param_shapes("model.")
    sd = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith("position_ids"):
            sd[k] = torch.arange(shp[1]).view(shp).float()
        elif k.endswith("LayerNorm.weight"):
            sd[k] = 1.0 + 0.05 * torch.randn(shp, generator=g)
        else:
            sd[k] = 0.03 * torch.randn(shp, generator=g)
    layout = ArenaLayout(shapes)
    flat = layout.pack(sd, dev)
    lens = torch.tensor([37, 5, 130, 64])
    B, L = len(lens), int(lens.max())
    ids = torch.full((B, L), spec.pad_id, dtype=torch.int64)
    mask = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = torch.randint(4, 1000, (n,), generator=g)
        ids[b, 0] = 0
        mask[b, :n] = 1
    batch = {"input_ids": ids, "attention_mask": mask}
    cfg = O.EncoderConfig(hidden=spec.hidden, heads=spec.heads, layers=spec.layers, intermediate=spec.intermediate, vocab=spec.vocab, max_pos=spec.max_pos,
                          pad_id=spec.pad_id, ln_eps=spec.ln_eps, token_type_size=spec.token_type_size, max_item_embeddings=spec.max_item_embeddings,
                          one_sided_window=max(spec.one_sided_window, 0))
    t0 = time.perf_counter()
    if spec.kind == "recformer":
        tt = torch.where(mask.bool(), torch.full_like(ids, 2), torch.full_like(ids, 3)); tt[:, 0] = 0
        ip = mask.clone(); ip[:, 0] = 0
        ga = torch.zeros_like(ids); ga[:, 0] = 1
        batch.update(token_type_ids=tt, item_position_ids=ip, global_attention_mask=ga)
        ref = O.recformer_encode(sd, ids, mask, ga, tt, ip, cfg, prefix="model.")
    else:
        ref = O.roberta_encode(sd, ids, mask, cfg, prefix="model.")
    ref = O.maybe_normalize(ref)
    cpu_s = time.perf_counter() - t0
    run = EncoderRunner(spec)
    pb = run.pack(batch, dev)
    res = {}
    for mode in ("f32", "bf16x6", "f16x3", "bf16x3"):
        W = WeightSet(layout, flat, mode).refresh()
        got = run.forward_packed(W, pb, normalize=True).cpu()
        res[mode] = (float((got - ref).abs().max()), float((got @ got.T - ref @ ref.T).abs().max()))
        if verbose:
            print(f"{kind:16s} {mode:7s}: max |embedding diff| vs CPU oracle {res[mode][0]:.2e}   cosine-logit diff {res[mode][1]:.2e}   (oracle {cpu_s:.1f} s)")
    return res


if __name__ == "__main__":
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    for kind_, spec_ in (("blair_large", EncoderSpec.blair_large()), ("recformer_large", EncoderSpec.recformer_large())):
        check(kind_, spec_)

"""The drop-in CLI at a real domain's scale, end to end: a SYNTHETIC domain with Pantry's measured sizes (4,968 items, 14,178 test
users, heavy-tailed sequence lengths) written in the reference's JSON format, tokenised by the local fixture tokenizer,
pushed through ``merge_test.py`` (2-way task-vector merge of BLaIR-base at true dims, full-catalog scoring, evaluator) once per GEMM
arithmetic.  Reports, per mode, the wall time of the whole CLI run (tokenisation included) and NDCG / Recall; the north star's
"NDCG@10 within 1e-3" is checked between the exact-fp32 kernels and the two split-precision modes on all 14,178 users."""
import json
import math
import os
import random
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

WORDS = ("solar garden lamp steel bottle organic green tea cotton towel wireless mouse ceramic mug leather wallet running shoes "
         "bamboo cutting board vitamin gummies scented candle yoga mat protein bar dark roast coffee beans almond butter crunchy "
         "sea salt chips sparkling water lemon ginger honey oat milk granola dried mango trail mix rice noodles soy sauce").split()
BRANDS = ["Acme", "Northwind", "Globex", "Initech", "Umbrella", "Hooli", "Stark", "Wayne"]
CATS = ["Grocery", "Snacks", "Beverages", "Home", "Kitchen", "Sports", "Office"]


def make_domain(root: Path, n_items: int, n_users: int, seed: int = 7):
    rng = random.Random(seed)
    root.mkdir(parents=True, exist_ok=True)
    smap = {f"B{1000 + i:07d}": i for i in range(n_items)}
    meta = {a: {"title": " ".join(rng.choice(WORDS) for _ in range(rng.randint(6, 28))).capitalize(), "brand": rng.choice(BRANDS),
                "category": " ".join(rng.sample(CATS, rng.randint(1, 3)))} for a in smap}
    umap = {f"U{i:06d}": i for i in range(n_users)}
    train, val, test = {}, {}, {}
    for u in range(n_users):
        n = 3 + min(50, math.ceil(rng.lognormvariate(1.6, 0.8)))  # interactions: >= 3 so that train / val / test are non-empty
        seq = [rng.randrange(n_items) for _ in range(n)]
        train[str(u)], val[str(u)], test[str(u)] = seq[:-2], [seq[-2]], [seq[-1]]
    for name, obj in (("smap", smap), ("umap", umap), ("meta_data", meta), ("train", train), ("val", val), ("test", test)):
        (root / f"{name}.json").write_text(json.dumps(obj))


def main():
    import merge_test

    n_items, n_users = int(os.environ.get("RS_ITEMS", 4968)), int(os.environ.get("RS_USERS", 14178))
    tmp = Path(tempfile.mkdtemp(prefix="realscale_"))
    t0 = time.perf_counter()
    make_domain(tmp / "PantryLike", n_items, n_users)
    print(f"synthetic domain: {n_items} items, {n_users} users written in {time.perf_counter() - t0:.1f} s")
    argv = ["--model_type", "blair_base", "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
            "--merge_type", "task_vector", "--learn_type", "task_wise", "--weight_file", "average", "--data_paths", str(tmp / "PantryLike"),
            "--tokenizer_path", str(ROOT / "tests" / "golden" / "mini_tokenizer"), "--batch_size", "32", "--test_data_split", "test"]
    results = {}
    for mode in ("f32", "bf16x6", "bf16x3"):
        os.environ["MERGEREC_GEMM_MODE"] = mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        metrics = merge_test.main(list(argv))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        results[mode] = (metrics[0], wall)
        print(f"[{mode}] merge_test.py end to end: {wall:.1f} s wall ({n_users / wall:.0f} users/s incl. checkpoint synthesis, merge, tokenisation, "
              f"catalog encode, scoring, metrics)  NDCG@10 {metrics[0]['test/NDCG@10']:.6f}  Recall@10 {metrics[0]['test/Recall@10']:.6f}  "
              f"loss {metrics[0]['test/loss']:.6f}")
    ref = results["f32"][0]
    for mode in ("bf16x6", "bf16x3"):
        m = results[mode][0]
        worst = max(abs(m[k] - ref[k]) for k in ref if "NDCG" in k or "Recall" in k)
        print(f"{mode} vs f32: max |metric difference| over NDCG/Recall@{{1,5,10,50}} = {worst:.2e}; |loss difference| = {abs(m['test/loss'] - ref['test/loss']):.2e}")
        assert abs(m["test/NDCG@10"] - ref["test/NDCG@10"]) <= 1e-3, "north star: NDCG@10 within 1e-3"


if __name__ == "__main__":
    main()

"""Several ranks behind the drop-in surface, with the REAL kernels (BASELINE configs[3], configs[4]).

Each test launches a fresh child ``python -m torch.distributed.run --nproc-per-node 2`` on the one GPU of the box with
MERGEREC_DIST_BACKEND=gloo (collectives staged through the host: RCCL needs one GPU per rank) and compares what rank 0 saved with a
single-process run of the same script: arena slices + all-gather, catalog shards + all-gather, token-balanced user shards and the
gathered per-user results must reproduce the single-process outputs BIT FOR BIT (the encoder kernels are batch-composition
invariant, the merge kernel is elementwise)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, ranks, timeout=900, extra_env=None):
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    if ranks > 1:
        env["MERGEREC_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port())] + cmd
    else:
        cmd = [sys.executable] + cmd
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(map(str, cmd))}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout


def test_two_ranks_reproduce_single_process_bit_for_bit(tmp_path):
    worker = str(ROOT / "tests" / "tools" / "dist_worker.py")
    _run([worker, str(tmp_path / "one.pt")], 1)
    _run([worker, str(tmp_path / "two.pt")], 2)
    one, two = torch.load(tmp_path / "one.pt"), torch.load(tmp_path / "two.pt")
    assert one.pop("world") == 1 and two.pop("world") == 2
    assert set(one) == set(two) and len(one) == 3
    for key in one:
        a, b = one[key], two[key]
        assert a["placement"] == "replicated" and b["placement"] == "sliced", key  # the N > 1 default is the north-star split
        for name in ("merged", "item_embeddings", "user_embeddings", "topk", "labels", "scores", "mm_forward_cls"):
            assert a[name].shape == b[name].shape and torch.equal(a[name], b[name]), (key, name)
        assert a["metrics"] == b["metrics"], key
        assert a["metrics"]["test/NDCG@10"] >= 0.0
        pa, pb = (torch.load(str(tmp_path / f) + ".pred_" + key.replace("/", "_")) for f in ("one.pt", "two.pt"))
        assert torch.equal(pa["Toy"]["scores"], pb["Toy"]["scores"]) and torch.equal(pa["Toy"]["labels"], pb["Toy"]["labels"])


def test_merge_test_cli_under_torch_distributed_run(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 merge_test.py ...` == `python merge_test.py ...` (metrics CSV, item and user
    embeddings, predictions), for an 8-domain task-wise merge (BASELINE configs[3]'s shape on a small model)."""
    def args(tag):
        return ["merge_test.py", "--model_type", "BLAIR_BASE", "--model_kwargs", "init_seed", "7", "spec.hidden", "128", "spec.heads", "2",
                "spec.layers", "2", "spec.intermediate", "256", "--finetune_checkpoint_paths",
                *[f"synthetic:{i}" for i in range(1, 9)], "--merge_type", "task_vector", "--learn_type", "task_wise", "--weight_file", "average",
                "--data_paths", "synthetic:Pantry:500:400", "synthetic:Toys:300:200", "--precision", "bf16-mixed",
                "--metrics_path", str(tmp_path / f"{tag}.csv"), "--item_embeddings_path", str(tmp_path / f"{tag}_items.pt"),
                "--user_embeddings_path", str(tmp_path / f"{tag}_users.pt"), "--predictions_path", str(tmp_path / f"{tag}_pred.pt")]

    _run(args("one"), 1)
    _run(args("two"), 2)
    assert (tmp_path / "one.csv").read_text() == (tmp_path / "two.csv").read_text()
    for name in ("items", "users"):
        a, b = torch.load(tmp_path / f"one_{name}.pt"), torch.load(tmp_path / f"two_{name}.pt")
        assert len(a) == len(b) == 2 and all(torch.equal(x, y) for x, y in zip(a, b)), name
    pa, pb = torch.load(tmp_path / "one_pred.pt"), torch.load(tmp_path / "two_pred.pt")
    assert pa.keys() == pb.keys() == {"Pantry", "Toys"}
    for k in pa:
        assert torch.equal(pa[k]["scores"], pb[k]["scores"]) and torch.equal(pa[k]["labels"], pb[k]["labels"])


def test_merge_train_two_ranks_keep_identical_alpha(tmp_path):
    """BASELINE configs[4]'s loop (collaborative-merging optimisation, pseudo-user batches data-parallel over the ranks) on a small
    model: both ranks must end every step with the SAME alpha (one all-reduce of d loss / d alpha per step), the alpha must have moved,
    and the test after training -- itself sharded over the two ranks -- must report what a single-process merge_test.py reports for
    that alpha file."""
    from tests.conftest import GOLDEN

    data, tok = str(GOLDEN / "mini_dataset"), str(GOLDEN / "mini_tokenizer")
    small = ["--model_type", "blair_base", "--model_kwargs", "init_seed", "7", "spec.hidden", "128", "spec.heads", "2", "spec.layers", "2",
             "spec.intermediate", "256", "--tokenizer_path", tok, "--max_seq_len", "96", "--max_attribute_len", "12", "--max_items", "20",
             "--batch_size", "8", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2", "--merge_type", "task_vector",
             "--learn_type", "layer_wise"]
    out = _run(["merge_train.py", *small, "--data_paths", data, data, "--item_embeddings_paths", "auto", "--sequence_embeddings_paths", "auto",
                "--train_data_split", "item", "--test_data_split", "test", "--loss_type", "SINGLE_PSEUDO_LABEL_KD", "--coefficient", "1000",
                "--learning_rate", "0.01", "--max_steps", "6", "--weights_dir", str(tmp_path / "w"), "--result_path", str(tmp_path / "res"),
                "--metrics_path", str(tmp_path / "train_test.csv")], 2)
    r0, r1 = (torch.load(f"{tmp_path}/res.rank{r}.pt", weights_only=False) for r in (0, 1))
    assert r0["world_size"] == r1["world_size"] == 2 and len(r0["history"]) == len(r1["history"]) == 6
    assert r0["weights"] == r1["weights"], "alpha diverged between the ranks"
    per = r0["weights"]["per_weights"]
    assert set(per) == {"0", "1", "others"} and any(abs(a - 0.2) > 1e-4 for v in per.values() for a in v)
    assert r0["history"] != r1["history"]  # the ranks really trained on different shards
    assert r0["test_metrics"] == r1["test_metrics"] and "Test metrics after training" in out
    # the alpha file the run logged, evaluated by a single process: same merged model, same metrics as the sharded test after training
    wfile = r0["weights_file"]
    assert wfile and Path(wfile).exists()
    last = len(Path(wfile).read_text().strip().splitlines()) - 1
    _run(["merge_test.py", *small, "--data_paths", data, data, "--weight_file", wfile, "--weight_file_line", str(last),
          "--metrics_path", str(tmp_path / "single.csv")], 1)
    import csv

    a = list(csv.DictReader(open(tmp_path / "train_test.csv")))
    b = list(csv.DictReader(open(tmp_path / "single.csv")))
    logged = eval(Path(wfile).read_text().strip().splitlines()[last], {"__builtins__": {}})["weights"]  # noqa: S307 - our own file
    if logged == r0["weights"]:  # the last logged step is the final alpha (log_every_steps divides max_steps): metrics must agree exactly
        assert a == b
    else:
        assert [r["dataset"] for r in a] == [r["dataset"] for r in b]


def test_rccl_initialises_and_accepts_the_products_collectives():
    """backend "nccl" (= RCCL) with one rank on the one GPU: process-group creation as parallel.init_from_env does it, and every collective
    call shape / dtype of the product path (tests/tools/nccl_single_rank.py)."""
    out = _run([str(ROOT / "tests" / "tools" / "nccl_single_rank.py"), str(_port())], 1, timeout=300)
    assert out.strip().endswith("OK")


def test_two_ranks_reproduce_single_process_at_true_dimensions(tmp_path):
    """The same comparison with BLaIR-base as it is (12 x 768, a 124.6 M-parameter arena cut into two slices, layer-wise coefficients, 3
    fine-tuned checkpoints; 1,500 items, 700 users): SHA-256 of the merged parameters, item / user embeddings, scores, top-k and metrics of
    the two-rank run equal the single process bit for bit."""
    worker = str(ROOT / "tests" / "tools" / "dist_worker.py")
    env = {"DIST_WORKER_TRUE_DIMS": "1"}
    _run([worker, str(tmp_path / "one.pt")], 1, extra_env=env)
    _run([worker, str(tmp_path / "two.pt")], 2, extra_env=env)
    one, two = torch.load(tmp_path / "one.pt"), torch.load(tmp_path / "two.pt")
    assert one.pop("world") == 1 and two.pop("world") == 2 and set(one) == set(two) and len(one) == 1
    for key in one:
        a, b = one[key], two[key]
        assert a["placement"] == "replicated" and b["placement"] == "sliced", key
        assert isinstance(a["merged"], str) and a["merged"] == b["merged"], "merged parameters differ"
        for name in ("item_embeddings", "user_embeddings", "topk", "labels", "scores", "mm_forward_cls"):
            assert a[name].shape == b[name].shape and torch.equal(a[name], b[name]), (key, name)
        assert a["metrics"] == b["metrics"], key


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run in the command (the driver's scaling invocation): the parent spawns
    the ranks as a child process before touching the GPU, relays rank 0's single JSON line and exits with the child's code.  Two ranks
    share the box's one GPU through gloo here; under "nccl" the same path reports rccl_ranks = N."""
    import json

    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0", MERGEREC_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for extra, placement in (([], "replicated"), (["--merge-placement", "sliced"], "sliced")):  # bench default at N > 1 / north_star's split
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
        assert out["dist_backend"] == "gloo" and out["rccl_ranks"] == 0 and out["merge_placement"] == placement
        assert out["value"] > 0 and out["config"]["parallelism"].startswith("dp2")
        pr = out["per_rank"]  # each rank's own step time and its time inside the data-path collectives, separately
        assert 0 < pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= out["ms_per_step"] * 1.05
        assert pr["collective_ms_per_step_max"] > 0 and pr["collective_bytes_received_per_rank_and_step"] > 0
        if placement == "sliced":  # the arena all-gather dominates the bytes: half of 499 MB per merge, two merges per epoch + one per step here
            assert pr["collective_bytes_received_per_rank_and_step"] > 200e6

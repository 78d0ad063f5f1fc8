"""merge_train.py end to end on the Pantry-sized synthetic domain (see realscale_cli_check.py), BLaIR-base at true dims, 2 domains, teachers
encoded first (``auto``): wall time per optimisation step INCLUDING the datamodule / collator / Python driver, next to the kernel-only step of
tests/tools/train_bench.py."""
import os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "tools"))
import torch
import realscale_cli_check as R
import merge_train
tmp = Path(tempfile.mkdtemp(prefix="realscale_"))
R.make_domain(tmp / "PantryLike", 4968, 14178)
steps = int(os.environ.get("RS_STEPS", 150))
argv = ["--model_type", "blair_base", "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
        "--data_paths", str(tmp / "PantryLike"), str(tmp / "PantryLike"), "--tokenizer_path", str(ROOT / "tests" / "golden" / "mini_tokenizer"),
        "--item_embeddings_paths", "auto", "--sequence_embeddings_paths", "auto", "--train_data_split", "item", "--test_data_split", "test",
        "--merge_type", "task_vector", "--learn_type", "task_wise", "--loss_type", "SINGLE_PSEUDO_LABEL_KD", "--coefficient", "1000",
        "--batch_size", "16", "--max_steps", str(steps), "--weights_dir", str(tmp / "w"), "--skip_test", "true"]
import mergerec_amd.utils as U
orig_fit = U.DistillTrainer.fit
def timed_fit(self, module, datamodule):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = orig_fit(self, module, datamodule)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"DistillTrainer.fit: {len(h)} steps in {dt:.2f} s -> {dt / len(h) * 1e3:.1f} ms / step end to end (datamodule setup, first-epoch catalog encode, collate, step)")
    return h
U.DistillTrainer.fit = timed_fit
merge_train.DistillTrainer = None  # (merge_train imports the class inside main)
t0 = time.perf_counter()
res = merge_train.main(argv)
print(f"merge_train.py total {time.perf_counter() - t0:.1f} s; alpha {res['weights']['per_weights']['all']}")

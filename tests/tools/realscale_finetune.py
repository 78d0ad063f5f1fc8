"""finetune_train.py end to end at real scale: one epoch of scripts/1_finetune/blair_base.sh's recipe (batch 64, in-batch negatives, accumulation
4, bf16-mixed -> bf16x3 training graph) on the Pantry-sized synthetic domain (see realscale_cli_check.py), BLaIR-base at true dims, then
validation, best checkpoint, test.  Reports wall time per phase as the CLI prints them."""
import os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "tools"))
import torch
import realscale_cli_check as R
import finetune_train
tmp = Path(tempfile.mkdtemp(prefix="realscale_"))
R.make_domain(tmp / "PantryLike", 4968, 14178)
argv = ["--model_type", os.environ.get("RS_MODEL", "blair_base"), "--model_kwargs", "init_seed", "7", "--tokenizer_path", str(ROOT / "tests" / "golden" / "mini_tokenizer"),
        "--data_path", str(tmp / "PantryLike"), "--batch_size", "64", "--negative_sample.in_batch", "--temperature", "0.05", "--warmup_steps", "10",
        "--learning_rate", "5e-5", "--max_epochs", os.environ.get("RS_EPOCHS", "1"), "--log_every_n_steps", "10", "--default_root_dir", str(tmp / "run")]
t0 = time.perf_counter()
trainer, metrics = finetune_train.main(argv)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(trainer.history)
print(f"finetune_train.py: {n} micro-steps ({trainer.global_step} optimizer steps) + validation + test in {dt:.1f} s total; first / last train loss "
      f"{sum(trainer.history[:5]) / 5:.4f} / {sum(trainer.history[-5:]) / 5:.4f}; val NDCG@10 {trainer.best_score:.4f}; test NDCG@10 {metrics[0]['test/NDCG@10']:.4f}")

"""cProfile of one merge_test.py run on the Pantry-sized synthetic domain (see realscale_cli_check.py): where the HOST time of the CLI goes."""
import cProfile, os, pstats, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "tools"))
import torch
import realscale_cli_check as R
import merge_test
tmp = Path(tempfile.mkdtemp(prefix="realscale_"))
R.make_domain(tmp / "PantryLike", 4968, 14178)
MODEL = os.environ.get("RS_MODEL", "blair_base")
argv = ["--model_type", MODEL, "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
        "--merge_type", "task_vector", "--learn_type", "task_wise", "--weight_file", "average", "--data_paths", str(tmp / "PantryLike"),
        "--tokenizer_path", str(ROOT / "tests" / "golden" / "mini_tokenizer"), "--batch_size", "32", "--test_data_split", "test"]
os.environ["MERGEREC_GEMM_MODE"] = "bf16x3"
merge_test.main(list(argv))  # warm (library load, first torch import costs)
pr = cProfile.Profile()
pr.enable()
merge_test.main(list(argv))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)

"""Host-side logic of the fine-tuning path (no GPU): config modes, schedule, CLI parsing."""
import pytest

from mergerec_amd.configs import NegativeSampleConfig, NegativeSampleOption
from mergerec_amd.optim import linear_schedule_with_warmup
from oracle import ref_cpu as O


def test_negative_sample_modes():
    """configs/finetune.py:16-24"""
    assert NegativeSampleConfig().mode == NegativeSampleOption.FULL
    assert NegativeSampleConfig(k=4).mode == NegativeSampleOption.SAMPLE
    assert NegativeSampleConfig(in_batch=True).mode == NegativeSampleOption.IN_BATCH
    assert NegativeSampleConfig(k=4, in_batch=True).mode == NegativeSampleOption.IN_BATCH_SAMPLE


@pytest.mark.parametrize("warm,total", [(0, 10), (3, 10), (2.5, 10), (100, 40)])
def test_schedule_matches_oracle(warm, total):
    for s in range(total + 3):
        assert linear_schedule_with_warmup(s, warm, total) == O.linear_warmup_multiplier(s, warm, total)


def test_cli_flags():
    import finetune_train as ft

    cfg = ft.parse(["--model_type", "blair_base", "--batch_size", "64", "--negative_sample.in_batch", "--temperature", "0.05",
                    "--warmup_steps", "100", "--data_path", "datasets/Arts", "--learning_rate", "5e-5", "--log_every_n_steps", "1"])
    assert cfg.model_type == "BLAIR_BASE" and cfg.negative_in_batch and cfg.negative_k is None
    assert cfg.warmup_steps == 100 and isinstance(cfg.warmup_steps, int)
    assert cfg.gradient_accumulation_steps == 4 and cfg.max_epochs == 100 and cfg.patience == 5 and cfg.valid_metric == "val/NDCG@10"
    cfg = ft.parse(["--model_type", "recformer_base", "--model_kwargs", "ckpt_path", "x.pt", "--warmup_steps", "0.1", "--data_path", "d"])
    assert cfg.warmup_steps == 0.1 and cfg.model_kwargs == {"ckpt_path": "x.pt"}

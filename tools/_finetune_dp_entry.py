"""Entry point of tools/finetune_dp_rehearsal.sh: finetune_train.main on a tiny BLaIR spec, printing each rank's weight checksum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import finetune_train
from mergerec_amd.engine import EncoderSpec
from mergerec_amd.module import models

models.BLaIRBase.SPEC = staticmethod(lambda: EncoderSpec(hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=514))
rank = int(os.environ.get("RANK", "0"))
trainer, metrics = finetune_train.main(sys.argv[1:])
flat = trainer.optimizer.param
print(f"RANK {rank} steps {trainer.global_step} weights sum {float(flat.double().sum()):.12f} abs {float(flat.double().abs().sum()):.12f} "
      f"test/NDCG@10 {metrics[0]['test/NDCG@10']:.6f}", flush=True)

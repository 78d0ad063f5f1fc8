"""Entry point of tools/train_dp_rehearsal.sh: merge_train.main on a tiny BLaIR spec, printing each rank's final alpha."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import merge_train
from mergerec_amd.engine import EncoderSpec
from mergerec_amd.module import models

models.BLaIRBase.SPEC = staticmethod(lambda: EncoderSpec(hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=514))
argv = [a for a in sys.argv[1:]]
# drop the placeholder pair "spec_overrides TINY" (the spec is patched above)
i = argv.index("spec_overrides")
del argv[i:i + 2]
res = merge_train.main(argv)
print(f"RANK {res['rank']}/{res['world_size']} alpha {res['weights']['per_weights']['all']} steps {len(res['history'])}", flush=True)

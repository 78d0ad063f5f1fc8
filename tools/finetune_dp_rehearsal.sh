#!/bin/bash
# Two data-parallel ranks of finetune_train.py sharing the one GPU of a gpurun box (gloo): every rank trains on its share of each
# epoch's batches, the gradient arena is averaged in one all-reduce per optimizer step, and both ranks must end with bit-identical
# weights (printed as a checksum).  Usage: bash tools/finetune_dp_rehearsal.sh
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--model_type blair_base --model_kwargs init_seed 7 --tokenizer_path tests/golden/mini_tokenizer --data_path tests/golden/mini_dataset \
  --batch_size 4 --negative_sample.in_batch --temperature 0.05 --warmup_steps 2 --learning_rate 1e-3 --gradient_accumulation_steps 2 \
  --gradient_clip_val 1.0 --max_epochs 2 --max_seq_len 96 --max_attribute_len 12 --max_items 20 --precision bf16-mixed --log_every_n_steps 1 \
  --default_root_dir gpurun_out/ft_dp"
MERGEREC_DIST_BACKEND=gloo MERGEREC_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29519 tools/_finetune_dp_entry.py $ARGS

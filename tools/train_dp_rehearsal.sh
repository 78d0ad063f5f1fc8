#!/bin/bash
# Two data-parallel ranks of merge_train.py sharing the one GPU of a gpurun box (gloo): checks that both ranks end with the same alpha
# and that it differs from a single-rank run only through the batch sharding.  Usage: bash tools/train_dp_rehearsal.sh
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
COMMON="--model_type blair_base --model_kwargs init_seed 7 spec_overrides TINY --finetune_checkpoint_paths synthetic:1 synthetic:2 \
  --data_paths tests/golden/mini_dataset tests/golden/mini_dataset --tokenizer_path tests/golden/mini_tokenizer \
  --item_embeddings_paths auto --sequence_embeddings_paths auto --train_data_split item --test_data_split test \
  --merge_type task_vector --learn_type task_wise --loss_type SINGLE_PSEUDO_LABEL_KD --coefficient 1000 --learning_rate 0.01 \
  --max_steps 6 --batch_size 8 --max_seq_len 96 --max_attribute_len 12 --max_items 20 --skip_test true"
MERGEREC_TINY=1 MERGEREC_DIST_BACKEND=gloo MERGEREC_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 tools/_train_dp_entry.py $COMMON --weights_dir gpurun_out/dp_weights

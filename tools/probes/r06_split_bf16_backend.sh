#!/bin/bash
# Round 6 side measurement (never part of `value`): the step with EVERY large GEMM on the own family and its experimental
# exact three-way bf16 split inner product (six bf16 MFMA products accumulated in fp32, error vs float64 below the fp32 MFMA
# chain's) -- what the power-bound fp32 matrix pipe leaves on the table.
cd "$(dirname "$0")/../.."
for cfg in "library 0" "own 0" "library 1" "own 1"; do
  set -- $cfg
  DATR_GEMM_BACKEND=$1 DATR_GEMM_SPLIT_BF16=$2 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('backend=$1 split_bf16=$2', d['ms_per_step'], d['value'], d['config']['gemm_backend'][:60])"
done

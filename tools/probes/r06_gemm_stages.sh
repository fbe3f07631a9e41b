#!/bin/bash
# Round 6: the own fp32-MFMA GEMM family with a deeper LDS ring (stages 2 / 3 / 4) on the 88 892-row shapes.
cd "$(dirname "$0")/../.."
run() { python tools/probes/gemm_one.py --iters ${ITERS:-10} "$@" 2>&1 | tail -1; }
for spec in "ffn1 nt none" "ffn1 nn none" "ffn1 nn gate_cs" "ffn2 nt none" "ffn2 nn none" "lin_256>256 nt none"; do
  set -- $spec
  layer=${1//_/ }
  [ "$2" = nt ] && run --layer "$layer" --form lib_nt
  [ "$2" = nn ] && [ "$3" = none ] && run --layer "$layer" --form lib_nn
  for plan in 1,2,16 1,2,32 2,2,16 2,2,32 2,1,32; do
    for ns in 2 3 4; do
      run --layer "$layer" --form $2 --epi $3 --plan $plan,0,$ns
    done
  done
done

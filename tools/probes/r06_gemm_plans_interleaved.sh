#!/bin/bash
# own GEMM family plans (DATR_GEMM_PLAN=tm,tn,bk,ksplit) against the tuned library, interleaved per launch
cd "$(dirname "$0")/../.."
for plan in ${PLANS:-auto 2,2,16,0 4,2,16,0 4,2,32,0}; do
  echo "== plan $plan"
  if [ $plan = auto ]; then unset DATR_GEMM_PLAN; else export DATR_GEMM_PLAN=$plan; fi
  python tools/probes/r06_gemm_interleaved.py --tuned --rounds 12 2>&1 | grep -E "ffn|lin 256>256" | grep -v wgrad
done

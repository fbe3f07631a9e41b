for plan in 128,1 128,2 128,4 64,1 64,2; do
  echo "== plan $plan"
  DATR_TAP_PLAN=$plan timeout 120 python tools/bench_conv_s2.py 2>&1 | grep "conv2\|input_proj" | cut -c1-110
done

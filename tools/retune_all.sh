# Re-create the shipped library selections on a MI355X after the set of GEMM / convolution shapes changed:
#   datr_amd/tuning/gemm_mi355x.csv (PyTorch TunableOp) and datr_amd/tuning/miopen/ (MIOpen user find-db).
# Results land in gpurun_out/tuning/ (copy them over the shipped files).
mkdir -p gpurun_out/tuning/miopen
python -m datr_amd.tuning.retune > gpurun_out/tuning/retune.log 2>&1; tail -1 gpurun_out/tuning/retune.log
cp datr_amd/tuning/gemm_mi355x.csv gpurun_out/tuning/
export MIOPEN_USER_DB_PATH=/tmp/miopen_db; rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
cp datr_amd/tuning/miopen/* $MIOPEN_USER_DB_PATH/
python tools/probes/bench_cudnn_benchmark.py > gpurun_out/tuning/find.log 2>&1; tail -1 gpurun_out/tuning/find.log | cut -c1-200
cp $MIOPEN_USER_DB_PATH/* gpurun_out/tuning/miopen/; ls -la gpurun_out/tuning/miopen

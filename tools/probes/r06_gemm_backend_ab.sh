cd /root/repo
python tools/probes/r06_gemm_interleaved.py --tuned 2>&1 | tail -24
for b in library own; do
DATR_GEMM_BACKEND=$b python bench.py --steps 12 --warmup 4 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$b', d['ms_per_step'], d['ms_per_step_percentiles'], d['config']['gemm_backend'])"
done
DATR_GEMM_BACKEND=library python bench.py --steps 12 --warmup 4 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('library again', d['ms_per_step'], d['ms_per_step_percentiles'])"

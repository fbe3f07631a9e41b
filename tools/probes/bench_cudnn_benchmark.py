import sys, runpy, torch
torch.backends.cudnn.benchmark = True
sys.argv = ["bench.py", "--steps", "20", "--warmup", "8", "--no-cpu-baseline"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")

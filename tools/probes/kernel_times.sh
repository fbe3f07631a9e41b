# per-kernel average / min durations of a command under rocprofv3:  bash tools/probes/kernel_times.sh <n rows> <cmd...>
rows=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- "$@" > /tmp/kt.log 2>&1
python - "$rows" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)
if not f:
    print(open("/tmp/kt.log").read()[-2000:]); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:int(sys.argv[1])]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} min_us {float(r["MinNs"])/1e3:8.1f}')
PY

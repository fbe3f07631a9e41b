import json, sys
for f in sys.argv[1:]:
    print(f)
    for l in open(f):
        d = json.loads(l)
        plans = {k: v for k, v in d.items() if "," in k}
        print(f"{d['layer']:22s} {d['form']:12s} lib {d['lib_us']:7.1f} ({d['lib_tf']:5.1f}TF) best {d['best']:7s} {d['best_us']:7.1f} ({d['best_tf']:5.1f}TF)  " + " ".join(f"{k}={v:.0f}" for k, v in plans.items()))

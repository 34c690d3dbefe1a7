"""Interleaved bench.py runs over several values of one environment variable (same box, alternating
processes).  usage: ab_values.py VAR v1,v2,... [rounds] [bench args...]"""
import json, os, subprocess, sys
var, vals = sys.argv[1], sys.argv[2].split(",")
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
extra = sys.argv[4:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        env = dict(os.environ)
        env[var] = v
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res[v].append((d["roofline"]["kernel_ms"], d["ms_per_step"]))
        print(var, v, res[v][-1], flush=True)
for v in vals:
    print("%s=%s kernel ms min %.4f  step ms min %.4f" % (var, v, min(x[0] for x in res[v]), min(x[1] for x in res[v])))

"""A/B of an environment switch read once per process: alternate child processes of bench.py (same box),
print kernel ms / step ms for each.  usage: ab_env.py VAR=VALUE [rounds] [bench args...]"""
import json, os, subprocess, sys
var, val = sys.argv[1].split("=", 1)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {"off": [], "on": []}
for r in range(rounds):
    for mode in ("off", "on"):
        env = dict(os.environ)
        env.pop(var, None)
        if mode == "on":
            env[var] = val
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        res[mode].append((d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"]["achieved"]))
        print(mode, res[mode][-1], flush=True)
for mode in res:
    print(mode, "kernel ms min %.4f  step ms min %.4f" % (min(x[0] for x in res[mode]), min(x[1] for x in res[mode])))

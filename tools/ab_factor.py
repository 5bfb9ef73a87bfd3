"""Same-session A/B of library builds on the headline step (n = 8192): assemble + factor time and KKT solves/s, each
variant in its own process (CIPKKT_LIB), alternating, `rounds` times.
usage: python tools/ab_factor.py name=path.so|default|env:VAR=VALUE [...] [--rounds 3] [--n 8192]
(env:VAR=VALUE runs the in-tree build with that environment variable set)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if "=" in a]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
n = sys.argv[sys.argv.index("--n") + 1] if "--n" in sys.argv else "8192"
res = {}
for r in range(rounds):
    for a in args:
        name, path = a.split("=", 1)
        env = dict(os.environ)
        if path.startswith("env:"):
            k, v = path[4:].split("=", 1)
            env[k] = v
        elif path != "default":
            env["CIPKKT_LIB"] = os.path.abspath(path)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-c5", "--no-secondary", "--no-plugin-boundary", "--n", n, "--steps", "20", "--warmup", "3"],
                             capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-400:]); continue
        d = json.loads(line[-1])
        res.setdefault(name, []).append((d["value"], d["breakdown_ms"]["ldlt_factor"], d["roofline"]["achieved"], d["converge"]["iters"]))
for name, v in res.items():
    print("%-10s KKT solves/s %s | factor ms %s | trailing TFLOP/s %s | iters %s" % (
        name, " ".join("%.1f" % x[0] for x in v), " ".join("%.3f" % x[1] for x in v), " ".join("%.1f" % x[2] for x in v), v[0][3]))

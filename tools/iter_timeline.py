"""Kernel timeline of the interior-point iterations of the LAST solve in a rocprofv3 kernel trace (csv): launches, kernel time and
idle time per iteration, by kernel, by queue and by (kernel -> next kernel) gap on the main queue.
usage: python tools/iter_timeline.py trace.csv [marker kernel prefix, default k_nt_scaling] [max gap between iterations, ms]"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_nt_scaling"
maxgap = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 50e6
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
idx = [i for i, r in enumerate(rows) if nm(r).startswith(marker)]
run = [idx[-1]]
for i in reversed(idx[:-1]):
    if int(rows[run[0]]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < maxgap:
        run.insert(0, i)
    else:
        break
a, b = run[0], run[-1]
its = len(run) - 1
seg = rows[a:b]
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print("%d iterations: %.3f ms each, %.1f launches each" % (its, span / its / 1e6, len(seg) / its))
queues = collections.Counter(r.get("Queue_Id", "?") for r in seg)
mainq = queues.most_common(1)[0][0]
print("queues:", dict(queues))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[nm(r)][0] += 1
    tot[nm(r)][1] += d
# union of busy intervals over all queues
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy = 0
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("per iteration: sum of kernel durations %.3f ms, time with any kernel running %.3f ms, idle %.3f ms" %
      (sum(t for _, t in tot.values()) / its / 1e6, busy / its / 1e6, (span - busy) / its / 1e6))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %-44s %6.1f x %8.1f us = %8.3f ms" % (n[:44], c / its, t / c / 1e3, t / its / 1e6))
mq = [r for r in seg if r.get("Queue_Id", "?") == mainq]
gg = collections.defaultdict(lambda: [0, 0.0])
for x, y in zip(mq, mq[1:]):
    g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
    key = (nm(x)[:28], nm(y)[:28])
    gg[key][0] += 1
    gg[key][1] += g
print("idle on the main queue by (kernel -> next kernel), per iteration:")
for (x, y), (c, t) in sorted(gg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-28s -> %-28s %5.1f x %6.2f us = %7.3f ms" % (x, y, c / its, t / c / 1e3, t / its / 1e6))

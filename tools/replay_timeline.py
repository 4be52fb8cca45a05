"""Read a rocprofv3 kernel trace (csv) of a hipGraph-replayed training leg and print where a replayed step's wall time goes:
kernel time on the critical stream, gaps between consecutive kernels, overlap.  usage: replay_timeline.py <kernel_trace.csv> [launches per step]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 108
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# the replayed steps are the last ones of the trace: take the last 40 steps' worth of launches
tail = ev[-40 * per_step:]
t0, t1 = tail[0][0], max(e[1] for e in tail)
busy = 0
cur_s, cur_e = tail[0][0], tail[0][1]
gaps = []
for s, e, n in tail[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
ksum = sum(e - s for s, e, _ in tail)
print(f"launches {len(tail)}, wall {wall / 1e6:.3f} ms = {wall / 40 / 1e3:.1f} us per step; union of kernel intervals "
      f"{busy / 1e6:.3f} ms ({busy / wall:.1%}); sum of kernel durations {ksum / 1e6:.3f} ms; idle gaps {len(gaps)}, "
      f"{sum(g for g, _ in gaps) / 1e6:.3f} ms, mean {sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3:.2f} us")
by = defaultdict(lambda: [0, 0, 0])
for s, e, n in tail:
    b = by[n.split("(")[0][:60]]
    b[0] += 1
    b[1] += e - s
for g, n in gaps:
    by[n.split("(")[0][:60]][2] += g
print(f"{'kernel':60s} {'calls/step':>10s} {'us/step':>9s} {'us/call':>8s} {'gap before, us/step':>20s}")
for n, (c, d, g) in sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:28]:
    print(f"{n:60s} {c / 40:10.1f} {d / 40 / 1e3:9.1f} {d / c / 1e3:8.2f} {g / 40 / 1e3:20.1f}")

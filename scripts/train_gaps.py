"""Idle time of the GPU inside the last `window_ms` of a rocprofv3 kernel trace of `bench.py --mode train`:
largest single gaps (previous kernel -> next kernel) and the gap total by kernel pair."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - win]
busy = sum(e - s for s, e, _ in rows)
gaps, pair = [], defaultdict(lambda: [0, 0])
hist = defaultdict(lambda: [0, 0])
last_end = rows[0][1]
for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
    last_end = max(last_end, e0)
    g = s1 - last_end
    if g > 0:
        gaps.append((g, n0[:60], n1[:60]))
        p = pair[(n0[:60], n1[:60])]
        p[0] += 1
        p[1] += g
        b = 1 if g < 2000 else 5 if g < 5000 else 10 if g < 10000 else 20 if g < 20000 else 100 if g < 100000 else 1000
        hist[b][0] += 1
        hist[b][1] += g
span = rows[-1][1] - rows[0][0]
print(f"# window {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms  idle {sum(g for g, _, _ in gaps) / 1e6:.1f} ms  launches {len(rows)}")
for b in sorted(hist):
    print(f"  gaps < {b:5d} us: {hist[b][0]:6d}  total {hist[b][1] / 1e6:7.2f} ms")
print("# largest single gaps")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"{g / 1e3:9.1f} us  {a} -> {b}")
print("# gap total by pair")
for (a, b), (c, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t / 1e6:8.2f} ms {c:5d}x  {a} -> {b}")

"""Idle time between kernels inside steady-state frames of a rocprofv3 kernel trace of bench.py:
frames are cut at mark_kernel launches; prints busy / span per frame and the largest gaps by (previous, next) kernel."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if r[2].startswith("void mark_kernel")]
frames = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    if any("gather_detections" in k[2] for k in ks):        # a full detector frame
        frames.append(ks + [rows[b]])
frames = frames[len(frames) // 3: 2 * len(frames) // 3]     # the middle third: the timed loop
gaps = defaultdict(lambda: [0, 0])
busy = span = 0
for ks in frames:
    span += ks[-1][0] - ks[0][0]
    busy += sum(e - s for s, e, _ in ks[:-1])
    for (s0, e0, n0), (s1, e1, n1) in zip(ks[:-1], ks[1:]):
        g = gaps[(n0[:40], n1[:40])]
        g[0] += 1
        g[1] += max(s1 - e0, 0)
n = len(frames)
print(f"frames {n}  span {span / n / 1e3:.1f} us  busy {busy / n / 1e3:.1f} us  idle {(span - busy) / n / 1e3:.1f} us  kernels {sum(len(k) - 1 for k in frames) / n:.1f}")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t / n / 1e3:7.2f} us/frame  {c / n:5.1f}x  {a} -> {b}")

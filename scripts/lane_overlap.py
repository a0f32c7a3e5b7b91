"""How well two frames in flight share the chip, from a rocprofv3 kernel trace of
`bench.py --no-cpu-baseline --steps 200` (default two lanes).  Over the longest stretch of back-to-back frames:
wall time per frame, time with 0 / 1 / >=2 kernels resident, and per kernel name the summed duration per frame
(compare with the one-lane profile: a kernel that stretches when overlapped was sharing CUs or bandwidth)."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if r[2].startswith("void mark_kernel")]
# the timed loop: the longest run of frames whose starts are < 3 ms apart
best, cur = (0, 0), 0
for i in range(1, len(marks)):
    if rows[marks[i]][0] - rows[marks[i - 1]][0] > 3_000_000:
        cur = i
    if i - cur > best[1] - best[0]:
        best = (cur, i)
a, b = best
a, b = a + (b - a) // 8, b - (b - a) // 8          # steady state
t0, t1 = rows[marks[a]][0], rows[marks[b]][0]
nfr = b - a
ev, per = [], defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s >= t0 and s < t1:
        ev += [(s, 1), (e, -1)]
        per[n[:70]][0] += 1
        per[n[:70]][1] += e - s
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[min(depth, 2)] += t - last
    last, depth = t, depth + d
wall = t1 - t0
print(f"frames {nfr}  wall {wall / nfr / 1e3:.1f} us/frame  resident kernels: 0 -> {hist[0] / nfr / 1e3:.1f} us  1 -> {hist[1] / nfr / 1e3:.1f} us  >=2 -> {hist[2] / nfr / 1e3:.1f} us")
tot = sum(v[1] for v in per.values())
print(f"sum of kernel durations {tot / nfr / 1e3:.1f} us/frame")
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{t / nfr / 1e3:8.1f} us/frame {c / nfr:5.1f}x  avg {t / c / 1e3:7.2f}  {n}")

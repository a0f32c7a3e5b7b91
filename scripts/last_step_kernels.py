"""Per-kernel GPU time of the LAST `window_ms` of a rocprofv3 kernel trace (steady state of a training run:
the first steps hold MIOpen's solver search).  usage: last_step_kernels.py <rocprof dir> <window_ms>"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6
rows = list(csv.DictReader(open(f)))
t_end = max(int(r["End_Timestamp"]) for r in rows)
agg, cnt = defaultdict(int), defaultdict(int)
for r in rows:
    if int(r["Start_Timestamp"]) >= t_end - win:
        agg[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[r["Kernel_Name"]] += 1
tot = sum(agg.values())
print(f"# last {sys.argv[2]} ms: busy {tot / 1e6:.1f} ms in {sum(cnt.values())} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v / 1e6:8.2f} ms {cnt[k]:6d} calls {v / cnt[k] / 1e3:9.1f} us  {k[:120]}")

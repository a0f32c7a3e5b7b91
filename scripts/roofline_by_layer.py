"""Per-layer durations of the roofline kernel template from a rocprofv3 --kernel-trace CSV of bench.py
(usage: roofline_by_layer.py <trace dir> [warm-up frames to skip]) -- the launches of one frame in order."""
import collections, csv, glob, sys
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void mark_kernel")][skip:]
per = collections.defaultdict(list)
for fi, s in enumerate(starts):
    e = starts[fi + 1] if fi + 1 < len(starts) else len(rows)
    k = 0
    for r in rows[s:e]:
        n = r["Kernel_Name"]
        if ("gg_win_kernelILi128ELi256ELi2ELi4ELi8E" in n and "ELi1EEv" not in n) or \
           ("gg_win_kernel<128, 256, 2, 4, 8" in n and not n.split("(")[0].rstrip(">").endswith(", 1")):
            per[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            k += 1
for k in sorted(per):
    print(f"{k:4d} {len(per[k]):6d} launches {sum(per[k]) / len(per[k]):8.2f} us")

"""Top kernels of a rocprofv3 --kernel-trace CSV (whole run)."""
import collections, csv, glob, sys
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(trace)):
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"# total kernel time {tot / 1e6:.1f} ms over {sum(a[0] for a in agg.values())} launches")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{n[:120]:120s} {a[0]:6d} {a[1] / 1e6:9.2f} ms {100 * a[1] / tot:5.1f}%")

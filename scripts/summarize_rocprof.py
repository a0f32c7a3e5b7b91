"""Condenses a rocprofv3 --kernel-trace --stats CSV pair into the per-frame table kept under
profiles/: steady-state frames only (everything after the warm-up's last reader launch)."""
import collections
import csv
import glob
import sys

d, skip = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void mark_kernel")]
starts = starts[skip:]
rows = rows[starts[0]:]
frames = len(starts)
agg = collections.OrderedDict()
by_grid = len(sys.argv) > 3 and sys.argv[3] == "grid"    # split every kernel by its grid size (one line per layer shape)
for r in rows:
    name = r["Kernel_Name"]
    if by_grid:
        name = f'{name[:60]} grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))}x{r.get("Grid_Size_Y", "")}x{r.get("Grid_Size_Z", "")}'
    a = agg.setdefault(name, [0, 0, r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(a[1] for a in agg.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"# frames {frames}  kernels/frame {len(rows) / frames:.1f}  busy ms/frame {busy / frames / 1e6:.3f}  span ms/frame {span / frames / 1e6:.3f}")
print(f"{'kernel':110s} {'calls/frame':>11s} {'us/frame':>10s} {'avg_us':>9s} {'%busy':>6s} vgpr lds scratch")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:110]:110s} {a[0] / frames:11.1f} {a[1] / frames / 1e3:10.1f} {a[1] / a[0] / 1e3:9.2f} {100 * a[1] / busy:6.2f} {a[2]} {a[3]} {a[4]}")

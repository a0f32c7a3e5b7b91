#!/bin/bash
# HBM traffic of a whole frame: FETCH_SIZE / WRITE_SIZE (KB) per kernel, separate passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_frame
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1 --no-graph > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob("gpurun_out/pmc_frame/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            calls[r["Kernel_Name"][:70]] += 1
frames = 7.0 + 1.0   # warm-up 2 + steps 5 + the probe/first frame
tot = 0.0
rows = []
for k, v in agg.items():
    mb = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) / 1024.0
    rows.append((mb, k, calls[k]))
    tot += mb
print(f"# HBM MB over the whole run (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction), all launches; ~{frames:.0f} frames")
for mb, k, c in sorted(rows, reverse=True)[:22]:
    print(f"{k:70s} launches {c:5d}  {mb:10.1f} MB  {mb / max(c, 1):8.2f} MB/launch")
print(f"# total {tot:.0f} MB")
PY

#!/bin/bash
# PMC passes over the weight-gradient kernel on one layer shape (separate runs per counter group, kernel-trace only).
# usage: pmc_wgrad.sh   (runs scripts/wgrad_one.py; averages over launches of wgrad_kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_wgrad
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/wgrad_one.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_wgrad/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_kernel" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:32s} launches {n:4d}  avg/launch {v / n:16.1f}")
PY
find $OUT -name "*.csv" -size +1M -delete

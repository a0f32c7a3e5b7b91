#!/bin/bash
# PMC passes over one gather-GEMM shape (separate runs per counter group, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gg
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/gg_one.py "$@" > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_gg/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not (("gather_gemm" in r["Kernel_Name"] or "gg_win" in r["Kernel_Name"]) and
                ("Li128ELi256E" in r["Kernel_Name"] or "128, 256" in r["Kernel_Name"])):
            continue
        a = agg[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:32s} launches {n:4d}  avg/launch {v / n:16.1f}")
PY

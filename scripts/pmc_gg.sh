#!/bin/bash
# PMC passes over one gather-GEMM shape (separate runs per counter group, kernel-trace only).
# PMC_MATCH = substring of the kernel name to average over (default: the 128 x 256 tiles), PMC_TAG = output suffix;
# PN_GG_WIN_CFG in the environment selects another tile for the same shape.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gg${PMC_TAG}
export PMC_OUT=$OUT
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/gg_one.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections, os
match = os.environ.get("PMC_MATCH", "")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.environ["PMC_OUT"] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not ("gather_gemm" in name or "gg_win" in name):
            continue
        if match:
            if match not in name:
                continue
        elif not ("Li128ELi256E" in name or "128, 256" in name):
            continue
        a = agg[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:32s} launches {n:4d}  avg/launch {v / n:16.1f}")
PY

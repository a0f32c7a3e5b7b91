#!/bin/bash
# Per-kernel time of the steady-state inference frame: rocprofv3 kernel trace of bench.py, condensed by
# scripts/summarize_rocprof.py into gpurun_out/<tag>_frame_kernels.txt (copy into profiles/ to keep).
TAG=${1:-prof}
shift        # further arguments go to bench.py (e.g. --lanes 1 --no-graph: one frame at a time, undisturbed kernel times)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 scripts/summarize_rocprof.py $OUT 6 > gpurun_out/${TAG}_frame_kernels.txt
python3 scripts/summarize_rocprof.py $OUT 6 grid > gpurun_out/${TAG}_frame_kernels_by_grid.txt
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
cat $OUT/bench.json
find $OUT -name "*.csv" -size +1M -delete

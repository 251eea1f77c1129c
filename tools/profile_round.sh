#!/bin/bash
# Collect the per-round profiles on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <outdir under gpurun_out>
# kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and SQ_* counters in SEPARATE passes (MI355X_MICROARCH.md).
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/${1:-prof}
mkdir -p $D
python3 bench.py --steps 10 --warmup 2 > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $D/bench_kt.log 2> $D/kt.err; echo "kt rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $D/bench_fetch.log 2> $D/fetch.err; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $D/bench_write.log 2> $D/write.err; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $D/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > $D/bench_sq.log 2> $D/sq.err; echo "sq rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/grbm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > $D/bench_grbm.log 2> $D/grbm.err; echo "grbm rc=$?"
find $D -name "*_kernel_trace.csv" -size +20M -delete
du -sh $D

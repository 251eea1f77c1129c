#!/bin/bash
# Collect the per-round profiles on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <outdir under gpurun_out> [workloads, default "2b 2a 4"]
# per workload: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and SQ_* counters in SEPARATE passes (MI355X_MICROARCH.md).
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/${1:-prof}
WL=${2:-"2b 2a 4"}
mkdir -p $D
python3 -c "import bench; print(bench.kernel_source_sha16())" > $D/kernel_source_sha16.txt
# what a VALU instruction costs (refreshed every round): whole-launch time of ncu*4*W one-wave blocks of 96 000 independent instructions each
(./tools/ubench_issue | grep "^#" | awk '{ printf "%s  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", $0, $9 * 1e-3 * 2.4e9 / ($11 * $5) }') > $D/valu_issue_ubench.txt 2>&1 || true
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"
for w in $WL; do
  B="python3 bench.py --workload $w --no-subconfigs --no-cpu-baseline --no-single-stream"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/$w/kt -- $B --steps 5 --warmup 2 > $D/$w.kt.log 2> $D/$w.kt.err; echo "$w kt rc=$?"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/$w/fetch -- $B --steps 2 --warmup 1 > $D/$w.fetch.log 2> $D/$w.fetch.err; echo "$w fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/$w/write -- $B --steps 2 --warmup 1 > $D/$w.write.log 2> $D/$w.write.err; echo "$w write rc=$?"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $D/$w/sq -- $B --steps 2 --warmup 1 --streams 1 > $D/$w.sq.log 2> $D/$w.sq.err; echo "$w sq rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/$w/kt1 -- $B --steps 3 --warmup 1 --streams 1 > $D/$w.kt1.log 2> $D/$w.kt1.err; echo "$w kt1 rc=$?"
done
find $D -name "*_kernel_trace.csv" -size +20M -delete
du -sh $D

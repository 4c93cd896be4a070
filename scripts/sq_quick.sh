#!/bin/bash
# instruction counters of one workload: scripts/sq_quick.sh TAG SIZE PIPE FRAMES MODEL GRID   (TR_LIBRARY honoured)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/sqq_$tag
rm -rf "$out"; mkdir -p "$out"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d "$out/$n" -o out --output-format csv -- python3 scripts/frame_loop.py "$@" > "$out/$n.log" 2>&1 || echo "refused: $set"
done
python3 scripts/pmc_summary.py $out/*/ 2>/dev/null | grep k_tile | sed "s/^/$tag /" | sed 's/gpurun_out[^ ]* //'

#!/bin/bash
# Runs on the GPU box (through gpurun): collects the round's judged evidence into gpurun_out/prof/.
#   bench line (with cpu_baseline), rocprofv3 kernel stats of the same bench command,
#   HBM traffic counters (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as
#   MI355X_MICROARCH.md prescribes) for the headline workload and the five BASELINE configs,
#   and SQ instruction / cycle counters of a plain frame loop of the headline workload.
# Copy what should be judged into profiles/ afterwards (scripts/summarise_profiles.py <tag>), which
# also stamps profiles/pmc_traffic.json with the fingerprint of the profiled sources.
set -o pipefail
out=gpurun_out/prof
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"

echo "== bench"
python3 bench.py > "$out/bench_4096_phong.log" 2> "$out/bench_4096_phong.err" || exit 1
tail -n 1 "$out/bench_4096_phong.log"

echo "== kernel trace"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o out --output-format csv -- python3 bench.py --no-cpu --no-extras \
    > "$out/bench_under_rocprof.log" 2> "$out/trace.err" || exit 1
tail -n 1 "$out/bench_under_rocprof.log"

# (frames: multiples of the workload's frames per launch -- 4, 32, 16, 4, 4, 4 -- so that every launch is a full group)
# tag            size pipeline frames model        grid
workloads=(
 "headline       4096 phong    32     diablo       1"
 "cfg0           800  default  128    african_head 1"
 "cfg1           2048 phong    32     diablo       1"
 "cfg2           4096 darboux  32     diablo       1"
 "cfg3           4096 shadow   32     diablo       1"
 "cfg4           8192 specular 8      diablo       8"
)
for w in "${workloads[@]}"; do
  set -- $w
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $1"
    rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_${c}_$1" -o out --output-format csv -- python3 scripts/frame_loop.py $2 $3 $4 $5 $6 \
        > "$out/pmc_${c}_$1.log" 2>&1 || exit 1
  done
done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-48)
  echo "== pmc $set"
  rocprofv3 --pmc $set --kernel-trace -d "$out/sq_$n" -o out --output-format csv -- python3 scripts/frame_loop.py 4096 phong 32 \
      > "$out/sq_$n.log" 2>&1 || echo "counter set refused: $set"
done
echo "== pmc SQ_INSTS configs[4]"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace -d "$out/sqcfg4_insts" -o out --output-format csv -- python3 scripts/frame_loop.py 8192 specular 8 diablo 8 \
    > "$out/sqcfg4_insts.log" 2>&1 || echo "counter set refused (configs[4])"
find "$out" -name "*.csv" | wc -l

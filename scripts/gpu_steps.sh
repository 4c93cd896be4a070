#!/bin/bash
# Runs the given "name::timeout::command" steps on the GPU box, logging each to gpurun_out/<name>.log.
# Stops at the first step that times out (a hung GPU step must not be followed by another),
# keeps going after an ordinary failure.
mkdir -p gpurun_out
: > gpurun_out/status.txt
for step in "$@"; do
  name="${step%%::*}"; rest="${step#*::}"; tmo="${rest%%::*}"; cmd="${rest#*::}"
  echo "=== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "$name rc=$rc secs=$(( $(date +%s) - start ))" | tee -a gpurun_out/status.txt
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out; stopping"; exit 1; fi
done
exit 0

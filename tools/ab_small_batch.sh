#!/bin/bash
# A/B of option settings on one small batch alone, ONE PROCESS PER MEASUREMENT (HSA queues accumulate in a process that creates
# contexts repeatedly, and a batch's loop / RNG / fit streams then share hardware queues by luck: +-3 ms from object to object)
# usage (through gpurun): bash tools/ab_small_batch.sh <E> <runs> "<cfg>" "<cfg>" ...   (cfg = name=value,name=value or "")
cd "$GRAFT_REPO_ROOT"
E=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for cfg in "$@"; do
    python3 tools/time_small_batch2.py $E 7 "$cfg" 2>&1 | tail -1
  done
done

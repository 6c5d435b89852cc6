#!/bin/bash
# kernel-trace of a small batch under two option settings (environment GPET_<OPTION>): per-kernel stats and one mid-trace iteration each
# usage (through gpurun): bash tools/prof_small_batch_ab.sh <E> "<ENV A>" "<ENV B>"
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=${1:-32}
for tag in A B; do
  if [ $tag = A ]; then ENVS="$2"; else ENVS="$3"; fi
  O=gpurun_out/prof_ab_$tag
  rm -rf $O && mkdir -p $O
  echo "== $tag: $ENVS"
  ( export $ENVS; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/single_edge_latency.py $E > $O/run.log 2>&1 )
  grep -E "^loop|^one|^eight" $O/run.log
  python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("  %-60s calls %6s  avg %9.1f us  total %9.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  python3 tools/dump_iteration_kernels.py $O/trace 'k_fit<true, false>' 46 > $O/iteration_46.txt 2>&1 || true
  cat $O/iteration_46.txt
  rm -rf $O/trace
done

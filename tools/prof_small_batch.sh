#!/bin/bash
# rocprofv3 --kernel-trace of one small batch alone (tools/single_edge_latency.py E) -> gpurun_out/profiles_new/<tag>_<E>_edges_timeline.json
# usage (through gpurun, from the repo root): bash tools/prof_small_batch.sh <E> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=${1:-32}
R=${2:-r06}
O=gpurun_out/prof_sb_$E
rm -rf $O && mkdir -p $O gpurun_out/profiles_new
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/single_edge_latency.py $E > $O/run.log 2>&1
cat $O/run.log
python3 tools/single_edge_timeline.py $O/trace gpurun_out/profiles_new/${R}_${E}_edges_timeline.json | tee $O/timeline.log
# one early, one middle and one late iteration of the last repetition, kernel by kernel
for w in 40 46 52; do python3 tools/dump_iteration_kernels.py $O/trace 'k_fit<true, false>' $w > $O/iteration_$w.txt 2>&1 || true; done
cp $O/iteration_46.txt gpurun_out/profiles_new/${R}_${E}_edges_iteration.txt || true
rm -rf $O/trace

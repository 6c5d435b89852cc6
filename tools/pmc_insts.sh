#!/bin/bash
# Instruction counts of one kernel (run through gpurun from the repo root): tools/pmc_insts.sh <kernel substring> <script + args>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=$1; shift
O=gpurun_out/pmci
rm -rf $O; mkdir -p $O
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F64" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/g$i -- python3 "$@" > $O/run$i.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, sys, collections
K = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmci/g*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, c), v in sorted(acc.items()):
        print("%-42s %-32s last %.4g  (n=%d)" % (kn, c, v[-1], len(v)))
PY

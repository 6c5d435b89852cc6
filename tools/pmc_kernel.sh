#!/bin/bash
# PMC counters of one kernel of the per-iteration pipeline (run through gpurun from the repo root):
#   tools/pmc_kernel.sh <kernel-name-substring[,substring...]> <python script + args ...>
# One counter group per rocprofv3 run (never together with a trace), program directly after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=$1; shift
O=gpurun_out/pmck
rm -rf $O; mkdir -p $O
i=0
for G in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/g$i -- python3 "$@" > $O/run$i.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, sys, collections
K = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmck/g*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in K.split(",")):
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, c), v in sorted(acc.items()):
        print("%-42s %-32s last %.4g  (n=%d)" % (kn, c, v[-1], len(v)))
PY

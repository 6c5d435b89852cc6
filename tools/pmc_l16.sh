#!/bin/bash
# PMC counters of the objective kernel at P problems (run through gpurun from the repo root): tools/pmc_l16.sh [n] [P]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-98}; P=${2:-13312}
O=gpurun_out/pmc_l16
rm -rf $O; mkdir -p $O
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/g$i -- python3 tools/prof_l16.py $N $P > $O/run$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_l16/g*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_lml" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-24:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, c), v in sorted(acc.items()):
        print("%-26s %-32s last %.5g  (n=%d)" % (kn, c, v[-1], len(v)))
PY

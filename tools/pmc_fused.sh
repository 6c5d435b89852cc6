#!/bin/bash
# PMC counters of k_sample_score (run through gpurun from the repo root): tools/pmc_fused.sh [edges]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcf
rm -rf $O; mkdir -p $O
i=0
for G in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/g$i -- python3 tools/prof_fused.py ${1:-1024} > $O/run$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcf/g*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if any(k in kn for k in ("k_sample_score", "k_sample_keep", "k_score_tile", "k_sample_gemm")):
            acc[(kn.split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, c), v in sorted(acc.items()):
        print("%-46s %-28s last %.4g  (n=%d)" % (kn, c, v[-1], len(v)))
PY

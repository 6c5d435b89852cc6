#!/bin/bash
# Store-side counters of the sample GEMM against a store-only kernel of the same shape (run through gpurun from the repo
# root): tools/pmc_gemm_stores.sh [edges]   ->   gpurun_out/r03_gemm_store_counters.txt
# One counter group per rocprofv3 run (never together with a trace), program directly after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=${1:-1024}
O=gpurun_out/pmcg
rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/hbm_write tools/ubench/hbm_write.hip
i=0
for G in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR" "TCC_WRITE_sum TCC_REQ_sum TCC_EA0_WR_UNCACHED_32B_sum" "SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/k$i -- python3 tools/prof_stages.py $E 2 > $O/runk$i.log 2>&1 || echo "group $i (GEMM): not collected"
  rocprofv3 --pmc $G --output-format csv -d $O/u$i -- $O/hbm_write > $O/runu$i.log 2>&1 || echo "group $i (ubench): not collected"
done
python3 - <<'PY' | tee gpurun_out/r03_gemm_store_counters.txt
import csv, glob, collections
print("counters per launch: sample GEMM (tools/prof_stages.py, last launch) | store-only kernels of tools/ubench/hbm_write.hip (every launch mode)")
for tag, pat, key in (("gemm", "gpurun_out/pmcg/k*/*/*_counter_collection.csv", "k_sample_gemm"), ("store-only", "gpurun_out/pmcg/u*/*/*_counter_collection.csv", "k_")):
    for f in sorted(glob.glob(pat)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (kn, c), v in sorted(acc.items()):
            print("%-11s %-30s %-30s last %.5g  mean %.5g (n=%d)" % (tag, kn, c, v[-1], sum(v) / len(v), len(v)))
PY
grep -h "TB/s\|GB/s\|ms" $O/runu1.log | head -12 >> gpurun_out/r03_gemm_store_counters.txt || true
rm -rf $O

#!/bin/bash
# PMC counters of one kernel, arbitrary counter groups (run through gpurun from the repo root):
#   tools/pmc_groups.sh <out-name> <kernel-name-substring> "<group 1>" "<group 2>" ... -- <python script + args ...>
# One counter group per rocprofv3 run (never together with a trace), the program directly after `--`; a group with a
# counter this GPU does not have is reported and skipped.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=$1; K=$2; shift 2
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
O=gpurun_out/pmc_$NAME
rm -rf $O; mkdir -p $O
i=0
for G in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/g$i -- python3 "$@" > $O/run$i.log 2>&1 || echo "group $i ($G) failed: $(tail -2 $O/run$i.log | tr '\n' ' ')"
done
python3 - "$K" "$O" <<'PY'
import csv, glob, sys, collections
K, O = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(O + "/g*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, c), v in sorted(acc.items()):
        print("%-42s %-32s last %.4g  (n=%d)" % (kn, c, v[-1], len(v)))
PY

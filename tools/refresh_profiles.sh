#!/bin/bash
# Regenerates profiles/<round>_* on the GPU box (run through gpurun from the repo root; round tag = $2, default r05):
#   bench line, rocprofv3 --kernel-trace --stats of the same command, the dominant kernel alone, PMC traffic.
# Every rocprofv3 pass is its own run (PMC passes never share a run with a trace), program directly after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=${1:-1024}
R=${2:-r05}
O=gpurun_out/prof
rm -rf $O gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_lml_f gpurun_out/pmc_lml_w
mkdir -p $O
echo "== bench"; python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
echo "== bench under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py > $O/bench_traced.json 2> $O/bench_traced.err
echo "== stages under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stage_trace -- python3 tools/prof_stages.py $E 5 > $O/stages.log 2>&1
echo "== converged fits alone under rocprofv3 --kernel-trace"
rocprofv3 --kernel-trace --output-format csv -d $O/lml_trace -- python3 tools/prof_final.py $E 0 > $O/final.log 2>&1
echo "== PMC passes"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_stages.py $E 2 > $O/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/prof_stages.py $E 2 > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_lml_f -- python3 tools/prof_final.py $E 0 > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_lml_w -- python3 tools/prof_final.py $E 0 > $O/pmc4.log 2>&1
echo "== full-rank factor (config 5 frame shape) under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eig_trace -- python3 tools/prof_matern_factor.py 1024 1 3 > $O/eig.log 2>&1
python3 tools/prof_matern_factor.py 1024 8 3 >> $O/eig.log 2>&1
echo "== config 3 (n = 1500, Lg = 2048) stage entry points under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_trace -- python3 tools/prof_config3.py 5 > $O/c3.log 2>&1
echo "== single-edge trace under rocprofv3 --kernel-trace (launch gaps)"
rocprofv3 --kernel-trace --output-format csv -d $O/se_trace -- python3 tools/single_edge_latency.py > $O/se.log 2>&1
echo "== instruction counters of the objective kernel"
bash tools/pmc_l16.sh 98 13312 > $O/pmc_l16.log 2>&1 || echo "(pmc_l16 failed: see $O/pmc_l16.log)"
echo "== instruction counters of the generator (k_mt_normals4 against k_mt_normals)"
bash tools/pmc_groups.sh rng4 k_mt_normals "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" -- tools/time_rng4.py $E > $O/pmc_rng4.log 2>&1 || echo "(pmc rng4 failed)"
echo "== summaries"
mkdir -p gpurun_out/profiles_new
python3 tools/make_traffic_profile.py $E 500 $R > $O/traffic.log
cp profiles/${R}_pmc_traffic.json gpurun_out/profiles_new/
grep '^{' $O/bench.json | tail -1 | python3 -c "import json,sys; json.dump(json.loads(sys.stdin.read()), open('gpurun_out/profiles_new/'+'$R'+'_bench_n1.json','w'), indent=1)"
cp $(ls $O/bench_trace/*/*kernel_stats.csv | head -1) gpurun_out/profiles_new/${R}_bench_kernel_stats.csv
cp $(ls $O/stage_trace/*/*kernel_stats.csv | head -1) gpurun_out/profiles_new/${R}_stage_kernel_stats.csv
cp $(ls $O/eig_trace/*/*kernel_stats.csv | head -1) gpurun_out/profiles_new/${R}_matern_factor_kernel_stats.csv
python3 tools/trace_gaps.py $O/eig_trace > gpurun_out/profiles_new/${R}_matern_factor_launches.txt
grep "^N=" $O/eig.log >> gpurun_out/profiles_new/${R}_matern_factor_launches.txt
cp $(ls $O/c3_trace/*/*kernel_stats.csv | head -1) gpurun_out/profiles_new/${R}_config3_kernel_stats.csv
python3 tools/single_edge_timeline.py $O/se_trace gpurun_out/profiles_new/${R}_single_edge_timeline.json > $O/se_timeline.log 2>&1
cp $O/pmc_l16.log gpurun_out/profiles_new/${R}_k_lml16_counters.txt
cp $O/pmc_rng4.log gpurun_out/profiles_new/${R}_k_mt_normals4_counters.txt
DOM=$(grep '^{' $O/bench.json | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel'])")
echo "dominant kernel: $DOM"
if [ "$DOM" = "k_lml" ]; then
  # the objective kernel of the bench's batches is gpet::k_lml16 (exact name: k_lml16_fit and the register-tile kernel k_lml are
  # other kernels); the last third of its launches = the last of prof_final.py's three repetitions
  NL=$(python3 -c "import csv,glob,re; f=glob.glob('$O/lml_trace/*/*_kernel_trace.csv')[0]; print(sum(re.sub(r'^void\\s+','',r['Kernel_Name'].strip('\"')).split('(')[0].split('<')[0]=='gpet::k_lml16' for r in csv.DictReader(open(f)))//3)")
  python3 tools/summarise_trace.py $O/lml_trace gpet::k_lml16 $NL gpurun_out/profiles_new/${R}_dominant_kernel.json "rocprofv3 --kernel-trace of tools/prof_final.py $E 0: the gpet::k_lml16 launches of one batch's converged fits ($E edges x 13 restarts), nothing else on the GPU"
else
  python3 tools/summarise_trace.py $O/stage_trace gpet::$DOM 5 gpurun_out/profiles_new/${R}_dominant_kernel.json "rocprofv3 --kernel-trace --stats of tools/prof_stages.py $E 5: the kernel alone on $E edges at the bench's mid-trace state (7 iterations in)"
fi
rm -rf $O/bench_trace $O/stage_trace $O/lml_trace $O/eig_trace $O/c3_trace $O/se_trace gpurun_out/pmc_l16 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_lml_f gpurun_out/pmc_lml_w
ls -la gpurun_out/profiles_new

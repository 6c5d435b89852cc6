#!/bin/bash
# Per-kernel evidence of the round beside tools/refresh_profiles.sh (run through gpurun from the repo root; round tag = $1):
# the two Jacobi forms of the 72 x 72 eigenproblem (times, rocprofv3 kernel stats, the microbenchmarks its design rests on),
# the any-rank factor cold and warm (hipEvents + rocprofv3 kernel stats of the WARM path alone), one small batch alone.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r05}
O=gpurun_out/evid
P=gpurun_out/profiles_new
rm -rf $O; mkdir -p $O $P
echo "== Jacobi forms"
python3 tools/time_jacobi_ahead.py 1 32 1024 > $P/${R}_jacobi_forms.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/jac -- python3 tools/prof_jacobi_ahead.py 1 > $O/jac.log 2>&1
cp $(ls $O/jac/*/*kernel_stats.csv | head -1) $P/${R}_jacobi_kernel_stats.csv
cat $O/jac.log >> $P/${R}_jacobi_forms.txt
echo "== microbenchmarks"
for u in f64_chain jac_params_chain round_skeleton lane_shift; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o $O/$u && { echo "## tools/ubench/$u.hip"; $O/$u; } >> $P/${R}_jacobi_ubench.txt 2>&1
done
echo "== any-rank factor, cold and warm"
python3 tools/time_matern_warm.py 1024 1 3 > $P/${R}_matern_factor_warm.txt 2>&1
python3 tools/time_matern_warm.py 1024 8 3 >> $P/${R}_matern_factor_warm.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ojw -- python3 tools/time_matern_warm.py 1024 1 3 1 > $O/ojw.log 2>&1
cp $(ls $O/ojw/*/*kernel_stats.csv | head -1) $P/${R}_matern_factor_warm_kernel_stats.csv
echo "== one 32-edge batch alone"
python3 tools/time_small_batch.py 32 7 > $P/${R}_small_batches.txt 2>&1
rm -rf $O
ls -la $P

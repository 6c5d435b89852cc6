#!/bin/bash
# An instrumented copy of the library next to the product one (which stays as built by __graft_entry__.build()):
#   tools/build_instrumented.sh -DGPET_JAC_PROF            -> gpurun_scratch/libgpet_prof.so
#   GPET_LIB_PATH=$PWD/gpurun_scratch/libgpet_prof.so python tools/prof_jacobi.py 1 256 1024
# Switches (csrc/gpet_kernels.hip): GPET_JAC_PROF (cycles per phase of a Jacobi round), GPET_JAC_TRACE (off-norm and
# largest relative coupling per sweep), GPET_SH_PROF / GPET_FIT_PROF (cycles per phase of k_struct_H / k_fit),
# GPET_NO_XCD_REMAP (plain workgroup -> (edge, part) mapping).
set -e
cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.build()"   # objects of the other sources
mkdir -p gpurun_scratch
C=gaussian_process_edge_trace_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/gpet_kernels.hip -o gpurun_scratch/gpet_kernels_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_scratch/libgpet_prof.so $C/build/gpet_api_ctx.hip.o $C/build/gpet_api_batch.hip.o $C/build/gpet_api_stages.hip.o $C/build/gpet_api_final.hip.o $C/build/gpet_api_loop.hip.o $C/build/gpet_api_comm.hip.o \
  gpurun_scratch/gpet_kernels_prof.o $C/build/gpet_eig.hip.o $C/build/gpet_lbfgsb.hip.o $C/build/gpet_rng.hip.o $C/build/gpet_options.hip.o -ldl
ls -la gpurun_scratch/libgpet_prof.so

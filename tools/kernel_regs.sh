#!/bin/bash
# VGPR / AGPR / spill / scratch figures of the kernels whose mangled name matches $1 (gfx950 build of gpet_kernels.hip)
set -e
mkdir -p /tmp/st && cd /tmp/st
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "${2:-/root/repo/gaussian_process_edge_trace_amd/csrc/gpet_kernels.hip}" -save-temps -o /tmp/st/k.o 2>/dev/null
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|agpr_count|private_segment_fixed_size):" *gfx950.s | paste - - - - - | grep -E "$1" | sed 's/\s\+/ /g'

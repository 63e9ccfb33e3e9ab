#!/bin/bash
# usage: tools/isa_stats.sh <file.hip> [extra hipcc flags]: registers / spills of every kernel of a translation unit (device-only compile)
f=$1; shift
out=/tmp/$(basename $f .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize --offload-device-only -S /root/repo/ws_unet_amd/csrc/$f -o $out "$@" 2>&1 | grep -E "error" 
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_count|name):" $out | paste - - - - | sed 's/ \+/ /g;s/_ZN12_GLOBAL__N_1//;s/EEvNS_[0-9A-Za-z]*E//'

#!/bin/bash
# usage: tools/micro/isa.sh <name>   (compiles tools/micro/<name>.hip with -save-temps and prints register use + the memory / MFMA / wait skeleton)
cd /root/repo/tools/micro/_bin && hipcc --offload-arch=gfx950 -O3 -I ../../../dreamer4_amd/csrc -save-temps=obj -o $1 ../$1.hip 2>&1 | grep -E "error" ; S=$1-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|group_segment)|vgpr_spill" $S | head -8
grep "s_waitcnt\|v_mfma\|global_load\|buffer_load\|ds_read\|s_cbranch\|s_barrier\|scratch\|global_store\|ds_write" $S | awk '{print $1,$2}' | uniq -c | head -${2:-150}

#!/bin/bash
# usage: tools/micro/kres.sh <file.hip in dreamer4_amd/csrc>   -> compiles the object, prints per-kernel register use / spills; the .s stays under csrc/_obj/
cd /root/repo/dreamer4_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $1 -o _obj/$1.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Name|VGPRs|Scratch|Spill|Occupancy" | sed 's/\[-Rpass.*//'

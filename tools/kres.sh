#!/bin/bash
# Prints per-kernel register / LDS / occupancy figures for one HIP source.
# usage: tools/kres.sh dsp_amd/csrc/file.hip [filter-regex] [extra hipcc flags...]
src=$1; filt=${2:-.}; shift; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "Function Name|    VGPRs:|VGPRs Spill|ScratchSize|LDS Size|Occupancy|TotalSGPRs" \
 | sed -E 's/^.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' \
 | sed -E 's/Function Name: //' | c++filt | sed -E 's/\(anonymous namespace\):://g; s/\(FirParams<[a-z]+>\)//' | grep -E "$filt" | cut -c1-240

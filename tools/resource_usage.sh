#!/bin/bash
# kernel resource usage (VGPR / AGPR / LDS / spills / occupancy) of one csrc file: tools/resource_usage.sh conv_halo8.hip [filter]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip -c yolort_amd/csrc/$1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/\[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" |"$0}END{print line}' | sed 's/ \+/ /g' | grep -E "${2:-.}"

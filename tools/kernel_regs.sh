#!/bin/bash
# VGPR / spill / scratch report of one .hip file:  tools/kernel_regs.sh targetdiff_amd/csrc/edge16.hip [filter]
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -fno-slp-vectorize -c "$1" -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize" |
  sed -E 's/.*remark: [^ ]+ +//; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - | grep -i "${2:-.}"

#!/bin/bash
# per-kernel VGPR / scratch (spill) report of one HIP source: tools/kernel_regs.sh mfcnet-tracker_amd/csrc/conv_igemm.hip [filter]
cd "$(dirname "$0")/../mfcnet-tracker_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -Wno-unused-variable -Rpass-analysis=kernel-resource-usage -c "$(basename "$1")" -o /tmp/regs.o 2>&1 \
 | python3 -c "
import sys,re
name=None; v=None
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: name=m.group(1)
    m=re.search(r' VGPRs: (\d+)',ln)
    if m: v=m.group(1)
    m=re.search(r'ScratchSize \[bytes/lane\]: (\d+)',ln)
    if m and name: print(name, 'vgpr', v, 'scratch', m.group(1))
" | grep -E "${2:-.}"

#!/bin/bash
# Developer aid: VGPRs / spills / LDS / occupancy of every kernel of one translation unit (hipcc remarks).
# usage: scripts/kernel_resources.sh dgl-kgat_amd/csrc/kgat_spmm_bi.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
python3 -c '
import re, sys
cur = {}
for ln in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs|VGPR Spill|SGPR Spill): (.*?) \[-Rpass", ln)
    if not m: 
        m = re.search(r"remark: (Function Name|    VGPRs|    AGPRs|    ScratchSize \[bytes/lane\]|    Occupancy \[waves/SIMD\]|    LDS Size \[bytes/block\]|    SGPRs): (\S+)", ln)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        if cur: print(cur)
        cur = {"fn": v}
    else:
        cur[k.split()[0]] = v
if cur: print(cur)
'

#!/bin/bash
# tools/pair_resources.sh -- VGPRs / scratch / occupancy of the paired-filter instantiations of filter_dna_kernel (both units)
cd "$(dirname "$0")/../sassy_amd/csrc"
for prof in 1 2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSASSY_SCAN_PROFILE=$prof -Rpass-analysis=kernel-resource-usage -c scan_kernel.hip -o /tmp/kr_$$.o 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); d={}; continue
    for key,short in (('VGPRs','vgpr'),('ScratchSize \[bytes/lane\]','scratch'),('Occupancy \[waves/SIMD\]','occ'),('SGPRs Spill','sspill'),('VGPRs Spill','vspill')):
        m=re.search(r'remark: +'+key+r': (\d+)',line)
        if m and cur: d[short]=int(m.group(1))
    if 'LDS Size' in line and cur:
        if 'filter_dna_kernel' in cur and 'Lb1ELi' in cur: print(cur[24:56], d)
        cur=None
"; done; rm -f /tmp/kr_$$.o

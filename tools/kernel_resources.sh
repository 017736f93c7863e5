#!/bin/bash
# tools/kernel_resources.sh [unit ...] -- VGPRs / scratch / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../sassy_amd/csrc"
units=${@:-"scan_kernel.hip:1 scan_kernel.hip:2 scan_kernel.hip:0 count_filter.hip trace_kernel.hip aux_kernels.hip seed_kernels.hip tiled_kernel.hip sort_kernels.hip"}
for u in $units; do
  f=${u%%:*}; d=""; [[ "$u" == *:* ]] && d="-DSASSY_SCAN_PROFILE=${u##*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $d -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kr_$$.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs Spill','VGPRs Spill'):
        m=re.search(r'    '+key+r': (\d+)',line)
        if m and cur is not None: cur[key.split()[0]]=int(m.group(1))
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    print(f\"$u  {n[:90]:90s} vgpr {r.get('VGPRs')} scratch {r.get('ScratchSize')} occ {r.get('Occupancy')} sspill {r.get('SGPRs')}\")
"
done
rm -f /tmp/kr_$$.o

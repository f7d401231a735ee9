#!/bin/bash
# compact per-kernel resource table of the PM translation unit: VGPRs, scratch bytes per lane, spills, occupancy
cd "$(dirname "$0")/../sea_ice_drift_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden $SID_DEFS \
  -Rpass-analysis=kernel-resource-usage -c ${1:-pm_kernel_mfma.hip} -o /dev/null 2>&1 | python3 -c "
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r'remark: +([A-Za-z \[\]/]+): (\S+)', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name': cur = {'name': v}
    cur[k] = v
    if k.startswith('LDS Size'):
        m2 = re.search(r'pm_kernel_(rp|mfma)I(.*?)EEvNS', cur['name'])
        if m2: print('%-5s %-28s VGPR %4s scratch %5s  sgpr-spill %3s vgpr-spill %3s occ %s' % (m2.group(1), m2.group(2).replace('Li','').replace('Lb',' b').replace('E',' '), cur.get('VGPRs'), cur.get('ScratchSize [bytes/lane]'), cur.get('SGPRs Spill'), cur.get('VGPRs Spill'), cur.get('Occupancy [waves/SIMD]')))
"

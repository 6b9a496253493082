#!/bin/bash
# VGPRs / spills / scratch / occupancy of the kernels of one translation unit, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage): tools/kernel_resources.sh [file.hip] ["extra flags"] [name filter regex]
cd "$(dirname "$0")/../imd_amd/csrc"
F=${1:-imdgpu.hip}; EXTRA=$2; PAT=${3:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $EXTRA -Rpass-analysis=kernel-resource-usage -c "$F" -o /tmp/kr_$$.o 2>&1 | python3 -c "
import sys, re, subprocess
name = None; d = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: name = m.group(1); d[name] = {}
    for k, t in (('  VGPRs:', 'vgpr'), ('VGPRs Spill', 'spill'), ('ScratchSize', 'scratch'), ('Occupancy', 'occ'), ('SGPRs:', 'sgpr')):
        if k in l and name: d[name][t] = re.sub(r'\s*\[-Rpass.*$', '', l.strip()).split()[-1]
names = list(d)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n') if names else []
print('%-70s %6s %6s %6s %8s %4s' % ('kernel', 'VGPRs', 'SGPRs', 'spill', 'scratchB', 'occ'))
for n, dn in zip(names, dem):
    dn = re.sub(r'^void ', '', dn); dn = re.sub(r'\(.*$', '', dn)
    if re.search(sys.argv[1], dn):
        v = d[n]; print('%-70s %6s %6s %6s %8s %4s' % (dn[:70], v.get('vgpr'), v.get('sgpr'), v.get('spill'), v.get('scratch'), v.get('occ')))
" "$PAT"
rm -f /tmp/kr_$$.o

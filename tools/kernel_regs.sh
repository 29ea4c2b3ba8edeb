#!/bin/bash
# Register / LDS / occupancy table of every kernel in the library (device-only compile, no GPU needed).
# usage: tools/kernel_regs.sh [filter-regex] [extra hipcc flags...]
FILTER=${1:-.}; shift
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -S -o /tmp/rt_capi.s rt_capi.hip \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'remark: [^:]*:\d+:\d+: (.*?) \[-Rpass', line) or re.search(r'remark: (.*?) \[-Rpass', line)
    if not m:
        if 'error' in line or 'warning' in line: sys.stderr.write(line)
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = n.replace('rt::', '').split('(')[0].replace('void ', '')
    if not re.search(sys.argv[1], n): continue
    print(f\"{n:48s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs', r.get('SGPRs','?')):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?'):>6s}\")
" "$FILTER"

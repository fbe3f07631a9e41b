# registers / spills / scratch per kernel of one .hip file:  tools/kres.sh datr_amd/csrc/gemm_f32.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I/root/repo/datr_amd/csrc -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
cur = None
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = m.group(1); d = {}; continue
    m = re.search(r'(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', l)
    if m and cur:
        d[m.group(1)] = m.group(2)
        if m.group(1).startswith('LDS'):
            print(cur[:70], 'vgpr', d.get('VGPRs'), 'sgpr', d.get('TotalSGPRs'), 'spill', d.get('VGPRs Spill'), 'scratch', d.get('ScratchSize [bytes/lane]'), 'occ', d.get('Occupancy [waves/SIMD]'))
"

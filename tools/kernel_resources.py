"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

    python tools/kernel_resources.py boxinstseg_amd/csrc/fused_eval.hip [-DNAME=VALUE ...]
"""
import re
import subprocess
import sys
import tempfile

src, extra = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', '-Wall', '-Wno-unused-function', '-mllvm',
           '-amdgpu-kernarg-preload-count=16', '-Rpass-analysis=kernel-resource-usage', *extra, src, '-o', d + '/x.o']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name)}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+(?:\[[A-Za-z/ ]*\])?): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    elif 'error' in line or 'warning' in line:
        print(line)
print('%-44s %6s %6s %6s %7s %7s %7s %4s %6s' % ('kernel', 'SGPRs', 'VGPRs', 'AGPRs', 'scratch', 'spillS', 'spillV', 'occ', 'LDS'))
for r in rows:
    print('%-44s %6d %6d %6d %7d %7d %7d %4d %6d' % (r['name'][:44], r.get('TotalSGPRs', -1), r.get('VGPRs', -1), r.get('AGPRs', -1),
                                                       r.get('ScratchSize [bytes/lane]', -1), r.get('SGPRs Spill', -1), r.get('VGPRs Spill', -1),
                                                       r.get('Occupancy [waves/SIMD]', -1), r.get('LDS Size [bytes/block]', -1)))

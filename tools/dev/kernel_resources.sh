#!/bin/bash
# Registers, occupancy and static LDS of every kernel of the library, from the compiler's own report
# (hipcc -Rpass-analysis=kernel-resource-usage): profiles/r04_kernel_resources.txt
cd "$(dirname "$0")/../../cedarx_h264_encoder_amd/csrc"
for f in enc_k_parallel enc_k_wavefront enc_k_deblock enc_k_intra enc_k_entropy; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -x hip --cuda-device-only -c $f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c "
import sys, re, subprocess
cur = {}
def flush():
    if cur.get('name'):
        n = subprocess.run(['c++filt', cur['name']], capture_output=True, text=True).stdout.strip().replace('cedar::', '').split('(')[0]
        print('%-34s vgpr %3s  sgpr %3s  waves/SIMD %s  scratch %s  static LDS %s' % (n[:34], cur.get('VGPRs'), cur.get('TotalSGPRs'), cur.get('Occupancy [waves/SIMD]'), cur.get('ScratchSize [bytes/lane]'), cur.get('LDS Size [bytes/block]')))
for line in sys.stdin:
    m = re.search(r'remark: +(.*?): (.*?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        flush(); cur = {'name': v}
    else:
        cur[k] = v
flush()
"
done

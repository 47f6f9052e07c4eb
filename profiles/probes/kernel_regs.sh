# VGPR / AGPR / spill / occupancy report of the kernels of one source file whose mangled name contains a pattern:
#   bash profiles/probes/kernel_regs.sh stin_gemm.hip k_gemm_nt_stream
cd $(dirname $0)/../../surface_texture_inpainting_net_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kernel_regs.o 2>&1 |
  python3 -c "
import sys, re
pat = sys.argv[1]
cur = None
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1) if pat in m.group(1) else None
        if cur: print(cur)
        continue
    if cur:
        m = re.search(r'(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)', l)
        if m: print('    %s = %s' % (m.group(1), m.group(2)), end='')
        if 'LDS Size' in l: print()
" $2

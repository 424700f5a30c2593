#!/bin/bash
# Build-container helper: registers, spills and scratch of every kernel of nrq_device.hip (code object metadata of the gfx950
# build) -> stdout.   bash tools/kernel_registers.sh > profiles/r4_kernel_registers.txt
set -e
D=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -Wno-pass-failed -I/root/repo/include -I/root/repo/nanorq_amd/csrc --cuda-device-only -S /root/repo/nanorq_amd/csrc/nrq_device.hip -o $D/dev.s 2>/dev/null
python3 - $D/dev.s <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
md = txt[txt.index('amdhsa.kernels:'):]
rows = []
for e in re.split(r'\n  - ', md)[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, e) or [None, '?'])[1]
    if g('name') == '?':
        continue
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    name = name.split('(')[0].replace('void ', '')
    rows.append((name, g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('max_flat_workgroup_size')))
print("# code object metadata of nrq_device.hip (hipcc --offload-arch=gfx950 -O3): registers, spilled registers and scratch bytes per kernel")
print("# nrq_solve_kernel<strip bytes, workgroup threads, waves per SIMD it is built for, lanes per element, aligned-only movers>; nrq_plan_kernel<threads, compact peeling state>")
print("%-52s %6s %10s %6s %10s %10s %6s" % ("kernel", "vgpr", "vgpr_spill", "sgpr", "sgpr_spill", "scratch_B", "wg"))
for r in sorted(rows):
    print("%-52s %6s %10s %6s %10s %10s %6s" % r)
PY
rm -rf $D

# Build-container helper: device assembly of nrq_device.hip (-> /tmp/t/dev.s) and the headline kernel's part of it (-> /tmp/t/k16.s)
mkdir -p /tmp/t && cd /tmp/t && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -Wno-pass-failed -I/root/repo/include -I/root/repo/nanorq_amd/csrc "$@" --cuda-device-only -S /root/repo/nanorq_amd/csrc/nrq_device.hip -o dev.s 2>&1 | grep -v "warning: argument unused" 
awk '/^_Z16nrq_solve_kernelILi16ELi768ELi1ELi1E[A-Za-z0-9_]*:/{p=1} p{print} p&&/s_endpgm/{exit}' dev.s > k16.s
grep -A30 "^    .name:           _Z16nrq_solve_kernelILi16ELi768ELi1ELi1E" dev.s | grep -E "vgpr_count|spill" 
wc -l k16.s

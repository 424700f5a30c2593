#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the reference's OWN benchmark.c, compiled unchanged against this library
# (oracle/_refprog/benchmark_hip, recipe oracle/Makefile), with the arguments of the reference's Makefile (graph.dat,
# /root/reference/Makefile:35-45): four columns in Mbit/s -- encode, precalc-encode, decode, decode with 5 % overhead.
# One source block per call through the unchanged per-block API (nanorq_generate_symbols / nanorq_repair_block), host
# memory to host memory; the harness ends in assert(in == out).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/reference_benchmark.sh > gpurun_out/r4_reference_benchmark.txt'
REPO=${GRAFT_REPO_ROOT:-$PWD}
B=$REPO/oracle/_refprog/benchmark_hip
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}   # (what include/nanorq.h asks a host process to export)
[ -x "$B" ] || { echo "no $B (built in the build container only)"; exit 1; }
$B 1280 1000 5.0 > /dev/null 2>&1   # first process on a fresh box: driver / code-object caches
echo "# reference benchmark.c on libnanorq_hip.so, $(rocm-smi --showproductname 2>/dev/null | grep -m1 -o 'MI[0-9A-Za-z]*' || echo MI355X), T=1280, 6 % loss; Mbit/s"
echo "K       encode   precalc  decode  decode-oh5"
for K in 100 500 1000 5000 10000 50000; do
  timeout 600 $B 1280 $K 5.0 || echo "$K FAILED rc=$?"
done

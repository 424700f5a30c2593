#!/bin/bash
# Runs ON THE GPU BOX (gpurun; one GPU): what an 8-GPU run asks of the HOST, measured where only one GPU is to be had.
#  (1) bench.py with N = 1, 2, 4, 8 RANKS on GPU 0 at constant total work (256 / N blocks per rank and step, gloo for the barrier):
#      N host processes -- N torch runtimes, N library contexts, N host plan builds per step, N launch paths -- feed one device.
#      The aggregate must stay near the one-rank figure, else something on the host serialises (the GPU's own work is the same).
#  (2) tools/bench_one_object.py with device 0 named 8 times at the cfg5 / cfg4 sizes of SURVEY 8(e): 64 blocks of K'=56403
#      (8 per "device") and 8 blocks of K=27000, T=65504 (one per "device", 1.77 GB each) -- the thread-per-device path, its
#      memory budgets and staging at the real block counts.
#   gpurun --timeout 2400 -- 'bash tools/ranks_on_one_gpu.sh > gpurun_out/r6_ranks_on_one_gpu.txt 2>&1'
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
export GPU_MAX_HW_QUEUES=8
echo "# host: $(nproc) hardware threads, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | xargs); NUMA nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)"
python - <<'PY'
from nanorq_amd import shard
g = shard.gpu_cpu_lists()
print("# GPUs by KFD topology:", [(n, "%d cpus" % len(c)) for n, c in g])
for n in (1, 2, 4, 8):
    print("#   %d ranks on GPU 0 ->" % n, [("%d-%d" % (s[0], s[-1]), len(s)) for s in shard.rank_cpus(n, gpus=g, device_of_rank=[0] * n)])
PY
run_n() { # N blocks-per-rank label
  N=$1; B=$2
  for rep in 1 2; do
    timeout 900 python bench.py --gpus $N --force-device 0 --dist-backend gloo --blocks $B --steps 10 --warmup 3 --cpu-sample 0 --alg-sample 0 --pmc off --no-e2e --one-object off 2> /tmp/r1g.err | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('ranks %d x %3d blocks: %7.1f Gbit/s aggregate  %7.2f ms/step  | rank 0: kernels enc %.2f dec %.2f planner %.2f ms, host CPU %.1f ms/step on cpus %s' % ($N, $B, d['value'], d['ms_per_step'],
          d['detail']['encode_solve_ms'] or 0, d['detail']['decode_solve_ms'] or 0, d['detail']['planner_ms'] or 0, d['detail']['host_cpu_ms_per_step_rank0'],
          (lambda c: c.split(',')[0] + '..' + c.split(',')[-1] if c else 'all')(d['detail']['rank_cpus'])))
except Exception as e:
    print('ranks $N FAILED', e, open('/tmp/r1g.err').read()[-400:])"
  done
}
echo "== (1a) N ranks on GPU 0, every rank the FULL per-GPU work (256 blocks per rank and step): what each of N ranks asks of the host is what it"
echo "        asks on its own GPU; the device is time-shared, so the aggregate should stay at the one-rank figure unless the host falls behind"
for N in 1 2 4 8; do run_n $N 256; done
echo "== (1b) N ranks on GPU 0, constant total work (256 blocks per step in total): N small launches per step that each want every CU's LDS"
for N in 2 4 8; do run_n $N $((256 / N)); done
[ "${SKIP_OBJ:-0}" = "1" ] && exit 0
echo "== (2) one object, device 0 named 8 times (a host thread, a context, streams and staging per name)"
timeout 900 python tools/bench_one_object.py --devices 0,0,0,0,0,0,0,0 --K 56403 --T 1280 --blocks 64 --loss 0.2 2> /tmp/o1.err | tail -1 | cut -c1-900 || tail -3 /tmp/o1.err
timeout 900 python tools/bench_one_object.py --devices 0 --K 56403 --T 1280 --blocks 64 --loss 0.2 2> /tmp/o1.err | tail -1 | cut -c1-900 || tail -3 /tmp/o1.err
FREE_GB=$(awk '/MemAvailable/ {print int($2 / 1048576)}' /proc/meminfo)
echo "# host memory available: ${FREE_GB} GB (the cfg4 object holds ~60 GB of page-locked host buffers: run only with 160 GB to spare)"
[ "$FREE_GB" -ge 160 ] && timeout 1200 python tools/bench_one_object.py --devices 0,0,0,0,0,0,0,0 --K 27000 --T 65504 --blocks 8 --loss 0.1 --reps 1 2> /tmp/o2.err | tail -1 | cut -c1-900 || tail -3 /tmp/o2.err
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2

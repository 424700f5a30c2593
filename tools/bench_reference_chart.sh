# The K values of the reference's published chart (graph.png: T=1280), through bench.py on one GPU (run on the GPU box).
cd $GRAFT_REPO_ROOT
for cfg in "100 8192 0.06" "500 4096 0.06" "1000 2048 0.06" "5000 512 0.06" "10000 256 0.06" "50000 16 0.06"; do set -- $cfg
  timeout 600 python bench.py --K $1 --T 1280 --blocks $2 --loss $3 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); de=d['detail']
print('K=%s blocks=%s: %.1f Gbit/s enc+dec (encode %.0f, decode incl. planner %.0f Gbit/s on device), %.1f ms/step, strip %sB wg %s' % ('$1','$2',d['value'],de['encode_gbps_device'],de['decode_gbps_device_incl_planner'],d['ms_per_step'],de['decode']['strip_bytes'],de['decode']['wg_threads']))"
done

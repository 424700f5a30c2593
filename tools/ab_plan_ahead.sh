mkdir -p gpurun_out/r4d
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "plan_issued_ahead or lazy or capacity" 2>&1 | tail -8) > gpurun_out/r4d/tests.log
B="python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pmc off --no-e2e"
for pa in off 1 2; do
  $B --K 56403 --blocks 8 --loss 0.2 --plan-ahead $( [ $pa = off ] && echo off || echo on ) --plan-ahead-depth $( [ $pa = off ] && echo 1 || echo $pa ) > gpurun_out/r4d/cfg5_pa_$pa.json 2> gpurun_out/r4d/cfg5_pa_$pa.err
  $B --K 27000 --T 65504 --blocks 1 --loss 0.1 --plan-ahead $( [ $pa = off ] && echo off || echo on ) --plan-ahead-depth $( [ $pa = off ] && echo 1 || echo $pa ) > gpurun_out/r4d/cfg4_pa_$pa.json 2> gpurun_out/r4d/cfg4_pa_$pa.err
  $B --K 20000 --blocks 64 --loss 0.1 --plan-ahead $( [ $pa = off ] && echo off || echo on ) --plan-ahead-depth $( [ $pa = off ] && echo 1 || echo $pa ) > gpurun_out/r4d/K20000_pa_$pa.json 2> gpurun_out/r4d/K20000_pa_$pa.err
  $B --K 50000 --blocks 16 --loss 0.06 --plan-ahead $( [ $pa = off ] && echo off || echo on ) --plan-ahead-depth $( [ $pa = off ] && echo 1 || echo $pa ) > gpurun_out/r4d/K50000_pa_$pa.json 2> gpurun_out/r4d/K50000_pa_$pa.err
done
cat gpurun_out/r4d/tests.log
for f in gpurun_out/r4d/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"],1), "Gbit/s", round(d["ms_per_step"],2), "ms/step planner_ms", d["detail"]["planner_ms"], "enc", d["detail"]["encode_solve_ms"], "dec", d["detail"]["decode_solve_ms"], d["config"].get("decode_found_plan_ahead"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
for f in gpurun_out/r4d/*.err; do tail -n 2 $f | grep -v amdgpu.ids; done

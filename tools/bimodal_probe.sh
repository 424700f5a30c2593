cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --K 500 --T 1280 --blocks 4096 --loss 0.06 --steps 5 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('%.1f Gbit/s %.2f ms/step enc %.2f dec %.2f planner %.2f' % (d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms'], de.get('planner_ms') or 0))"; done

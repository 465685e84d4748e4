( timeout 1500 python -m pytest tests/test_late_regime.py tests/test_lowrank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 ) 2>&1 | sed "s/^/tests: /"
for vb in 1 0; do
( MLP_VBRANCH=$vb timeout 900 python bench.py --no-full-solve --no-factor-transport --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d['windows']
print(d['value'], {k:(w[k]['us_per_pivot'], w[k].get('kernels',{}).get('w_pass_v')) for k in ('mid','late')}, d['roofline']['frac'], d['roofline'].get('pivot_level',{}).get('frac'))" ) 2>&1 | sed "s/^/vb=$vb: /"
done

# sweep_env.sh VAR "values" "cfgs": bench ms_per_step for each value of an environment knob, two rounds
cd $GRAFT_REPO_ROOT
for cfg in $3; do for round in 1 2; do for v in $2; do
  env $1=$v timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg $1=$v ms_per_step %.3f '%d['ms_per_step']+' '.join('%s %.3f'%(n,x['ms_per_step']) for n,x in k.items()))"
done; done; done

# A/B of one environment knob over bench configs: ab_env.sh VAR "cfgs"   (e.g. HRPT_WF_SPECULATIVE "2 5")
cd $GRAFT_REPO_ROOT
for cfg in $2; do for v in 1 0 1 0; do
  env $1=$v timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg $1=$v ms_per_step %.3f '%d['ms_per_step']+' '.join('%s %.3f'%(n,x['ms_per_step']) for n,x in k.items()))"
done; done

# env_sweep.sh VAR "values" "cfgs": bench.py per value of an environment knob, kernel class times per step
cd $GRAFT_REPO_ROOT
for cfg in $3; do for v in $2; do
  extra=""; [ "$cfg" = "5" ] && extra="--spp 8"
  env $1=$v timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg $1=%-4s ms/step %.3f one-frame %.3f | '%('$v', d['ms_per_step'], d.get('one_frame_in_flight',{}).get('ms_per_step',0))+' '.join('%s %.3f'%(n[3:],x['ms_per_step']) for n,x in k.items()))"
done; done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for round in 1 2; do for sp in 0 2 1; do
  HRPT_WF_SHADOW_PATH=$sp timeout -k 10 300 python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config 4 HRPT_WF_SHADOW_PATH=$sp: ms/step %.3f one-frame %.3f | '%(d['ms_per_step'], d.get('one_frame_in_flight',{}).get('ms_per_step',0))+' '.join('%s %.3f'%(n[3:],x['ms_per_step']) for n,x in k.items()))"
done; done 2>&1 | tee gpurun_out/r03/shadow_path_config4.txt

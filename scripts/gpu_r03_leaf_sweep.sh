# GPU builder leaf size (1..4 triangles) with the per-scene node format, configs 4 and 5 (8 spp), interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for cfg in 4 5; do for round in 1 2; do for leaf in 2 1 3 4; do
  extra=""; [ "$cfg" = "5" ] && extra="--spp 8"
  HRPT_GPU_BVH_MAX_LEAF=$leaf timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg max leaf $leaf: ms/step %.3f one-frame %.3f | '%(d['ms_per_step'], d.get('one_frame_in_flight',{}).get('ms_per_step',0))+' '.join('%s %.3f'%(n[3:],x['ms_per_step']) for n,x in k.items()))"
done; done; done 2>&1 | tee gpurun_out/r03/leaf_sweep.txt

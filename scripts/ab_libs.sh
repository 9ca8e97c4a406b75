# ab_libs.sh "LIB_A LIB_B ..." "cfgs" [rounds]: bench several builds of the library (paths relative to the repo root; "-" = the in-tree
# one), interleaved inside one GPU job -- the only comparison that survives the +-2 % box-to-box spread. Config 5 runs at 8 spp here.
cd $GRAFT_REPO_ROOT
for cfg in $2; do for round in $(seq 1 ${3:-2}); do for lib in $1; do
  extra=""; [ "$cfg" = "5" ] && extra="--spp 8"
  path=""; [ "$lib" != "-" ] && path=$GRAFT_REPO_ROOT/$lib
  env HRPT_LIBRARY=$path timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg %-44s ms/step %.3f one-frame %.3f | '%('$lib', d['ms_per_step'], d.get('one_frame_in_flight',{}).get('ms_per_step',0))+' '.join('%s %.3f'%(n[3:],x['ms_per_step']) for n,x in k.items()))"
done; done; done

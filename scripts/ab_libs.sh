# ab_libs.sh "name=lib.so name2=lib2.so ..." "cfgs" [rounds]: interleaved bench of several builds of the same ABI inside one GPU job; "new" = in-tree.
# Paths are relative to the repository root. Config 5 runs at 8 spp (its per-launch figures do not change with spp).
cd $GRAFT_REPO_ROOT
for cfg in $2; do for round in $(seq 1 ${3:-2}); do for ent in $1 new=; do
  name=${ent%%=*}; lib=${ent#*=}; [ -n "$lib" ] && lib=$GRAFT_REPO_ROOT/$lib
  extra=""; [ "$cfg" = "5" ] && extra="--spp 8"
  env HRPT_LIBRARY=$lib timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg lib=%-6s ms/step %.3f one-frame %.3f | '%('$name', d['ms_per_step'], d.get('one_frame_in_flight',{}).get('ms_per_step',0))+' '.join('%s %.3f'%(n[3:],x['ms_per_step']) for n,x in k.items()))"
done; done; done

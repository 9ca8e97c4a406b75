# ab_lib.sh OTHER_LIB "cfgs": bench the in-tree library against another build of the same ABI, interleaved, in one call
cd $GRAFT_REPO_ROOT
for cfg in $2; do for round in 1 2; do for lib in "$1" ""; do
  env HRPT_LIBRARY=$lib timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('config $cfg lib=%-10s ms_per_step %.3f '%('${lib:+base}' or 'new', d['ms_per_step'])+' '.join('%s %.3f'%(n,x['ms_per_step']) for n,x in k.items()))"
done; done; done

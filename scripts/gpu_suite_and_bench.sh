# full GPU suite, then bench configs 2/4/5 (device ms per kernel class)
set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/parity.log 2>&1 || { tail -40 gpurun_out/parity.log; exit 1; }
tail -2 gpurun_out/parity.log
rm -f gpurun_out/bench_cfgs.log
for cfg in 2 4 5; do
  timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('config $cfg ms_per_step %.3f Mrays/s %.0f '%(d['ms_per_step'],d['value'])+' '.join('%s %.3f'%(n,v['ms_per_step']) for n,v in k.items()))
" | tee -a gpurun_out/bench_cfgs.log
done

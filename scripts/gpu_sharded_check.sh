# GPU suite + the N>1 code path of bench.py with a single rank (RCCL world 1): pipelined and serial gather
set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/parity.log 2>&1 || { tail -30 gpurun_out/parity.log; exit 1; }
tail -3 gpurun_out/parity.log
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], 'ms_per_step %.3f Mrays/s %.0f kernel %s frac %.2f'%(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['frac']), d['config']['sharding'])
" "$1"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | show "plain N=1     "
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-sharded 2>/dev/null | show "sharded pipelined"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-sharded --serial-gather 2>/dev/null | show "sharded serial "

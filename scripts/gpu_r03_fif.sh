# frames in flight: 2 (default on one GPU) against 3 and 4, interleaved, configs 2 / 4 / 5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for cfg in 2 4 5; do for round in 1 2; do for fif in 2 3 4; do
  timeout -k 10 300 python3 bench.py --config $cfg --steps 12 --warmup 4 --no-cpu-baseline --frames-in-flight $fif 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $cfg frames in flight $fif: ms/step %.3f  one-frame %.3f  pool %.1f GB'%(d['ms_per_step'], d.get('one_frame_in_flight',{}).get('ms_per_step',0), d['roofline'].get('queue_pool_bytes',0)/1e9))"
done; done; done 2>&1 | tee gpurun_out/r03/frames_in_flight.txt

mkdir -p gpurun_out/short
B="python bench.py --bank-cache /tmp/bank --cpu-baseline 0"
$B --steps 20 > /dev/null 2>&1
for cfg in "20 5" "20 5" "20 200" "20 200" "200 5" "200 5" "100 5" "50 5" "20 25"; do set -- $cfg; $B --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('steps %s warmup %s: %.1f M  %.4f ms/step' % (d['steps'], d['warmup'], d['value']/1e6, d['ms_per_step']))"; done

# steps per launch / chains / lead / lag for the headline and the driver's command (round 5)
mkdir -p gpurun_out/r05
one() { # one <tag> <flags...>
  tag=$1; shift
  timeout -k 10 200 python bench.py --bank-cache /tmp/bank --cpu-baseline 0 "$@" > gpurun_out/r05/sweep_$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r05/sweep_$tag.json')); k=list(d['roofline']['kernels'].values())[0]
print('%-34s %6.1f M  ms/step %.5f  kernel avg_ms %s legs %s' % ('$tag', d['value']/1e6, d['ms_per_step'], k['avg_ms'], d['roofline']['legs']))"
}
for sub in 1 2 4; do for T in 16 64; do one s2000_sub${sub}_T$T --sub-batches $sub --multi $T; done; done
one s2000_sub2_T64_l6_g16 --sub-batches 2 --multi 64 --multi-lead 6 --multi-lag 16
one s2000_sub1_T64_l8_g24 --sub-batches 1 --multi 64 --multi-lead 8 --multi-lag 24
one s2000_sub1_T64_l16_g40 --sub-batches 1 --multi 64 --multi-lead 16 --multi-lag 40
for sub in 1 2 4; do for T in 5 10 20; do one drv_sub${sub}_T$T --gpus 1 --steps 20 --warmup 5 --sub-batches $sub --multi $T; done; done
one drv_sub4_T1 --gpus 1 --steps 20 --warmup 5 --sub-batches 4 --multi 1
one drv_sub1_T20_l6_g12 --gpus 1 --steps 20 --warmup 5 --sub-batches 1 --multi 20 --multi-lead 6 --multi-lag 12
one drv_sub2_T20_l4_g10 --gpus 1 --steps 20 --warmup 5 --sub-batches 2 --multi 20 --multi-lead 4 --multi-lag 10

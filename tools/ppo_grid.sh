mkdir -p gpurun_out/r03/ppo
timeout -k 10 500 python -m pytest tests/test_gpu_ppo.py -m gpu -x -q 2>&1 | tail -12
timeout -k 10 200 python examples/ppo.py --task pathfollow --envs 2048 --rollout 32 --updates 120 --log-every 10 > gpurun_out/r03/ppo/ppo_pathfollow_2048x32x120.log 2>&1; tail -2 gpurun_out/r03/ppo/ppo_pathfollow_2048x32x120.log | cut -c1-300

"""-m gpu: the reporting collective on RCCL itself.  A 1-GPU box cannot run two RCCL ranks (ranks sharing a device are
refused), but a process group of ONE rank on backend "nccl" (= RCCL on ROCm) still goes through RCCL's all_gather /
all_reduce / barrier on the device: the library loads, the communicator initialises, the tensors of
gather_episode_stats take the device path (no host staging as under gloo).  The multi-rank logic (padding, trimming,
rank order) is covered on CPU with gloo (tests/test_distributed.py); the N-GPU run is the driver's."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_episode_stats_gather_through_rccl_single_rank():
    import torch.distributed as dist
    from gym_auv_amd import distributed as D
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    saved = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        n = 4096
        idx = torch.arange(n, dtype=torch.float32, device="cuda:0")
        stats = {"episode_return": idx * 2.0, "episode_length": idx + 100.0, "collision": (idx % 3 == 0).float()}
        full = D.gather_episode_stats(stats)
        assert all(v.is_cuda and v.shape == (n,) for v in full.values())
        assert torch.equal(full["episode_return"], idx * 2.0) and torch.equal(full["episode_length"], idx + 100.0)
        assert D.max_over_ranks(3.5, torch.device("cuda:0")) == 3.5
        D.barrier()
        torch.cuda.synchronize()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

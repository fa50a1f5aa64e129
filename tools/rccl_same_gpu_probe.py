"""Can two RCCL ranks share ONE GPU on this pool?  (If yes, the nccl variants of tests/test_dist_gpu.py can be
rehearsed on a one-GPU box; NCCL normally refuses with "Duplicate GPU detected".)  Prints the outcome, exit code 0
either way."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((4,), float(rank + 1), device="cuda", dtype=torch.float64)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok -> {t.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: RCCL on a shared GPU refused: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29571), nprocs=2, join=True)
    sys.exit(0)

"""Can RCCL bring up TWO ranks on ONE device?  (It refuses duplicate devices by default; this probes whether any switch allows a real multi-rank
RCCL run on a one-GPU box.)   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 scratch/rccl_two_ranks_one_gpu.py"""
import datetime, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=40))
    x = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over {world} ranks on one device -> {x.tolist()}", flush=True)
    outs = [torch.empty(3, device="cuda") for _ in range(world)]
    dist.all_gather(outs, torch.full((3,), float(rank), device="cuda"))
    print(f"rank {rank}: all_gather -> {[o.tolist() for o in outs]}", flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(1)

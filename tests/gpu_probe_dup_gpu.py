import os, sys, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world)
    x=torch.full((4,), float(rank+1), device="cuda:0")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank",rank,"allreduce ok",x.tolist(), flush=True)
except Exception as e:
    print("rank",rank,"FAILED:",str(e)[:300], flush=True)
# IPC probe: share a tensor between the two processes through torch's IPC reductions

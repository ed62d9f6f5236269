import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(1<<20, device="cuda"); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
e=torch.tensor([1.5],dtype=torch.float64,device="cuda"); dist.all_reduce(e, op=dist.ReduceOp.MAX)
print("rccl ok", float(t[0]), float(e)); dist.destroy_process_group()

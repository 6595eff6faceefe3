import os, sys, time
sys.path.insert(0, os.getcwd())
import stringdecomposer_amd
stringdecomposer_amd.prefer_queue_thread_dispatch()  # AMD_DIRECT_DISPATCH=0 unless chosen
print("AMD_DIRECT_DISPATCH =", os.environ.get("AMD_DIRECT_DISPATCH"))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(1024, device="cuda")
t0 = time.time()
for _ in range(20):
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
print("rccl ok: 20 all_reduce + barrier in %.3f s, value %.1f" % (time.time() - t0, float(t[0])))
dist.destroy_process_group()

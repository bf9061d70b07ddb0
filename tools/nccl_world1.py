import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
from mspl_amd import dist as mdist
t = torch.arange(8, device='cuda', dtype=torch.float32)
mdist.all_reduce_mean(t); mdist.barrier()
h = mdist.reduce_histogram(torch.tensor([1, 2, 3, 4, 5], device='cuda'))
print('nccl world 1 ok', t.tolist(), h.tolist(), mdist.world())
lists = mdist.gather_lists(['a', 'b'])
print(lists)
dist.destroy_process_group()

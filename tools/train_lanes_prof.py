"""The graphed uest train step as bench.py runs it (micro-batch lanes), a few replays: for rocprofv3 --kernel-trace (tools/train_lanes_prof.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
step, x, y = bench.train_step_build('cuda:0', 0)
for _ in range(8):
    step(x, y)
torch.cuda.synchronize()

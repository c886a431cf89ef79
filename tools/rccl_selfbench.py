#!/usr/bin/env python3
"""Floor of one ghost exchange on this box: RCCL send-to-self + receive-from-self groups, microseconds per group.
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/rccl_selfbench.py"""
import torch
import torch.distributed as dist

from geometricmultigridpressuresolver_amd.distributed import RcclComm

torch.cuda.set_device(0)
dist.init_process_group("gloo")
comm = RcclComm(device=0)
for floats in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22):
    print(f"{floats * 4 / 1024:10.0f} KiB: {comm.selfbench(floats):8.1f} us per exchange", flush=True)
comm.close()
dist.destroy_process_group()

"""RCCL self-check of the collectives bench.py and dist.gather_events use, runnable on a 1-GPU box:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/nccl_selfcheck.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from spectrogram_midi_amd import dist as adist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
t = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
s = torch.tensor([2.0], dtype=torch.float64, device=dev)
dist.all_reduce(s, op=dist.ReduceOp.SUM)
events = [{"note": 60 + rank, "start": 3, "end": 9, "confidence": 0.9, "velocity": 100, "track": "main",
           "rms_energy": -12.5, "technique": None, "slope": 0.0}]
rows = adist.pack_events(7 + rank, events)
allr = adist.gather_events(rows, dst=0, device=dev)
if rank == 0:
    back = adist.unpack_events(allr)
    print("nccl ok", float(t.item()), float(s.item()), allr.shape, back)
dist.barrier()
dist.destroy_process_group()

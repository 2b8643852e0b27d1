"""Host time of one training step: how long the Python thread needs to ENQUEUE a step (no metric read, nothing waited for),
eager against the recorded step program (include/bgan.h bg_dstep / bg_gstep).  Usage: python tools/host_time.py [arch] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

arch = sys.argv[1] if len(sys.argv) > 1 else "mnist"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = {"celeba64": 256, "celeba128": 128, "mnist": 64}[arch]
from blurred_gan_amd.models import IMAGE_SHAPE
H, W, C = IMAGE_SHAPE[arch]
out = {"arch": arch, "batch": B, "steps": steps}
for mode in ("eager", "replay", "graph"):
    os.environ.pop("BGAN_STEP_GRAPH", None)
    if mode == "graph":
        os.environ["BGAN_STEP_GRAPH"] = "1"
    gan = bench.build_gan(arch, B, 1, 5.0)
    gan.step_replay = mode != "eager"
    gan.sync_metrics = False                      # measuring aid: no device->host read, the host never waits
    reals = torch.rand(B, H, W, C, device="cuda") * 2 - 1
    for _ in range(5):
        gan.train_on_batch(reals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gan.train_on_batch(reals)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[mode] = {"host_ms_per_step": round((t1 - t0) / steps * 1e3, 4), "gpu_drain_ms_per_step": round((t2 - t0) / steps * 1e3, 4),
                 "programs": dict(gan._programs.stats)}
    del gan
print(json.dumps(out))

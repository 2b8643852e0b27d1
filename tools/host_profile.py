"""Host-side profile of one training step (where the Python time goes when the GPU work is tiny: MNIST).
Usage: python tools/host_profile.py [arch] [steps]"""
import cProfile
import pstats
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

arch = sys.argv[1] if len(sys.argv) > 1 else "mnist"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = {"celeba64": 256, "celeba128": 128, "mnist": 64}[arch]
gan = bench.build_gan(arch, B, 1, 5.0)
from blurred_gan_amd.models import IMAGE_SHAPE
H, W, C = IMAGE_SHAPE[arch]
reals = torch.rand(B, H, W, C, device="cuda") * 2 - 1
for _ in range(5):
    gan.train_on_batch(reals)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    gan.train_on_batch(reals)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

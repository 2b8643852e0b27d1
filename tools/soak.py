"""Soak run: N training steps on synthetic data with the product's own RNG; reports loss statistics and checks for NaN/Inf.
Usage: python tools/soak.py [arch] [batch] [steps]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from blurred_gan_amd import callbacks
from blurred_gan_amd.models import IMAGE_SHAPE

arch = sys.argv[1] if len(sys.argv) > 1 else "celeba64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
gan = bench.build_gan(arch, B, 1, 23.5)
gan.sync_metrics = True
H, W, C = IMAGE_SHAPE[arch]
g = torch.Generator(device="cuda").manual_seed(1)
# a fixed "dataset" of smooth images so the critic has something to learn
base = torch.rand(8 * B, 1, 1, C, device="cuda", generator=g) * 2 - 1
yy = torch.linspace(-1, 1, H, device="cuda").view(1, H, 1, 1)
xx = torch.linspace(-1, 1, W, device="cuda").view(1, 1, W, 1)
data = (base * 0.5 + 0.5 * torch.sin(3 * yy * base.abs() + 2 * xx)).clamp(-1, 1).contiguous()
ctl = callbacks.BlurDecayController(total_n_training_examples=steps * B, max_value=23.5)
ctl.set_model(gan)
names = gan.metrics_names
hist = []
for i in range(steps):
    ctl.on_batch_begin(i, {})
    reals = data[(i % 8) * B:(i % 8 + 1) * B]
    out = dict(zip(names, gan.train_on_batch(reals)))
    bad = [k for k, v in out.items() if isinstance(v, float) and not math.isfinite(v)]
    assert not bad, (i, bad, out)
    hist.append(out)
    if i % max(1, steps // 10) == 0 or i == steps - 1:
        print(i, {k: round(v, 4) for k, v in out.items() if k in ("disc_loss", "gen_loss", "gp_term", "real_scores", "fake_scores", "std")})
for net in (gan.generator, gan.discriminator):
    th = net.store.theta
    assert torch.isfinite(th).all(), "non-finite weights"
print("soak ok:", steps, "steps; final std", round(hist[-1]["std"], 4))

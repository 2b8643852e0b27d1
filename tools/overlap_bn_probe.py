"""Does the BatchNorm-backward chain of stage i-1 (HBM-bound) hide under the filter gradient of stage i (matrix-bound) when the two are
issued on two streams?  The generator's backward runs, per stage:  data gradient(i) -> [ BN backward(i-1) | filter gradient(i) ] -> ...
python tools/overlap_bn_probe.py  -> per C2 generator stage: filter gradient alone, BN backward alone, one after the other, side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops

B = int(os.environ.get("PROBE_BATCH", "256"))
side = torch.cuda.Stream()
# (name, filter-gradient geometry of stage i as a conv: B,H,W,Ci,Co,k,s with x = conv input side, and the BN tensor below it: M x C)
STAGES = [
    ("G6 conv32->3  | BN 64x64x32", (64, 64, 32, 3, 1), (64 * 64, 32)),
    ("G5 CT64->32   | BN 32x32x64", (64, 64, 32, 64, 2), (32 * 32, 64)),      # ConvT as the conv it transposes: x side = 64x64x32
    ("G4 CT128->64  | BN 16x16x128", (32, 32, 64, 128, 2), (16 * 16, 128)),
    ("G3 CT256->128 | BN 8x8x256", (16, 16, 128, 256, 2), (8 * 8, 256)),
    ("G2 CT512->256 | BN 4x4x512", (8, 8, 256, 512, 2), (4 * 4, 512)),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, (H, W, Ci, Co, s), (Mpix, C) in STAGES:
    Ho, Wo = -(-H // s), -(-W // s)
    x = torch.rand(B, H, W, Ci, device="cuda")
    dy = torch.rand(B, Ho, Wo, Co, device="cuda")
    dw = torch.empty(5, 5, Ci, Co, device="cuda")
    nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
    ws_w = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    M = B * Mpix
    g = torch.randn(M, C, device="cuda")
    z = torch.randn(M, C, device="cuda")
    dz = torch.empty_like(g)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    mean, inv = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ws_b = torch.empty(ops._lib.load().bg_bn_workspace_bytes(M, C) // 4 + 4, device="cuda")

    def wgrad():
        ops.conv2d_bwd_filter(x, dy, dw, 5, s, 0.0, 1.0, ws_w)

    def bn():
        ops.bn_train_bwd(g, None, z, dz, M, C, gamma, mean, inv, dg, db, ws_b, lrelu_alpha=0.0, beta=beta)

    def serial():
        bn(); wgrad()

    def par():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            wgrad()
            e2 = torch.cuda.Event(); e2.record()
        bn()
        torch.cuda.current_stream().wait_event(e2)

    tw, tb, ts, tp = timeit(wgrad), timeit(bn), timeit(serial), timeit(par)
    print(f"{name:30s} wgrad {tw:7.1f}  BN bwd {tb:7.1f}  serial {ts:7.1f}  two streams {tp:7.1f} us   saved {ts - tp:6.1f} us of the BN's {tb:6.1f}")

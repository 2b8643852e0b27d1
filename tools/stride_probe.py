"""Does the 32-KiB row stride of position-major tiles on 4x4 maps hurt?  Same conv, channel counts that make the per-image
stride a power of two (512) or not (480 / 544): time per MAC."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops

B, H, W, s = 256, 4, 4, 1
for Ci, Co in [(512, 512), (480, 512), (544, 512), (512, 480), (256, 512), (224, 512)]:
    x = torch.rand(B, H, W, Ci, device="cuda") - 0.5
    w = torch.rand(5, 5, Ci, Co, device="cuda") - 0.5
    wT = ops.transpose_last2(w, torch.empty(w.numel(), device="cuda"), 25, Ci, Co)
    y = torch.empty(B, H, W, Co, device="cuda")
    nb = ops.conv2d_splitk_workspace_bytes(False, B, H, W, Ci, Co, 5, s)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    epi = ops.epilogue(ws=ws)
    for _ in range(3):
        ops.conv2d_fwd(x, wT, y, 5, s, epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_fwd(x, wT, y, 5, s, epi)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gf = 2.0 * B * H * W * Ci * Co * 25 / 1e9
    print(f"Ci={Ci:4d} Co={Co:4d}  image stride {H*W*Ci*4/1024:6.1f} KiB  {ms*1e3:7.1f} us  {gf/ms:7.1f} algorithmic TFLOP/s")

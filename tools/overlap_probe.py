"""Do a data-gradient and a filter-gradient launch of the same layer overlap usefully when issued on two streams?
python tools/overlap_probe.py  -> time of (dgrad; wgrad) on one stream against dgrad || wgrad on two, per C2 layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_conv import LAYERS

B = int(os.environ.get("PROBE_BATCH", "256"))
# PROBE_PRIO=1: the data gradient (the backward's critical chain) on a HIGH-priority stream, the filter gradient on a default one:
# does the low-priority kernel fill the under-occupied tail of the other instead of taking its share of every CU?
PRIO = os.environ.get("PROBE_PRIO") == "1"
side = torch.cuda.Stream()
main = torch.cuda.Stream(priority=-1) if PRIO else None
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", "high-priority main:", PRIO)
for name, H, W, Ci, Co, s in LAYERS["celeba64"]:
    if Ci < 16 or Co < 16:
        continue
    Ho, Wo = -(-H // s), -(-W // s)
    x = torch.rand(B, H, W, Ci, device="cuda"); dy = torch.rand(B, Ho, Wo, Co, device="cuda")
    w = torch.rand(5, 5, Ci, Co, device="cuda") / 100
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    nk = ops.conv2d_splitk_workspace_bytes(True, B, H, W, Ci, Co, 5, s) if hasattr(ops, "conv2d_splitk_workspace_bytes") else 0
    epi = None
    if nk:
        kws = torch.empty(nk // 4 + 4, device="cuda")
        from blurred_gan_amd._lib import EPI_NONE
        epi = ops.epilogue(EPI_NONE, ws=kws)
    def serial():
        ops.conv2d_bwd_data(dy, w, dx, 5, s, epi)
        ops.conv2d_bwd_filter(x, dy, dw, 5, s, 0.0, 1.0, ws)
    def par():
        ev = torch.cuda.Event(); ev.record()
        if PRIO:
            with torch.cuda.stream(main):
                main.wait_event(ev)
                ops.conv2d_bwd_data(dy, w, dx, 5, s, epi)
                e1 = torch.cuda.Event(); e1.record()
        else:
            ops.conv2d_bwd_data(dy, w, dx, 5, s, epi)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.conv2d_bwd_filter(x, dy, dw, 5, s, 0.0, 1.0, ws)
            e2 = torch.cuda.Event(); e2.record()
        torch.cuda.current_stream().wait_event(e2)
        if PRIO:
            torch.cuda.current_stream().wait_event(e1)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    ts, tp = timeit(serial), timeit(par)
    print(f"{name:18s} serial {ts:7.1f} us   two streams {tp:7.1f} us   {100 * (1 - tp / ts):5.1f} % saved")

#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv kernels at the C2 (celeba64, B=256) layer shapes: forward, data-gradient and
filter-gradient of every layer, timed with the library's own HIP-event hooks.  Usage: python tools/bench_conv.py [--arch celeba64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from blurred_gan_amd import ops  # noqa: E402

LAYERS = {  # (name, B-mult, H, W, Cin, Cout, stride): conv geometry (H,W,Cin = conv input side)
    "celeba64": [("D1 3->32", 64, 64, 3, 32, 2), ("D2 32->64", 32, 32, 32, 64, 2), ("D3 64->128", 16, 16, 64, 128, 2),
                 ("D4 128->256", 8, 8, 128, 256, 2), ("D5 256->512", 4, 4, 256, 512, 2),
                 # generator ConvT layers expressed as the underlying conv (input side = ConvT output)
                 ("G1 CT512->512 s1", 4, 4, 512, 512, 1), ("G2 CT512->256", 8, 8, 256, 512, 2), ("G3 CT256->128", 16, 16, 128, 256, 2),
                 ("G4 CT128->64", 32, 32, 64, 128, 2), ("G5 CT64->32", 64, 64, 32, 64, 2), ("G6 conv32->3", 64, 64, 32, 3, 1)],
    # demo_mnist.py:48-86: critic Conv 1->64, 64->128 (stride 2); generator ConvT 256->128 (s1, 7x7), 128->64 (s2), 64->1 (s2, tanh)
    "mnist": [("D1 1->64", 28, 28, 1, 64, 2), ("D2 64->128", 14, 14, 64, 128, 2), ("G1 CT256->128 s1", 7, 7, 128, 256, 1),
              ("G2 CT128->64", 14, 14, 64, 128, 2), ("G3 CT64->1", 28, 28, 1, 64, 2)],
    "celeba128": [("D1 3->16", 128, 128, 3, 16, 2), ("D2 16->32", 64, 64, 16, 32, 2), ("D3 32->64", 32, 32, 32, 64, 2),
                  ("D4 64->128", 16, 16, 64, 128, 2), ("D5 128->256", 8, 8, 128, 256, 2), ("D6 256->512", 4, 4, 256, 512, 2),
                  ("G1 CT512->512 s1", 4, 4, 512, 512, 1), ("G2 CT512->256", 8, 8, 256, 512, 2), ("G3 CT256->128", 16, 16, 128, 256, 2),
                  ("G4 CT128->64", 32, 32, 64, 128, 2), ("G5 CT64->32", 64, 64, 32, 64, 2), ("G6 CT32->16", 128, 128, 16, 32, 2),
                  ("G7 conv16->3", 128, 128, 16, 3, 1)],
}


def run(fn, iters):
    fn()
    torch.cuda.synchronize()
    ops.prof_reset()
    ops.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    recs = ops.prof_records(with_useful=True)
    ops.prof_enable(False)
    ops.prof_reset()
    ms = sum(r[1] for r in recs) / iters
    fl = sum(r[2] for r in recs) / iters
    us = sum(r[4] for r in recs) / iters
    per = {}
    for r in recs:
        per[r[0]] = per.get(r[0], 0.0) + r[1] / iters
    names = [f"{k}={v * 1e3:.1f}us" for k, v in sorted(per.items())]
    return ms, fl, us, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="celeba64")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--epi", action="store_true", help="the step's epilogues: bias + LeakyReLU + dropout mask (forward), LeakyReLU gradient x mask (data gradient)")
    a = ap.parse_args()
    B = a.batch
    torch.manual_seed(0)
    # Rates are priced by USEFUL flops (bg_prof_get_useful: only the taps that meet real data -- on 4x4 maps 51 % of the 25 taps land on
    # the SAME zero padding and are skipped, not multiplied): the column that cannot exceed 100.  SURVEY 8d's count (every tap at every
    # output position; the figure roofline.achieved keeps) follows in brackets and CAN read above the peak on small maps.
    print(f"{'layer':<20}{'op':<8}{'ms':>9}{'TFLOP/s':>10}{'%peak':>8}  {'[8d count: TFLOP/s, %]':<24}  kernels")
    tot = {}
    for name, H, W, Ci, Co, s in LAYERS[a.arch]:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        Ho, Wo = -(-H // s), -(-W // s)
        x = torch.rand(B, H, W, Ci, device="cuda") - 0.5
        dy = torch.rand(B, Ho, Wo, Co, device="cuda") - 0.5
        w = torch.rand(5, 5, Ci, Co, device="cuda") - 0.5
        wT = ops.transpose_last2(w, torch.empty(w.numel(), device="cuda"), 25, Ci, Co)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
        nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
        ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
        nf, nd = ops.conv2d_splitk_workspace_bytes(False, B, H, W, Ci, Co, 5, s), ops.conv2d_splitk_workspace_bytes(True, B, H, W, Ci, Co, 5, s)
        wsk = torch.empty(max(nf, nd) // 4 + 4, device="cuda")
        ef, ed = ops.epilogue(ws=wsk if nf else None), ops.epilogue(ws=wsk if nd else None)     # split-K scratch as the engine passes it
        if a.epi:
            bias = torch.rand(Co, device="cuda") - 0.5
            keep_y = (torch.rand(dy.shape, device="cuda") < 0.7).to(torch.uint8)
            keep_x = (torch.rand(x.shape, device="cuda") < 0.7).to(torch.uint8)
            ref_x = torch.rand(x.shape, device="cuda") - 0.5
            ef = ops.epilogue(ops.EPI_BIAS_LRELU, bias=bias, keep=keep_y, scale=1 / 0.7, ws=wsk if nf else None)
            ed = ops.epilogue(ops.EPI_MUL_GRAD, ref=ref_x, keep=keep_x, scale=1 / 0.7, ws=wsk if nd else None)
        for op, fn in (("fwd", lambda: ops.conv2d_fwd(x, wT, y, 5, s, ef)), ("dgrad", lambda: ops.conv2d_bwd_data(dy, w, dx, 5, s, ed)),
                       ("wgrad", lambda: ops.conv2d_bwd_filter(x, dy, dw, 5, s, 0.0, 1.0, ws))):
            ms, fl, us, names = run(fn, a.iters)
            tf = fl / (ms * 1e-3) / 1e12
            tot.setdefault(op, [0.0, 0.0, 0.0])
            tot[op][0] += ms
            tot[op][1] += fl
            tot[op][2] += us
            tu = us / (ms * 1e-3) / 1e12
            print(f"{name:<20}{op:<8}{ms:9.4f}{tu:10.2f}{100 * tu / 157.3:8.1f}  [{tf:7.2f}, {100 * tf / 157.3:5.1f}]{'':<8}  {','.join(names)}")
    for op, (ms, fl, us) in tot.items():
        print(f"{'TOTAL':<20}{op:<8}{ms:9.4f}{us / (ms * 1e-3) / 1e12:10.2f}{100 * us / (ms * 1e-3) / 1e12 / 157.3:8.1f}  [{fl / (ms * 1e-3) / 1e12:7.2f}, {100 * fl / (ms * 1e-3) / 1e12 / 157.3:5.1f}]")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- images/sec of one full WGAN-GP + blur training step (D-step with gradient penalty + G-step,
Adam on both), BASELINE.json's metric, on synthetic CelebA-shaped batches.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

N=1 workload = BASELINE.json configs[1]: CelebA 64x64 RGB, batch 256, blur sigma schedule on
(BlurDecayController(max_value=5) -> 31 taps), 64-arch (build-side definition).
N>1 = configs[2]: the same per-GPU work, batch 256 x N sharded data-parallel, RCCL SUM all-reduce of the
critic and generator gradients ("weak" scaling).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; read at the first HIP call (see blurred_gan_amd/__init__.py)
import torch  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0
CONV_GFLOP_PER_IMAGE = {"celeba64": 3.908, "celeba128": 4.769, "mnist": 0.659}


def same_live_taps(n, k=5, s=2):
    """(output position, tap) pairs of a 1-D k-tap stride-s TF-SAME window over n samples whose source sample exists (is not zero
    padding): out = ceil(n/s), pad_total = max((out-1)s + k - n, 0), pad_before = pad_total // 2."""
    o = -(-n // s)
    before = max((o - 1) * s + k - n, 0) // 2
    return sum(1 for i in range(o) for t in range(k) if 0 <= i * s + t - before < n)


def conv_layers(arch):
    """The conv / transposed-conv / dense layers of both stacks as (net, kind, H, W, Cin, Cout, stride): conv INPUT-side geometry
    (a Conv2DTranspose is listed as the convolution it is the data gradient of, i.e. H, W = its output side), dense as
    (net, "dense", 1, 1, in, out, 1).  demo_celeba.py:51-124, demo_mnist.py:48-86."""
    from blurred_gan_amd import models
    base, ch, convt, last = models._G[arch]
    out = [("G", "dense", 1, 1, models.LATENT[arch], base * base * ch, 1)]
    hw, c = base, ch
    for filters, stride, _ in convt:
        hw *= stride
        out.append(("G", "convt", hw, hw, filters, c, stride))
        c = filters
    if last is not None:
        out.append(("G", "conv", hw, hw, c, last, 1))
    H, W, C = models.IMAGE_SHAPE[arch]
    hw, c = H, C
    for f in models._D[arch]:
        out.append(("D", "conv", hw, hw, c, f, 2))
        hw, c = -(-hw // 2), f
    out.append(("D", "dense", 1, 1, hw * hw * c, 1, 1))
    return out


def conv_macs_per_image(arch, useful=False):
    """Forward-MAC equivalents of one train_on_batch per image: generator 4 passes (D-step forward, G-step forward,
    data gradient, filter gradient; its Dense has no data gradient), critic 12 (D-step: 3 forwards, 2 x (filter + data gradient),
    the penalty's data-gradient chain, its linearised forward and filter gradient; G-step: forward + data gradient), the first
    critic conv 10 (no data gradient to the image on the [fakes; reals] rows).  useful=False: F_l = Ho*Wo*Cin*Cout*25 (every tap at
    every output position); useful=True: only the (output pixel, tap) pairs whose source pixel exists --
    the taps on TF's zero padding multiply nothing."""
    total, first_d = 0, True
    for net, kind, H, W, Ci, Co, s in conv_layers(arch):
        if kind == "dense":
            f = Ci * Co
            total += (3 if net == "G" else 12) * f
            continue
        if useful:
            f = same_live_taps(H, 5, s) * same_live_taps(W, 5, s) * Ci * Co
        else:
            f = (-(-H // s)) * (-(-W // s)) * Ci * Co * 25
        if net == "G":
            total += 4 * f
        else:
            total += (10 if first_d else 12) * f
            first_d = False
    return total


def build_gan(arch, B, world, sigma):
    import blurred_gan_amd as bg
    from blurred_gan_amd import models, dist
    bg.set_seed(123123)                                   # demo_celeba.py:132; every rank starts from identical weights
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=sigma, batch_size=B, global_batch_size=B * world)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir="/tmp/bg_bench_logs"))
    gan._rng_seed = 123123 + dist.rank()
    return gan


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; the contract is ONE JSON line there.  File
    descriptor 1 points at stderr while the process group (and its first collective, which is what creates the
    communicator) comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(arch, batch, sigma, seconds_budget=30.0):
    """The oracle's torch-CPU port of the same step at the SAME batch as the GPU line, timed on this box's host cores
    (reported, not a target).  One small-batch step warms oneDNN up, then whole steps at the quoted batch are timed until
    the budget is spent (at ~10 images/s one batch-256 step is the whole sample)."""
    from oracle.torch_ref import TorchTrainer
    from oracle import models as OM
    torch.manual_seed(0)
    H, W, C = OM.image_shape(arch)
    gen = torch.Generator().manual_seed(0)
    warm = TorchTrainer(arch, seed=0, std=sigma, hp=dict(global_batch_size=8))
    warm.train_on_batch(torch.rand(8, H, W, C, generator=gen) * 2 - 1, warm.draw(8, gen))
    tr = TorchTrainer(arch, seed=0, std=sigma, hp=dict(global_batch_size=batch))
    reals = torch.rand(batch, H, W, C, generator=gen) * 2 - 1
    n, t0 = 0, time.time()
    while True:
        tr.train_on_batch(reals, tr.draw(batch, gen))
        n += 1
        if time.time() - t0 > seconds_budget or n >= 20:
            break
    dt = time.time() - t0
    return {"value": round(batch * n / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"{n} step(s) of batch {batch} of the same {arch} step in {dt:.1f} s (oracle/torch_ref.py, torch-CPU fp32 on "
                      f"{torch.get_num_threads()} threads, autograd double backward for the penalty)"}


def cpu_baseline_blur(B, H, W, C, sigma, apps, seconds_budget=15.0):
    """The blur alone on the host cores: oracle/torch_ref.blur, i.e. the two SAME depthwise convolutions of
    gaussian_blur.py:116-130 as torch-CPU grouped convs (the closest stand-in for TF's depthwise_conv2d on the CPU), on a
    BOUNDED sample of the same workload: whole applications over the same 64 images until the budget is spent.
    Reported in the line's own unit (algorithmic GB/s, 8*H*W*C bytes per image per application), not a target."""
    from oracle import torch_ref
    gen = torch.Generator().manual_seed(0)
    sb = B
    x = torch.rand(sb, H, W, C, generator=gen) * 2 - 1
    with torch.no_grad():
        torch_ref.blur(x[:1], sigma)                      # warm-up (oneDNN primitive creation)
        n, t0 = 0, time.time()
        while True:
            torch_ref.blur(x, sigma)
            n += 1
            if time.time() - t0 > seconds_budget or n >= 20 * apps:
                break
    dt = time.time() - t0
    return {"value": round(8.0 * sb * H * W * C * n / dt / 1e9, 4), "unit": "GB/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"{n} application(s) of the blur on {sb} of the {B} images ({sb}x{H}x{W}x{C}) in {dt:.1f} s (oracle/torch_ref.blur: two "
                      f"depthwise torch-CPU convs, fp32, {torch.get_num_threads()} threads)"}


TRAFFIC_FILE = "profiles/hbm_traffic.json"      # refreshed by tools/pmc_step.sh + tools/traffic_update.py; a per-round copy is kept as rNN_hbm_traffic.json


def hbm_traffic(kernel, arch, batch):
    """HBM-side bytes per launch of ``kernel`` from the committed PMC passes (tools/pmc_step.sh / tools/pmc_blur.sh over this
    very workload: separate --pmc runs as MI355X_MICROARCH.md prescribes; the counters cannot be read live).  None when no
    profile exists for the workload OR the kernel sources have changed since it was taken (the entry carries their hash)."""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    try:
        with open(os.path.join(root, TRAFFIC_FILE)) as f:
            d = json.load(f)
    except OSError:
        return None
    for e in d.get("entries", []):
        if e.get("arch") != arch or e.get("batch") != batch or kernel not in e.get("kernels", {}):
            continue
        h = hashlib.sha1()
        try:
            for src in e["sources"]:
                with open(os.path.join(root, src), "rb") as f:
                    h.update(f.read())
        except OSError:
            return None
        if h.hexdigest()[:16] != e.get("sources_sha"):
            return None                     # stale: the kernel was edited after the profile was taken
        return round(e["kernels"][kernel]["hbm_bytes_per_launch"])
    return None


def bench_blur(args):
    """BASELINE.json configs[4] (C5), one rank's share: 64 synthetic 256x256x3 images, a step = the 7 blur applications of one
    training step (D-step forward x3, blur^T, GP second order; G-step forward, blur^T) at the given sigma
    (5 -> 31 taps, the bandwidth-bound control; 23.5 -> 143 and 42.34 -> 255 taps, the callback default / the largest
    reasonable sigma, gaussian_blur.py:15-18).  value = algorithmic GB/s, 8*H*W*C bytes per image per application
    (SURVEY.md 8d), over the whole timed region; roofline = the same figure over the kernels' own HIP-event durations."""
    from blurred_gan_amd import dist, ops
    with stdout_to_stderr():
        world = dist.init_from_env()
        torch.cuda.set_device(dist.local_rank())
        dist.barrier()
    B, H, W, C = args.batch or 64, 256, 256, 3
    apps = 7
    g = torch.Generator(device="cuda").manual_seed(123123 + dist.rank())
    x = torch.rand(B, H, W, C, device="cuda", generator=g) * 2 - 1
    y = torch.empty_like(x)
    ks, se, nt = ops.blur_policy(args.sigma, H, W)
    taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
    nb = ops.blur_workspace_bytes(B, H, W, C, nt)
    tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None

    def step():
        for i in range(apps):                  # ping-pong so that no application reads what the cache still holds of its own output
            ops.blur_nhwc(x if i % 2 == 0 else y, y if i % 2 == 0 else x, taps, nt, tmp)

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    dt = dist.max_over_ranks(time.perf_counter() - t0)
    dp_evidence = dist.evidence()
    alg_bytes = 8.0 * B * H * W * C                              # per application
    value = alg_bytes * apps * args.steps * world / dt / 1e9
    ops.prof_reset()
    ops.prof_enable(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    recs = ops.prof_records(with_exec=True)
    ops.prof_enable(False)
    ops.prof_reset()
    kern = {}
    exec_fl = 0.0
    for name, ms, fl, by, ex in recs:
        k = kern.setdefault(name, [0, 0.0])
        k[0] += 1; k[1] += ms
        exec_fl += ex if name.startswith(("blur_band", "blur_panel")) else 0.0
    total_ms = sum(k[1] for k in kern.values())
    per_app_ms = total_ms / (3 * apps)
    ach = alg_bytes / (per_app_ms * 1e-3) / 1e9
    # at wide kernels the banded Toeplitz product on the fp32 matrix pipe is the binding roof, not HBM: report both
    dom = max(kern, key=lambda n: kern[n][1])
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
            "traffic": hbm_traffic(f"blur{nt}", "blur256", B), "algorithmic_bytes_per_launch": round(alg_bytes),
            "launches_per_application": len(recs) // (3 * apps), "avg_application_ms": round(per_app_ms, 5),
            "kernels_ms_per_application": {n: round(k[1] / (3 * apps), 5) for n, k in kern.items()}}
    if exec_fl > 0:
        # above 65 taps the banded Toeplitz products (both passes of the 32-row panel kernel in one launch; the two transposing
        # band passes for other channel counts) run on the fp32 matrix pipe and THAT is the binding roof: the flops they issue
        # (every 32x32x2 MFMA of the bands, image-clipped) over their own durations.
        # The HBM view of the same launches stays in the object as "hbm".
        tf = exec_fl / (total_ms * 1e-3) / 1e12
        hbm_view = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
        roof.update({"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / PEAK_MFMA_F32_TFLOPS, 4), "issued_gflop_per_application": round(exec_fl / (3 * apps) / 1e9, 3),
                     "time_at_peak_ms_per_application": round(exec_fl / (3 * apps) / (PEAK_MFMA_F32_TFLOPS * 1e12) * 1e3, 5),
                     "hbm": hbm_view})
    if dist.rank() == 0:
        out = {"metric": "blur GB/s (8*H*W*C bytes per image per application, 7 applications per step)", "value": round(value, 1),
               "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"blur256: {B}x{H}x{W}x{C} per GPU, sigma {args.sigma} ({nt} taps), {apps} blur applications per step",
                          "global_batch": B * world, "parallelism": f"dp{world}"},
               "roofline": roof}
        if dp_evidence:
            out["dp"] = dp_evidence
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_blur(B, H, W, C, args.sigma, apps)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="celeba64", choices=["celeba64", "celeba128", "mnist", "blur256"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 256; 128 for celeba128; 64 for mnist)")
    ap.add_argument("--sigma", type=float, default=None, help="blur sigma (default: the demos' schedule start: 5 for CelebA, 0.05 for MNIST)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the GLOBAL batch stays 2048 (BASELINE.json configs[2]) and is "
                    "split over the ranks, instead of 256 images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=None, help="batch of the CPU baseline (default: the GPU line's batch)")
    ap.add_argument("--kernels-out", default=None, help="write the full per-kernel table (and one step's launch sequence) to this JSON file")
    args = ap.parse_args()
    if args.sigma is None:
        args.sigma = 0.05 if args.arch == "mnist" else 5.0      # demo_mnist.py:199 (initial_blur_std), demo_celeba.py:226

    if args.gpus > 1 and "RANK" not in os.environ:
        # started bare with --gpus N: become the launcher (one rank per GPU over RCCL) -- as a child process, before anything
        # here has touched the GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.run(cmd).returncode)

    if args.arch == "blur256":
        return bench_blur(args)
    from blurred_gan_amd import dist, ops, callbacks
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    with stdout_to_stderr():
        world = dist.init_from_env()
        torch.cuda.set_device(dist.local_rank())
        dist.barrier()                                    # first collective: the communicator (and its banner) is created here
        if dist.is_initialized():
            dist.max_over_ranks(0.0)
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    B = args.batch or {"celeba64": 256, "celeba128": 128, "mnist": 64}[args.arch]
    if args.strong:
        gb = 2048 if args.batch is None else args.batch
        assert gb % world == 0, f"global batch {gb} does not divide over {world} ranks"
        B = gb // world
    gan = build_gan(args.arch, B, world, args.sigma)
    from blurred_gan_amd.models import IMAGE_SHAPE
    H, W, C = IMAGE_SHAPE[args.arch]
    g = torch.Generator(device="cuda").manual_seed(123123 + dist.rank())
    reals = torch.rand(B, H, W, C, device="cuda", generator=g) * 2 - 1
    total_examples = (60000 if args.arch == "mnist" else 202599) * 10          # dataset x 10 epochs (demo_mnist.py:99,199; demo_celeba.py:135,226)
    ctl = callbacks.BlurDecayController(total_n_training_examples=total_examples, max_value=args.sigma)
    ctl.set_model(gan)

    def step():
        ctl.on_batch_begin(0, {})
        gan.train_on_batch(reals)

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    # one event per step on the launch stream (a marker packet per 100+ kernel launches): the spread of the per-step times says
    # whether `value` -- the mean over a region of a few hundred milliseconds -- carries a host hiccup of the shared box
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = dist.max_over_ranks(dt)
    value = B * world * args.steps / dt
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    step_ms = {"p10": round(per_step[len(per_step) // 10], 4), "p50": round(per_step[len(per_step) // 2], 4),
               "p90": round(per_step[(len(per_step) * 9) // 10], 4), "max": round(per_step[-1], 4)} if per_step else None
    dp_evidence = dist.evidence()          # every rank takes part (an all-reduce and an all-gather through the step's communicator)

    # ---- per-kernel HIP-event timing on the launch stream (separate, un-timed steps)
    roof = None
    kern = {}
    if not args.no_profile:
        ops.prof_reset()
        ops.prof_enable(True)
        nprof = 2
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        recs = ops.prof_records(with_exec=True, with_useful=True)
        ops.prof_enable(False)
        ops.prof_reset()
        for name, ms, fl, by, ex, us in recs:
            k = kern.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            k[0] += 1; k[1] += ms; k[2] += fl; k[3] += by; k[4] += ex; k[5] += us
        # every conv kernel family runs on the MFMA pipe (gather-GEMM, row-staged, filter-gradient) except the direct fallback;
        # their split-K / slab reduce passes are counted with them
        mfma = {n: k for n, k in kern.items() if n.startswith("conv_") and n != "conv_wgrad_direct"}
        if mfma:
            dom = max(mfma, key=lambda n: mfma[n][1])
            cnt, ms, fl, by, ex, us = mfma[dom]
            ach = fl / (ms * 1e-3) / 1e12
            ach_x = ex / (ms * 1e-3) / 1e12
            ach_u = us / (ms * 1e-3) / 1e12
            traffic = hbm_traffic(dom, args.arch, B)
            all_ms = sum(k[1] for k in mfma.values())
            all_fl = sum(k[2] for k in mfma.values())
            all_us = sum(k[5] for k in mfma.values())
            # the step's convolution work per image, both counts, from the closed forms (the per-launch records of the library
            # must add up to them: checked below and in tests/test_bench_cpu.py)
            gf_alg = 2e-9 * conv_macs_per_image(args.arch)
            gf_use = 2e-9 * conv_macs_per_image(args.arch, useful=True)
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4),
                    # the same launches priced by the MFMA flops they ISSUE: whole tiles (padding rows / columns in), minus the
                    # zero-padding taps that position-major tiles skip -- how busy the matrix pipe is, not how useful
                    "executed": {"achieved": round(ach_x, 3), "frac": round(ach_x / PEAK_MFMA_F32_TFLOPS, 4)},
                    # ... and by the flops that multiply REAL data: SURVEY 8d's count above charges all 25 taps at every output
                    # position, but on a 4x4 map 51 % of them (8x8: 28 %) land on the SAME zero padding and are work nobody has to
                    # do (bg_prof_get_useful: exact (output pixel, tap) pairs inside the image, no tile padding).  This is the
                    # figure that cannot read above 1; "frac" keeps SURVEY 8d's definition.
                    "useful": {"achieved": round(ach_u, 3), "frac": round(ach_u / PEAK_MFMA_F32_TFLOPS, 4),
                               "share_of_algorithmic": round(us / fl, 4) if fl else None},
                    "traffic": traffic,
                    "traffic_unit": f"HBM-side bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, {TRAFFIC_FILE})" if traffic else None,
                    "algorithmic_bytes_per_launch": round(by / cnt) if by else None,
                    "launches_per_step": cnt // nprof, "avg_launch_ms": round(ms / cnt, 5),
                    "all_mfma_kernels": {"achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 3),
                                         "frac": round(all_fl / (all_ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, 4),
                                         "useful_frac": round(all_us / (all_ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, 4),
                                         "ms_per_step": round(all_ms / nprof, 4)},
                    "step_conv_frac": round(CONV_GFLOP_PER_IMAGE[args.arch] * value / world / 1e3 / PEAK_MFMA_F32_TFLOPS, 4),
                    "step_conv_frac_useful": round(gf_use * value / world / 1e3 / PEAK_MFMA_F32_TFLOPS, 4),
                    "conv_gflop_per_image": {"algorithmic": round(gf_alg, 4), "useful": round(gf_use, 4),
                                             # what the library's per-launch records of one step add up to (conv launches only;
                                             # the closed forms also hold the Dense layers, ~0.1 % of the total)
                                             "recorded_algorithmic": round(all_fl / nprof / B / 1e9, 4),
                                             "recorded_useful": round(all_us / nprof / B / 1e9, 4)}}

    if dist.rank() == 0:
        out = {"metric": "images/sec (G+D+GP step)", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{args.arch} {H}x{W}x{C} batch {B}/GPU, blur sigma {args.sigma} "
                                      f"({ops.blur_policy(float(gan.std), H, W)[2]} taps), D-step+GP+G-step+Adam",
                          "global_batch": B * world, "parallelism": f"dp{world}"}}
        if step_ms:
            out["step_ms"] = step_ms       # rank 0's per-step times over the timed region (GPU events): spread, not the metric
        if dp_evidence:
            # the collective layer's own account of the group: peers counted by an all-reduce through the communicator the
            # gradients use, every rank's physical card -- so that an N-GPU line shows N ranks on N distinct devices over RCCL
            out["dp"] = dp_evidence
        if roof is not None:
            out["roofline"] = roof
        if kern:
            top = sorted(kern.items(), key=lambda kv: -kv[1][1])[:12]
            out["kernels_ms_per_step"] = {n: round(k[1] / 2, 4) for n, k in top}
        if kern and args.kernels_out:
            table = [{"kernel": n, "launches_per_step": k[0] / 2, "ms_per_step": round(k[1] / 2, 5),
                      "tflops": round(k[2] / (k[1] * 1e-3) / 1e12, 2) if k[2] else None} for n, k in sorted(kern.items(), key=lambda kv: -kv[1][1])]
            seq = [{"kernel": n, "us": round(ms * 1e3, 2), "gflop": round(fl / 1e9, 3), "useful_gflop": round(us / 1e9, 3), "alg_mb": round(by / 1e6, 3)}
                   for n, ms, fl, by, _, us in recs[:len(recs) // 2]]
            with open(args.kernels_out, "w") as f:
                json.dump({"ms_per_step_sum": round(sum(k[1] for k in kern.values()) / 2, 4), "kernels": table, "sequence": seq}, f, indent=1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.arch, args.cpu_sample_batch or B, args.sigma)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.shutdown()


if __name__ == "__main__":
    main()

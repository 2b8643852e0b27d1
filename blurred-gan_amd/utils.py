"""Host utilities mirroring reference utils.py (JSON config round trip, result dirs, image helpers) and the
``simple_parsing.ParseableFromCommandLine`` mixin the reference imports (wgan.py:15) -- re-created with
stdlib argparse + dataclasses because simple_parsing is not a dependency here."""
from __future__ import annotations

import argparse
import dataclasses
import glob
import json
import os
from typing import Dict


def create_result_subdir(result_dir: str, run_name: str) -> str:
    """utils.py:14-24."""
    paths = glob.glob(os.path.join(result_dir, f"*-{run_name}"))
    run_ids = [int(os.path.basename(p).split("-")[0]) for p in paths]
    run_id = max(run_ids, default=0) + 1
    path = os.path.join(result_dir, f"{run_id:02d}-{run_name}")
    print(f"Creating result subdir at '{path}'")
    os.makedirs(path)
    return path


def normalize_images(images):
    """utils.py:50-52: [-1, 1] -> [0, 1]."""
    return (images + 1) / 2


def run_id(path_string: str) -> int:
    """utils.py:27-28: ``results/07-mnist/model_12.hdf5`` -> 7."""
    return int(path_string.split("/")[-2].split("-")[0])


def epoch(path_string: str) -> int:
    """utils.py:31-32: ``.../model_12.hdf5`` -> 12."""
    return int(path_string.split("/")[-1].split("_")[1].split(".")[0])


def locate_model_file(result_dir: str, run_name: str, suffix="hdf5") -> str:
    """utils.py:35-47: newest ``model_<epoch>.<suffix>`` of the latest run of ``run_name``; FileNotFoundError when none."""
    paths = glob.glob(os.path.join(result_dir, f"*-{run_name}/model_*.{suffix}"))
    if not paths:
        raise FileNotFoundError
    latest = max(run_id(p) for p in paths)
    return max((p for p in paths if run_id(p) == latest), key=epoch)


def samples_grid(samples):
    """utils.py:73-88: the first 64 samples as an 8x8 grid.  The reference returns a matplotlib figure for TensorBoard;
    here the grid is the image itself, [8H, 8W, C] (or [8H, 8W] for one channel), same dtype as the samples."""
    import numpy as np
    s = samples.detach().cpu().numpy() if hasattr(samples, "detach") else np.asarray(samples)
    if s.shape[0] < 64:
        raise ValueError(f"samples_grid needs 64 samples, got {s.shape[0]}")
    n, h, w, c = s.shape
    grid = s[:64].reshape(8, 8, h, w, c).transpose(0, 2, 1, 3, 4).reshape(8 * h, 8 * w, c)
    return grid[..., 0] if c == 1 else grid


def plot_to_image(figure):
    """utils.py:55-70: an image (here: the array samples_grid returns, values in [0, 1]) -> uint8 RGBA with a batch
    dimension, [1, H, W, 4], the shape the reference hands to ``tf.summary.image``."""
    import numpy as np
    g = np.asarray(figure)
    if g.ndim == 2:
        g = g[..., None]
    if g.shape[-1] == 1:
        g = np.repeat(g, 3, axis=-1)
    if g.dtype != np.uint8:
        g = (np.clip(g, 0, 1) * 255).astype(np.uint8)
    if g.shape[-1] == 3:
        g = np.concatenate([g, np.full(g.shape[:-1] + (1,), 255, np.uint8)], axis=-1)
    return g[None]


def NHWC_to_NCHW(image):
    """utils.py:91-92 (torch tensor or ndarray)."""
    return image.permute(0, 3, 1, 2) if hasattr(image, "permute") else image.transpose(0, 3, 1, 2)


def NCHW_to_NHWC(image):
    """utils.py:95-96."""
    return image.permute(0, 2, 3, 1) if hasattr(image, "permute") else image.transpose(0, 2, 3, 1)


def to_dataset(t):
    """utils.py:99-103: a dataset (anything iterable that is not an array/tensor) passes through; an array or tensor becomes
    the sequence of its slices along axis 0, as ``tf.data.Dataset.from_tensor_slices`` would give."""
    if hasattr(t, "shape") and hasattr(t, "__getitem__"):
        return _Slices(t)
    return t


class _Slices:
    def __init__(self, t):
        self.t = t

    def __len__(self):
        return int(self.t.shape[0])

    def __iter__(self):
        return (self.t[i] for i in range(len(self)))


def read_json(file_path: str) -> Dict:
    with open(file_path, "r") as f:
        return json.load(f)


class JsonSerializable:
    """utils.py:116-135."""

    def asdict(self):
        d = dataclasses.asdict(self)
        out = {}
        for k, v in d.items():
            if hasattr(v, "numpy") or hasattr(v, "item"):
                v = float(v.item() if hasattr(v, "item") else v.numpy())
            out[k] = v
        return out

    def save_json(self, file_path: str) -> None:
        with open(file_path, "w") as f:
            json.dump(self.asdict(), f, indent=1)

    @classmethod
    def from_json(cls, file_path: str):
        return cls(**read_json(file_path))


class ParseableFromCommandLine:
    """The two classmethods the reference demos call (demo_mnist.py:104-111): one ``--field`` flag per
    dataclass field, defaults from the dataclass, help from the field name."""

    @classmethod
    def add_arguments(cls, parser: argparse.ArgumentParser):
        group = parser.add_argument_group(cls.__qualname__, (cls.__doc__ or "").strip())
        for f in dataclasses.fields(cls):
            default = f.default if f.default is not dataclasses.MISSING else None
            typ = f.type if isinstance(f.type, type) else {"int": int, "float": float, "str": str, "bool": bool}.get(str(f.type), str)
            if typ is bool:
                group.add_argument(f"--{f.name}", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=default)
            else:
                try:
                    group.add_argument(f"--{f.name}", type=typ, default=default)
                except argparse.ArgumentError:
                    pass   # a parent dataclass already registered this flag

    @classmethod
    def from_args(cls, args: argparse.Namespace):
        kw = {f.name: getattr(args, f.name) for f in dataclasses.fields(cls) if hasattr(args, f.name)}
        return cls(**kw)


@dataclasses.dataclass
class HyperParams(JsonSerializable):
    """utils.py:139-156: base for hyper-parameter dataclasses that travel with checkpoints.  The reference turns every number
    into a tf.constant so that tf.train.Checkpoint tracks it; here the values stay plain Python numbers and
    ``checkpoint.Checkpoint`` stores ``asdict()``."""

    def __repr__(self):
        return self.asdict().__repr__()

    def __str__(self):
        return str(self.asdict())

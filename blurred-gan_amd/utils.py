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

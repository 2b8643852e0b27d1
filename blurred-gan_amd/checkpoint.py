"""Checkpoint / resume: stands in for ``tf.train.Checkpoint(gan=gan)`` + ``CheckpointManager``
(demo_mnist.py:145-163, callbacks.py:239-246).  One ``.npz`` per checkpoint holding both networks' variables
(weights + BN moving statistics), both Adams' slots and step counts, ``n_img``, ``n_batches``, ``blur.std`` and the
positions of the step's random streams (latents / alpha / dropout), so a resumed run continues the uninterrupted one.

Like ``tf.train.CheckpointManager`` the manager orders checkpoints by SAVE ORDER, not by the number in the file name
(``SaveModelCallback`` numbers files with a counter that restarts at every ``fit``, callbacks.py:245): the order lives in
a small ``checkpoint`` index file next to the ``.npz`` files, as TF keeps it."""
from __future__ import annotations

import json
import os
import re

import numpy as np
import torch

INDEX = "checkpoint"


_CKPT_NAME = re.compile(r"^ckpt-\d+\.npz$")


class CheckpointManager:
    def __init__(self, gan, directory, max_to_keep=5, keep_checkpoint_every_n_hours=None):
        self.gan, self.directory = gan, directory
        self.max_to_keep = None if not max_to_keep else int(max_to_keep)        # None / 0: keep everything

    # ---- save-order index
    def _index_path(self):
        return os.path.join(self.directory, INDEX)

    def _read_index(self):
        """File names in save order (oldest first).  A directory without an index (written by hand, or by the first
        version of this module) is ordered by modification time."""
        try:
            with open(self._index_path()) as f:
                names = [n for n in json.load(f)["all_model_checkpoint_paths"]]
        except (OSError, ValueError, KeyError):
            names = None
        if names is None:
            if not os.path.isdir(self.directory):
                return []
            names = [n for n in os.listdir(self.directory) if _CKPT_NAME.match(n)]      # never the ckpt-N.npz.tmp.npz of a crashed save
            names.sort(key=lambda n: os.path.getmtime(os.path.join(self.directory, n)))
        return [n for n in names if os.path.exists(os.path.join(self.directory, n))]

    def _write_index(self, names):
        tmp = self._index_path() + ".tmp"
        with open(tmp, "w") as f:
            json.dump({"model_checkpoint_path": names[-1] if names else None, "all_model_checkpoint_paths": names}, f)
        os.replace(tmp, self._index_path())

    @property
    def checkpoints(self):
        return [os.path.join(self.directory, n) for n in self._read_index()]

    @property
    def latest_checkpoint(self):
        ps = self.checkpoints
        return ps[-1] if ps else None

    # ---- save / restore
    def state_dict(self):
        g = self.gan
        d = {"n_img": np.int64(int(g.n_img)), "n_batches": np.int64(int(g.n_batches)),
             "rng_seed": np.int64(int(g._rng_seed)), "rng_off": np.int64(int(g._rng_off))}
        if hasattr(g, "blur"):
            d["std"] = np.float32(float(g.std))
        for tag, model in (("g", g.generator), ("d", g.discriminator)):
            st = model.store
            st.ensure_opt_state()
            d[f"{tag}_theta"] = st.theta.cpu().numpy()
            d[f"{tag}_state"] = st.state.cpu().numpy()
            d[f"{tag}_m"] = st.m.cpu().numpy()
            d[f"{tag}_v"] = st.v.cpu().numpy()
            d[f"{tag}_iterations"] = np.int64(model.optimizer.iterations)
            d[f"{tag}_rng_offset"] = np.int64(int(model.net().rng_offset))        # dropout-mask stream of this network
        return d

    def save(self, checkpoint_number=None):
        g = self.gan
        os.makedirs(self.directory, exist_ok=True)
        n = int(g.n_img) if checkpoint_number is None else int(checkpoint_number)
        name = f"ckpt-{n}.npz"
        path = os.path.join(self.directory, name)
        tmp = path + ".tmp.npz"
        for stale in os.listdir(self.directory):                     # temp files a crashed save left behind
            if stale.endswith(".tmp.npz") and stale.startswith("ckpt-"):
                try:
                    os.remove(os.path.join(self.directory, stale))
                except OSError:
                    pass
        np.savez(tmp, **self.state_dict())
        os.replace(tmp, path)                   # a crash mid-write never leaves a truncated "latest" checkpoint
        names = [x for x in self._read_index() if x != name] + [name]
        if self.max_to_keep is not None:
            for old in names[:-self.max_to_keep]:
                try:
                    os.remove(os.path.join(self.directory, old))
                except OSError:
                    pass
            names = names[-self.max_to_keep:]
        self._write_index(names)
        return path

    def restore(self, path):
        g = self.gan
        d = np.load(path)
        g.n_img.assign(int(d["n_img"]))
        g.n_batches.assign(int(d["n_batches"]))
        if "rng_off" in d.files:
            g._rng_seed, g._rng_off = int(d["rng_seed"]), int(d["rng_off"])
        if "std" in d.files and hasattr(g, "blur"):
            g.std.assign(float(d["std"]))
        for tag, model in (("g", g.generator), ("d", g.discriminator)):
            st = model.store
            st.ensure_opt_state()
            for name, buf in (("theta", st.theta), ("state", st.state), ("m", st.m), ("v", st.v)):
                buf.copy_(torch.from_numpy(d[f"{tag}_{name}"]))
            model.optimizer.iterations = int(d[f"{tag}_iterations"])
            if f"{tag}_rng_offset" in d.files:
                model.net().rng_offset = int(d[f"{tag}_rng_offset"])
            st.tr_dirty = True
        return path

"""Checkpoint / resume: stands in for ``tf.train.Checkpoint(gan=gan)`` + ``CheckpointManager``
(demo_mnist.py:145-163, callbacks.py:239-246).  One ``.npz`` per checkpoint holding both networks' variables
(weights + BN moving statistics), both Adams' slots and step counts, ``n_img``, ``n_batches`` and ``blur.std``."""
from __future__ import annotations

import glob
import os
import re

import numpy as np
import torch


class CheckpointManager:
    def __init__(self, gan, directory, max_to_keep=5, keep_checkpoint_every_n_hours=None):
        self.gan, self.directory, self.max_to_keep = gan, directory, max_to_keep

    def _paths(self):
        ps = glob.glob(os.path.join(self.directory, "ckpt-*.npz"))
        return sorted(ps, key=lambda p: int(re.findall(r"ckpt-(\d+)\.npz", p)[0]))

    @property
    def latest_checkpoint(self):
        ps = self._paths()
        return ps[-1] if ps else None

    def save(self, checkpoint_number=None):
        g = self.gan
        os.makedirs(self.directory, exist_ok=True)
        n = int(g.n_img) if checkpoint_number is None else int(checkpoint_number)
        d = {"n_img": np.int64(int(g.n_img)), "n_batches": np.int64(int(g.n_batches))}
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
        path = os.path.join(self.directory, f"ckpt-{n}.npz")
        np.savez(path, **d)
        for old in self._paths()[:-self.max_to_keep]:
            os.remove(old)
        return path

    def restore(self, path):
        g = self.gan
        d = np.load(path)
        g.n_img.assign(int(d["n_img"]))
        g.n_batches.assign(int(d["n_batches"]))
        if "std" in d.files and hasattr(g, "blur"):
            g.std.assign(float(d["std"]))
        for tag, model in (("g", g.generator), ("d", g.discriminator)):
            st = model.store
            st.ensure_opt_state()
            for name, buf in (("theta", st.theta), ("state", st.state), ("m", st.m), ("v", st.v)):
                buf.copy_(torch.from_numpy(d[f"{tag}_{name}"]))
            model.optimizer.iterations = int(d[f"{tag}_iterations"])
            st.tr_dirty = True
        return path

"""Counterpart of reference demo_celeba.py (128x128 CelebA stack, demo_celeba.py:51-124), on the HIP kernels.
Dataset: `$DATASETS_DIR/celeba_128.npy` (float32 [N,128,128,3] in [-1,1]) when present, else synthetic batches."""
import argparse
import os

import numpy as np
import torch

import blurred_gan_amd as blurred_gan
from blurred_gan_amd import BlurredWGANGP, TrainingConfig, callbacks, layers, utils
from blurred_gan_amd.checkpoint import CheckpointManager


def make_dataset(batch_size, n_batches=None, seed=0):
    """demo_celeba.py:15-48 (normalise to [-1,1], resize to 128x128, shuffle); preprocessing is expected offline."""
    path = os.path.join(os.environ.get("DATASETS_DIR", "/tmp/datasets"), "celeba_128.npy")
    rng = np.random.default_rng(seed)
    if os.path.exists(path):
        x = np.load(path, mmap_mode="r")
        idx = rng.permutation(len(x))
        n = (len(x) // batch_size) if n_batches is None else n_batches
        return (torch.from_numpy(np.ascontiguousarray(x[np.sort(idx[i * batch_size:(i + 1) * batch_size])])) for i in range(n))
    return [torch.from_numpy(rng.uniform(-1, 1, size=(batch_size, 128, 128, 3)).astype(np.float32)) for _ in range(n_batches or 16)]


class DCGANGenerator(layers.Sequential):
    """demo_celeba.py:51-93."""

    def __init__(self, latent_size=100, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.latent_size = latent_size
        self.add(layers.Dense(4 * 4 * 512, use_bias=False, input_shape=(self.latent_size,)))
        self.add(layers.BatchNormalization())
        self.add(layers.LeakyReLU())
        self.add(layers.Reshape((4, 4, 512)))
        assert self.output_shape == (None, 4, 4, 512)
        hw = 4
        for filters, stride in ((512, 1), (256, 2), (128, 2), (64, 2), (32, 2), (16, 2)):
            self.add(layers.Conv2DTranspose(filters, (5, 5), strides=(stride, stride), padding='same', use_bias=False))
            hw *= stride
            assert self.output_shape == (None, hw, hw, filters), self.output_shape
            self.add(layers.BatchNormalization())
            self.add(layers.LeakyReLU())
        self.add(layers.Conv2D(3, (5, 5), padding='same', use_bias=False, activation='tanh'))
        assert self.output_shape == (None, 128, 128, 3), self.output_shape


class DCGANDiscriminator(layers.Sequential):
    """demo_celeba.py:96-124."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for i, c in enumerate((16, 32, 64, 128, 256, 512)):
            kw = dict(input_shape=[128, 128, 3]) if i == 0 else {}
            self.add(layers.Conv2D(c, 5, strides=2, padding='same', **kw))
            self.add(layers.LeakyReLU())
            self.add(layers.Dropout(0.3))
        self.add(layers.Flatten())
        self.add(layers.Dense(1, activation="linear"))


def main(argv=None):
    """demo_celeba.py:127-246; multi-GPU decisions as in demo_mnist.main (global batch = per-GPU batch x replicas, one run
    directory made by rank 0, file-writing callbacks on rank 0 only)."""
    blurred_gan.set_seed(123123)
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    BlurredWGANGP.HyperParameters.add_arguments(parser)
    TrainingConfig.add_arguments(parser)
    parser.add_argument("--epochs", type=int, default=10)
    parser.add_argument("--max_batches", type=int, default=None)
    parser.add_argument("--results_dir", default="results")
    args = parser.parse_args(argv)
    hyperparameters = BlurredWGANGP.HyperParameters.from_args(args)
    config = TrainingConfig.from_args(args)
    dist = blurred_gan.dist
    num_gpus = dist.init_from_env()
    rank0 = dist.rank() == 0
    if rank0:
        print("Num gpus:", num_gpus)
    hyperparameters.global_batch_size = hyperparameters.batch_size * num_gpus
    dataset = make_dataset(hyperparameters.batch_size, n_batches=args.max_batches, seed=dist.rank())
    total_n_examples = 202_599
    config.log_dir = dist.broadcast_object(utils.create_result_subdir(args.results_dir, "celeba") if rank0 else None)
    config.checkpoint_dir = config.log_dir + "/checkpoints"
    gen, disc = DCGANGenerator(), DCGANDiscriminator()
    gan = blurred_gan.BlurredWGANGP(gen, disc, hyperparams=hyperparameters, config=config)
    manager = CheckpointManager(gan, directory=config.checkpoint_dir, max_to_keep=5)
    if manager.latest_checkpoint:
        manager.restore(manager.latest_checkpoint)
    cbs = [callbacks.BlurDecayController(total_n_training_examples=total_n_examples * args.epochs, max_value=5)]
    if rank0:
        gan.hparams.save_json(os.path.join(config.log_dir, "hyper_parameters.json"))
        gan.config.save_json(os.path.join(config.log_dir, "train_config.json"))
        cbs = [callbacks.GenerateSampleGridCallback(log_dir=config.log_dir, every_n_examples=5_000), *cbs,
               callbacks.SaveModelCallback(manager, n=10_000), callbacks.LogMetricsCallback()]
    try:
        gan.fit(x=dataset, y=None, epochs=args.epochs, initial_epoch=gan.n_img // total_n_examples, callbacks=cbs)
    except KeyboardInterrupt:
        if rank0:
            manager.save()
    dist.barrier()
    if rank0:
        print("Done training.")
    return gan


if __name__ == "__main__":
    main()
    blurred_gan.dist.shutdown()

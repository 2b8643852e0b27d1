"""Counterpart of reference demo_mnist.py: same classes, flags and call sequence, on the HIP kernels.

The reference loads MNIST through tensorflow_datasets (network); here the dataset is `$DATASETS_DIR/mnist.npz`
(key `x_train`, uint8 [N,28,28]) when present, otherwise synthetic U(-1,1) images of the same shape."""
import argparse
import os

import numpy as np
import torch

import blurred_gan_amd as blurred_gan
from blurred_gan_amd import BlurredWGANGP, TrainingConfig, callbacks, layers, utils
from blurred_gan_amd.checkpoint import CheckpointManager


def make_dataset(batch_size, n_batches=None, shuffle_buffer_size=256, seed=0):
    """demo_mnist.py:17-45: take image, cast, (x - 127.5) / 127.5, shuffle, batch."""
    path = os.path.join(os.environ.get("DATASETS_DIR", "/tmp/datasets"), "mnist.npz")
    rng = np.random.default_rng(seed)
    if os.path.exists(path):
        x = (np.load(path)["x_train"].astype(np.float32) - 127.5) / 127.5
        x = x.reshape(-1, 28, 28, 1)
    else:
        x = rng.uniform(-1, 1, size=(batch_size * (n_batches or 64), 28, 28, 1)).astype(np.float32)
    idx = rng.permutation(len(x))
    return [torch.from_numpy(x[idx[i:i + batch_size]]) for i in range(0, len(x) - batch_size + 1, batch_size)][:n_batches]


class DCGANGenerator(layers.Sequential):
    """demo_mnist.py:48-71."""

    def __init__(self, latent_size=100, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.latent_size = latent_size
        self.add(layers.Dense(7 * 7 * 256, use_bias=False, input_shape=(self.latent_size,)))
        self.add(layers.BatchNormalization())
        self.add(layers.LeakyReLU())
        self.add(layers.Reshape((7, 7, 256)))
        assert self.output_shape == (None, 7, 7, 256)
        self.add(layers.Conv2DTranspose(128, (5, 5), strides=(1, 1), padding='same', use_bias=False))
        assert self.output_shape == (None, 7, 7, 128)
        self.add(layers.BatchNormalization())
        self.add(layers.LeakyReLU())
        self.add(layers.Conv2DTranspose(64, (5, 5), strides=(2, 2), padding='same', use_bias=False))
        assert self.output_shape == (None, 14, 14, 64)
        self.add(layers.BatchNormalization())
        self.add(layers.LeakyReLU())
        self.add(layers.Conv2DTranspose(1, (5, 5), strides=(2, 2), padding='same', use_bias=False, activation='tanh'))
        assert self.output_shape == (None, 28, 28, 1)


class DCGANDiscriminator(layers.Sequential):
    """demo_mnist.py:74-86."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.add(layers.Conv2D(64, (5, 5), strides=(2, 2), padding='same', input_shape=[28, 28, 1]))
        self.add(layers.LeakyReLU())
        self.add(layers.Dropout(0.3))
        self.add(layers.Conv2D(128, (5, 5), strides=(2, 2), padding='same'))
        self.add(layers.LeakyReLU())
        self.add(layers.Dropout(0.3))
        self.add(layers.Flatten())
        self.add(layers.Dense(1))


def main(argv=None):
    """demo_mnist.py:91-219.  Under ``torchrun`` (one process per GPU) every rank trains on its own shard; what the reference
    leaves undone for multi-GPU (demo_mnist.py:116 "TODO") is decided here: ``global_batch_size`` = per-GPU batch x replicas
    (wgan.py:130,157 scale the losses by it), ONE run directory created by rank 0 and shared, and the host-side callbacks that
    write files (checkpoints, sample grids, scalar logs) on rank 0 only."""
    blurred_gan.set_seed(123123)
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    BlurredWGANGP.HyperParameters.add_arguments(parser)
    TrainingConfig.add_arguments(parser)
    parser.add_argument("--epochs", type=int, default=10)
    parser.add_argument("--max_batches", type=int, default=None, help="truncate the epoch (smoke runs)")
    parser.add_argument("--results_dir", default="results")
    args = parser.parse_args(argv)
    hyperparameters = BlurredWGANGP.HyperParameters.from_args(args)
    config = TrainingConfig.from_args(args)

    dist = blurred_gan.dist
    num_gpus = dist.init_from_env()
    rank0 = dist.rank() == 0
    if rank0:
        print(hyperparameters)
        print(config)
        print("Num gpus:", num_gpus)
    batch_size_per_gpu = hyperparameters.batch_size
    hyperparameters.global_batch_size = batch_size_per_gpu * num_gpus          # demo_mnist.py:123 computes it and forgets to store it
    dataset = make_dataset(batch_size_per_gpu, n_batches=args.max_batches, seed=dist.rank())
    total_n_examples = 60_000
    config.log_dir = dist.broadcast_object(utils.create_result_subdir(args.results_dir, "mnist") if rank0 else None)
    config.checkpoint_dir = config.log_dir + "/checkpoints"

    gen = DCGANGenerator()
    disc = DCGANDiscriminator()
    gan = blurred_gan.BlurredWGANGP(gen, disc, hyperparams=hyperparameters, config=config)
    manager = CheckpointManager(gan, directory=config.checkpoint_dir, max_to_keep=5)
    if manager.latest_checkpoint:
        manager.restore(manager.latest_checkpoint)
        print(f"Model was previously trained on {gan.n_img.numpy()} images")
    cbs = [callbacks.BlurDecayController(total_n_training_examples=total_n_examples * args.epochs,
                                         max_value=hyperparameters.initial_blur_std)]
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
        samples = gan.generate_samples()
        print(tuple(samples.shape))
    return gan


if __name__ == "__main__":
    main()
    blurred_gan.dist.shutdown()

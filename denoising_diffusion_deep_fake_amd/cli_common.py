"""Shared plumbing of the three click commands (`d3f train`, `d3f denoise`, `d3f balance`): YAML -> hparams and
one place that builds the Trainer.  The command / option names stay the reference's
(d3f/train_deep_fake/start_training.py:8-31, d3f/train_denoiser/train_denoiser.py:7-20,
d3f/balance_training_images/balance_training_images.py:7-24); everything behind them is this module."""
import click
import yaml

from .trainer import LearningRateMonitor, Trainer

max_steps_option = click.option("--max_steps", default=-1, type=int,
                                help="Stop after this many optimiser steps (smoke runs).")


def load_hparams(yaml_path, **extra):
    """the config file as a dict, with command-line supplied keys (image lists) laid over it"""
    with open(yaml_path) as stream:
        hparams = yaml.safe_load(stream) or {}
    if not isinstance(hparams, dict):
        raise click.BadParameter(f"{yaml_path}: expected a mapping of hyper-parameters at the top level")
    hparams.update(extra)
    return hparams


def describe(hparams, out=print):
    out("\nHyper Parameters:")
    for key in hparams:
        out(f"\t{key}: {hparams[key]}")
    out("")


def fit(lit_module, ckpt_path=None, max_steps=-1, lr_monitor=False, verbose=False):
    """Trainer(log_every_n_steps=1, max_epochs=hparams.max_epochs).fit(lit_module[, ckpt_path]) -- the flags every
    reference command passes; one HIP device per process (torch.distributed.run adds data-parallel ranks)."""
    hp = lit_module.hparams
    if verbose:
        describe(hp)
    trainer = Trainer(max_epochs=hp.max_epochs, max_steps=max_steps, log_every_n_steps=1,
                      callbacks=[LearningRateMonitor(logging_interval="step")] if lr_monitor else [],
                      default_root_dir=hp.get("default_root_dir", "lightning_logs"))
    trainer.fit(model=lit_module, ckpt_path=ckpt_path)
    return trainer

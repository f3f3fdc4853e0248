"""`d3f train new|resume|modify` -- d3f/train_deep_fake/start_training.py:8-65 on the HIP path."""
import click
import yaml

from ..trainer import Trainer
from .lit_module import LitModule


@click.group()
def train():
    pass


@train.command()
@click.option("--config_path", required=True, help="Path to the config yaml.")
@click.option("--max_steps", default=-1, type=int, help="Stop after this many optimiser steps (smoke runs).")
def new(config_path, max_steps):
    hparams_dict = read_yaml_file_into_dict(config_path)
    lit_module = LitModule(**hparams_dict)
    start_training(lit_module, max_steps=max_steps)


@train.command()
@click.option("--checkpoint_path", required=True, help="Path to model checkpoint.")
@click.option("--max_steps", default=-1, type=int)
def resume(checkpoint_path, max_steps):
    lit_module = LitModule.load_from_checkpoint(checkpoint_path)
    start_training(lit_module, resume_from_checkpoint=checkpoint_path, max_steps=max_steps)


@train.command()
@click.option("--config_path", required=True, help="Path to the config yaml.")
@click.option("--checkpoint_path", required=True, help="Path to model checkpoint.")
@click.option("--max_steps", default=-1, type=int)
def modify(config_path, checkpoint_path, max_steps):
    hparams_dict = read_yaml_file_into_dict(config_path)
    lit_module = LitModule.load_from_checkpoint(checkpoint_path, strict=False, **hparams_dict)
    start_training(lit_module, max_steps=max_steps)


def read_yaml_file_into_dict(yaml_file_path):
    with open(yaml_file_path) as f:
        return yaml.safe_load(f)


def start_training(lit_module, resume_from_checkpoint=None, max_steps=-1):
    p = lit_module.hparams
    print_hparams(p)
    trainer = Trainer(
        accelerator="gpu",
        devices=1,
        log_every_n_steps=1,
        max_epochs=p.max_epochs,
        max_steps=max_steps,
        default_root_dir=p.get("default_root_dir", "lightning_logs"),
    )
    trainer.fit(model=lit_module, ckpt_path=resume_from_checkpoint)
    return trainer


def print_hparams(p):
    print()
    print("Hyper Parameters:")
    for k, v in p.items():
        print(f"\t{k}: {v}")
    print()


if __name__ == "__main__":
    train()

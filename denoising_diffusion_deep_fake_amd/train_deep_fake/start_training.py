"""`d3f train new|resume|modify` (d3f/train_deep_fake/start_training.py:8-31) on the HIP path."""
import click

from .. import cli_common
from .lit_module import LitModule


@click.group()
def train():
    pass


@train.command()
@click.option("--config_path", required=True, help="Path to the config yaml.")
@cli_common.max_steps_option
def new(config_path, max_steps):
    start_training(LitModule(**cli_common.load_hparams(config_path)), max_steps=max_steps)


@train.command()
@click.option("--checkpoint_path", required=True, help="Path to model checkpoint.")
@cli_common.max_steps_option
def resume(checkpoint_path, max_steps):
    start_training(LitModule.load_from_checkpoint(checkpoint_path), resume_from_checkpoint=checkpoint_path,
                   max_steps=max_steps)


@train.command()
@click.option("--config_path", required=True, help="Path to the config yaml.")
@click.option("--checkpoint_path", required=True, help="Path to model checkpoint.")
@cli_common.max_steps_option
def modify(config_path, checkpoint_path, max_steps):
    # new hyper-parameters over old weights; strict=False because denoise -> swap adds the EMA copies
    overrides = cli_common.load_hparams(config_path)
    start_training(LitModule.load_from_checkpoint(checkpoint_path, strict=False, **overrides), max_steps=max_steps)


def start_training(lit_module, resume_from_checkpoint=None, max_steps=-1):
    return cli_common.fit(lit_module, ckpt_path=resume_from_checkpoint, max_steps=max_steps, verbose=True)


if __name__ == "__main__":
    train()

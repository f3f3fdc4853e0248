"""`d3f balance --config ... --input_list ... --output_list ...` -- d3f/balance_training_images/balance_training_images.py:7-56."""
import click
import yaml

from ..trainer import LearningRateMonitor, Trainer
from .lit_module import LitModule


@click.command()
@click.option("--config", required=True, help="Path to config yaml file")
@click.option("--input_list", required=False, default=None,
              help="Path to text file that lists relative paths to each image (omit with `synthetic: true`)")
@click.option("--output_list", required=False, default=None,
              help="Path of the text file to write: one '<image path> TAB <difficulty class>' line per image")
@click.option("--max_steps", default=-1, type=int, help="Stop after this many optimiser steps (smoke runs).")
def balance(**options):
    """Assign difficulty class to each image for balanced sampling.

    This trains a model to denoise images.
    It then bins the images based on the reconstuction loss.
    The bin indexes become the difficulty classes.
    """
    print(options)
    hparams_dict = read_yaml_file_into_dict(options["config"])
    hparams_dict["input_image_list_path"] = options["input_list"]
    hparams_dict["output_image_list_path"] = options["output_list"]
    start_training(hparams_dict, max_steps=options["max_steps"])


def read_yaml_file_into_dict(yaml_file_path):
    with open(yaml_file_path) as f:
        return yaml.safe_load(f)


def start_training(hparams_dict, max_steps=-1):
    lit_module = LitModule(**hparams_dict)
    p = lit_module.hparams
    trainer = Trainer(gpus=1, log_every_n_steps=1, max_epochs=p.max_epochs, max_steps=max_steps,
                      callbacks=[LearningRateMonitor(logging_interval="step")],
                      default_root_dir=p.get("default_root_dir", "lightning_logs"))
    trainer.fit(model=lit_module)
    return trainer

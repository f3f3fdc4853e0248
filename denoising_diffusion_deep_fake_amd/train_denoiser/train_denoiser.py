"""`d3f denoise --config ... --input_list ...` -- d3f/train_denoiser/train_denoiser.py:7-52 on the HIP path."""
import click
import yaml

from ..trainer import LearningRateMonitor, Trainer
from .lit_module import LitModule


@click.command()
@click.option("--config", required=True, help="Path to config yaml file")
@click.option("--input_list", required=False, default=None,
              help="Path to text file that lists relative paths to each image (omit with `synthetic: true`)")
@click.option("--max_steps", default=-1, type=int, help="Stop after this many optimiser steps (smoke runs).")
def denoise(**options):
    """This trains a model to denoise images."""
    print(options)
    hparams_dict = read_yaml_file_into_dict(options["config"])
    hparams_dict["input_image_list_path"] = options["input_list"]
    start_training(hparams_dict, max_steps=options["max_steps"])


def read_yaml_file_into_dict(yaml_file_path):
    with open(yaml_file_path) as f:
        return yaml.safe_load(f)


def start_training(hparams_dict, max_steps=-1):
    lit_module = LitModule(**hparams_dict)
    p = lit_module.hparams
    callback_list = [LearningRateMonitor(logging_interval="step")]
    trainer = Trainer(
        gpus=1,
        log_every_n_steps=1,
        max_epochs=p.max_epochs,
        max_steps=max_steps,
        callbacks=callback_list,
        default_root_dir=p.get("default_root_dir", "lightning_logs"),
    )
    trainer.fit(model=lit_module)
    return trainer

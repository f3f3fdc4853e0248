"""`d3f denoise --config ... --input_list ...` (d3f/train_denoiser/train_denoiser.py:7-52) on the HIP path."""
import click

from .. import cli_common
from .lit_module import LitModule


@click.command()
@click.option("--config", required=True, help="Path to config yaml file")
@click.option("--input_list", required=False, default=None,
              help="Path to text file that lists relative paths to each image (omit with `synthetic: true`)")
@cli_common.max_steps_option
def denoise(**options):
    """This trains a model to denoise images."""
    print(options)
    start_training(cli_common.load_hparams(options["config"], input_image_list_path=options["input_list"]),
                   max_steps=options["max_steps"])


def start_training(hparams_dict, max_steps=-1):
    return cli_common.fit(LitModule(**hparams_dict), max_steps=max_steps, lr_monitor=True)

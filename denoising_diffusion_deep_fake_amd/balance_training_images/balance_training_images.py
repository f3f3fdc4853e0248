"""`d3f balance --config ... --input_list ... --output_list ...`
(d3f/balance_training_images/balance_training_images.py:7-56) on the HIP path."""
import click

from .. import cli_common
from .lit_module import LitModule


@click.command()
@click.option("--config", required=True, help="Path to config yaml file")
@click.option("--input_list", required=False, default=None,
              help="Path to text file that lists relative paths to each image (omit with `synthetic: true`)")
@click.option("--output_list", required=False, default=None,
              help="Path of the text file to write: one '<image path> TAB <difficulty class>' line per image")
@cli_common.max_steps_option
def balance(**options):
    """Assign difficulty class to each image for balanced sampling.

    Trains a denoiser at one fixed noise ratio, scores every image by its reconstruction error and bins the
    scores; the bin indexes are the difficulty classes.
    """
    print(options)
    start_training(cli_common.load_hparams(options["config"], input_image_list_path=options["input_list"],
                                           output_image_list_path=options["output_list"]),
                   max_steps=options["max_steps"])


def start_training(hparams_dict, max_steps=-1):
    return cli_common.fit(LitModule(**hparams_dict), max_steps=max_steps, lr_monitor=True)

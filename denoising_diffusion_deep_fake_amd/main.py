"""`d3f` console entry point (d3f/main.py:6-12): `d3f train new|resume|modify`, `d3f denoise`, `d3f balance`."""
import click

from .balance_training_images.balance_training_images import balance
from .train_deep_fake.start_training import train
from .train_denoiser.train_denoiser import denoise


@click.group()
def cli():
    pass


cli.add_command(train)
cli.add_command(denoise)
cli.add_command(balance)

if __name__ == "__main__":
    cli()

"""Alias package: `import d3f...` resolves to the MI355X implementation, keeping the reference's
module paths (d3f.train_denoiser.lit_module, d3f.loss_functions, d3f.dataset.image_dataset, ...)."""
import importlib
import sys

_impl = "denoising_diffusion_deep_fake_amd"
_pkg = importlib.import_module(_impl)
__path__ = _pkg.__path__  # sub-modules are found in the implementation package


class _Alias:
    """meta-path finder mapping d3f.X -> denoising_diffusion_deep_fake_amd.X"""

    @staticmethod
    def find_spec(name, path=None, target=None):
        if not name.startswith("d3f."):
            return None
        real = _impl + name[3:]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError:
            return None
        sys.modules[name] = mod
        return importlib.util.spec_from_loader(name, loader=_Loader(mod))


class _Loader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


import importlib.util  # noqa: E402

sys.meta_path.insert(0, _Alias)

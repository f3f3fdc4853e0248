"""Just enough of the pytorch_lightning 1.x surface the reference's LitModules lean on
(`save_hyperparameters`, `self.hparams`, `self.log`, `self.device`, `global_step`, `current_epoch`,
`load_from_checkpoint`) -- pytorch_lightning itself is not installed on the MI355X image
(SURVEY.md 8c).  Behaviour follows SURVEY.md Appendix A.4; the training loop is in trainer.py.
"""
import inspect

import torch
import torch.nn as nn


class AttributeDict(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.__dict__["_hparams"] = AttributeDict()
        self.__dict__["trainer"] = None
        self.__dict__["_logged"] = {}
        self.__dict__["_manual_optimizers"] = None
        self.automatic_optimization = True  # False: training_step runs backward + optimiser step itself

    def optimizers(self):
        """the optimizer(s) of this module, as pytorch_lightning's `self.optimizers()` (manual optimisation)"""
        opts = self.trainer.optimizers if self.trainer is not None else self._manual_optimizers
        if opts is None:
            raise RuntimeError("optimizers(): no trainer attached and attach_optimizers() was not called")
        return opts[0] if len(opts) == 1 else list(opts)

    def attach_optimizers(self, optimizers):
        """stand-alone use (no Trainer): the optimizers `self.optimizers()` hands to a manual-optimisation step"""
        self.__dict__["_manual_optimizers"] = list(optimizers)

    def save_hyperparameters(self, *args, **kwargs):
        # pytorch_lightning collects the caller's __init__ arguments; the reference's modules are all
        # `def __init__(self, **kwargs): ... self.save_hyperparameters()`
        frame = inspect.currentframe().f_back
        local = frame.f_locals
        hp = {}
        for k, v in local.items():
            if k in ("self", "__class__") or k.startswith("_"):
                continue
            if k == "kwargs" and isinstance(v, dict):
                hp.update(v)
            elif not isinstance(v, (nn.Module, torch.Tensor)):
                hp[k] = v
        hp.update(kwargs)
        self._hparams.update(hp)

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @property
    def global_step(self):
        return self.trainer.global_step if self.trainer is not None else 0

    @property
    def current_epoch(self):
        return self.trainer.current_epoch if self.trainer is not None else 0

    @property
    def logger(self):
        return self.trainer.logger if self.trainer is not None else None

    def log(self, name, value, **kwargs):
        # kept as a device tensor: no host sync on the step path (the Trainer reads it when it prints)
        self._logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

    # ---- checkpoints: Lightning's {"state_dict", "hyper_parameters", ...} layout ---------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", strict=True, **overrides):
        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        module = cls(**hp)
        module.load_state_dict(ckpt["state_dict"], strict=strict)
        return module

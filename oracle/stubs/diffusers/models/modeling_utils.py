import torch


class ModelMixin(torch.nn.Module):
    """nn.Module + `.config` (provided by ConfigMixin.register_to_config)."""

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

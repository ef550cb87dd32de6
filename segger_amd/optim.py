"""``Adam``: ``torch.optim.Adam`` (the reference's optimizer, lightning_model.py:300-303) whose ``step()`` runs the
hand-written kernel (``segger_adam_step``: every parameter tensor in two launches instead of torch's three multi-tensor
launches, and one ctypes call instead of torch's per-step tensor-list grouping on the host) whenever the configuration
is the plain one -- fp32 parameters on the GPU, device-side step counters (``capturable=True``), no weight decay /
amsgrad / maximize.  Same class hierarchy, same state (``exp_avg``, ``exp_avg_sq``, ``step``), same ``state_dict``: a
checkpoint moves freely between this class and ``torch.optim.Adam``.  Anything else falls through to torch's own step."""
from __future__ import annotations

import torch


class Adam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        loss = None
        if closure is not None:                      # Lightning's automatic optimisation hands the whole step in
            with torch.enable_grad():
                loss = closure()
        if not ops.adam_step(self):                  # first step (no state yet) or a configuration the kernel does not cover
            super().step()
        return loss

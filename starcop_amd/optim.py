"""Fused Adam for the HIP network: one kernel over the flat parameter / gradient / moment buffers.

Replaces ``torch.optim.Adam(self.network.parameters(), self.lr)`` (reference model_module.py:174; torch
defaults betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False).  It *is* a ``torch.optim.Adam`` (so
``ReduceLROnPlateau``, Lightning and ``state_dict()`` treat it as one; the per-parameter ``exp_avg`` /
``exp_avg_sq`` state entries are views of the flat buffers) whose ``step()`` launches ``sc_adam_step``.
Bias corrections and the learning rate live in device memory (``sc_adam_prepare``) so that a captured
hipGraph of the training step replays correctly while the step count and the scheduler's lr change.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream


class FusedAdam(torch.optim.Adam):
    def __init__(self, network, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.network = network
        params = network._ensure_flat()
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        flat = network.flat_parameters()
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=flat.device)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._hp_dev = torch.zeros(4, dtype=torch.float32, device=flat.device)
        self._lr_host = None
        self._flat_ptr = flat.data_ptr()
        self._bind_state_views()

    def _bind_state_views(self):
        """(re)point the per-parameter state entries at the flat moment buffers; tensors found there that are NOT those views
        (what torch.optim.Optimizer.load_state_dict leaves: copies of a checkpoint's exp_avg / exp_avg_sq / step) are first
        copied into the flat buffers, so a resumed run continues with the checkpoint's moments and bias corrections"""
        off, step = 0, None
        for p in self.network.parameters():
            n = p.numel()
            mv, vv = self._m[off:off + n].view(p.shape), self._v[off:off + n].view(p.shape)
            st = self.state.get(p)
            if st is not None and "exp_avg" in st and st["exp_avg"].data_ptr() != mv.data_ptr():
                mv.copy_(st["exp_avg"].to(mv.device, torch.float32).reshape(p.shape))
                vv.copy_(st["exp_avg_sq"].to(vv.device, torch.float32).reshape(p.shape))
                s = int(round(float(st.get("step", 0))))
                step = s if step is None else max(step, s)
            self.state[p] = {"step": self._step_dev.view(()),   # shared scalar step (tensor, like capturable Adam)
                             "exp_avg": mv, "exp_avg_sq": vv}
            off += n
        if step is not None:
            self._step_dev.fill_(step)
        self._lr_host = None                                     # param_groups may carry a different lr now

    def load_state_dict(self, state_dict):
        """Checkpoint resume (Lightning restores the optimiser this way; reference train.py:137).  Accepts this class's own
        state and a plain torch.optim.Adam state of the same parameter list."""
        super().load_state_dict(state_dict)
        self._bind_state_views()

    def __setstate__(self, state):
        super().__setstate__(state)
        if hasattr(self, "_m"):
            self._bind_state_views()

    def _sync_lr(self):
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    @torch.no_grad()
    def step_flat(self, grad_scale=1.0, sync_ranks=True):
        """Adam update straight from the network's flat gradient buffer.  ``sync_ranks=False``: this step is rank-local (the other
        ranks of an initialised process group are NOT stepping): it takes no part in the periodic range check."""
        net = self.network
        flat = net.flat_parameters()
        if flat.data_ptr() != self._flat_ptr:
            raise RuntimeError("FusedAdam: the network's parameter storage moved (e.g. .to(device) after the optimiser "
                               "was built); create the optimiser after moving the module")
        g = net.flat_grads()
        lib = _lib.load()
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        self._sync_lr()
        st = stream()
        check(lib.sc_adam_prepare(ptr(self._step_dev), ptr(self._lr_dev), float(b1), float(b2), ptr(self._hp_dev), st))
        check(lib.sc_adam_step(ptr(flat), ptr(g), ptr(self._m), ptr(self._v), flat.numel(), 0.0, float(b1), float(b2),
                               float(grp["eps"]), float(grp["weight_decay"]), 1.0, 1.0, float(grad_scale),
                               ptr(self._hp_dev), st))
        net.mark_parameters_changed()
        if not sync_ranks:
            # rank-local step (the other ranks of an initialised process group are not stepping): it must not enter the common check's
            # collective, nor shift its cadence -- but a job that ONLY ever takes rank-local steps (independent per-rank fine-tuning
            # under torchrun) would otherwise never run the range check at all: own counter, local check, no all-reduce
            self._local_steps_since_check = getattr(self, "_local_steps_since_check", 0) + 1
            if (net.range_check_every and self._local_steps_since_check >= net.range_check_every
                    and not torch.cuda.is_current_stream_capturing()):
                self._local_steps_since_check = 0
                net.check_split_range(sync_ranks=False)
            return
        self._steps_since_check = getattr(self, "_steps_since_check", 0) + 1
        if net.range_check_every and self._steps_since_check >= net.range_check_every and not torch.cuda.is_current_stream_capturing():
            self._steps_since_check = 0
            # filters, BatchNorm gains and the device-side sticky activation records (every step since the last check) still inside
            # the fp16-split range?  MAX over the ranks of a data-parallel job: all replicas switch together or not at all
            net.check_split_range(sync_ranks=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        net = self.network
        net._ensure_flat()
        # gradients normally already live in the flat buffer (the autograd node returns views of it)
        for p in net.parameters():
            gv = net._grad_view(p)
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
        self.step_flat()
        return loss

"""``ModelModule``: the LightningModule surface of the reference, on the HIP network.

Mirrors /root/reference/starcop/models/model_module.py:
  ModelModule(settings) :24-67, training_step :69-88, forward :90-98, val_step :110-135,
  val_epoch_end :147-164, configure_optimizers :172-185, batch_with_preds :191-208,
  pred_classification :210-212, configure_architecture :224-256, load_weights :258-266, differences :268-269.

What is different underneath: ``network`` is :class:`starcop_amd.network.HyperStarcopUNet` (hand-written HIP
kernels, no torch conv/bn ops), the normaliser is fused into the stem convolution, the weighted BCE loss and the
integer masks are HIP kernels, and the optimiser is a fused single-pass Adam over a flat parameter buffer.
``pytorch_lightning`` / ``torchmetrics`` / ``wandb`` are optional: when Lightning is importable the class derives
from ``pl.LightningModule`` (so it drops into scripts/train.py), otherwise from ``torch.nn.Module`` with the same
methods plus a small ``fit`` loop.
"""
import ctypes as C
import os
from typing import Dict

import numpy as np
import torch
import torch.nn

from . import _lib, metrics
from ._lib import check, ptr, stream
from .network import HyperStarcopUNet
from .normalizer import DataNormalizer
from .optim import FusedAdam

try:  # optional: the reference pins pytorch_lightning 1.6.4 (requirements.txt:4)
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # pragma: no cover - Lightning is absent in the build image
    pl = None
    _Base = torch.nn.Module
    HAVE_LIGHTNING = False


class Settings(dict):
    """Tiny attribute-access config (OmegaConf ``DictConfig`` stand-in): ``s.model.lr``, ``"k" in s.dataset``."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = Settings(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_settings(**model_overrides):
    """The hot-path subset of starcop/config.yaml (:20-52)."""
    s = Settings(
        dataset=dict(input_products=["mag1c", "TOA_AVIRIS_640nm", "TOA_AVIRIS_550nm", "TOA_AVIRIS_460nm"],
                     output_products=["labelbinary"], use_weight_loss=True, weight_loss="weight_mag1c"),
        model=dict(model_mode="segmentation_output", model_type="unet_semseg", semseg_backbone="mobilenet_v2",
                   num_classes=1, optimizer="adam", lr=1e-4, lr_decay=0.5, lr_patience=4,
                   loss="BCEWithLogitsLoss", pos_weight=15, early_stopping_patience=8, train=True, test=False),
        wandb=dict(images_logging="local"),
    )
    for k, v in model_overrides.items():
        s.model[k] = v
    return s


# ----------------------------------------------------------------------------------------------
class _BCEFunction(torch.autograd.Function):
    """BCEWithLogits(pos_weight) [* weight] -> per-pixel loss or mean; forward and gradient in one HIP pass."""

    @staticmethod
    def forward(ctx, logits, target, weight, pos_weight, reduce_mean):
        _lib.require_device(logits)
        lib = _lib.load()
        z = logits.contiguous().float()
        t = target.contiguous().float()
        w = weight.contiguous().float() if weight is not None else None
        n = z.numel()
        dz = torch.empty_like(z)
        if reduce_mean:
            acc = torch.zeros(1, dtype=torch.float64, device=z.device)
            check(lib.sc_bce_logits_weighted(ptr(z), ptr(t), ptr(w), float(pos_weight), n, ptr(acc), ptr(dz), None, stream()))
            ctx.save_for_backward(dz)
            ctx.mean = True
            return (acc / n).float().reshape(())
        px = torch.empty_like(z)
        # gradient of the *sum* w.r.t. logits per pixel: undo the 1/n the kernel applies
        check(lib.sc_bce_logits_weighted(ptr(z), ptr(t), ptr(w), float(pos_weight), n, None, ptr(dz), ptr(px), stream()))
        ctx.save_for_backward(dz)
        ctx.mean, ctx.n = False, n
        return px

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        if ctx.mean:
            return dz * g, None, None, None, None
        return dz * (g * ctx.n), None, None, None, None


class HipBCEWithLogitsLoss(torch.nn.Module):
    """``torch.nn.BCEWithLogitsLoss(pos_weight, reduction)`` (model_module.py:57) on the HIP kernel.
    Registers ``pos_weight`` as a buffer so the ``loss_function.pos_weight`` checkpoint key exists."""

    def __init__(self, pos_weight, reduction="mean"):
        super().__init__()
        if reduction not in ("none", "mean"):
            raise ValueError(f"reduction {reduction!r} not supported")
        self.reduction = reduction
        self.register_buffer("pos_weight", pos_weight.detach().clone() if isinstance(pos_weight, torch.Tensor)
                             else torch.tensor(float(pos_weight)))

    def pos_weight_host(self):
        """host copy of pos_weight, refreshed only when the buffer changes (no device sync per step)."""
        key = (self.pos_weight._version, self.pos_weight.data_ptr())
        if getattr(self, "_pw_key", None) != key:
            self._pw_val, self._pw_key = float(self.pos_weight), key
        return self._pw_val

    def forward(self, logits, target):
        return _BCEFunction.apply(logits, target, None, self.pos_weight_host(), self.reduction == "mean")

    def weighted_mean(self, logits, target, weight):
        """mean(loss_none(logits, target) * weight) in one pass (model_module.py:76-79)."""
        return _BCEFunction.apply(logits, target, weight, self.pos_weight_host(), True)


# ----------------------------------------------------------------------------------------------
def masks_from_logits(logits, target=None, ge0=False):
    """HIP masks: returns dict(prediction f32, pred_binary i64, differences i64|None, pred_classification i64 (B,1)).

    ``ge0=True`` is the validation rule ``logits >= 0`` (model_module.py:124); ``ge0=False`` is
    ``sigmoid(logits) > .5`` (:204).  They differ only at logit == 0."""
    _lib.require_device(logits)
    lib = _lib.load()
    z = logits.contiguous().float()
    B, K, H, W = z.shape
    pred = torch.empty_like(z)
    pb = torch.empty(z.shape, dtype=torch.int64, device=z.device)
    diff = None
    tgt = None
    if target is not None:
        tgt = target.contiguous().float()
        diff = torch.empty(z.shape, dtype=torch.int64, device=z.device)
    cnt = torch.zeros(B * K, dtype=torch.int64, device=z.device)
    cls = torch.empty(B * K, dtype=torch.int64, device=z.device)
    st = stream()
    check(lib.sc_threshold_masks(ptr(z), ptr(tgt), 1 if ge0 else 0, ptr(pred), ptr(pb), ptr(diff), ptr(cnt), B * K, H * W, st))
    check(lib.sc_pred_classification(ptr(cnt), ptr(cls), B * K, H, W, st))
    return dict(prediction=pred, pred_binary=pb, differences=diff, pred_classification=cls.reshape(B, K))


def pred_classification(pred_binary: torch.Tensor) -> torch.Tensor:
    """tile has a plume iff sum(pred_binary) > 10*H*W/64**2  (model_module.py:210-212)."""
    _lib.require_device(pred_binary)
    lib = _lib.load()
    pbl = pred_binary.contiguous().long()
    H, W = pbl.shape[-2:]
    lead = pbl.shape[:-2]
    n = int(np.prod(lead)) if len(lead) else 1
    cnt = pbl.reshape(n, -1).sum(dim=1)
    cls = torch.empty(n, dtype=torch.int64, device=pbl.device)
    check(lib.sc_pred_classification(ptr(cnt), ptr(cls), n, H, W, stream()))
    return cls.reshape(lead)


def differences(y_pred_binary: torch.Tensor, y_gt: torch.Tensor) -> torch.Tensor:
    """2*pred + (gt == 1) in {0,1,2,3}  (model_module.py:268-269)."""
    return 2 * y_pred_binary.long() + (y_gt == 1).long()


def configure_architecture(architecture, num_channels, num_classes, extra_settings_model):
    """Only ``unet_semseg`` with the ``mobilenet_v2`` backbone exists on the reference's path (:224-256)."""
    if architecture == "unet_semseg":
        backbone = extra_settings_model.semseg_backbone
        if backbone != "mobilenet_v2":
            raise Exception(f"No HIP model implemented for semseg_backbone: {backbone}")
        net = HyperStarcopUNet(in_channels=num_channels, classes=num_classes)
        # extension (not in the reference's config.yaml): settings.model.precision = "fp32" (default, parity mode) | "fp32-bwd2" | "fp32-2" | "bf16"
        # (bf16 matrix math for the 3x3 convolutions, BASELINE configs[3]); absent key -> fp32
        try:
            prec = extra_settings_model["precision"] if "precision" in extra_settings_model else "fp32"
        except TypeError:
            prec = getattr(extra_settings_model, "precision", "fp32")
        net.precision = prec
        _ = net._terms           # validates
        return net
    raise Exception(f"No model implemented for model_type: {architecture}")


def load_weights(path_weights: str, map_location="cpu"):
    if os.path.exists(path_weights):
        with open(path_weights, "rb") as fh:
            return torch.load(fh, map_location=map_location)
    raise ValueError(f"Pretrained weights file: {path_weights} does not exists")


# ----------------------------------------------------------------------------------------------
def _multi_rank():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class ModelModule(_Base):

    def __init__(self, settings):
        super().__init__()
        if HAVE_LIGHTNING:
            self.save_hyperparameters()
        self.settings_model = settings.model
        self.settings_wandb = settings.wandb if "wandb" in settings else None
        self.normalizer = DataNormalizer(settings)
        self.num_classes = self.settings_model.num_classes
        self.num_channels = len(settings.dataset.input_products)
        architecture = self.settings_model.model_type
        self.network = configure_architecture(architecture, self.num_channels, self.num_classes, self.settings_model)
        self.lr = self.settings_model.lr
        self.lr_decay = self.settings_model.lr_decay
        self.lr_patience = self.settings_model.lr_patience
        self.loss_name = self.settings_model.loss
        use_weight_loss = "use_weight_loss" not in settings.dataset or settings.dataset.use_weight_loss
        if self.settings_model.loss == "BCEWithLogitsLoss":
            self.reduction = "none" if use_weight_loss else "mean"
            self.pos_weight = torch.nn.Parameter(torch.tensor(float(self.settings_model.pos_weight)), requires_grad=False)
            self.loss_function = HipBCEWithLogitsLoss(self.pos_weight, self.reduction)
        else:
            raise NotImplementedError(f"loss {self.settings_model.loss!r}: the segmentation path trains with "
                                      "BCEWithLogitsLoss (config.yaml:47)")
        if self.settings_model.model_mode == "segmentation_output":
            self.confusion_matrix = metrics.BinaryConfusionMatrix()
            self.classification_confusion_matrix = metrics.BinaryConfusionMatrix()
        elif self.settings_model.model_mode == "regression_output":
            raise NotImplementedError("Not implemented yet")
        self._logged = {}
        self._optimizer = None

    # -- Lightning shims when Lightning is absent -------------------------------------------------
    if not HAVE_LIGHTNING:
        @property
        def device(self):
            return next(self.parameters()).device

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, settings=None, map_location="cpu", strict=True, **kw):
            ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
            model = cls(settings)
            model.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt, strict=strict)
            return model

    def log(self, name, value=None, *args, **kwargs):
        try:
            if HAVE_LIGHTNING:
                super().log(name, value, *args, **kwargs)
            else:
                self._logged[name] = value
        except Exception as e:  # the reference swallows logging errors (:103-107)
            print(f"Bug logging {e}")

    # ---------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, num_channels, H, W) raw products -> (B, 1, H, W) logits; normalize_x is fused into the stem."""
        return self.network(x, self.normalizer.consts(x.device))

    def _loss(self, predictions, y, batch):
        if self.reduction == "none":
            return self.loss_function.weighted_mean(predictions, y, batch["weight_loss"])
        return self.loss_function(predictions, y)

    def training_step(self, batch: Dict, batch_idx) -> torch.Tensor:
        x, y = batch["input"], batch["output"]
        predictions = self.forward(x)
        loss = self._loss(predictions, self.normalizer.normalize_y(y), batch)
        if (batch_idx % 100) == 0:
            self.log(f"train_{self.loss_name}", loss)
        return loss

    def pred_classification(self, pred_binary: torch.Tensor) -> torch.Tensor:
        return pred_classification(pred_binary)

    def val_step(self, batch, batch_idx: int, prefix: str = "val"):
        x, y = batch["input"], batch["output"]
        predictions = self.forward(x)
        y = self.normalizer.normalize_y(y)
        loss = self._loss(predictions, y, batch)
        self.log(f"{prefix}_loss", loss, on_epoch=True)
        if self.settings_model.model_mode == "segmentation_output":
            m = masks_from_logits(predictions, None, ge0=True)       # (predictions >= 0).long()
            self.confusion_matrix.update(m["pred_binary"], y.long())
            y_classification = torch.as_tensor(batch["has_plume"], device=predictions.device).reshape(-1)[:, None]
            self.classification_confusion_matrix.update(m["pred_classification"], y_classification)
        return loss

    def validation_step(self, batch, batch_idx: int):
        return self.val_step(batch, batch_idx, prefix="val")

    def test_step(self, batch, batch_idx: int):
        return self.val_step(batch, batch_idx, prefix="test")

    def val_epoch_end(self, outputs, prefix):
        outs = {}
        for cmobj, tag in ((self.confusion_matrix, ""), (self.classification_confusion_matrix, "classification_")):
            if hasattr(cmobj, "sync"):
                cmobj.sync()
            cm = cmobj.compute()
            for fun in metrics.METRICS_CONFUSION_MATRIX:
                val = fun(cm)
                outs[f"{prefix}_{tag}{fun.__name__}"] = val
                self.log(f"{prefix}_{tag}{fun.__name__}", val)
            cmobj.reset()
        return outs

    def validation_epoch_end(self, outputs) -> None:
        self.val_epoch_end(outputs, prefix="val")

    def test_epoch_end(self, outputs) -> None:
        self.val_epoch_end(outputs, prefix="test")

    def configure_optimizers(self):
        if self.settings_model.optimizer == "adam":
            optimizer = FusedAdam(self.network, lr=self.lr)
        else:
            raise Exception(f"No optimizer implemented for : {self.settings_model.optimizer}")
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=self.lr_decay,
                                                               patience=self.lr_patience)
        return {"optimizer": optimizer, "lr_scheduler": scheduler, "monitor": "val_loss"}

    def debug(self):
        print("Model debug:")
        print(self)

    def batch_with_preds(self, batch):
        logits = self(batch["input"])
        batch = batch.copy()
        batch["input_norm"] = self.normalizer.normalize_x(batch["input"])
        batch["output_norm"] = self.normalizer.normalize_y(batch["output"])
        m = masks_from_logits(logits, batch["output_norm"], ge0=False)
        batch["prediction"] = m["prediction"]
        batch["logits"] = logits
        if self.reduction == "none":
            with torch.no_grad():
                batch["loss_per_pixel"] = self.loss_function(logits, batch["output_norm"])
            batch["loss_per_pixel_weighted"] = batch["weight_loss"] * batch["loss_per_pixel"]
        batch["pred_binary"] = m["pred_binary"]
        batch["differences"] = m["differences"]
        batch["pred_classification"] = m["pred_classification"]
        return batch

    def apply_threshold(self, pred, threshold):
        return (pred > threshold).long()

    def threshold_spec(self) -> int:
        """``apply_threshold`` is a plain ``pred > threshold`` (no morphology): see validation.run_validation."""
        return 0

    # ---------------------------------------------------------------------------------------------
    def predict(self, tensor: np.ndarray) -> np.ndarray:
        """np (C,H,W) raw products -> np (H,W) plume probability, any H,W (reflect-padded to x32); the
        convenience the notebooks spell as ``padded_predict(x, lambda x: sigmoid(model(x)))``."""
        from .padding import padded_predict
        was = self.training
        if was:            # (nn.Module.train() walks ~250 submodules: 0.3 ms per toggle -- skipped when the module is in eval mode already)
            self.eval()
        try:
            out = padded_predict(np.asarray(tensor, dtype=np.float32),
                                 lambda t: masks_from_logits(self(t))["prediction"], 32, self.device)
        finally:
            if was:
                self.train(True)
        return out[0]

    # ---------------------------------------------------------------------------------------------
    # fused training step: forward + loss + backward + (all-reduce) + Adam with no autograd graph.
    # Same arithmetic as training_step -> backward -> optimizer.step; hipGraph-capturable.
    def fused_train_step(self, batch, optimizer=None, grad_sync=None):
        if optimizer is None:
            if self._optimizer is None:
                self._optimizer = self.configure_optimizers()["optimizer"]
            optimizer = self._optimizer
        lib = _lib.load()
        net = self.network
        if not net.training:
            raise RuntimeError("fused_train_step needs the module in train() mode")
        x = batch["input"]
        y = self.normalizer.normalize_y(batch["output"]).contiguous().float()
        w = batch["weight_loss"].contiguous().float() if self.reduction == "none" else None
        plan = net._forward_impl(x, self.normalizer.consts(x.device), True, True)
        logits = plan.buf["logits"]
        if not hasattr(plan, "loss_acc"):
            plan.loss_acc = torch.zeros(1, dtype=torch.float64, device=x.device)
        plan.loss_acc.zero_()
        check(lib.sc_bce_logits_weighted(ptr(logits), ptr(y), ptr(w), self.loss_function.pos_weight_host(), logits.numel(),
                                         ptr(plan.loss_acc), ptr(plan.dlogits), None, stream()))
        scale = 1.0
        if grad_sync is not None and hasattr(grad_sync, "begin"):
            # two buckets: decoder + head gradients are reduced while the encoder's backward runs, the rest afterwards
            started = []
            g = net.flat_grads()
            net._backward_impl(plan, plan.dlogits,
                               on_tail_ready=lambda lo, hi: started.append((lo, grad_sync.begin(g[lo:hi]))))
            lo = started[0][0] if started else g.numel()
            scale = grad_sync.finish(g[:lo], [h for _, h in started])
        else:
            net._backward_impl(plan, plan.dlogits, exchange_follows=grad_sync is not None)      # (an all-reduce of these gradients follows)
            if grad_sync is not None:
                scale = grad_sync(net.flat_grads())
        # a step without gradient exchange inside an initialised multi-rank job is rank-local by definition (bench.py's
        # instrumented passes on rank 0): its periodic range check must not enter a collective the other ranks never join
        optimizer.step_flat(grad_scale=scale, sync_ranks=grad_sync is not None or not _multi_rank())
        plan.loss_n = logits.numel()
        return plan.loss_acc   # device double: sum of weighted per-pixel losses; divide by plan.loss_n for the mean

    def fit(self, train_batches, epochs=1, val_batches=None):
        """Minimal trainer used when Lightning is absent: Adam + ReduceLROnPlateau on val_loss."""
        cfg = self.configure_optimizers()
        opt, sched = cfg["optimizer"], cfg["lr_scheduler"]
        history = []
        for _ in range(epochs):
            self.train()
            for i, b in enumerate(train_batches):
                acc = self.fused_train_step(b, opt)
                if i % 100 == 0:
                    history.append(float(acc.item()) / self.network._plans[next(iter(self.network._plans))].loss_n)
            if val_batches is not None:
                self.eval()
                tot = []
                with torch.no_grad():
                    for i, b in enumerate(val_batches):
                        tot.append(float(self.validation_step(b, i)))
                self.val_epoch_end(None, "val")
                sched.step(float(np.mean(tot)))
        return history

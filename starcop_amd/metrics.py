"""Binary confusion-matrix metrics used for the F1 / IoU parity numbers.

Same names, argument (a 2x2 matrix ``cm[target, prediction]``) and formulas as
/root/reference/starcop/metrics.py:20-85; written around one (tn, fp, fn, tp) unpacking.
The matrices are accumulated from the integer masks the HIP ``sc_threshold_masks`` kernel writes, so
equal masks give equal metrics by construction.
"""
import torch


def _cells(cm):
    assert cm.shape == (2, 2), f"Expected binary found {cm.shape}"
    return cm[0, 0], cm[0, 1], cm[1, 0], cm[1, 1]      # tn, fp, fn, tp


def TN(cm): return _cells(cm)[0]          # noqa: E704
def FP(cm): return _cells(cm)[1]          # noqa: E704
def FN(cm): return _cells(cm)[2]          # noqa: E704
def TP(cm): return _cells(cm)[3]          # noqa: E704


def precision(cm):
    _, fp, _, tp = _cells(cm)
    return tp / (tp + fp)


def recall(cm):
    _, _, fn, tp = _cells(cm)
    return tp / (tp + fn)


def f1score(cm):
    p, r = precision(cm), recall(cm)
    return 2 * (p * r) / (p + r)


def iou(cm):
    _, fp, fn, tp = _cells(cm)
    return tp / (tp + fn + fp)


def accuracy(cm):
    tn, _, _, tp = _cells(cm)
    return (tp + tn) / cm.sum()


def FPR(cm):
    tn, fp, _, _ = _cells(cm)
    return fp / (fp + tn)


def balanced_accuracy(cm):
    tn, fp, _, _ = _cells(cm)
    return 0.5 * (recall(cm) + tn / (tn + fp))


def cohen_kappa(cm):
    """1 - (off-diagonal mass) / (expected off-diagonal mass under independence)."""
    m = cm if cm.is_floating_point() else cm.float()
    expected = m.sum(dim=1, keepdim=True) @ m.sum(dim=0, keepdim=True) / m.sum()
    off = 1.0 - torch.eye(2, dtype=m.dtype, device=m.device)
    return 1 - (off * m).sum() / (off * expected).sum()


def user_accuracy(cm): return precision(cm)          # noqa: E704   (own functions: run_validation keys results by __name__)
def producer_accuracy(cm): return recall(cm)      # noqa: E704
def TPR(cm): return recall(cm)                    # noqa: E704

METRICS_CONFUSION_MATRIX = [precision, recall, f1score, iou, accuracy, cohen_kappa, balanced_accuracy]


class BinaryConfusionMatrix:
    """Stand-in for ``torchmetrics.ConfusionMatrix(task='binary')`` (reference model_module.py:62-63) with the
    three calls the module uses: ``update(preds, target)``, ``compute()`` -> [[TN, FP], [FN, TP]], ``reset()``.
    ``sync()`` sums the matrix over data-parallel ranks (what torchmetrics does under DDP)."""

    def __init__(self):
        self.mat = torch.zeros(2, 2, dtype=torch.int64)

    def update(self, preds, target):
        code = (target.reshape(-1).long() * 2 + preds.reshape(-1).long())
        self.mat = self.mat.to(code.device) + torch.bincount(code, minlength=4).reshape(2, 2)

    def compute(self):
        return self.mat

    def reset(self):
        self.mat = torch.zeros(2, 2, dtype=torch.int64)

    def sync(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            m = self.mat.cuda() if dist.get_backend() == "nccl" else self.mat
            dist.all_reduce(m, op=dist.ReduceOp.SUM)
            self.mat = m

"""Noam learning-rate schedule (reference: promptttspp/utils/lr_scheduler.py:18-41):
lr = base_lr * sqrt(warmup) * min(step^-0.5, step * warmup^-1.5)."""
from torch.optim.lr_scheduler import _LRScheduler


def noam_scale(step, warmup_steps):
    step = max(1, step)
    return warmup_steps**0.5 * min(step ** (-0.5), step * warmup_steps ** (-1.5))


class NoamLR(_LRScheduler):
    def __init__(self, optimizer, warmup_steps):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer)

    def get_lr(self):
        s = noam_scale(self.last_epoch, self.warmup_steps)
        return [base_lr * s for base_lr in self.base_lrs]

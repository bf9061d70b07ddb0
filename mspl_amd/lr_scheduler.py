"""Epoch-wise learning-rate schedules of the supervised loop (host logic, no GPU work).

Same class names, constructor arguments and `step(epoch) -> lr` values as utilities/lr_scheduler.py (used at
train_segmentation.py:318-354): CyclicLR :8-53, FixedMultiStepLR :64-78, PolyLR :88-101, LinearLR :111-119,
HybirdLR :128-147 (the reference's spelling), CosineLR :159-166.  Values are pinned by tests/golden/lr_schedules.json,
written from the reference's classes by tests/golden/make_golden.py.
"""
import bisect
import math


class CyclicLR(object):
    """Warm restarts: after one warm-up epoch at min_lr the rate starts at cycle_len*min_lr, drops by min_lr per epoch and
    restarts after cycle_len epochs; at each epoch that is a multiple of the current entry of `steps` (epoch > 1) min_lr is
    multiplied by gamma and the next entry becomes current.  The warm-up happens once only (its interval is zeroed after
    the first use), so a decay epoch continues the running cycle with the smaller min_lr."""

    def __init__(self, min_lr=0.1, cycle_len=5, steps=[51, 101, 131, 161, 191, 221, 251, 281], gamma=0.5, step=True):
        assert len(steps) > 0, 'Please specify step intervals.'
        assert 0 < gamma <= 1, 'Learing rate decay factor should be between 0 and 1'
        self.min_lr = min_lr
        self.m = cycle_len
        self.steps = steps
        self.decayFactor = gamma
        self.stepping = step
        self._warmup_left = 1          # epochs still to be served at min_lr before the first cycle
        self._since_decay = 0          # epochs handled since the last decay (the reference's count_cycles)
        self._pos = 0                  # position inside the current cycle
        self._step_idx = 0

    def step(self, epoch):
        if self.stepping and epoch > 1 and epoch % self.steps[self._step_idx] == 0:
            self.min_lr = self.min_lr * self.decayFactor
            self._since_decay = 0
            if self._step_idx < len(self.steps) - 1:
                self._step_idx += 1
            else:
                self.stepping = False
        if self._since_decay < self._warmup_left:
            self._since_decay += 1
            if self._since_decay == self._warmup_left:
                self._warmup_left = 0
            return self.min_lr
        if self._pos >= self.m:
            self._pos = 0
        lr = round(self.min_lr * self.m - self._pos * self.min_lr, 5)
        self._pos += 1
        self._since_decay += 1
        return lr


class FixedMultiStepLR(object):
    def __init__(self, base_lr=0.1, steps=[30, 60, 90], gamma=0.1, step=True):
        assert len(steps) > 1, 'Please specify step intervals.'
        self.base_lr, self.steps, self.decayFactor, self.stepping = base_lr, steps, gamma, step

    def step(self, epoch):
        return round(self.base_lr * self.decayFactor ** bisect.bisect(self.steps, epoch), 5)


class PolyLR(object):
    def __init__(self, base_lr, max_epochs, power=0.99):
        assert 0 < power < 1
        self.base_lr, self.max_epochs, self.power = base_lr, max_epochs, power

    def step(self, epoch):
        return round(self.base_lr * (1 - float(epoch) / self.max_epochs) ** self.power, 6)


class LinearLR(object):
    def __init__(self, base_lr, max_epochs):
        self.base_lr, self.max_epochs = base_lr, max_epochs

    def step(self, epoch):
        return round(self.base_lr - self.base_lr * (epoch / self.max_epochs), 6)


class HybirdLR(object):
    """Cyclic for the first clr_max epochs (one decay-free CyclicLR), then linear decay over the remaining epochs."""

    def __init__(self, base_lr, clr_max, max_epochs, cycle_len=5):
        self.linear_epochs = max_epochs - clr_max + 1
        self.clr = CyclicLR(min_lr=base_lr, cycle_len=cycle_len, steps=[clr_max], gamma=1)
        self.decay_lr = LinearLR(base_lr=base_lr, max_epochs=self.linear_epochs)
        self.cyclic_epochs = clr_max
        self.base_lr, self.max_epochs, self.clr_max, self.cycle_len = base_lr, max_epochs, clr_max, cycle_len

    def step(self, epoch):
        if epoch < self.cyclic_epochs:
            return round(self.clr.step(epoch), 6)
        return round(self.decay_lr.step(epoch - self.cyclic_epochs + 1), 6)


class CosineLR(object):
    def __init__(self, base_lr, max_epochs):
        self.base_lr, self.max_epochs = base_lr, max_epochs

    def step(self, epoch):
        return round(self.base_lr * (1 + math.cos(math.pi * epoch / self.max_epochs)) / 2, 6)

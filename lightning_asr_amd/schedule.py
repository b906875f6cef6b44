"""Host-side LR schedule of the hot path (pure scalar math, no device work).

``CosineAnnealingWarmupRestarts`` keeps the reference's constructor
(scheduler/cosine_annearing_with_warmup.py:19-28) and is stepped once per batch
(``'interval': 'step'``, train.py:57-61).  It accepts either a torch optimizer (its
``param_groups[*]['lr']`` is then kept in sync, as the reference does) or ``None``."""
from __future__ import annotations

import math


class CosineAnnealingWarmupRestarts:
    def __init__(self, optimizer=None, first_cycle_steps: int = 1, cycle_mult: float = 1.0, max_lr: float = 0.1,
                 min_lr: float = 0.001, warmup_steps: int = 0, gamma: float = 1.0, last_epoch: int = -1):
        if not warmup_steps < first_cycle_steps:
            raise AssertionError("warmup_steps must be < first_cycle_steps")
        self.optimizer = optimizer
        self.first_cycle_steps = first_cycle_steps
        self.cycle_mult = cycle_mult
        self.base_max_lr = max_lr
        self.max_lr = max_lr
        self.min_lr = min_lr
        self.warmup_steps = warmup_steps
        self.gamma = gamma
        self.cur_cycle_steps = first_cycle_steps
        self.cycle = 0
        self.step_in_cycle = last_epoch
        self.last_epoch = last_epoch
        self.lr = min_lr
        self.step()        # the torch base class steps once on construction ...
        self.lr = min_lr   # ... after which the reference resets every group to min_lr (init_lr, :47-51)
        self._publish()

    def _publish(self) -> None:
        if self.optimizer is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = self.lr

    def get_lr(self) -> float:
        if self.step_in_cycle == -1:
            return self.min_lr
        if self.step_in_cycle < self.warmup_steps:
            return (self.max_lr - self.min_lr) * self.step_in_cycle / self.warmup_steps + self.min_lr
        return self.min_lr + (self.max_lr - self.min_lr) * (
            1 + math.cos(math.pi * (self.step_in_cycle - self.warmup_steps) / (self.cur_cycle_steps - self.warmup_steps))) / 2

    def step(self) -> float:
        self.last_epoch += 1
        self.step_in_cycle += 1
        if self.step_in_cycle >= self.cur_cycle_steps:
            self.cycle += 1
            self.step_in_cycle -= self.cur_cycle_steps
            self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        self.lr = self.get_lr()
        self._publish()
        return self.lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd) -> None:
        self.__dict__.update(sd)

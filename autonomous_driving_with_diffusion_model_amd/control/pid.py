"""Windowed PID term of the post-sampling controller (reference: control/pid.py:6-28).

    step(e) = K_P * e + K_I * mean(last n errors; the history starts as n zeros) + K_D * (e - previous error)

Kept as a preallocated ring buffer instead of the reference's deque.  The windowed mean reproduces numpy's result
for that deque bit for bit: same chronological order, and fp32 arithmetic once every sample in the window is an fp32
scalar (np.mean over a sequence of np.float32 stays in fp32; the initial integer zeros promote it to fp64).
"""
from __future__ import annotations

import numpy as np


class PIDController:
    def __init__(self, K_P: float = 1.0, K_I: float = 0.0, K_D: float = 0.0, n: int = 20):
        if n < 1:
            raise ValueError("PID window must hold at least one sample")
        self.gains = (K_P, K_I, K_D)
        self.history = np.zeros(n, dtype=np.float64)   # fp32 samples are stored exactly
        self.is_f32 = np.zeros(n, dtype=bool)          # which slots hold an np.float32 sample
        self.head = 0                    # slot of the NEXT sample
        self.last = 0                    # newest sample as it was given (the reference's window starts as integer zeros)

    def step(self, error):
        # `error` keeps its numpy type: the reference's P and D terms are evaluated in the sample's own precision
        # (fp32 when it comes from fp32 waypoints) and only the windowed mean in fp64
        previous = self.last
        self.history[self.head] = error
        self.is_f32[self.head] = isinstance(error, np.float32)
        self.head = (self.head + 1) % self.history.size
        self.last = error
        kp, ki, kd = self.gains
        # a window of one sample has no "previous" entry inside it: the reference then uses zero I and D terms
        if self.history.size < 2:
            return kp * error
        window = np.concatenate((self.history[self.head:], self.history[:self.head]))   # oldest first
        integral = window.astype(np.float32).mean() if self.is_f32.all() else window.mean()
        return kp * error + ki * integral + kd * (error - previous)

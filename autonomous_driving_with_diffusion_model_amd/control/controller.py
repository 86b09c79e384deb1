"""Waypoints -> (throttle, steer, brake): the step after the sampling loop in the reference's agents
(control/controller.py:29-76, called at interact.py:235 and e2e_driving/diffusion_agent.py), plus the agents'
`post_process_control` (e2e_driving/diffusion_agent.py:268-277).  Scalar host arithmetic on one trajectory; it is
restated here so that the package closes the loop sampling -> control without importing the reference."""
from __future__ import annotations

import numpy as np

from .pid import PIDController


def _heading(v) -> float:
    """Angle of v from the forward (y) axis in units of 90 degrees, controller.py:49-51."""
    return np.degrees(np.pi / 2 - np.arctan2(v[1], v[0])) / 90


class Controller:
    def __init__(self, cfg):
        p, c = cfg.PID, cfg.CONTROL
        self.turn_controller = PIDController(K_P=p.TURN_KP, K_I=p.TURN_KI, K_D=p.TURN_KD, n=p.TURN_N)
        self.speed_controller = PIDController(K_P=p.SPEED_KP, K_I=p.SPEED_KI, K_D=p.SPEED_KD, n=p.SPEED_N)
        self.aim_dist, self.angle_thresh, self.dist_thresh = c.AIM_DIST, c.ANGLE_THRESH, c.DIST_THRESH
        self.brake_speed, self.brake_ratio = c.BRAKE_SPEED, c.BRAKE_RATIO
        self.clip_delta, self.max_throttle = c.CLIP_DELTA, c.MAX_THROTTLE

    def control_pid(self, waypoints, velocity, target):
        """waypoints [T, 2] and target [2] in the ego frame, velocity [1] (tensors, any device)."""
        wp = waypoints.data.cpu().numpy()
        tgt = target.data.cpu().numpy()
        pairs = len(wp) - 1
        # desired speed: mean segment length x 2, accumulated pair by pair in the reference's order
        desired_speed = 0
        for i in range(pairs):
            desired_speed += np.linalg.norm(wp[i + 1] - wp[i]) * 2.0 / pairs
        # aim point: start of the segment whose midpoint lies closest to AIM_DIST (first one wins ties)
        aim, best = wp[0], 1e5
        for i in range(pairs):
            norm = np.linalg.norm((wp[i + 1] + wp[i]) / 2.0)
            if abs(self.aim_dist - best) > abs(self.aim_dist - norm):
                aim, best = wp[i], norm
        angle, angle_last, angle_target = _heading(aim), _heading(wp[-1] - wp[-2]), _heading(tgt)
        to_target = np.abs(angle_target) < np.abs(angle)
        to_target = to_target or (np.abs(angle_target - angle_last) > self.angle_thresh and tgt[1] < self.dist_thresh)
        steer = np.clip(self.turn_controller.step(angle_target if to_target else angle), -1.0, 1.0)

        speed = velocity[0].data.cpu().numpy()
        brake = desired_speed < self.brake_speed or (speed / desired_speed) > self.brake_ratio
        delta = np.clip(desired_speed - speed, 0.0, self.clip_delta)
        throttle = np.clip(self.speed_controller.step(delta), 0.0, self.max_throttle)
        throttle = throttle if not brake else 0.0
        return throttle, steer, brake


def post_process_control(throttle_res, steer_res, brake_res):
    """e2e_driving/diffusion_agent.py:268-277."""
    if brake_res < 0.05:
        brake_res = 0.0
    if throttle_res > brake_res:
        brake_res = 0.0
    if brake_res > 0.5:
        throttle_res = float(0)
    return throttle_res, steer_res, brake_res

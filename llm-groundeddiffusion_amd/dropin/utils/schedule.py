"""utils/schedule.py of the reference (host side)."""
import warnings

import torch


def get_fast_schedule(origial_timesteps, fast_after_steps, fast_rate):
    """schedule.py:4-8: stride `fast_rate` after `fast_after_steps`."""
    if fast_after_steps >= len(origial_timesteps) - 1:
        return origial_timesteps
    return torch.cat((origial_timesteps[:fast_after_steps], origial_timesteps[fast_after_steps + 1::fast_rate]), dim=0)


def dynamically_adjust_inference_steps(scheduler, index, t):
    """schedule.py:10-19: keep DDIM's prev_t = t - 1000//n consistent with an irregular schedule."""
    prev_t = scheduler.timesteps[index + 1] if index + 1 < len(scheduler.timesteps) else -1
    scheduler.num_inference_steps = scheduler.config.num_train_timesteps // int(t - prev_t)
    if index + 1 < len(scheduler.timesteps):
        if scheduler.config.num_train_timesteps // scheduler.num_inference_steps != t - prev_t:
            warnings.warn(f"({scheduler.config.num_train_timesteps} // {scheduler.num_inference_steps}) != ({t} - {prev_t}), so the step sizes may not be accurate")
    elif scheduler.config.num_train_timesteps // scheduler.num_inference_steps > t - prev_t:
        warnings.warn(f"({scheduler.config.num_train_timesteps} // {scheduler.num_inference_steps}) > ({t} - {prev_t}), so the step sizes may not be accurate")

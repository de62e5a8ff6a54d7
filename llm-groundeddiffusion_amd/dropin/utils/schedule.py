"""`utils.schedule` of the reference (optional fast tail of the per-box generations) — adapters over
lgd_amd.scheduler.DDIMScheduler, whose `fast_schedule` / `dynamic_step_sizes` are pinned bit-exactly against
the reference's own functions (tests/test_schedule.py)."""
import warnings

from lgd_amd.scheduler import DDIMScheduler as _Sched


def get_fast_schedule(origial_timesteps, fast_after_steps, fast_rate):
    """Same (misspelt) argument names as schedule.py:4."""
    return _Sched.fast_schedule(origial_timesteps, fast_after_steps, fast_rate)


def dynamically_adjust_inference_steps(scheduler, index, t):
    """schedule.py:10-19: sets `scheduler.num_inference_steps` so that DDIM's own `prev_t = t - N // n` lands
    on the next timestep of an irregular schedule (N = num_train_timesteps); warns when it cannot."""
    ts = scheduler.timesteps
    last = index + 1 >= len(ts)
    gap = int(t) - (-1 if last else int(ts[index + 1]))
    n_train = scheduler.config.num_train_timesteps
    scheduler.num_inference_steps = n_train // gap
    step = n_train // scheduler.num_inference_steps
    if (step > gap) if last else (step != gap):
        warnings.warn(f"DDIM step {step} ({n_train} // {scheduler.num_inference_steps}) does not match the "
                      f"schedule gap {gap} at index {index}; step sizes may not be accurate")

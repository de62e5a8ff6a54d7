"""DDIM scheduler (eta = 0) as the reference configures it ([ext] diffusers 0.18.0 DDIMScheduler with
the SD scheduler_config: scaled_linear betas 0.00085..0.012, 1000 train steps, steps_offset=1,
set_alpha_to_one=False, clip_sample=False; used at models/pipelines.py:150,196,221,357,443,545,583).

Keeps the attribute surface the reference touches (`timesteps`, `alphas_cumprod`, `init_noise_sigma`,
`num_inference_steps`, `config.num_train_timesteps`, `scale_model_input`, `set_timesteps`, `step`) so
utils/schedule.py works unchanged, and exports the per-step coefficient table the fused HIP step
kernel reads (`coef_table`).
"""
import numpy as np
import torch


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, prediction_type="epsilon"):
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule="scaled_linear", steps_offset=steps_offset,
                           prediction_type=prediction_type, clip_sample=False, set_alpha_to_one=False)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts) + self.config.steps_offset

    def prev_timestep(self, t, index=None):
        return int(t) - self.config.num_train_timesteps // self.num_inference_steps

    def alpha_pair(self, t):
        prev_t = self.prev_timestep(t)
        a_t = float(self.alphas_cumprod[int(t)])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p

    def step(self, model_output, timestep, sample):
        """Host/torch form (used by the hook-compatible slow path and by tests)."""
        a_t, a_p = self.alpha_pair(timestep)
        if self.config.prediction_type == "epsilon":
            x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
            e = model_output
        else:
            x0 = a_t ** 0.5 * sample - (1 - a_t) ** 0.5 * model_output
            e = a_t ** 0.5 * model_output + (1 - a_t) ** 0.5 * sample

        class _O:
            pass
        o = _O()
        o.prev_sample = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e
        return o

    def coef_table(self, guidance_scale: float, device, timesteps=None, step_ratios=None) -> torch.Tensor:
        """fp32 [T][4] = {alpha_bar_t, alpha_bar_prev, guidance_scale | sqrt(1-alpha_bar_t), v_pred}.

        Column 2 holds the CFG scale for the step kernel; the guidance update (pipelines.py:62-69)
        uses sqrt(1 - alpha_bar_t), exported separately by `guidance_step_table`."""
        ts = self.timesteps if timesteps is None else timesteps
        rows = []
        n = len(ts)
        for i, t in enumerate(ts):
            t = int(t)
            if step_ratios is not None:
                prev_t = t - step_ratios[i]
            else:
                prev_t = self.prev_timestep(t)
            a_t = float(self.alphas_cumprod[t])
            a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
            rows.append([a_t, a_p, guidance_scale, 1.0 if self.config.prediction_type == "v_prediction" else 0.0])
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def guidance_step_table(self, device, timesteps=None) -> torch.Tensor:
        """fp32 [T][4] with column 0 = sqrt(1 - alpha_bar_t): DDIM has no `sigmas`, so the latent
        update of backward guidance is scaled this way (pipelines.py:62-69)."""
        ts = self.timesteps if timesteps is None else timesteps
        rows = [[float((1 - self.alphas_cumprod[int(t)]) ** 0.5), 0.0, 0.0, 0.0] for t in ts]
        return torch.tensor(rows, dtype=torch.float32, device=device)

    # ---- utils/schedule.py (the optional fast tail of the per-box generations)
    @staticmethod
    def fast_schedule(timesteps, fast_after_steps, fast_rate=2):
        """schedule.py:4-8: keep the first `fast_after_steps` timesteps, then every `fast_rate`-th."""
        if fast_after_steps >= len(timesteps) - 1:
            return timesteps
        return torch.cat((timesteps[:fast_after_steps], timesteps[fast_after_steps + 1::fast_rate]), dim=0)

    def dynamic_step_sizes(self, timesteps):
        """schedule.py:10-12 followed by the scheduler's own prev_timestep rule: before each step the
        reference sets num_inference_steps = N_train // (t - next_t) (next_t = -1 after the last one) and
        DDIM then steps by N_train // num_inference_steps.  Returns that step size per index."""
        n_train = self.config.num_train_timesteps
        out = []
        for i, t in enumerate(timesteps):
            nxt = int(timesteps[i + 1]) if i + 1 < len(timesteps) else -1
            out.append(n_train // (n_train // (int(t) - nxt)))
        return out

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



class DPMSolverMultistepScheduler(DDIMScheduler):
    """[ext] diffusers 0.18.0 DPMSolverMultistepScheduler with its defaults (algorithm_type "dpmsolver++",
    solver_order 2, solver_type "midpoint", lower_order_final, no thresholding) — what `load_sd(...,
    use_dpm_multistep_scheduler=True)` selects (models/models.py:46-47).  Restated from the published algorithm
    (DPM-Solver++, Lu et al. 2022, eq. 2M) — diffusers is absent from the sandbox: parity unpinned at this boundary;
    the first-order case is pinned against DDIM, which it equals identically (tests/test_schedule.py).

    The update is linear in (x, x0, x0_prev), so the device side is the fused `lgd_cfg_multistep_step_f32` kernel
    reading one coefficient row per step (`multistep_table`); nothing about the captured hipGraphs changes.
    Backward guidance scales its latent update by sqrt(1 - alpha_bar_t) as with DDIM: the 0.18.0 class has no
    `sigmas` attribute unless Karras sigmas are switched on (pipelines.py:60-69)."""
    multistep = True

    def __init__(self, *a, solver_order=2, lower_order_final=True, **k):
        super().__init__(*a, **k)
        self.config.update(solver_order=solver_order, lower_order_final=lower_order_final,
                           algorithm_type="dpmsolver++", solver_type="midpoint")

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.config.num_train_timesteps
        ts = np.linspace(0, n - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def _als(self, t):
        a = float(self.alphas_cumprod[int(t)])
        alpha, sigma = a ** 0.5, (1.0 - a) ** 0.5
        return alpha, sigma, float(np.log(alpha) - np.log(sigma))

    def multistep_rows(self, timesteps=None):
        """Per step: (c0, c1, A, B, C) of  x0 = c0 x + c1 m ;  x' = A x + B x0 + C x0_prev."""
        ts = [int(t) for t in (self.timesteps if timesteps is None else timesteps)]
        n = len(ts)
        rows = []
        for i, t in enumerate(ts):
            t_next = ts[i + 1] if i + 1 < n else 0                          # the last step lands on timestep 0
            a_t, s_t, l_t = self._als(t)
            a_n, s_n, l_n = self._als(t_next)
            h = l_n - l_t
            if self.config.prediction_type == "v_prediction":
                c0, c1 = a_t, -s_t
            else:
                c0, c1 = 1.0 / a_t, -s_t / a_t
            first = (i == 0 or self.config.solver_order == 1 or
                     (i == n - 1 and self.config.lower_order_final and n < 15))
            A = s_n / s_t
            e = float(np.expm1(-h))                                          # exp(-h) - 1
            if first:
                B, C = -a_n * e, 0.0
            else:
                _, _, l_p = self._als(ts[i - 1])
                r = (l_t - l_p) / h
                B, C = -a_n * e * (1.0 + 0.5 / r), 0.5 * a_n * e / r
            rows.append((c0, c1, A, B, C))
        return rows

    def multistep_table(self, guidance_scale: float, device, timesteps=None) -> torch.Tensor:
        rows = [[c0, c1, A, B, C, guidance_scale, 0.0, 0.0] for c0, c1, A, B, C in self.multistep_rows(timesteps)]
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def coef_table(self, guidance_scale, device, timesteps=None, step_ratios=None):
        raise RuntimeError("DPMSolverMultistepScheduler drives lgd_cfg_multistep_step_f32 (multistep_table)")

    def prev_timestep(self, t, index=None):
        raise RuntimeError("multistep scheduler: the next timestep is the next entry of `timesteps`")

    def step_host(self, model_output, index, sample, x0_prev=None):
        """Torch form of one step (tests): returns (prev_sample, x0)."""
        c0, c1, A, B, C = self.multistep_rows()[index]
        x0 = c0 * sample + c1 * model_output
        out = A * sample + B * x0 + (C * x0_prev if C != 0.0 else 0.0)
        return out, x0


class EulerDiscreteScheduler:
    """[ext] diffusers EulerDiscreteScheduler as the SDXL refiner configures it (scheduler_config of
    stabilityai/stable-diffusion-xl-refiner-1.0: scaled_linear betas 0.00085..0.012, 1000 train steps, timestep_spacing
    "leading", steps_offset 1, epsilon prediction, linear sigma interpolation, no Karras sigmas, s_churn 0) — the sampler
    behind generation/sdxl_refinement.py:29.  diffusers is absent from the sandbox: restated from the published
    algorithm (Karras et al. 2022, Algorithm 2 without churn), parity unpinned at this boundary.

    In sigma space (x = x0 + sigma * eps) one Euler step is linear in (x, x0):
        x0 = x - sigma eps ;  x' = x + (sigma' - sigma) (x - x0) / sigma = (sigma'/sigma) x + (1 - sigma'/sigma) x0
    so the device side is the fused `lgd_cfg_multistep_step_f32` kernel with rows {1, -sigma, sigma'/sigma,
    1 - sigma'/sigma, 0, guidance_scale, c_in, 0}; c_in = 1/sqrt(sigma^2 + 1) (scale_model_input) is applied by
    `lgd_scale_rows_f32` from the same row."""
    multistep = True
    C_IN_COL = 6

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 prediction_type="epsilon"):
        if prediction_type != "epsilon":
            raise NotImplementedError("EulerDiscreteScheduler: the refiner predicts epsilon")
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule="scaled_linear", steps_offset=steps_offset, prediction_type=prediction_type,
                           timestep_spacing="leading", interpolation_type="linear", use_karras_sigmas=False)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_inference_steps = None
        self.timesteps = None
        self.sigmas = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.config.num_train_timesteps
        ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy().astype(np.float32)
        ts += self.config.steps_offset
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)

    @property
    def init_noise_sigma(self):
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)                   # "leading" spacing

    def img2img_start(self, num_inference_steps, strength):
        """StableDiffusionXLImg2ImgPipeline.get_timesteps (no denoising_start): index of the first step that runs."""
        init = min(int(num_inference_steps * strength), num_inference_steps)
        return max(num_inference_steps - init, 0)

    def index_of(self, t):
        return int((self.timesteps == float(t)).nonzero()[0])

    def scale_model_input(self, sample, timestep):
        s = float(self.sigmas[self.index_of(timestep)])
        return sample / ((s * s + 1) ** 0.5)

    def add_noise(self, original, noise, timestep):
        return original + noise * float(self.sigmas[self.index_of(timestep)])

    def multistep_rows(self, first=0):
        rows = []
        for i in range(first, len(self.timesteps)):
            s, sn = float(self.sigmas[i]), float(self.sigmas[i + 1])
            rows.append((1.0, -s, sn / s, 1.0 - sn / s, 0.0, 1.0 / (s * s + 1.0) ** 0.5))
        return rows

    def multistep_table(self, guidance_scale: float, device, first=0) -> torch.Tensor:
        rows = [[c0, c1, A, B, C, guidance_scale, c_in, 0.0] for c0, c1, A, B, C, c_in in self.multistep_rows(first)]
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def step_host(self, model_output, index, sample):
        """Torch form of one step exactly as the scheduler class writes it (tests)."""
        s, sn = self.sigmas[index], self.sigmas[index + 1]
        pred_original = sample - s * model_output
        return sample + (sample - pred_original) / s * (sn - s)

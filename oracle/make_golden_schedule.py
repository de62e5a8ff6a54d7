"""Golden vectors for the optional fast schedule, produced by the reference's own
utils/schedule.py (unmodified, imported from /root/reference) driving the stub DDIMScheduler the oracle
harness uses ([ext] diffusers 0.18.0 arithmetic, oracle/stubs/diffusers/schedulers).

    python oracle/make_golden_schedule.py        # writes tests/golden/schedule_fast.json

Runs only in the build container (needs /root/reference); the JSON is committed."""
import importlib.util, json, os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "stubs"))
from diffusers.schedulers import DDIMScheduler  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_schedule", "/root/reference/utils/schedule.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

cases = []
for T in (50, 30, 10):
    for fast_after in (None, 0, 5, 10, 25, 30, T - 2, T - 1, T + 3):
        sch = DDIMScheduler()
        sch.set_timesteps(T)
        if fast_after is not None:
            sch.timesteps = ref.get_fast_schedule(sch.timesteps, fast_after, 2)
        rows = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for index, t in enumerate(sch.timesteps):
                ref.dynamically_adjust_inference_steps(sch, index, t)          # pipelines.py:217-218,439-440
                t = int(t)
                prev_t = t - sch.config.num_train_timesteps // int(sch.num_inference_steps)
                a_t = float(sch.alphas_cumprod[t])
                a_p = float(sch.alphas_cumprod[prev_t]) if prev_t >= 0 else float(sch.final_alpha_cumprod)
                rows.append([t, prev_t, a_t, a_p])
        cases.append(dict(T=T, fast_after_steps=fast_after, fast_rate=2, steps=rows))
out = os.path.join(HERE, "..", "tests", "golden", "schedule_fast.json")
json.dump(cases, open(out, "w"))
print("wrote", out, len(cases), "cases")

"""ORACLE TEST INFRASTRUCTURE — minimal stand-in for `diffusers==0.18.0` (requirements.txt:5 of the
reference), which is not installed in this sandbox and has no source under /root/reference.

Only the symbols the reference's hot-path modules import are provided, restated from the published
diffusers 0.18.0 behaviour: ResnetBlock2D / Downsample2D / Upsample2D,
Timesteps / TimestepEmbedding, DDIMScheduler, ModelMixin / ConfigMixin / register_to_config.
Parity at this boundary is UNPINNED (no reference test holds golden vectors for it).

Never imported by the product (lgd_amd); used only by oracle/ref_harness.py to run the reference's
own unmodified models/*.py, utils/*.py on CPU.
"""
from . import schedulers  # noqa: F401
from .schedulers import DDIMInverseScheduler, DDIMScheduler, DPMSolverMultistepScheduler  # noqa: F401


class AutoencoderKL:  # only referenced by models/models.py:41 (load_sd), never built in the harness
    @classmethod
    def from_pretrained(cls, *a, **k):
        raise RuntimeError("no weights in the sandbox")


__version__ = "0.18.0-oracle-stub"

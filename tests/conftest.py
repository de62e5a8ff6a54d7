import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def gate(name, value, limit, *, at_least=False):
    """Parity gate that PRINTS what it measured next to its limit (pytest -rP shows it), so that a limit can be kept
    within 3x of the measurement it guards: `value < limit`, or `value > limit` with at_least (cosines)."""
    ok = value > limit if at_least else value < limit
    slack = ((1.0 - limit) / max(1.0 - value, 1e-30)) if at_least else (limit / max(value, 1e-30))
    print(f"[gate] {name}: {value:.4e} (limit {'>' if at_least else '<'} {limit:.4e}, slack x{slack:.1f})")
    assert ok, f"{name}: {value:.4e} violates {'>' if at_least else '<'} {limit:.4e}"
    return value

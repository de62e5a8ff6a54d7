"""`utils` package of the reference (utils/__init__.py: `from .utils import *`), HIP-backed where the
hot path needs it; everything else falls through to the reference's own files when available."""
import os as _os

from .utils import *  # noqa: F401,F403

_ref = _os.environ.get("LGD_REFERENCE_ROOT")
if _ref and _os.path.isdir(_os.path.join(_ref, "utils")):
    __path__.append(_os.path.join(_ref, "utils"))   # parse, cache, vis, llm, eval ... (out of scope)

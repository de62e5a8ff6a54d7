"""`models` package of the reference (models/__init__.py: `from .models import *`)."""
import os as _os

from .models import *  # noqa: F401,F403

_ref = _os.environ.get("LGD_REFERENCE_ROOT")
if _ref and _os.path.isdir(_os.path.join(_ref, "models")):
    __path__.append(_os.path.join(_ref, "models"))   # sam.py etc. (out of scope) resolve to the reference

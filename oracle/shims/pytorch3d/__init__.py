"""Import shim (build container only): pytorch3d v0.7.2 is not installed; the four transforms GenPose uses are
restated in oracle/rot.py from their published algorithm (see that file's header)."""
from . import transforms  # noqa: F401
from . import io  # noqa: F401

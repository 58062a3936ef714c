"""MI355X-native WorldMirror forward pass (HIP kernels behind a C ABI)."""
from .config import WMConfig, param_spec  # noqa: F401

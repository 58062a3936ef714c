"""MI355X-native WorldMirror forward pass (hand-written HIP kernels behind a C ABI)."""
from .config import WMConfig, param_spec  # noqa: F401
from .worldmirror import WorldMirror, extract_priors  # noqa: F401
from .geometry import create_confidence_mask, depth_to_world_coords_points  # noqa: F401
from .ingest import load_and_preprocess_images, preprocess_rgb  # noqa: F401
from .rasterization import Rasterizer  # noqa: F401

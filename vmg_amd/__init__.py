"""vmg_amd: MI355X-native VMG hot path (hand-written HIP kernels behind the reference's nn.Module surface)."""
from .create import create_model  # noqa: F401
from .model import VMG, SPyNet  # noqa: F401

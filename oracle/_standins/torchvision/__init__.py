"""Stand-in for `torchvision` (only deform_conv2d is named by the reference; it is never called on the configured path)."""
from . import ops

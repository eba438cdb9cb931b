"""vmg_amd: MI355X-native VMG hot path (HIP kernels behind the reference nn.Module surface)."""

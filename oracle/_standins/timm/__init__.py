"""Stand-in for the `timm` names the reference imports (test infrastructure only; never shipped in the product path)."""

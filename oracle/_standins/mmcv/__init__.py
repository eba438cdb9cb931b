"""Stand-in for the `mmcv` names the reference imports (test infrastructure only)."""
__version__ = '1.3.18'

"""Stand-in for scikit-image (absent, version not pinned by the reference): only skimage.metrics.peak_signal_noise_ratio is
used (tools/Tester.py:208).  TEST INFRASTRUCTURE ONLY."""

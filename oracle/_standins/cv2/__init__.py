"""Stand-in for OpenCV (absent from the image): the reference's utils/ package imports it at module level; none of the
functions used for fixtures (tools/Tester.py test_image / test_clips / test_clips_max) calls into it.  TEST INFRASTRUCTURE ONLY."""

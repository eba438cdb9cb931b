"""Stand-in for fvcore (absent): tools/Tester.py imports FlopCountAnalysis / parameter_count_table at module level only."""

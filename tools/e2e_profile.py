import cProfile, pstats, sys, os, runpy
sys.argv = ["tools/e2e_driver_bench.py", "512", "bf16"]
cProfile.run("runpy.run_path('tools/e2e_driver_bench.py', run_name='__main__')", "/tmp/e2e.prof")
p = pstats.Stats("/tmp/e2e.prof"); p.sort_stats("cumulative").print_stats(35)

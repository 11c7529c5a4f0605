import cProfile, pstats, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py", "--steps", "200", "--warmup", "10", "--cpu-rays", "0", "--render-frames", "0"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(sys.path[0], "bench.py"), run_name="__main__")
finally:
    pr.disable()
    st = pstats.Stats(pr).sort_stats("cumulative")
    st.print_stats(45)

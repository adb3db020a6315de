"""Stream ceiling of the GPU with this library's copy / read kernels for a few grid sizes.
usage: python tools/exp_ceiling.py"""
import os, subprocess, sys, json
if len(sys.argv) > 1:
    from dddmr_navigation_amd import scenes
    from dddmr_navigation_amd.local_planner import LocalPlanner
    sc = scenes.bench_scene("C1")
    with LocalPlanner([sc.theory]) as lp:
        print(sys.argv[1], ["%.0f" % v for v in lp.stream_ceiling(1 << 30, 10)], ["%.0f" % v for v in lp.stream_ceiling(1 << 31, 6)])
else:
    for b in (1024, 2048, 4096, 8192, 16384, 65536):
        subprocess.call([sys.executable, __file__, str(b)], env=dict(os.environ, DDDMR_CEIL_BLOCKS=str(b)))

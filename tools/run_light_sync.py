"""bench.py with HyperStarcopUNet.light_stream_sync on (1) / off (0), for traced same-box comparisons:
rocprofv3 --kernel-trace ... -- python tools/run_light_sync.py 1"""
import os, sys, runpy
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import starcop_amd.network as n
n.HyperStarcopUNet.light_stream_sync = (sys.argv[1] == "1")
sys.argv = [os.path.join(R, 'bench.py'), '--steps', '16', '--warmup', '4', '--no-cpu-baseline', '--no-extras']
runpy.run_path(os.path.join(R, 'bench.py'), run_name='__main__')

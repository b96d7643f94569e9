import sys, runpy
sys.path.insert(0, ".")
import generativedensification_amd.rasterizer as R
R.K.DEEP_MAX_BUSY = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")

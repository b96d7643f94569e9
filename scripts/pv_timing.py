import sys, os, time, atexit
sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd())
from generativedensification_amd import viewgroup as G, rasterizer as R
pc = time.perf_counter
import gc
if os.environ.get("NOGC"): gc.disable()
_gc = {"t": 0.0, "n": [0, 0, 0], "tt": [0.0, 0.0, 0.0], "start": 0.0}
def _gc_cb(phase, info):
    if phase == "start": _gc["start"] = pc()
    else:
        dt = pc() - _gc["start"]; g = info["generation"]; _gc["n"][g] += 1; _gc["tt"][g] += dt
gc.callbacks.append(_gc_cb)
atexit.register(lambda: print("gc collections per generation", _gc["n"], "seconds", [round(x, 4) for x in _gc["tt"]], "objects", len(gc.get_objects()), file=sys.stderr))
seg, cnt = {}, {}
def timed(name, fn):
    def wrap(*a, **k):
        t = pc(); r = fn(*a, **k); seg[name] = seg.get(name, 0) + pc() - t; cnt[name] = cnt.get(name, 0) + 1; return r
    return wrap
G._find_group = timed("find_group", G._find_group)
G._same_as_pairs = timed("same_as", G._same_as_pairs)
G._signature = timed("signature", G._signature)
R.forward_raw = timed("forward_raw", R.forward_raw)
R._CountReadback.wait = timed("readback_wait", R._CountReadback.wait)
R._RasterizeGaussians.backward = staticmethod(timed("classic_bwd", R._RasterizeGaussians.backward))
G._Hub.backward = staticmethod(timed("hub_bwd", G._Hub.backward))
G._GroupView.backward = staticmethod(timed("gv_bwd", G._GroupView.backward))
G.grouped_call = timed("grouped_call", G.grouped_call)
atexit.register(lambda: print({k: (round(v / cnt[k] * 1e6), cnt[k]) for k, v in seg.items()}, file=sys.stderr))
sys.argv = ["bench.py"] + sys.argv[1:]
exec(compile(open("bench.py").read(), "bench.py", "exec"))

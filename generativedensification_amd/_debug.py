"""Test and A/B knobs of the host side — NOT product configuration.

Everything here selects between code paths that give the same results: schedules that were measured and rejected but stay
reachable because a test compares them with the default (DESIGN.md section 3 has every measurement), the tested fallbacks
of the binning stage, and the overrides the parity tests use to reach rare paths (cut lists, the deep forward, a capacity
overflow) on small scenes.  The product reads none of them from its callers; the two switches a USER may need live elsewhere:
`rasterizer.DEPTH_TO_MEAN` (parity-risk switch R1) and `viewgroup.GROUP_VIEWS`.  Environment variables (GDR_*) preset the
knobs for the A/B scripts under scripts/; the library itself reads no environment variable.
"""
import os as _os


def _int_env(name):
    return int(_os.environ[name]) if _os.environ.get(name) else None


# ---- streams of a multi-view node ---------------------------------------------------------------------------------------
RENDER_SIDE = int(_os.environ.get("GDR_RENDER_SIDE", "1"))   # 0: everything on the caller's stream (bench.py's serial pass)
FWD_STREAMS = int(_os.environ.get("GDR_FWD_STREAMS", "4"))   # streams carrying the views' forward chains (binning + K6)
BWD_STREAMS = _int_env("GDR_BWD_STREAMS")                    # side streams of per-view K7 launches; None = side_count()
BIN_STREAM = None                                            # tests: force side_count() (None = by image size)
# K7 of a multi-view 3DGS node: 0 = one launch per view on side streams (rounds 1-3), 1 = ONE launch, the views interleaved
# (default, round 4: C2 +5-8 %, C3 +3-5 %, C4 +1.5 %), 2 = one launch, one view after the other; None = the default
K7_VIEWS = _int_env("GDR_K7_VIEWS")
EARLY_CLEAR = True        # gradient records cleared at the end of the forward, on the K7 streams

# ---- binning / cut lists / deep forward ---------------------------------------------------------------------------------
SEG_LEN = _int_env("GDR_SEG_LEN")          # None = policy (256; 512 on busy 800x800 images); 0 = lists are never cut
DEEP_MAX_BUSY = None                       # None = library default (768 busy tiles); 0 = never the deep forward
DEEP_MIN_MEAN = _int_env("GDR_DEEP_MIN_MEAN")
FORCE_GLOBAL_SORT = False                  # one global radix sort instead of tile partition + per-tile LDS sort (tested fallback)
FORCE_RADIX_PARTITION = False              # radix partition on the tile bits instead of the direct tile binning (tested fallback)
LAUNCH_HINTS = True                        # launch-size feedback between calls of a scene shape

# ---- the duplicate count --------------------------------------------------------------------------------------------------
DEFER_D = _os.environ.get("GDR_DEFER_D", "1") != "0"   # False: read the count back before sizing anything (upstream's flow)

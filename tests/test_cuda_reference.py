"""The CUDA-reference dump, consumed (round-4 verdict, missing #1 / next #8).

`scripts/dump_reference_cuda.py` run on a CUDA box with the reference's real extension writes
tests/golden/cuda_reference_*.npz.  When such a file is present: the oracle (this CPU test) and the HIP path (the -m gpu
twin) are compared with it under north_star's bars and the test prints which of the open conventions R1 / R3 / R4 the data
selects (tests/cuda_reference.py) — and FAILS if the data selects something else than what the product ships.  With no file
both skip with a message naming the script; the parity header of the oracle stays "unpinned".

The consumer itself is tested here either way: the dump script is run on the oracle stand-in (`--standin oracle`, CPU) into
a temporary directory and both backends are run against that synthetic dump — it must select the shipped conventions and
meet every bar (it is the oracle's own output).  A synthetic dump pins nothing and is never committed.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import cuda_reference as CR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMPS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cuda_reference_*.npz")))
SKIP_MSG = ("no tests/golden/cuda_reference_*.npz: run `python scripts/dump_reference_cuda.py --out "
            "tests/golden/cuda_reference_c1.npz` on a CUDA box with the reference's extension installed (parity stays unpinned)")


def _synthetic(tmp_path, n=3000, size=96, deg=2):
    out = str(tmp_path / "synthetic_dump.npz")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dump_reference_cuda.py"), "--standin", "oracle", "--out", out,
                          "--n", str(n), "--size", str(size), "--deg", str(deg), "--seed", "4"], capture_output=True, text=True,
                         env=env, cwd=ROOT, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return dict(np.load(out))


def _check(dump, backend, must_match_defaults=True):
    sel, ok, report = CR.compare(dump, backend)
    print("\n".join(report))
    print("selected conventions:", sel)
    assert ok, "\n".join(report)
    if must_match_defaults:
        assert {k: sel[k] for k in CR.DEFAULTS} == CR.DEFAULTS, sel
    return sel


@pytest.mark.skipif(not DUMPS, reason=SKIP_MSG)
@pytest.mark.parametrize("path", DUMPS or ["-"], ids=lambda p: os.path.basename(p))
def test_oracle_vs_cuda_reference_dump(oracle_built, path):
    dump = dict(np.load(path))
    assert "standin" not in str(dump.get("source", "cuda")), "a synthetic dump must not be committed as a cuda_reference_* fixture"
    _check(dump, CR.oracle_backend(CR.case_of(dump)))


def test_consumer_on_a_synthetic_dump_selects_the_shipped_conventions(oracle_built, tmp_path):
    dump = _synthetic(tmp_path)
    assert "standin" in str(dump["source"])
    _check(dump, CR.oracle_backend(CR.case_of(dump)))


def test_consumer_tells_the_variants_apart(oracle_built, tmp_path):
    """A dump from a fork with the OTHER conventions (normalised depth, no depth -> centre path, signed columns 2:4) is
    recognised as such — and fails the defaults check — instead of passing or failing for an unrelated reason."""
    dump = _synthetic(tmp_path)
    other = dict(dump)
    other["depth"] = dump["depth"] / np.maximum(dump["alpha"], 1e-10)
    sel, _, report = CR.compare(other, CR.oracle_backend(CR.case_of(other)))
    assert sel["R4"] == "normalized", report
    other = dict(dump)
    other["grad_color_only_means2D"] = np.concatenate([dump["grad_color_only_means2D"][:, :2]] * 2, axis=1)
    sel, _, report = CR.compare(other, CR.oracle_backend(CR.case_of(other)))
    assert sel["R3"] == "signed", report
    # R1 "no": the depth-only gradient of means3D without the centre path, from the oracle itself
    case = CR.case_of(dump)
    import torch
    up = [torch.from_numpy(dump[k]) for k in ("upstream_color", "upstream_depth", "upstream_alpha")]
    _, g0 = CR.oracle_backend(case)((torch.zeros_like(up[0]), up[1], torch.zeros_like(up[2])), False)
    other = dict(dump)
    other["grad_depth_only_means3D"] = g0["means3D"]
    sel, _, report = CR.compare(other, CR.oracle_backend(case))
    assert sel["R1"] == "no", report


@pytest.mark.gpu
@pytest.mark.skipif(not DUMPS, reason=SKIP_MSG)
@pytest.mark.parametrize("path", DUMPS or ["-"], ids=lambda p: os.path.basename(p))
def test_hip_vs_cuda_reference_dump(path):
    dump = dict(np.load(path))
    _check(dump, CR.hip_backend(CR.case_of(dump)))


@pytest.mark.gpu
def test_hip_consumer_on_a_synthetic_dump(oracle_built, tmp_path):
    dump = _synthetic(tmp_path, n=20_000, size=160, deg=3)
    _check(dump, CR.hip_backend(CR.case_of(dump)))

"""CPU replay of the tile kernel's addressing (tests/cpp/eval_tiles_test.cpp over amatsukaze_amd/csrc/eval_tiles.hpp):
every mask pixel's 5x5 window must read the samples CalcCorrelation5x5 reads (LogoScan.hpp:24-41), once, in raster order."""
import os
import subprocess

import numpy as np
import pytest

import amtlib
import amt_synth as S

ROOT = amtlib.ROOT


@pytest.fixture(scope="module")
def replay_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tiles") / "eval_tiles_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "amatsukaze_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "eval_tiles_test.cpp"), "-o", out])
    return out


def test_synthetic_masks(replay_bin):
    r = subprocess.run([replay_bin], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("kind,ratio", [("deint", 0.35), ("field", 0.35), ("deint", 0.1), ("deint", 1.0)])
def test_bench_logo_masks(replay_bin, tmp_path, kind, ratio):
    """the masks CreateLogoMask (LogoScan.hpp:112-229, through the oracle) gives the bench logo"""
    O = amtlib.Oracle()
    data, _, _ = S.make_logo(256, 128)
    hl = O.make_logo(data, 256, 128, 1440, 1080, 1120, 64)
    d = O.lib.orc_logo_deint(hl) if kind == "deint" else O.lib.orc_logo_field(hl, 0)
    O.lib.orc_logo_create_mask(d, ratio, 0)
    info = O.logo_info(d)
    w, h = int(info[0]), int(info[1])
    mask = O.logo_arrays(d)[1].reshape(h, w)
    ys, xs = np.nonzero(mask[2:h - 2, 2:w - 2])
    pos = ((ys + 2).astype(np.uint32) << 16) | (xs + 2).astype(np.uint32)
    fn = tmp_path / "pos.bin"
    with open(fn, "wb") as f:
        np.array([len(pos), w, h], np.int32).tofile(f)
        pos.astype(np.uint32).tofile(f)
    r = subprocess.run([replay_bin, str(fn)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

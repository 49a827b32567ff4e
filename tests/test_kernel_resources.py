"""The tape walkers stay off scratch (round 5: the float formatter of MarshalJSON kept its digits in byte buffers -- 36-132 B of
scratch per lane in every k_ms_tile variant -- and k_ser_scan_cnt spilled; sj_ftoa.h now keeps the digits in registers).
Compile-only: hipcc's resource remarks for marshal.hip and serialize.hip (tools/kernel_resources.py; the whole library:
profiles/r06_kernel_resources.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402


@pytest.mark.parametrize("src", ["marshal.hip", "serialize.hip"])
def test_no_scratch_and_occupancy_as_asked(src):
    rows = KR.kernels_of(src)
    assert len(rows) >= 8, rows
    for name, vgprs, scratch, occ, lds in rows:
        assert scratch == 0, (name, scratch)
        if name.startswith("k_ms_tile<"):  # k_ms_tile<MODE, waves per SIMD asked for, window>: the LDS window leaves at least that
            asked = int(name.split(",")[1])
            assert occ >= asked, (name, occ)

"""w2l_clock_probe (bench.py's sustained-clock measurement): the two counters it reports are the shader clock and the 100 MHz
reference clock of one spin, so their ratio is a plausible gfx950 clock and the reference ticks match the requested spin."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_clock_probe_reports_a_plausible_shader_clock(cuda):
    from wav2lip_amd import _lib
    lib = _lib.load()
    ticks = torch.zeros(2, dtype=torch.int64, device=cuda)
    _lib.check(lib.w2l_clock_probe(_lib.current_stream(), 2000, _lib.ptr(ticks)), "clock_probe")
    torch.cuda.synchronize()
    shader, ref = ticks.tolist()
    assert 200000 <= ref <= 260000, "a 2000 us spin is 200 000 ticks of the 100 MHz clock, got %d" % ref
    mhz = 100.0 * shader / ref
    assert 300.0 <= mhz <= 2600.0, "shader clock %.0f MHz out of the gfx950 range" % mhz


def test_clock_probe_rejects_bad_arguments(cuda):
    from wav2lip_amd import _lib
    lib = _lib.load()
    ticks = torch.zeros(2, dtype=torch.int64, device=cuda)
    assert lib.w2l_clock_probe(_lib.current_stream(), 0, _lib.ptr(ticks)) != 0
    assert lib.w2l_clock_probe(_lib.current_stream(), 1000, None) != 0

"""__graft_entry__ as the driver may call it: build() and smoke() in ONE fresh process.  build() loads libicd_amd.so; before round 3
it did so before anything had imported torch, which left /opt/rocm's HIP runtime and the torch wheel's bundled ROCm stack side by side
and made every later launch fail ("no ROCm-capable device is detected") - masked, on top, by a secondary "tensor ... is not bound"
message of the executor walking on after the failed launch.  _lib.load() now imports torch first and the executor keeps its first error."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE-OK')"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SMOKE-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_library_loaded_before_torch_by_the_caller_still_works():
    code = ("import sys; from invertible_cd_amd import _lib; _lib.load(); assert 'torch' in sys.modules; "
            "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SMOKE-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

"""The C++ host side (smarties_amd/host/vracer_hip.h) driven by a C++ program that reads like a smarties
learner test (tests/cpp/host_parity.cpp): same episodes into the HIP learner and the CPU oracle, sampled
indices bit-exact, weights / beta / generator state after 200 steps, Learner::select on a live agent
(action draw order), checkpoint round trip, die() behaviour."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_host_header_compiles_and_links():
    """CPU: the host class compiles against the C-ABI header and links with the two libraries."""
    import __graft_entry__ as ge
    ge.build_hip()
    ge.build_oracle()
    exe = ge.build_host_cpp()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_host_parity_program():
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "host_parity")
    if not os.path.exists(exe):
        import __graft_entry__ as ge
        exe = ge.build_host_cpp()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0 and "host_parity: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

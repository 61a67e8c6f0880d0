"""The library a product host ships (`make -C tendrils_amd/csrc release`) carries none of the test machinery: no in-process
transport (th_loopback.hip, th_comm_loopback_id), no fault injection (TH_OPT_INJECT_FAILURE) - and still every other entry
point include/tendrils_hip.h declares.  The default build (TESTING=1: what __graft_entry__.build() makes and the suites
run against) has both."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tendrils_amd", "csrc")
RELEASE = os.path.join(ROOT, "tendrils_amd", "lib", "release", "libtendrils_hip.so")
TESTING = os.path.join(ROOT, "tendrils_amd", "lib", "libtendrils_hip.so")


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(set(re.findall(r"\bT (th_[a-z0-9_]+)$", out, flags=re.M)))


def header_symbols(testing):
    text = open(os.path.join(ROOT, "include", "tendrils_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    if not testing:
        text = re.sub(r"#ifdef TH_TESTING.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(th_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def release():
    import __graft_entry__ as g
    if not os.path.exists(TESTING):
        g.build()
    # (the three units that read TH_TESTING are compiled again, every other object is the test build's: under a minute)
    subprocess.check_call(["make", "-j3", "-C", CSRC, "release"], stdout=subprocess.DEVNULL)
    return RELEASE


def test_release_library_exports_the_header_minus_the_test_machinery(release):
    want = header_symbols(testing=False)
    assert "th_comm_loopback_id" not in want and "th_draw_sharded" in want and len(want) >= 80
    assert exported(release) == want
    assert exported(TESTING) == header_symbols(testing=True) == sorted(want + ["th_comm_loopback_id"])


def test_release_library_holds_no_injection_and_no_loopback_code(release):
    blob, test_blob = open(release, "rb").read(), open(TESTING, "rb").read()
    for needle in (b"injected failure", b"TH_OPT_INJECT_FAILURE", b"TH_LOOPBACK_TIMEOUT_MS"):
        assert needle not in blob and needle in test_blob, needle


def test_ctypes_binding_loads_a_release_library(release):
    """tendrils_amd/_capi.py binds every prototype; the one symbol a release build lacks is skipped, nothing else may be"""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['TH_LIB'] = %r\n"
            "from tendrils_amd import _capi\n"
            "lib = _capi.load()\n"
            "assert not hasattr(lib, 'th_comm_loopback_id') and lib.th_abi_version() == 14\n"
            "print('ok')" % (ROOT, release))
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-1500:]

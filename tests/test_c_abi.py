"""include/ohw.h from C: tests/c/abi_smoke.c is compiled as C99 with -Wall -Werror against the header and linked with libohw.so
the way a C or Rust host links it (no ctypes in between).  The host-only part runs here; the GPU part builds a synthetic model
and runs mel -> encoder -> greedy through the staged API."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def smoke_binary(tmp_path_factory):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    from openhush_amd import engine as E
    E.lib()                                        # builds libohw.so when it is missing
    out = str(tmp_path_factory.mktemp("cabi") / "abi_smoke")
    lib_dir = os.path.join(ROOT, "openhush_amd")
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-L", lib_dir, "-lohw", f"-Wl,-rpath,{lib_dir}", "-lm", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_header_compiles_as_c99_and_host_entries_work(smoke_binary):
    r = subprocess.run([smoke_binary, "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_staged_api_and_engine_from_c_on_the_gpu(smoke_binary, tmp_models):
    r = subprocess.run([smoke_binary, "gpu", tmp_models("micro")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gpu ok" in r.stdout and "engine ok: 3 windows" in r.stdout, (r.returncode, r.stdout, r.stderr)

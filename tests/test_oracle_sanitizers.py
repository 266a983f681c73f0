"""The CPU oracle (oracle/rsim_oracle.c: the checker every parity claim rests on) under AddressSanitizer + UndefinedBehaviorSanitizer.  An out-of-bounds read in the
checker would make a parity test compare the kernel with garbage that happens to agree; this runs the oracle's own test file (controller replays against the
reference-recorded fixtures, documented-model known answers, narrow phase against elementary geometry, sensors, the Newton / PGS cross-check: 52 tests) on a
sanitized build of the same source, in a child interpreter with the ASan runtime preloaded.  (The whole CPU suite passes on that build too -- 139 tests, 140 s --
when run by hand: RSIM_ORACLE_LIB=<san.so> LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 pytest tests -m "not gpu"
--ignore tests/test_mjcf_cpp.py; the C++ compiler has its own sanitizer test, tests/test_mjcf_sanitizers.py.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")


def test_oracle_test_file_passes_on_an_asan_ubsan_build_of_the_oracle(tmp_path):
    so = str(tmp_path / "librsim_oracle_san.so")
    r = subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-std=gnu99", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                        "-shared", "-o", so, os.path.join(ROOT, "oracle", "rsim_oracle.c"), "-lm"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find" in r.stderr:
        pytest.skip("libasan / libubsan not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, RSIM_ORACLE_LIB=so, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    print(r.stdout.strip().splitlines()[-1])

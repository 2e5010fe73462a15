"""The checker itself is checked: oracle/rf_oracle.c under AddressSanitizer + UBSan, every entry
point on small odd-shaped inputs (tests/oracle_sanitize.c).  CPU only; skipped when the compiler
has no sanitizer runtime."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_sanitize")
    build = subprocess.run(
        [gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
         "-fno-omit-frame-pointer", "-fopenmp", os.path.join(ROOT, "tests", "oracle_sanitize.c"),
         os.path.join(ROOT, "oracle", "rf_oracle.c"), "-lm", "-o", exe],
        capture_output=True, text=True)
    if build.returncode != 0 and ("asan" in build.stderr or "ubsan" in build.stderr):
        pytest.skip("sanitizer runtime not installed: " + build.stderr.strip()[-200:])
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="2")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "0 failing call(s)" in run.stdout

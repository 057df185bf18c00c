"""Host-side sanitizer run of the C ABI glue: `make -C csrc asan` builds the HOST half
of every translation unit (argument validation, launch-geometry arithmetic, error strings) with
-fsanitize=address,undefined -- no kernels, CPU only -- and tests/capi_error_walk.py drives every entry point down its
NNS_ERR_* paths in a child process with the ASan runtime pre-loaded.  A sanitizer report aborts that process."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

CSRC = os.path.join(PKG, 'csrc')


def _asan_runtime():
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/clang', '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def test_error_paths_under_address_and_ub_sanitizer():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no clang ASan runtime in this image")
    b = subprocess.run(['make', '-C', CSRC, '-j', '8', 'asan'], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    lib = os.path.join(CSRC, 'libnns_hip_asan.so')
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'capi_error_walk.py'), lib], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:] + '\n' + r.stderr[-3000:])
    assert 'walked' in r.stdout and 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr

"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no
sanitizer or race tooling; the build adds a -fsanitize host build of the CPU restatement).  The golden-vector tests of
tests/test_oracle_golden.py run once more in a child process whose oracle is oracle/liboracle_san.so
(-fsanitize=address,undefined, -fno-sanitize-recover): an out-of-bounds index, a use-after-free or a signed overflow in
the restatement aborts the child."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        p = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_golden_vectors_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip('gcc has no libasan.so here')
    sys.path.insert(0, REPO)
    from oracle import oracle
    so = oracle.build_sanitized()
    env = dict(os.environ, HSK_ORACLE_SO=so, LD_PRELOAD=asan,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(REPO, 'tests', 'test_oracle_golden.py'), '-x', '-q',
                        '-p', 'no:cacheprovider'], env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert 'passed' in r.stdout and 'ERROR: AddressSanitizer' not in tail and 'runtime error' not in tail, tail

"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer run of the oracle, of the product's host-side table and pass
planning and of the kernel templates as the emulator executes them (reference: cmake/compilation-flags.cmake:24-57,
tests/pre-commit-script.sh:28-33).  Sanitizers cannot run on the GPU of this pool; this covers everything that
exists on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asan_ubsan_clean():
    d = os.path.join(ROOT, "tests", "sanitize")
    subprocess.check_call(["make", "-C", d, "san_main"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([os.path.join(d, "san_main")], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "sanitize: ok" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr

"""SURVEY 5 (race detection / sanitizers): the CPU code of this repo -- the C++ host layer above the C
ABI (hsearch_amd/host: file readers and writers, plane generator, Evaluate, the programs' option
handling) and the oracle's restatement -- under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU
sanitizers are not available on the pool, so this is the CPU job: the host programs are rebuilt with
-fsanitize=address,undefined into hsearch_amd/bin_san, the restatement into oracle/libhs_oracle_san.so,
and the CPU tests that drive them run again in a python started with the sanitizer runtime preloaded.
Any report aborts the child (halt_on_error / -fno-sanitize-recover) and fails this test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_and_oracle_under_asan_ubsan(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "gcc has no libasan.so"
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "hsearch_amd", "host"), "san"], check=True,
                   stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": asan,
        # python itself leaks by design; container overflow checks need every library instrumented
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_container_overflow=0",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
        "HS_ORACLE_LIB": os.path.join(ROOT, "oracle", "libhs_oracle_san.so"),
        "HS_HOST_BIN_DIR": os.path.join(ROOT, "hsearch_amd", "bin_san"),
    })
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_host_cli.py"),
                        os.path.join(ROOT, "tests", "test_tables.py")],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail

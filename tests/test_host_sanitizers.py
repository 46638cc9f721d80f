"""Host-side C code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not available on the
pool).  htk_amd/host/*.c is compiled with gcc -fsanitize=address,undefined into a library of its own (plus a stub for the error
string kept in the HIP part), and tests/host_sanitize_driver.py drives the readers / writers / editors through it in a child
process with the sanitizer runtime preloaded."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
static char g_err[1024];
void htkamd_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
const char *htkamd_last_error(void) { return g_err; }
long long htkamd_fb_frame_states(const void *fb) { (void)fb; return 0; }     /* capi.lib() sets its prototype at load time */
'''


def test_host_code_under_asan_ubsan(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    stub = tmp_path / "stub.c"; stub.write_text(STUB)
    lib = tmp_path / "libhtk_host_asan.so"
    srcs = sorted(glob.glob(os.path.join(ROOT, "htk_amd", "host", "*.c")))
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-fno-sanitize-recover=undefined", "-o", str(lib), str(stub)] + srcs + ["-lm"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_sanitize_driver.py"), str(lib)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])

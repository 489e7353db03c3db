"""Host-side C++ (graphgan_amd/csrc/{ingest,tree_builder,emb_writer,synth}.cpp) rebuilt with g++
-fsanitize=address,undefined and driven with well-formed and malformed inputs (tests/support/fuzz_host.py):
no heap overflow / UB on bad edge files, bad roots, short capacities, non-finite values."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graphgan_amd", "csrc")
HIP_INC = "/opt/rocm/include"

STUBS = r'''
#include <stdarg.h>
#include <stdio.h>
#include <string>
struct gg_ctx;
namespace gg { thread_local std::string g_last_error;
int fail(gg_ctx *, int code, const char *fmt, ...) { char b[1024]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); g_last_error = b; return code; } }
extern "C" int gg_get_embeddings(gg_ctx *, int, float *) { return -3; }
'''


def test_host_code_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan) or not os.path.isdir(HIP_INC):
        pytest.skip("g++ AddressSanitizer runtime or the HIP headers are not available")
    objs = []
    (tmp_path / "stubs.cpp").write_text(STUBS)
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-fPIC",
             "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    for name, src in [(n, os.path.join(CSRC, n + ".cpp")) for n in ("tree_builder", "emb_writer", "ingest", "synth")] + [("stubs", str(tmp_path / "stubs.cpp"))]:
        obj = str(tmp_path / (name + ".o"))
        r = subprocess.run(["g++"] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        objs.append(obj)
    lib = str(tmp_path / "libhost_asan.so")
    r = subprocess.run(["g++", "-shared", "-fsanitize=address,undefined", "-o", lib] + objs + ["-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "support", "fuzz_host.py"), lib], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "host sanitizer run: ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

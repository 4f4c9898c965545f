"""CPU build under AddressSanitizer + UndefinedBehaviorSanitizer (never on the GPU box's card: the sanitizers are for the host C++).

What can be compiled without HIP is exactly the host-side logic that handles caller-sized inputs: the restated libstdc++
selection (csrc/pp_topk_aten.h: pp_topk_aten_host, the kNN tie path) and the checkpoint rewrites (csrc/pp_rebalance.h:
pp_rebalance_weights_host, pp_plan_create), plus the std:: comparison harness of tests/test_topk_aten.py."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def gxx():
    cc = shutil.which("g++")
    if cc is None:
        pytest.skip("g++ not found")
    return cc


def test_host_code_under_asan_ubsan(gxx, tmp_path):
    exe = str(tmp_path / "host_sanitize")
    subprocess.run([gxx, *SAN, "-o", exe, os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")], check=True)
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout[-500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_topk_comparison_harness_under_asan_ubsan(gxx, tmp_path):
    """tests/native/topk_aten_check.cpp (the restatement next to the real std:: algorithms) built with the sanitizers and driven
    from a sanitized executable -- a shared object with ASan cannot be loaded into an unsanitized python."""
    main = tmp_path / "drive.cpp"
    main.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
extern "C" void mine_topk(const float *v, int n, int k, int *out);
extern "C" void std_topk(const float *v, int n, int k, int *out);
extern "C" void mine_piece(int which, const float *v, int n, int k, int *out);
extern "C" void std_piece(int which, const float *v, int n, int k, int *out);
int main() {
    std::mt19937 rng(5);
    for (int t = 0; t < 400; t++) {
        const int n = t % 3 ? 1 + (int)(rng() % 3000) : 2048 + (int)(rng() % 3000), k = n < 32 ? n : 32;
        const int levels = (int[]){2, 5, 20, 1000}[rng() % 4];
        std::vector<float> v(n);
        for (auto &x : v) x = 0.37f * (float)(rng() % levels);
        std::vector<int> a(n), b(n);
        mine_topk(v.data(), n, k, a.data()); std_topk(v.data(), n, k, b.data());
        for (int j = 0; j < k; j++) if (a[j] != b[j]) { std::printf("topk differs at row %d\n", t); return 1; }
        for (int which = 0; which < 4; which++) {
            mine_piece(which, v.data(), n, k, a.data()); std_piece(which, v.data(), n, k, b.data());
            for (int j = 0; j < n; j++) if (a[j] != b[j]) { std::printf("piece %d differs at row %d\n", which, t); return 1; }
        }
    }
    std::printf("ok\n");
    return 0;
}
''')
    exe = str(tmp_path / "topk_check")
    subprocess.run([gxx, *SAN, "-o", exe, str(main), os.path.join(ROOT, "tests", "native", "topk_aten_check.cpp")], check=True)
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout[-500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]

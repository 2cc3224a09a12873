"""mf_fastdiv (csrc/mf_fastdiv.h): the multiply-shift constants the launchers hand to the kernels in place of integer divisions.  Compiled with g++ and checked
against the C division: every divisor up to 5000 and a spread of larger ones (the UNet / VAE / Wav2Lip geometries: pixels per image, widths, channel quads, tile and
split counts), numerators at every multiple of the divisor +- 1 up to 2^31 - 1 plus a dense low range."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "mf_fastdiv.h"
#include <cstdio>
#include <cstdint>
static long bad = 0;
static void check(uint32_t d) {
    uint32_t mul, shr;
    mf_fastdiv(d, &mul, &shr);
    auto one = [&](int64_t n) {
        if (n < 0 || n > 0x7fffffffll) return;
        if (mf_fdiv_host((int)n, mul, shr) != (int)(n / d)) { if (bad++ < 5) std::printf("d %u n %lld\n", d, (long long)n); }
    };
    for (int64_t n = 0; n < 70000; ++n) one(n);
    const int64_t step = ((0x7fffffffll / d) / 4001 + 1) * (int64_t)d;
    for (int64_t q = 0; q <= 0x7fffffffll; q += step) { one(q - 1); one(q); one(q + 1); one(q + d - 1); }
    one(0x7fffffffll); one(0x7ffffffell);
}
int main() {
    for (uint32_t d = 1; d <= 5000; ++d) check(d);
    const uint32_t big[] = {8192, 9216, 16384, 36864, 65535, 65536, 65537, 147456, 262144, 524288, 1000003, 16777216, 0x3fffffffu, 0x40000000u, 0x40000001u, 0x7fffffffu};
    for (uint32_t d : big) check(d);
    std::printf("bad %ld\n", bad);
    return bad ? 1 : 0;
}
"""


def test_fastdiv_matches_integer_division(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mere-fusion_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-500:]

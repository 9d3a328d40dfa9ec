"""The library's one environment switch (tpl_amd/csrc/tplhip_debug.h): `TPLHIP_DEBUG="token[=int],..."`.  The parser is a
header-only host function, so it is compiled here with g++ into a tiny probe and driven with different values of the
variable (no GPU involved)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = r'''
#include <cstdio>
#include "tplhip_debug.h"
int main(int argc, char** argv)
{
    for (int i = 1; i < argc; ++i) {
        int v = -7;
        const bool hit = tplhip::debug_opt(argv[i], &v);
        std::printf("%s %d %d\n", argv[i], hit ? 1 : 0, v);
    }
    return 0;
}
'''


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    d = tmp_path_factory.mktemp("dbg")
    src = d / "probe.cpp"
    src.write_text(PROBE)
    exe = d / "probe"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tpl_amd", "csrc"), "-o", str(exe), str(src)])
    return str(exe)


def _ask(probe, value, *names):
    env = {k: v for k, v in os.environ.items() if k != "TPLHIP_DEBUG"}
    if value is not None:
        env["TPLHIP_DEBUG"] = value
    out = subprocess.check_output([probe, *names], env=env, text=True)
    return {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[2])) for ln in out.splitlines()}


def test_tokens_values_and_prefixes(probe):
    r = _ask(probe, "no_v3,tile_v=4,kchunk=3,ilqr_lds_kb=39", "no_v3", "tile_v", "kchunk", "ilqr_lds_kb", "no_s3", "tile", "tile_a", "no_v")
    assert r["no_v3"] == (1, -7)                  # present, no value: the caller's variable is left alone
    assert r["tile_v"] == (1, 4) and r["kchunk"] == (1, 3) and r["ilqr_lds_kb"] == (1, 39)
    assert r["no_s3"] == (0, -7) and r["tile_a"] == (0, -7)
    assert r["tile"] == (0, -7) and r["no_v"] == (0, -7)          # a prefix of a token is not the token


def test_unset_empty_and_odd_strings(probe):
    for value in (None, "", ",", ",,no_fastdiv,,"):
        r = _ask(probe, value, "no_fastdiv", "big")
        assert r["big"] == (0, -7)
        assert r["no_fastdiv"][0] == (1 if value and "no_fastdiv" in value else 0)
    r = _ask(probe, "big=0,big=1", "big")
    assert r["big"] == (1, 0)                     # the first occurrence wins
    r = _ask(probe, "big=", "big")
    assert r["big"] == (1, 0)                     # '=' with nothing behind it reads as 0

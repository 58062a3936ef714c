"""Run pytest in one process with wm_set_tuning keys applied first (the library's tuning table is process-global): the parity suites under an
opt-in kernel form.   usage: python tools/pytest_with_tuning.py key=val[,key=val] <pytest args ...>"""
import sys
sys.path.insert(0, '.')
import pytest
from hunyuanworld_mirror_amd import _lib
L = _lib.lib()
for kv in sys.argv[1].split(","):
    k, v = kv.split("=")
    assert L.wm_set_tuning(k.encode(), int(v)) == 0, kv
sys.exit(pytest.main(sys.argv[2:]))

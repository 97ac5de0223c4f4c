"""CPU: the C-ABI library builds, loads and exports every symbol include/gsx.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_pkg


def header_functions():
    src = open(os.path.join(ROOT, "include", "gsx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    g = load_pkg()
    g.build()
    lib = ctypes.CDLL(g._lib.SO_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsx.h but not exported"
    # and the ctypes binding covers exactly the header
    assert names == g._lib.declared_symbols()
    assert g.lib().gsx_abi_version() == 2


def test_camera_struct_layout():
    g = load_pkg()
    assert ctypes.sizeof(g.Camera) == 8 * 2 + 4 * 2 + 8 * 9 + 8 * 3
    c = g.Camera.from_dict({"fx": 1, "fy": 2.5, "width": 3, "height": 4, "rotation": [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                            "position": [7, 8, 9]})
    assert (c.fx, c.fy, c.width, c.height, c.R[4], c.p[2]) == (1.0, 2.5, 3, 4, 1.0, 9.0)
    with pytest.raises(ValueError):
        g.Camera.from_dict({"fx": 1, "fy": 1, "width": 1, "height": 1, "rotation": [1, 2, 3], "position": [0, 0, 0]})


def test_no_cpu_fallback_and_null_handling():
    """Without a gfx950 device the product refuses to run (no silent fallback)."""
    import torch
    g = load_pkg()
    lib = g.lib()
    assert lib.gsx_vote_begin(None, 150, 0, 1) == g._lib.GSX_E_INVALID
    assert b"NULL" in lib.gsx_last_error(None)
    if not torch.cuda.is_available():
        with pytest.raises(g.GsxError) as e:
            g.Context(0)
        assert e.value.code == g._lib.GSX_E_HIP and "no CPU fallback" in str(e.value)


def test_default_host_threads_shares_the_node_between_ranks():
    """gsx_default_host_threads: min(16, usable CPUs / ranks on this node); env GSX_HOST_THREADS wins.  One process per GPU."""
    import subprocess
    import sys
    g = load_pkg()
    g.build()
    code = ("import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); l.gsx_default_host_threads.restype=ctypes.c_int; "
            "print(l.gsx_default_host_threads())")

    def ask(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("GSX_HOST_THREADS", "LOCAL_WORLD_SIZE")}
        e.update(env)
        return int(subprocess.run([sys.executable, "-c", code, g._lib.SO_PATH], env=e, check=True, capture_output=True,
                                  text=True).stdout)

    cpus = len(os.sched_getaffinity(0))
    alone = ask()
    assert 1 <= alone <= min(16, cpus)   # a cgroup quota may lower it further
    assert ask(LOCAL_WORLD_SIZE="1") == alone
    shared = ask(LOCAL_WORLD_SIZE="8")
    assert 1 <= shared <= max(1, cpus // 8) and shared <= alone
    assert ask(LOCAL_WORLD_SIZE="100000") == 1
    assert ask(GSX_HOST_THREADS="3", LOCAL_WORLD_SIZE="8") == 3

"""cubemapslam_amd/host/batch_driver.cpp (bench.py's step without the interpreter): the library exports the driver's entry points, and bench.py's ctypes
descriptions of its plan structures have the sizes the library was compiled with -- a field added on one side only would otherwise show up as garbage
pointers on the GPU box.  No GPU: nothing is run."""
import ctypes as C
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_plan_structures_match_the_library():
    from cubemapslam_amd import api, build
    api.lib()                                             # (the host library links against the HIP one)
    L = C.CDLL(build.HOST_LIB)
    for name in ("cbd_create", "cbd_begin", "cbd_steps", "cbd_collect_inflight", "cbd_finish", "cbd_drain", "cbd_stats_get", "cbd_destroy", "cbd_sizes", "cbd_last_error"):
        assert hasattr(L, name), name
    VP, FrameSet, Group, STEP_DONE, Plan, Stats = _bench().cbd_types()
    sz = (C.c_int * 4)()
    L.cbd_sizes(sz)
    assert list(sz) == [C.sizeof(FrameSet), C.sizeof(Group), C.sizeof(Plan), C.sizeof(Stats)], list(sz)
    # a plan that names no groups is refused, not dereferenced
    L.cbd_create.restype = C.c_void_p
    p = Plan()
    assert L.cbd_create(C.byref(p), 1) is None

"""The C-ABI libraries load and export exactly what include/*.h declares; without a GPU the product fails loudly
(no CPU fallback).  CPU only — no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests.common import Case


def declared_functions(header):
    text = open(os.path.join(A.REPO_ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rbp_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_exports_every_symbol_of_rbp_h():
    L = planner.lib()
    names = declared_functions("rbp.h")
    assert set(names) == set(planner.EXPORTED_SYMBOLS)
    for n in names:
        assert getattr(L, n) is not None


def test_host_library_exports_every_symbol_of_rbp_host_h():
    L = host.lib()
    for n in declared_functions("rbp_host.h"):
        assert getattr(L, n) is not None


def test_param_defaults_match_param_hpp():
    p = A.rbp_param()
    planner.lib().rbp_param_defaults(C.byref(p))
    d = Param()  # param.hpp:44-70
    assert list(p.world_min) == [d.world_x_min, d.world_y_min, d.world_z_min]
    assert list(p.world_max) == [d.world_x_max, d.world_y_max, d.world_z_max]
    assert (p.box_xy_res, p.box_z_res, p.downwash, p.time_step) == (0.1, 0.1, 2.0, 1.0)
    assert (p.grid_xy_res, p.grid_z_res, p.grid_margin, p.ecbs_w) == (0.3, 0.6, 0.2, 1.3)
    assert (p.n, p.phi, p.sequential, p.batch_size, p.batch_iter, p.iteration, p.time_scale, p.log) == (5, 3, 0, 4, 0, 1, 1, 0)


def test_struct_layout_roundtrip():
    """the ctypes mirrors and the C structs agree on layout: the oracle (C) reads what Python wrote."""
    from tests import oracle_lib as O
    c = Case("c1_4agents_empty_joint")
    pr = c.inputs()
    rc, _ = O.corridor_update(c.world, c.mission, c.param, pr)
    assert rc == 0 and pr.sfc_count.min() >= 1


def test_product_has_no_cpu_fallback():
    if planner.lib().rbp_device_count() > 0:
        pytest.skip("a HIP device is present")
    c = Case("c1_4agents_empty_joint")
    pr = c.inputs()
    cor = planner.Corridor(c.world, c.mission, c.param)
    assert cor.update(False, pr) is False
    assert cor.rc == A.RBP_ERR_NO_DEVICE and "no CPU fallback" in cor.last_error
    pl = planner.RBPPlanner(c.mission, c.param)
    assert pl.update(False, c.with_corridor()) is False and pl.rc == A.RBP_ERR_NO_DEVICE
    assert np.all(pr.sfc_count == 0)  # nothing was computed
    from swarm_simulator_amd import host
    keys, res, _ = host.load_octomap("empty.bt")
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # the GPU distance grid (rbp_edt_build) neither
        planner.build_world(keys, res, c.param)


def test_product_does_not_import_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(A.REPO_ROOT, "swarm_simulator_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(root, f), errors="ignore").read()
                for needle in ("oracle_lib", "rbp_oracle", "oracle/", "import oracle", "from oracle"):
                    assert needle not in text, (f, needle)


def test_planner_call_without_corridor_is_refused():
    """RBPPlanner::update reads SFC/RSFC (rbp_planner.hpp:435-511): a plan without them is a caller error, reported as
    such before any device work (argument check only: runs without a GPU)."""
    c = Case("c1_4agents_empty_joint")
    pr = c.with_corridor()
    pr.rsfc_normal = None
    pl = planner.RBPPlanner(c.mission, c.param)
    assert pl.update(False, pr) is False
    assert pl.rc == A.RBP_ERR_BAD_ARGUMENT and "corridor" in pl.last_error


def test_abi_version_and_struct_sizes_match_the_binding():
    """rbp_plan / rbp_counters are written by the library: a binding built against another header must not run"""
    L = planner.lib()
    hdr = open(os.path.join(A.REPO_ROOT, "include", "rbp.h")).read()
    assert int(re.search(r"#define RBP_ABI_VERSION (\d+)", hdr).group(1)) == A.RBP_ABI_VERSION == L.rbp_abi_version()
    for which, t in enumerate((A.rbp_world, A.rbp_mission, A.rbp_param, A.rbp_plan, A.rbp_counters)):
        assert L.rbp_sizeof(which) == C.sizeof(t), t.__name__
    assert L.rbp_sizeof(99) == 0
    assert b"0.3" in L.rbp_version()
    L.rbp_release_thread_context()   # nothing to release: must be harmless without a device


def test_the_library_reads_no_environment_variables():
    """what a caller may choose about the solvers travels in rbp_solver_opts (include/rbp.h, ABI 4); the release library never calls
    getenv (the developer build of kernels/jqp.hip does, behind RBP_DEV_KNOBS: `make dev`)"""
    base = os.path.join(A.REPO_ROOT, "swarm_simulator_amd", "csrc")
    for root, _, files in os.walk(base):
        if os.sep + "lib" in root:
            continue
        for f in files:
            if not f.endswith((".hip", ".inc", ".cpp", ".h", ".hpp")):
                continue
            text = open(os.path.join(root, f), errors="ignore").read()
            # strip the developer block
            text = re.sub(r"#ifdef RBP_DEV_KNOBS.*?#endif", "", text, flags=re.S)
            code = "\n".join(l.split("//")[0] for l in text.splitlines())
            assert "getenv" not in code, f
    import subprocess
    so = os.path.join(A.LIB_DIR, "librbp_hip.so")
    syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    assert "getenv" not in syms.split(), "librbp_hip.so imports getenv"


def test_solver_opts_defaults_and_checks():
    o = planner.solver_opts()
    assert o.size == C.sizeof(A.rbp_solver_opts) == planner.lib().rbp_sizeof(6)
    assert (o.polish, o.joint_wide_min_agents, o.joint_corrector, o.joint_schedule) == (1, 16, 1, 0)
    assert (o.qp_schedule, o.qp_variant, o.qp_block_order, o.qp_groups, o.qp_rounds, o.qp_far_slack) == (0, 0, 1, 0, 0, 0.7)
    bad = planner.solver_opts()
    bad.size = 8
    assert planner.lib().rbp_session_set_solver_opts(None, C.byref(bad)) == A.RBP_ERR_BAD_ARGUMENT   # (null session)
    with pytest.raises(TypeError):
        planner.solver_opts(no_such_field=1)


def test_shard_joint_entry_point_without_a_gpu():
    """rbp_session_shard_joint (ABI 5) is exported with the hook type the binding declares; a null session is refused before any device
    work (the sharing itself is a GPU test: tests/test_gpu_joint_shard.py), and the error code of a failing hook has its text"""
    L = planner.lib()
    hook = planner.EXCHANGE_FN(lambda user, send, recv, nbytes: 0)
    assert L.rbp_session_shard_joint(None, 0, 2, hook, None) == A.RBP_ERR_BAD_ARGUMENT
    assert b"null session" in L.rbp_last_error()
    assert A.RBP_ERR_EXCHANGE == 32 and A.RBP_ERR_EXCHANGE in planner.ERROR_TEXT
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rbp.h")).read()
    assert "typedef int (*rbp_exchange_fn)(void* user, void* send_dev, void* recv_dev, size_t bytes);" in hdr
    assert "RBP_ERR_EXCHANGE = 32" in hdr


def test_rccl_library_exports_every_symbol_of_rbp_rccl_h():
    """lib/librbp_rccl.so (the RCCL exchange hook of a sharded joint solve; a library of its own so that the core does not link RCCL):
    every function include/rbp_rccl.h declares is exported, and the hook has the signature of rbp_exchange_fn.  Checked with nm: loading it
    here would bring a second HIP runtime into this process."""
    import shutil, subprocess
    so = os.path.join(A.LIB_DIR, "librbp_rccl.so")
    if not os.path.exists(so):
        pytest.skip("librbp_rccl.so not built (no RCCL headers where build() ran)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "rbp_rccl.h")).read()
    declared = set(re.findall(r"\b(rbp_rccl_[a-z_]+)\s*\(", hdr.split("#ifdef __cplusplus")[1]))
    assert declared == {"rbp_rccl_unique_id", "rbp_rccl_pair_create", "rbp_rccl_exchange", "rbp_rccl_pair_destroy", "rbp_rccl_last_error",
                        "rbp_rccl_pair_set_timeout", "rbp_rccl_exchange_stream", "rbp_rccl_abort"}
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.run([nm, "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (rbp_rccl_[a-z_]+)", syms))
    assert declared <= exported, declared - exported
    assert "int rbp_rccl_exchange(void* pair, void* send_dev, void* recv_dev, size_t bytes);" in hdr   # = rbp_exchange_fn of rbp.h
    core = subprocess.run([nm, "-D", os.path.join(A.LIB_DIR, "librbp_hip.so")], capture_output=True, text=True).stdout
    assert "nccl" not in core.lower()   # the core library stays free of RCCL

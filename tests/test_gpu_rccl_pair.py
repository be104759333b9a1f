"""include/rbp_rccl.h: the exchange hook of rbp_session_shard_joint over RCCL (lib/librbp_rccl.so).  A one-GPU box cannot host two RCCL ranks,
so what runs here is the plumbing with a pair of ONE rank (send to self): unique id, communicator, grouped ncclSend / ncclRecv of a vector, a
64-agent inverse and a 256-agent inverse on the pair's stream, bitwise comparison on the device (csrc/rccl/selftest.hip, a binary of its
own: the python process holds torch's bundled HIP runtime, RCCL brings /opt/rocm's)."""
import os
import subprocess

import pytest

from swarm_simulator_amd import _abi as A

pytestmark = pytest.mark.gpu


def test_rccl_exchange_hook_self_pair():
    exe = os.path.join(A.LIB_DIR, "rbp_rccl_selftest")
    if not os.path.exists(exe):
        pytest.skip("lib/rbp_rccl_selftest not built (no RCCL headers where build() ran)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and "rbp_rccl selftest ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.count("0 words differ") == 3

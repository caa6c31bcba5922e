"""psp_allreduce / psp_comm_* (include/psp.h) on the MI355X: the RCCL binding of the C ABI.

A one-GPU box can only form a single-rank communicator (RCCL refuses two ranks on one device), so this covers the
plumbing -- lazy librccl binding, id / init / in-place all-reduce on the caller's stream / destroy; the multi-rank path
is exercised by `bench.py --gpus N` (its warm-up cross-checks psp_allreduce against torch.distributed's all-reduce and
reports the outcome in the JSON line)."""
import ctypes as C

import pytest
import torch

from util_cases import psp

pytestmark = pytest.mark.gpu
nat = psp.native


def test_single_rank_communicator_allreduce_is_identity_on_the_stream():
    lib = nat.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ident = (C.c_ubyte * nat.COMM_ID_BYTES)()
    nat.check(lib.psp_comm_unique_id(ident), "psp_comm_unique_id")
    comm = C.c_void_p()
    nat.check(lib.psp_comm_init(C.byref(comm), 1, 0, ident), "psp_comm_init")
    assert comm.value
    try:
        g = torch.randn(17188, device=dev)
        s = torch.tensor([3.25, -7.5], dtype=torch.float64, device=dev)
        g0, s0 = g.clone(), s.clone()
        st = nat.stream_ptr(dev)
        nat.check(lib.psp_allreduce(nat.ptr(s), 2, nat.DT_F64, comm, st), "psp_allreduce f64")
        nat.check(lib.psp_allreduce(nat.ptr(g), g.numel(), nat.DT_F32, comm, st), "psp_allreduce f32")
        g.mul_(2.0)                                  # ordered after the collective on the same stream
        torch.cuda.synchronize()
        assert torch.equal(s, s0) and torch.equal(g, 2.0 * g0)
    finally:
        nat.check(lib.psp_comm_destroy(comm), "psp_comm_destroy")

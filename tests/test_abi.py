"""The C-ABI shared library loads and exports every symbol include/psp.h declares.
No compute calls here (no GPU needed): only size queries and argument validation."""
import ctypes as C
import os
import re

import pytest

from util_cases import psp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nat = psp.native


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "psp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not nat.is_built():
        import __graft_entry__
        __graft_entry__.build()
    return nat.load()


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert "psp_hjb_rollout_fwd" in syms and "psp_adam_step" in syms
    raw = C.CDLL(nat.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), s
    assert set(syms) == set(nat.SIGNATURES), (set(syms) ^ set(nat.SIGNATURES))


def test_version_and_support_table(lib):
    assert lib.psp_version() == 400
    assert nat.supported(100, 64) and nat.supported(2, 30)
    assert not nat.supported(3, 7)


def _cfg(d=100, H=64, K=1024, N=50):
    c = nat.HjbConfig()
    c.d, c.H, c.K_local, c.N, c.K_global = d, H, K, N, K
    c.dt, c.sqrt_dt = 0.01, 0.1
    c.drift_kind, c.sigma_kind = nat.DRIFT_DENSE, nat.SIGMA_DENSE
    c.adaptive, c.store_path = 1, 1
    return c


def test_query_sizes_without_gpu(lib):
    s = nat.query(_cfg())
    p = (100 + 1) * 64 + 64 + 64 * 64 + 64 + 64 * 100 + 100
    assert s.n_params == p == 17188
    # path store: N steps x K/16 tiles x (X, xi, h1, h2 register images padded to 16-feature blocks:
    # 2*4*ceil(d/16) + 2*4*ceil(H/16) k-steps) x 64 lanes x 4 B
    assert s.path_bytes == 50 * 64 * (2 * 28 + 2 * 16) * 64 * 4
    assert s.fwd_workgroups >= 1 and s.bwd_workgroups >= 1
    assert s.grad_partial_bytes == s.bwd_workgroups * p * 4


def test_query_reports_the_cooperative_wide_forward(lib, monkeypatch):
    """psp_hjb_sizes.fwd_coop_tiles (0.4.0): d > 160 in split-product mode with on-device noise runs hjbc_fwd_kernel -- four tiles per
    workgroup where that fills the chip, else two; the grid counts those workgroups; PSP_FWD_COOP=0 keeps the tile-per-wave kernel
    (host-only calls: no GPU needed)."""
    monkeypatch.delenv("PSP_FWD_COOP", raising=False)
    for d, K, want in ((500, 16384, 4), (500, 4096, 2), (200, 32768, 4), (200, 1024, 2), (100, 65536, 0), (128, 65536, 0)):
        c = _cfg()
        c.d, c.K_local, c.mlp_dtype, c.noise_mode = d, K, nat.MLP_F16X3, nat.NOISE_PHILOX
        s = nat.query(c)
        assert s.fwd_coop_tiles == want, (d, K, s.fwd_coop_tiles)
        if want:
            assert s.fwd_workgroups == -(-(K // 16) // want)
    c = _cfg()
    c.d, c.K_local, c.mlp_dtype, c.noise_mode = 500, 16384, nat.MLP_F16X3, nat.NOISE_PHILOX
    monkeypatch.setenv("PSP_FWD_COOP", "0")
    assert nat.query(c).fwd_coop_tiles == 0
    monkeypatch.setenv("PSP_FWD_COOP", "2")
    assert nat.query(c).fwd_coop_tiles == 2
    monkeypatch.delenv("PSP_FWD_COOP")
    c.noise_mode = nat.NOISE_SUPPLIED                    # the cooperative kernel generates its increments on the device
    assert nat.query(c).fwd_coop_tiles == 0


def test_unsupported_shape_reports_error(lib):
    with pytest.raises(nat.NativeCallError) as e:
        nat.query(_cfg(d=3, H=7))
    assert "no compiled HJB kernel instance" in str(e.value)


def test_null_buffers_are_rejected_before_any_launch(lib):
    c = _cfg()
    rc = lib.psp_hjb_rollout_fwd(C.byref(c), None, None, 0, None, None, 1, 0, None, None, None, None, None, None)
    assert rc != 0 and "missing" in nat.last_error() or "null" in nat.last_error()
    rc = lib.psp_adam_step(None, None, None, None, 10, 1, 1e-3, 0.9, 0.999, 1e-8, None)
    assert rc != 0


def test_struct_layouts_match_the_header():
    """The ctypes structures of native.py against sizeof() of the C structs (psp_abi_struct_sizes); load() itself refuses a
    library whose layouts differ, this test names the struct."""
    import ctypes as C

    from util_cases import psp
    nat = psp.native
    lib = nat.load()
    sizes = (C.c_int32 * 6)()
    assert lib.psp_abi_struct_sizes(C.byref(sizes)) == 0
    names = ("HjbConfig", "HjbSizes", "GenConfig", "GenSizes", "DnetConfig", "DnetSizes")
    for n, got in zip(names, sizes):
        assert C.sizeof(getattr(nat, n)) == got, n


def test_collective_entry_points_validate_arguments(lib):
    """psp_allreduce / psp_comm_* (include/psp.h): argument checks only -- a communicator needs a GPU."""
    assert lib.psp_allreduce(None, 4, nat.DT_F32, None, None) != 0
    assert "null" in nat.last_error()
    buf = (C.c_float * 4)()
    fake = C.c_void_p(1)
    assert lib.psp_allreduce(buf, 0, nat.DT_F32, fake, None) != 0
    assert lib.psp_allreduce(buf, 4, 7, fake, None) != 0 and "dtype" in nat.last_error()
    comm = C.c_void_p()
    ident = (C.c_ubyte * nat.COMM_ID_BYTES)()
    assert lib.psp_comm_init(C.byref(comm), 2, 5, ident) != 0
    assert lib.psp_comm_destroy(None) == 0


def test_genl_query_without_gpu(lib):
    """Value nets of any depth (include/psp.h 0.4.0): sizes, activation / layout switches and limits -- queries only."""
    def cfg(d, dims, has_time=1, act=nat.ACT_RELU2, K=200, N=25):
        c = nat.GenlConfig()
        c.base.d, c.base.K_local, c.base.N, c.base.store_path = d, K, N, 1
        c.has_time, c.n_hidden, c.activation = has_time, len(dims), act
        for i, h in enumerate(dims):
            c.widths[i] = h
        return c
    sz = nat.GenlSizes()
    c = cfg(100, [110, 110, 50])                                 # Allen-Cahn.ipynb:72
    assert lib.psp_genl_query(C.byref(c), C.byref(sz)) == 0
    assert sz.n_params == 101 * 110 + 110 + 211 * 110 + 110 + 321 * 50 + 50 + 371 + 1
    nt = (200 + 15) // 16
    assert sz.n_blocks == 26 * nt and sz.path_bytes == 26 * nt * 2 * 7 * 256 * 4
    assert sz.ahat_bytes == 26 * nt * 16 * 4 + nt * 4             # coefficients, then one step count per tile
    assert sz.waves_per_tile == 8 and sz.grad_partial_bytes % (sz.n_params * 4) == 0
    assert list(sz.seg_block_offset)[:4] == [0, 7, 14, 21]
    c = cfg(10, [20, 10, 10, 10], has_time=0, act=nat.ACT_TANH2, N=5000)     # Committor function.ipynb: tanh^2, N = 5000
    assert lib.psp_genl_query(C.byref(c), C.byref(sz)) == 0 and sz.waves_per_tile == 1
    assert sz.n_params == 10 * 20 + 20 + 30 * 10 + 10 + 40 * 10 + 10 + 50 * 10 + 10 + 60 + 1
    c = cfg(10, [20, 10], act=7)
    assert lib.psp_genl_query(C.byref(c), C.byref(sz)) != 0 and "activation" in nat.last_error()
    c = cfg(10, [200])
    assert lib.psp_genl_query(C.byref(c), C.byref(sz)) != 0
    c = cfg(10, [20, 10])
    c.base.domain_kind, c.base.dom_a, c.base.dom_b = nat.DOM_ANNULUS, 2.0, 1.0
    assert lib.psp_genl_query(C.byref(c), C.byref(sz)) != 0 and "annulus" in nat.last_error()
    rc = lib.psp_genl_rollout_bwd(C.byref(cfg(10, [20, 10])), None, None, None, None, None, None, None, None, None)
    assert rc != 0 and "null" in nat.last_error()

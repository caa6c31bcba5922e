"""The on-device noise stream against its definition (oracle/philox_oracle.py).
CPU: the numpy Philox4x32 reproduces the Random123 known-answer vectors of the 7-round generator the kernels run (and of the
10-round one of rounds 1 - 3), and its Box-Muller normals have the moments of N(0, 1).
GPU: psp_philox_normal_fill (the same device function the rollout kernels call: csrc/hjb_kernels.h philox_block) gives these
normals -- counter layout, key, round count, uniform construction and feature mapping pinned; the tolerance covers the fp32
hardware log / sqrt / sin / cos of the device against float64."""
import numpy as np
import pytest
import torch

from util_cases import psp
import oracle.philox_oracle as po


def test_numpy_philox_reproduces_the_random123_known_answers():
    assert po.ROUNDS == 7
    for rounds, kat in ((7, po.KAT7), (10, po.KAT)):
        for ctr, key, want in kat:
            got = po.philox4x32(*[np.uint32(c) for c in ctr], key[0], key[1], rounds=rounds)
            assert tuple(int(g) for g in got) == want, (rounds, ctr, key, [hex(int(g)) for g in got])


def test_numpy_normals_are_standard_normal():
    z = po.normal_stream(N=8, K=4096, d=20, seed=7)[1:]
    assert abs(z.mean()) < 3e-3 and abs(z.var() - 1.0) < 6e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3.0) < 6e-2
    assert np.abs(z).max() < 6.0
    a, b = z[:, :, 0].ravel(), z[:, :, 1].ravel()
    assert abs(np.corrcoef(a, b)[0, 1]) < 1e-2                          # neighbouring features: different calls


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,N,k_offset,seed,it", [(100, 48, 5, 0, 42, 0), (20, 33, 3, 65536, (7 << 32) | 12345, 9),
                                                    (500, 16, 2, 3, 2 ** 63 + 11, 4)])
def test_device_stream_matches_its_definition(d, K, N, k_offset, seed, it):
    nat = psp.native
    dev = torch.device("cuda:0")
    xi = torch.empty(N + 1, K, d, device=dev)
    nat.check(nat.load().psp_philox_normal_fill(nat.ptr(xi), N, K, d, k_offset, seed, it, None), "fill")
    torch.cuda.synchronize()
    want = po.normal_stream(N, K, d, k_offset=k_offset, seed=seed, iteration=it)
    got = xi.cpu().double().numpy()
    assert np.all(got[0] == 0.0)
    err = np.abs(got - want).max()
    print("d=%d: max |device - float64 definition| = %.2e" % (d, err))
    assert err <= 2e-5, err                                            # (a wrong counter / key / round gives O(1))

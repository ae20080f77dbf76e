"""cafe_scores_noalloc_dev (include/cafe_hip.h): the variant of cafe_scores_dev that allocates nothing -- its
per-lookup norms live in a fixed device array inside the library.  Called through the C ABI and compared bit for
bit with cafe_scores_dev (same kernels, so the bar is equality) on two calls in a row, the second with a smaller
input (the array keeps the first call's contents), then against the oracle's scores within 1e-5 relative."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _tables(L, F, D, hotn):
    w_hot = torch.zeros(hotn, D, device="cuda")
    cold = [torch.zeros(3, D, device="cuda") for _ in range(F)]
    ptrs = (ctypes.c_void_p * F)(*[w.data_ptr() for w in cold])
    offs = (ctypes.c_int64 * F)(*[1000 * f for f in range(F)])
    hs = (ctypes.c_int32 * F)(*([3] * F))
    h = ctypes.c_void_p()
    assert L.cafe_tables_create(ctypes.byref(h), F, D, hotn, ctypes.c_void_p(w_hot.data_ptr()), ptrs, offs, hs) == 0
    return h, (w_hot, cold)


def test_scores_noalloc_matches_scores_dev(oracle_mod):
    from cafe_amd import _lib
    L = _lib.load()
    torch.cuda.set_device(0)
    F, D = 5, 64
    h, keep = _tables(L, F, D, 16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator(device="cuda").manual_seed(11)
    try:
        for B in (777, 130):                      # second call smaller: stale norms of the first stay behind it
            g = torch.randn(F, B, D, device="cuda", generator=gen) * 3.0
            want = torch.full((F * B,), -1.0, device="cuda")
            got = torch.full((F * B,), -2.0, device="cuda")
            assert L.cafe_scores_dev(h, ctypes.c_void_p(g.data_ptr()), B, ctypes.c_void_p(want.data_ptr()), st) == 0
            rc = L.cafe_scores_noalloc_dev(h, ctypes.c_void_p(g.data_ptr()), B, ctypes.c_void_p(got.data_ptr()), st)
            assert rc == 0, L.cafe_last_error()
            torch.cuda.synchronize()
            assert torch.equal(got, want), f"B={B}: differs from cafe_scores_dev"
            gh = g.cpu().numpy()
            sc = got.cpu().numpy().reshape(F, B)
            for f in range(F):
                ref = oracle_mod.ref_scores(gh[f])
                assert np.all(np.abs(sc[f] - ref) <= RTOL * np.abs(ref) + 1e-12), f"B={B} field {f}"
        # beyond the fixed maximum: a clear error, nothing launched
        big = (1 << 20) // F + 1
        rc = L.cafe_scores_noalloc_dev(h, ctypes.c_void_p(keep[0].data_ptr()), big, ctypes.c_void_p(keep[0].data_ptr()), st)
        assert rc != 0 and b"exceed" in L.cafe_last_error()
    finally:
        L.cafe_tables_destroy(h)

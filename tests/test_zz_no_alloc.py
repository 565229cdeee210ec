"""gjx_resample_systematic_logw_static (the entry point that allocates nothing) against
gjx_resample_systematic_logw, through the C ABI: ancestor indices bit-exact, on two calls in a
row, the second with a smaller input (the static array keeps the first call's contents)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _logw(n, seed, spread):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.normal(size=n) * spread - 40.0).astype(np.float32)).cuda()


def test_static_resample_matches_ctx_entry_point_on_two_calls():
    from genjax_amd import backend as B

    assert torch.cuda.is_available()
    big, small = B.STATIC_MAX_N, 2048 * 37 + 5
    outs = [torch.empty(big, dtype=torch.int32, device="cuda"), torch.empty(small, dtype=torch.int32, device="cuda")]
    cases = [(_logw(big, 1, 6.0), 0.37454012, outs[0]), (_logw(small, 2, 0.2), 0.99999994, outs[1])]
    # reference answers first (this entry point owns a gjx_ctx and may allocate)
    refs = []
    for lw, u, _ in cases:
        stats = B.logsumexp_ess(lw)
        refs.append((stats, B.resample_systematic_logw(lw, stats[1:2], u)))
    lses = [stats[1:2].clone() for stats, _ in refs]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for (lw, u, out), lse in zip(cases, lses):           # the two calls, back to back, nothing in between
        got = B.resample_systematic_logw_static(lw, lse, u, out=out)
        assert got.data_ptr() == out.data_ptr()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0      # the device's free memory did not move: nothing was allocated
    for (_, _, out), (_, ref) in zip(cases, refs):
        assert torch.equal(out, ref), int((out != ref).sum())
    # ... and again in the opposite order, on a non-default stream
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for (lw, u, out), (stats, ref) in reversed(list(zip(cases, refs))):
            out.zero_()
            assert torch.equal(B.resample_systematic_logw_static(lw, stats[1:2], u, out=out), ref)
    s.synchronize()


def test_static_resample_small_sizes_and_limit():
    from genjax_amd import backend as B
    from genjax_amd._native import GjxError

    for n in (1, 2, 2047, 2048, 2049, 100003):
        lw = _logw(n, n, 3.0)
        stats = B.logsumexp_ess(lw)
        assert torch.equal(B.resample_systematic_logw_static(lw, stats[1:2], 0.6180339),
                           B.resample_systematic_logw(lw, stats[1:2], 0.6180339))
    lw = torch.zeros(B.STATIC_MAX_N + 1, device="cuda")
    with pytest.raises(GjxError, match="GJX_STATIC_MAX_N"):
        B.resample_systematic_logw_static(lw, torch.zeros(1, device="cuda"), 0.5)


def test_static_resample_rejects_outputs_it_cannot_write_in_place():
    """The wrapper hands raw pointers to the C call: an `out` that is strided, of another dtype or length, or not
    on the log-weights' device is refused before the call (each would be written wrongly, not reported)."""
    from genjax_amd import backend as B
    from genjax_amd._native import GjxError

    n = 4096
    lw = _logw(n, 5, 3.0)
    lse = B.logsumexp_ess(lw)[1:2]
    bad_outs = {
        "strided": torch.empty(2 * n, dtype=torch.int32, device="cuda")[::2],
        "host": torch.empty(n, dtype=torch.int32),
        "int64": torch.empty(n, dtype=torch.int64, device="cuda"),
        "short": torch.empty(n - 1, dtype=torch.int32, device="cuda"),
    }
    for name, out in bad_outs.items():
        with pytest.raises(GjxError, match="`out` must"):
            B.resample_systematic_logw_static(lw, lse, 0.25, out=out)
    with pytest.raises(GjxError, match="log-sum-exp"):
        B.resample_systematic_logw_static(lw, lse.cpu(), 0.25)
    with pytest.raises(GjxError, match="contiguous"):
        B.resample_systematic_logw_static(torch.zeros(2 * n, device="cuda")[::2], lse, 0.25)
    good = torch.empty(n, dtype=torch.int32, device="cuda")
    assert torch.equal(B.resample_systematic_logw_static(lw, lse, 0.25, out=good), B.resample_systematic_logw(lw, lse, 0.25))

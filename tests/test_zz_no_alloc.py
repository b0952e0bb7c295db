"""The entry point that allocates nothing (mgx_colsum_bf16_static, include/mgx.h): called through the C ABI with raw
pointers, checked bit-for-bit against mgx_colsum_bf16 (the same kernels with a caller-provided workspace), twice in a
row, the second time with a smaller input (the static scratch keeps the first call's contents)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, x):
    T, n = x.shape
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(int(lib.mgx_colsum_workspace_floats(n)), dtype=torch.float32, device="cuda")
    want = torch.empty(n, dtype=torch.float32, device="cuda")
    assert lib.mgx_colsum_bf16(x.data_ptr(), want.data_ptr(), ws.data_ptr(), T, n, st) == 0
    got = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.mgx_colsum_bf16_static(x.data_ptr(), got.data_ptr(), T, n, st)
    assert rc == 0, lib.mgx_last_error_string().decode()
    torch.cuda.synchronize()
    return got, want


def test_colsum_static_matches_workspace_form_on_two_calls():
    from musicgeneration_amd._lib import lib
    g = torch.Generator().manual_seed(77)
    big = torch.randn(4097, 3072, generator=g).bfloat16().cuda()
    small = torch.randn(129, 520, generator=g).bfloat16().cuda()
    for x in (big, small, big[:1000, :].contiguous()):
        got, want = _both(lib, x)
        assert torch.equal(got, want)
        ref = x.float().sum(0)
        assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-3


def test_colsum_static_rejects_sizes_beyond_its_scratch():
    from musicgeneration_amd._lib import lib
    nmax = lib.mgx_colsum_static_max_n()
    assert nmax == 8192
    x = torch.zeros(4, nmax + 8, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(nmax + 8, dtype=torch.float32, device="cuda")
    rc = lib.mgx_colsum_bf16_static(x.data_ptr(), out.data_ptr(), 4, nmax + 8, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"exceeds the static scratch" in lib.mgx_last_error_string()
    # the limit itself is accepted
    x = torch.ones(4, nmax, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(nmax, dtype=torch.float32, device="cuda")
    assert lib.mgx_colsum_bf16_static(x.data_ptr(), out.data_ptr(), 4, nmax, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert float(out.min()) == 4.0 and float(out.max()) == 4.0

"""Kernel-level checks on the GPU through the C ABI's diagnostic entry points.

A floating-point kernel is compared with a plain torch fp32 reference of the same op on the same
(already 16-bit-rounded) inputs; tolerances are stated per check.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from openhush_amd import engine
    engine.lib()
    return engine


def _tdtype(E, dt):
    return torch.bfloat16 if dt == E.OHW_DTYPE_BF16 else torch.float16


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (3000, 384, 1152), (128, 128, 64), (1, 128, 64), (4096, 1280, 1280), (3000, 256, 768), (1100, 512, 64), (6000, 768, 3072)])
def test_gemm_epilogues(E, dt, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    td = _tdtype(E, dt)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().T + bias
    s = torch.cuda.current_stream().cuda_stream
    # fp32 out
    out = torch.zeros(M, N, device="cuda")
    assert E.lib().ohw_dbg_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_F32, s) == 0, E.last_error()
    torch.cuda.synchronize()
    # fp32 accumulate of exact 16-bit products: only summation order differs
    assert (out - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    # 16-bit out (+ gelu)
    for epi, fn in ((E.EPI_BIAS_T, lambda x: x), (E.EPI_BIAS_GELU_T, lambda x: torch.nn.functional.gelu(x))):
        o16 = torch.zeros(M, N, device="cuda", dtype=td)
        assert E.lib().ohw_dbg_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), o16.data_ptr(), M, N, K, epi, s) == 0, E.last_error()
        torch.cuda.synchronize()
        want = fn(ref)
        tol = (2 ** -7 if dt == 0 else 2 ** -10) * max(1.0, want.abs().max().item())
        assert (o16.float() - want).abs().max().item() <= tol
    # residual accumulate into fp32
    res = torch.randn(M, N, device="cuda", generator=g)
    acc = res.clone()
    assert E.lib().ohw_dbg_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), acc.data_ptr(), M, N, K, E.EPI_BIAS_RESID_F32, s) == 0
    torch.cuda.synchronize()
    assert (acc - (res + ref)).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


def _check_gemm_all_epilogues(E, dt, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    td = _tdtype(E, dt)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().T + bias
    s = torch.cuda.current_stream().cuda_stream
    L = E.lib()
    out = torch.zeros(M, N, device="cuda")
    assert L.ohw_dbg_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_F32, s) == 0, E.last_error()
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    for epi, fn in ((E.EPI_BIAS_T, lambda x: x), (E.EPI_BIAS_GELU_T, lambda x: torch.nn.functional.gelu(x))):
        o16 = torch.zeros(M + 3, N, device="cuda", dtype=td)          # three guard rows behind the last one
        assert L.ohw_dbg_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), o16.data_ptr(), M, N, K, epi, s) == 0, E.last_error()
        torch.cuda.synchronize()
        want = fn(ref)
        tol = (2 ** -7 if dt == 0 else 2 ** -10) * max(1.0, want.abs().max().item())
        assert (o16[:M].float() - want).abs().max().item() <= tol
        assert not o16[M:].any()                                       # the ragged last tile wrote nothing past row M - 1
    res = torch.randn(M + 3, N, device="cuda", generator=g)
    acc = res.clone()
    assert L.ohw_dbg_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), acc.data_ptr(), M, N, K, E.EPI_BIAS_RESID_F32, s) == 0
    torch.cuda.synchronize()
    assert (acc[:M] - (res[:M] + ref)).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    assert torch.equal(acc[M:], res[M:])


GEMM256_SHAPES = [(8200, 1280, 1280), (33000, 256, 192), (16500, 512, 128)]   # at least 128 tiles of 256 x 256: gemm256_kernel; ragged M; 20 / 3 / 2 K-tiles


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("M,N,K", GEMM256_SHAPES)
def test_gemm256_epilogues_and_ragged_last_tile(E, dt, M, N, K):
    """the 256 x 256 kernel by itself (test_gemm_epilogues' shapes stay below its 128-tile threshold): every epilogue against
    torch fp32, a last m-tile that is only partly inside M, an odd K-tile count"""
    _check_gemm_all_epilogues(E, dt, M, N, K)


def test_gemm_rejects_bad_shapes(E):
    a = torch.zeros(16, 64, device="cuda", dtype=torch.bfloat16)
    assert E.lib().ohw_dbg_gemm(0, a.data_ptr(), a.data_ptr(), None, a.data_ptr(), 16, 100, 64, E.EPI_BIAS_T, None) == E.OHW_E_INVALID_ARG
    assert "multiple" in E.last_error()


def test_gemm_asymmetric_identity(E):
    # A = I, asymmetric W: catches a transposed C write (cdna_hip_programming.md section 3)
    K = N = 128
    A = torch.eye(K, device="cuda", dtype=torch.bfloat16)
    W = (torch.arange(N, device="cuda")[:, None] * 0.5 + torch.arange(K, device="cuda")[None, :] * 0.03125).to(torch.bfloat16)
    out = torch.zeros(K, N, device="cuda")
    assert E.lib().ohw_dbg_gemm(0, A.data_ptr(), W.data_ptr(), None, out.data_ptr(), K, N, K, E.EPI_F32, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, W.float().T)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("B,T,H", [(2, 1500, 2), (1, 200, 4), (3, 129, 1)])
def test_encoder_attention(E, dt, B, T, H):
    td = _tdtype(E, dt)
    d = 64 * H
    g = torch.Generator(device="cuda").manual_seed(B * 100 + T + H)
    qkv = (torch.randn(B * T, 3 * d, device="cuda", generator=g)).to(td)
    # spike one key against one query so the running max jumps mid-sequence (online-softmax rescale path)
    qkv[T // 2, :64] *= 6.0
    qkv[min(T - 1, 100), d:d + 64] = qkv[T // 2, :64]
    out = torch.zeros(B * T, d, device="cuda", dtype=td)
    assert E.lib().ohw_dbg_attention(dt, qkv.data_ptr(), out.data_ptr(), B, T, H, torch.cuda.current_stream().cuda_stream) == 0, E.last_error()
    torch.cuda.synchronize()
    x = qkv.float().view(B, T, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B * T, d)
    # P is rounded to 16 bits before P.V (like every flash kernel): error ~ 2^-8 (bf16) / 2^-11 (f16) of |V|
    tol = 3e-2 if dt == 0 else 4e-3
    assert (out.float() - ref).abs().max().item() < tol

import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from openhush_amd import engine as E
L = E.lib()
M, N = 48000, 3840
for K in (64, 128, 256, 512, 1280, 2560):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.ohw_dbg_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_BIAS_T, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.ohw_dbg_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, E.EPI_BIAS_T, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"K={K}: {ms*1e3:.1f} us  per tile-round {ms*1e3/12:.2f} us  phases {K//32}", flush=True)

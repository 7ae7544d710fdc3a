import os, sys
sys.path.insert(0, "/root/repo")
import torch, jatsr_amd._lib as L
OP = torch.float16 if L.OPERAND_DTYPE == "fp16" else torch.bfloat16
print("dtype", L.operand_dtype())
torch.manual_seed(0)
M, N, K = 256, 1792, 1280
A = torch.randn(M, K, device="cuda").to(OP); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(OP)
outs = {}
for v in (20, 28, 34, 33, 35, 21):
    o = torch.zeros(M, N, device="cuda")
    L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), None, L.ptr(o), M, N, K, 0, None, 0, M, v, L.stream_ptr()))
    torch.cuda.synchronize(); outs[v] = o
ref = A.double() @ W.double().T
for v, o in outs.items():
    print(v, "bit-equal to v20:", torch.equal(o, outs[20]), "max|diff|", float((o - outs[20]).abs().max()), "rel err vs fp64", float((o.double() - ref).norm() / ref.norm()))

import torch, time
torch.manual_seed(0)
M = 131072
for (N, K, name) in ((2304, 768, 'QKV'), (768, 768, 'OUT'), (3072, 768, 'FFN1'), (768, 3072, 'FFN2')):
    a = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    w = torch.randn(N, K, device='cuda', dtype=torch.bfloat16) * 0.02
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): c = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: torch bf16 matmul (hipBLASLt/rocBLAS) {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TF", flush=True)

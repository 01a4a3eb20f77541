import time, torch, numpy as np
for n in (1024, 4096, 8192):
    a = torch.randn(n, n, dtype=torch.complex128, device='cuda')
    a = a @ a.conj().T / n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w = torch.linalg.eigvalsh(a)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("gpu eigvalsh n=%d: %.2f s" % (n, t1 - t0), flush=True)
    if n <= 4096:
        an = a.cpu().numpy()
        t0 = time.perf_counter(); wn = np.linalg.eigvalsh(an); t1 = time.perf_counter()
        print("   numpy: %.2f s, max |dw| = %.2e" % (t1 - t0, np.abs(wn - w.cpu().numpy()).max()), flush=True)

#!/usr/bin/env python
"""Read-bandwidth vs working-set size (L2 / Infinity Cache / HBM) with a plain
streaming reduction, to see where on-chip reuse can pay."""
import sys
import torch

def main():
    dev = torch.device("cuda")
    for mb in (2, 8, 16, 24, 32, 64, 128, 192, 256, 384, 512, 1024, 4096):
        n = mb * 1024 * 1024 // 8
        x = torch.ones(n, dtype=torch.float64, device=dev)
        reps = max(4, min(200, 16384 // mb))
        for _ in range(3):
            x.sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            x.sum()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("read %5d MB: %8.4f ms  %8.1f GB/s" % (mb, ms, mb * 1.048576 / ms), flush=True)
        y = torch.empty_like(x)
        for _ in range(3):
            y.copy_(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("copy %5d MB: %8.4f ms  %8.1f GB/s (r+w)" % (mb, ms, 2 * mb * 1.048576 / ms), flush=True)

if __name__ == "__main__":
    main()

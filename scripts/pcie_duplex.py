#!/usr/bin/env python3
"""Is the host link full duplex for us?  16 MiB uploads on one stream, 19 MB downloads on another (page-locked memory),
alone and together.  (torch is only the plumbing here: streams, pinned buffers, copies.)"""
import time
import torch

dev = torch.device("cuda", 0)
up_h = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
dn_h = torch.empty(19 << 20, dtype=torch.uint8).pin_memory()
up_d = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
dn_d = torch.empty(19 << 20, dtype=torch.uint8, device=dev)
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()


def run(do_up, do_dn, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if do_up:
            with torch.cuda.stream(s_up):
                up_d.copy_(up_h, non_blocking=True)
        if do_dn:
            with torch.cuda.stream(s_dn):
                dn_h.copy_(dn_d, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    a, b, c = run(True, False), run(False, True), run(True, True)
    print("upload alone %.3f ms (%.1f GB/s)  download alone %.3f ms (%.1f GB/s)  both %.3f ms (sum %.3f, max %.3f)" % (
        a, 16.78 / a, b, 19.92 / b, c, a + b, max(a, b)))

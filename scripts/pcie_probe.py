"""Host-link rates that bound the drop-in entry points (bssm_bsf with host buffers): device -> pinned DMA (one and two
streams), then pinned -> fresh pageable memory with several CPU threads.  python scripts/pcie_probe.py"""
import threading
import time

import numpy as np
import torch

dev = torch.device("cuda")
nbytes = 1 << 30
src = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
pin = [torch.empty(nbytes // 2, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
torch.cuda.synchronize()
for label, streams in (("one stream", 1), ("two streams", 2)):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for q in range(2):
            with torch.cuda.stream(ss[q % streams]):
                pin[q].copy_(src[q * (nbytes // 2):(q + 1) * (nbytes // 2)], non_blocking=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f"D2H pinned, {label}: {nbytes / el / 1e9:.1f} GB/s")
# pageable destination through the runtime
pg = torch.empty(nbytes, dtype=torch.uint8)
t0 = time.perf_counter(); pg.copy_(src); torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"D2H pageable (first touch): {nbytes / el / 1e9:.1f} GB/s")
t0 = time.perf_counter(); pg.copy_(src); torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"D2H pageable (touched):     {nbytes / el / 1e9:.1f} GB/s")
# CPU: pinned -> fresh pageable, T threads
a = pin[0].numpy()
for T in (1, 4, 8, 16, 32):
    dst = np.empty(a.size, dtype=np.uint8)  # untouched pages
    bounds = np.linspace(0, a.size, T + 1).astype(np.int64)
    def work(q):
        dst[bounds[q]:bounds[q + 1]] = a[bounds[q]:bounds[q + 1]]
    th = [threading.Thread(target=work, args=(q,)) for q in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    th = [threading.Thread(target=work, args=(q,)) for q in range(T)]
    for x in th: x.start()
    for x in th: x.join()
    el2 = time.perf_counter() - t1
    print(f"CPU copy pinned->pageable, {T:2d} threads: first touch {a.size / el / 1e9:.1f} GB/s, touched {a.size / el2 / 1e9:.1f} GB/s")

import torch, time, numpy as np
n = 1<<27  # 1 GiB of float64
d = torch.ones(n, dtype=torch.float64, device="cuda")
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/reps
hp = torch.empty(n, dtype=torch.float64)
print("pageable (touched) GB/s", 8*n/t(lambda: hp.copy_(d))/1e9)
hpin = torch.empty(n, dtype=torch.float64).pin_memory()
print("pinned GB/s", 8*n/t(lambda: hpin.copy_(d, non_blocking=True))/1e9)
# fresh untouched numpy zeros each time (what a caller's np.zeros output buffer is)
def fresh():
    a = np.zeros(n); torch.from_numpy(a).copy_(d)
print("fresh np.zeros GB/s", 8*n/t(fresh, 2)/1e9)
a = np.zeros(n)
t0=time.perf_counter(); a[:] = 1.0; print("first-touch fill GB/s", 8*n/(time.perf_counter()-t0)/1e9)
t0=time.perf_counter(); hp.numpy()[:] = hpin.numpy(); print("host memcpy 1 thread GB/s", 8*n/(time.perf_counter()-t0)/1e9)

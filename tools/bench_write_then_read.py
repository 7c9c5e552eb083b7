"""Read bandwidth of torch.sum over a buffer that was just written (fill_) or just read, by size: up to 256 MiB both stay in the Infinity
Cache (5.2 TB/s); beyond it a read of freshly written data streams at 3.0-3.1 TB/s, of clean data at 3.6-3.9 TB/s (MI355X). The binned grid
backward's reduce reads 0.72 GB of records its scatter has just written: that is its bound (NOTEBOOK.md, rounds 1-4 §5)."""
import torch
big = torch.empty(1024 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    x = big[: mb * 1024 * 1024 // 4]
    for mode in ("write-then-read", "read-then-read"):
        ts = []
        for it in range(12):
            if mode == "write-then-read": x.fill_(float(it))
            else: x.sum()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); s = x.sum(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        print(f"{mb:5d} MiB {mode:16s}: sum {ms*1e3:8.1f} us  {mb * 1.048576e-3 / ms:6.2f} TB/s")

#!/usr/bin/env python3
"""Which torch-native ops run inside one headline training step, with their input shapes and GPU time (torch.profiler): the part of the
step that is not this library's kernels — optimizer, GradScaler, loss, and the casts / fills of the wrappers.

    python tools/step_op_table.py [steps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    model = bench.build_model(1, dev, seed=0).train()
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    poses, intr = bench.make_training_rays(dev, 1, 8, 0)
    gen = torch.Generator().manual_seed(1)
    batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(4)]
    for i in range(6):
        bench.train_step(model, opt, scaler, *batches[i % 4])
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for i in range(steps):
            bench.train_step(model, opt, scaler, *batches[i % 4])
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = getattr(e, "self_cuda_time_total", 0)
        if dt > 0:
            rows.append((dt / steps, e.count / steps, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"GPU time attributed to ops: {total:.1f} us per step")
    for dt, cnt, key, shapes in rows[:60]:
        print(f"{dt:9.1f} us  x{cnt:5.1f}  {key:45s} {shapes}")


if __name__ == "__main__":
    main()

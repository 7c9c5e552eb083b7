#!/usr/bin/env python3
"""Which torch-native ops run inside one headline training step, with their input shapes and GPU time (torch.profiler): the part of the
step that is not this library's kernels — optimizer, GradScaler, loss, and the casts / fills of the wrappers.

    python tools/step_op_table.py [steps] [occ]        # occ: the occupancy-grid step (configs[2]) instead of the headline step
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    occ = len(sys.argv) > 2 and sys.argv[2] == "occ"
    dev = torch.device("cuda:0")
    model = bench.build_model(2 if occ else 1, dev, cuda_ray=occ, seed=0).train()
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    poses, intr = bench.make_training_rays(dev, 2 if occ else 1, 8, 0)
    gen = torch.Generator().manual_seed(1)
    batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(4)]
    step = bench.cuda_ray_train_step if occ else bench.train_step
    for i in range(17 if occ else 6):
        step(model, opt, scaler, *batches[i % 4])
        if occ and i == 15:
            model.mean_count = int(model.step_counter[:16, 0].sum().item() / 16)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for i in range(steps):
            step(model, opt, scaler, *batches[i % 4])
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

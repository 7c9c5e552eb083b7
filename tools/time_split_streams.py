#!/usr/bin/env python3
"""Experiment: does the headline step's forward + backward gain from running as TWO half batches on two streams (kernels with different
bounds side by side: request-bound encoder against issue-bound MLPs against the LDS-atomic reduce)?  Each form is captured as one HIP graph
(no host in the loop) and replayed; no optimizer (it would be the same in every form).

    python tools/time_split_streams.py [replays]

Forms: full = one 4096-ray batch; seq2 = two 2048-ray halves one after the other on one stream; par2 = the halves on two streams inside one
graph (two models with the same weights, so no gradient accumulation crosses the streams); par4 = quarters on four streams.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from focnerf_amd.graph import GraphedStep                      # noqa: E402


def fb(model, o, d, t):
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(o, d, staged=False, num_steps=bench.NUM_STEPS, upsample_steps=0, perturb=True, bg_color=None, fused=True)
        loss = torch.nn.functional.mse_loss(out["image"], t)
    (loss * 1024.0).backward()
    return loss.detach()          # (a live loss keeps the AccumulateGrad nodes, and with them the stream they were made on)


def main():
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    models = [bench.build_model(1, dev, seed=0).train() for _ in range(4)]
    for m in models[1:]:
        m.load_state_dict(models[0].state_dict())
    poses, intr = bench.make_training_rays(dev, 1, 8, 0)
    gen = torch.Generator().manual_seed(1)
    o, d, t = bench.sample_batch(poses, intr, dev, gen)
    n = o.shape[-2]
    print("batch", tuple(o.shape), tuple(d.shape), tuple(t.shape), flush=True)

    def part(x, i, k):
        return x[..., i * (n // k):(i + 1) * (n // k), :].contiguous()

    streams = [torch.cuda.Stream() for _ in range(3)]

    def full(o, d, t):
        return fb(models[0], o, d, t)

    def seq(k):
        def f(*a):
            for i in range(k):
                loss = fb(models[i], a[3 * i], a[3 * i + 1], a[3 * i + 2])
            return loss
        return f

    def par(k):
        def f(*a):
            cur = torch.cuda.current_stream()
            for s in streams[:k - 1]:
                s.wait_stream(cur)
            loss = fb(models[0], a[0], a[1], a[2])
            for i in range(1, k):
                with torch.cuda.stream(streams[i - 1]):
                    fb(models[i], a[3 * i], a[3 * i + 1], a[3 * i + 2])
            for s in streams[:k - 1]:
                cur.wait_stream(s)
            return loss
        return f

    def inputs(k):
        out = []
        for i in range(k):
            out += [part(o, i, k), part(d, i, k), part(t, i, k)]
        return tuple(out)

    forms = [("full", full, (o, d, t)), ("seq2", seq(2), inputs(2)), ("par2", par(2), inputs(2)), ("par4", par(4), inputs(4))]
    if os.environ.get("SPLIT_EAGER") == "1":                     # the forms without capture (debugging aid)
        for name, fn, inp in forms:
            fn(*inp)
            torch.cuda.synchronize()
            print("eager ok", name, flush=True)
    graphs = []
    for name, fn, inp in forms:
        try:
            graphs.append((name, GraphedStep(fn, inp), inp))
            print("captured", name, flush=True)
        except Exception as e:                                    # a form that does not capture is reported, the others still run
            print("capture failed", name, repr(e), flush=True)
    for rep in range(3):
        for name, g, inp in graphs:
            for _ in range(5):
                g(*inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(replays):
                g(*inp)
            torch.cuda.synchronize()
            print(f"rep {rep} {name}: {1e3 * (time.perf_counter() - t0) / replays:.4f} ms per 4096 rays (fwd + bwd, no optimizer)", flush=True)
    # same weights, same rays: the forms' table gradients agree up to the half-batch summation order
    g0 = models[0].encoder.embeddings.grad
    if g0 is not None:
        print("grad table norm", float(g0.float().norm()))


if __name__ == "__main__":
    main()

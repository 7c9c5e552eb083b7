"""HIP-graph capture of a whole training step (forward, backward, GradScaler, fused Adam) for launch-bound steps.

The occupancy-grid step is ~85 launches for ~1.1 ms of GPU work and the host needs ~1.25 ms to enqueue them from Python; replayed
as one graph it runs at GPU speed. Every kernel of libfocnerf_hip.so goes to torch's current stream through the C ABI and the scratch
buffers are persistent, so the step captures like any torch code. Requirements, checked where they can be: static shapes (for
`march_rays_train` that means `mean_count > 0` and `force_all_rays=False`), an optimizer built with `capturable=True`, no host
synchronisation inside the step (no `.item()`).

    opt = torch.optim.Adam(model.get_params(lr), ..., fused=True, capturable=True)
    step = GraphedStep(lambda o, d, t: train_step(model, opt, scaler, o, d, t), (rays_o, rays_d, target))
    loss = step(rays_o, rays_d, target)        # copies the batch into the static buffers, replays the graph
"""
import torch


class GraphedStep:
    def __init__(self, step_fn, example_inputs, warmup=3):
        self.static_inputs = [t.clone() for t in example_inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                       # warm-up on a side stream, as torch.cuda.graphs requires
            for _ in range(warmup):
                step_fn(*self.static_inputs)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # capture on the warm-up stream: the library's scratch buffers are per stream (backend._Scratch), so the capture finds the
        # buffers the warm-up sized instead of allocating a second set; buffers a capture has seen are pinned for the process' lifetime
        with torch.cuda.graph(self.graph, stream=side):
            self.static_output = step_fn(*self.static_inputs)

    def __call__(self, *inputs):
        for s, t in zip(self.static_inputs, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.static_output

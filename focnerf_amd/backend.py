"""Tensor-level shims with the exact names and positional orders of the reference's four
pybind11 extension modules:

    _raymarching   raymarching/src/bindings.cpp:5-19
    _gridencoder   gridencoder/src/bindings.cpp
    _freqencoder   freqencoder/src/bindings.cpp
    _ffmlp         ffmlp/src/bindings.cpp

Each function converts torch tensors to raw device pointers, picks torch's current HIP
stream and calls the C ABI of libfocnerf_hip.so (include/focnerf.h); a non-zero return
becomes a RuntimeError, like TORCH_CHECK in the reference. Outputs are written in place
into caller-allocated tensors, exactly as in the reference.
"""
import math

import torch

from . import _lib
from ._lib import lib, ptr, stream_of, check, require_cuda, dtype_code, raw_stream


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"focnerf_amd: expected float32 tensor, got {t.dtype}")


def _contig(*ts):
    for t in ts:
        if t is not None and not t.is_contiguous():
            raise RuntimeError("focnerf_amd: tensor must be contiguous")   # CHECK_CONTIGUOUS in the reference


class _Scratch:
    """Grow-only scratch buffers (counters, workspaces), one per (purpose, device, STREAM): two streams running the same op never share
    a workspace (nor the precount header that lives in one), and work queued on a stream is ordered against its own earlier uses.

    A HIP graph captured over a call keeps the raw address it saw. A buffer that was handed out during a stream capture is therefore
    never given back to the allocator: when a larger request supersedes it, it moves to `_pinned` and lives as long as the process
    (a graph may replay at any later time), and the capture itself never grows a buffer it has already baked in — it raises instead."""

    def __init__(self):
        self._bufs = {}            # key -> [tensor, seen_by_capture]
        self._pinned = []

    def get(self, key, nbytes, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        k = (key, idx, raw_stream(idx))
        capturing = torch.cuda.is_current_stream_capturing()
        ent = self._bufs.get(k)
        if ent is None or ent[0].numel() < nbytes:
            if ent is not None and ent[1]:
                if capturing:
                    raise RuntimeError(f"focnerf_amd: scratch '{key}' would have to grow from {ent[0].numel()} to {int(nbytes)} bytes inside a stream "
                                       f"capture that already recorded its address; warm the step up at its largest shapes before capturing")
                self._pinned.append(ent[0])                  # a captured graph still writes there
            ent = self._bufs[k] = [torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device), False]
        if capturing:
            ent[1] = True
        return ent[0]


_scratch = _Scratch()


class _raymarching:
    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        require_cuda(rays_o, rays_d, aabb, nears, fars); _f32(rays_o, rays_d, aabb, nears, fars); _contig(rays_o, rays_d, aabb, nears, fars)
        check(lib.foc_near_far_from_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), N, min_near, ptr(nears), ptr(fars), stream_of(rays_o)),
              "near_far_from_aabb")

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        require_cuda(rays_o, rays_d, coords); _f32(rays_o, rays_d, coords); _contig(rays_o, rays_d, coords)
        check(lib.foc_sph_from_ray(ptr(rays_o), ptr(rays_d), radius, N, ptr(coords), stream_of(rays_o)), "sph_from_ray")

    @staticmethod
    def morton3D(coords, N, indices):
        require_cuda(coords, indices); _contig(coords, indices)
        assert coords.dtype == torch.int32 and indices.dtype == torch.int32
        check(lib.foc_morton3D(ptr(coords), N, ptr(indices), stream_of(coords)), "morton3D")

    @staticmethod
    def morton3D_invert(indices, N, coords):
        require_cuda(coords, indices); _contig(coords, indices)
        assert coords.dtype == torch.int32 and indices.dtype == torch.int32
        check(lib.foc_morton3D_invert(ptr(indices), N, ptr(coords), stream_of(coords)), "morton3D_invert")

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        require_cuda(grid, bitfield); _f32(grid); _contig(grid, bitfield)
        assert bitfield.dtype == torch.uint8
        check(lib.foc_packbits(ptr(grid), N, density_thresh, ptr(bitfield), stream_of(grid)), "packbits")

    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
        require_cuda(rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
        _f32(rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises); _contig(rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
        assert grid.dtype == torch.uint8 and rays.dtype == torch.int32 and counter.dtype == torch.int32
        scratch = _scratch.get("march", lib.foc_march_rays_train_scratch_bytes(N, max_steps), rays_o.device)
        check(lib.foc_march_rays_train(ptr(rays_o), ptr(rays_d), ptr(grid), bound, dt_gamma, max_steps, N, C, H, M, ptr(nears), ptr(fars),
                                       ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(counter), ptr(noises), ptr(scratch),
                                       stream_of(rays_o)), "march_rays_train")

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        require_cuda(sigmas, rgbs, deltas, rays, weights_sum, depth, image); _f32(sigmas, rgbs, deltas, weights_sum, depth, image)
        _contig(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        check(lib.foc_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, T_thresh, ptr(weights_sum), ptr(depth),
                                                   ptr(image), stream_of(sigmas)), "composite_rays_train_forward")

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        ts = (grad_weights_sum, grad_image, sigmas, rgbs, deltas, weights_sum, image, grad_sigmas, grad_rgbs)
        require_cuda(*ts, rays); _f32(*ts); _contig(*ts, rays)
        check(lib.foc_composite_rays_train_backward(ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays),
                                                    ptr(weights_sum), ptr(image), M, N, T_thresh, ptr(grad_sigmas), ptr(grad_rgbs),
                                                    stream_of(sigmas)), "composite_rays_train_backward")

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises):
        ts = (rays_t, rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
        require_cuda(*ts, rays_alive, grid); _f32(*ts); _contig(*ts, rays_alive, grid)
        check(lib.foc_march_rays(n_alive, n_step, ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), bound, dt_gamma, max_steps, C, H,
                                 ptr(grid), ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises), stream_of(rays_o)), "march_rays")

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        ts = (rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
        require_cuda(*ts, rays_alive); _f32(*ts); _contig(*ts, rays_alive)
        check(lib.foc_composite_rays(n_alive, n_step, T_thresh, ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                     ptr(weights_sum), ptr(depth), ptr(image), stream_of(sigmas)), "composite_rays")

    # extension (no reference binding): ordered device-side compaction of rays_alive >= 0
    @staticmethod
    def compact_alive(rays_alive, n_alive, out, n_out):
        require_cuda(rays_alive, out, n_out)
        scratch = _scratch.get("compact", 4 * (n_alive // 1024 + 2), rays_alive.device)
        check(lib.foc_compact_alive(ptr(rays_alive), n_alive, ptr(out), ptr(n_out), ptr(scratch), stream_of(rays_alive)), "compact_alive")


class _gridencoder:
    @staticmethod
    def _common(inputs, embeddings, offsets):
        require_cuda(inputs, embeddings, offsets)          # CHECK_CUDA, gridencoder.cu:449-452
        _contig(inputs, embeddings, offsets)               # CHECK_CONTIGUOUS :455-458
        if inputs.dtype != torch.float32:
            raise RuntimeError("inputs must be a float32 tensor")
        if offsets.dtype != torch.int32:
            raise RuntimeError("offsets must be an int tensor")   # CHECK_IS_INT :463

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, out_bl=False):
        _gridencoder._common(inputs, embeddings, offsets)
        require_cuda(outputs, dy_dx); _contig(outputs, dy_dx)
        dt = dtype_code(embeddings)
        if outputs.dtype != embeddings.dtype or (dy_dx is not None and dy_dx.dtype != embeddings.dtype):
            raise RuntimeError("grid_encode_forward: outputs/dy_dx must share the embeddings dtype")
        fn = lib.foc_grid_encode_forward_bl if out_bl else lib.foc_grid_encode_forward
        check(fn(ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, float(S), H, ptr(dy_dx), gridtype, int(bool(align_corners)),
                 interp, dt, None, stream_of(inputs)), "grid_encode_forward")

    @staticmethod
    def planes_to_rows(planes, rows, B, L, unit_bytes):
        require_cuda(planes, rows); _contig(planes, rows)
        check(lib.foc_grid_planes_to_rows(ptr(planes), ptr(rows), B, L, unit_bytes, stream_of(planes)), "grid_planes_to_rows")

    @staticmethod
    def rows_to_planes(rows, planes, B, L, unit_bytes):
        require_cuda(planes, rows); _contig(planes, rows)
        check(lib.foc_grid_rows_to_planes(ptr(rows), ptr(planes), B, L, unit_bytes, stream_of(rows)), "grid_rows_to_planes")

    @staticmethod
    def _host_entry(offsets):
        """Host copy of the (tiny, immutable) level-offset table, kept ON the tensor object so that it dies with it and a new tensor at a
        recycled address can never be mistaken for it; cached so that the backward does not synchronise on every call."""
        h = getattr(offsets, "_foc_host", None)
        if h is None or h[0] != offsets._version:
            import ctypes
            arr = offsets.detach().cpu().numpy().astype("int32")
            h = (offsets._version, arr, arr.ctypes.data_as(ctypes.c_void_p), int((arr[1:] - arr[:-1]).max()) if arr.size > 1 else 0)
            offsets._foc_host = h
        return h

    @staticmethod
    def _host_offsets(offsets):
        return _gridencoder._host_entry(offsets)[2]

    @staticmethod
    def _binned_ok(offsets, S, H, L, gridtype):
        """The binned backward serves hash grids whose levels have at most 64 segments of 8192 rows (log2_hashmap_size <= 19, the NeRF
        default) and a resolution below 8191 (gb_check in csrc/gridencoder.hip says why); everything else takes the scattered-atomic kernel."""
        finest = math.ceil(2.0 ** (float(S) * (L - 1)) * H - 1.0) + 1
        return gridtype == 0 and finest <= 8190 and _gridencoder._host_entry(offsets)[3] <= 8192 * 64

    # ---- the gradient-independent half of the binned backward (record counts -> record ranges), run ahead of time ----------------
    # It needs the sample positions only and is VALU/LDS work, while the forward gathers are bound by cache requests and leave the VALU
    # idle: a training forward (`grid_encode_forward_counted`) lets it ride along in the same launch. The result lives in the header of
    # the shared backward workspace; a ticket tells the backward whether that header is still the one of ITS forward — any other use of
    # the workspace in between invalidates it, and the backward then counts again itself. (`standalone=True`: the count as its own
    # launch, foc_grid_encode_backward_count, for callers whose forward is not the [L,B,C] kernel.)
    _pre = {}                                               # device index -> dict(ticket, key)

    @staticmethod
    def _pre_state(device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = _gridencoder._pre.get(idx)
        if st is None:
            st = _gridencoder._pre[idx] = dict(ticket=0, key=None)
        return idx, st

    @staticmethod
    def _invalidate_precount(device):
        _, st = _gridencoder._pre_state(device)
        st["ticket"] += 1
        st["key"] = None

    @staticmethod
    def grid_encode_forward_counted(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, gridtype, align_corners, interp, standalone=False):
        """[L,B,C] forward + the backward's count pass. Returns a ticket for `grid_encode_backward(..., precount=ticket)`, or None when
        the binned path does not apply (then nothing was computed: call grid_encode_forward)."""
        import os
        if os.environ.get("FOC_GRID_PRECOUNT", "1") == "0" or os.environ.get("FOCNERF_GRID_ATOMIC", "0") == "1":
            return None
        _gridencoder._common(inputs, embeddings, offsets)
        require_cuda(outputs); _contig(outputs)
        dt = dtype_code(embeddings)
        ws_bytes = lib.foc_grid_encode_backward_workspace_bytes(B, D, C, L, dt)
        if not ws_bytes or B * 8 * L >= 2 ** 32 or B == 0 or not _gridencoder._binned_ok(offsets, S, H, L, gridtype):
            return None
        if outputs.dtype != embeddings.dtype or outputs.numel() != L * B * C:
            raise RuntimeError("grid_encode_forward_counted: outputs must be [L,B,C] of the embeddings' dtype")
        ws = _scratch.get("grid_bwd", ws_bytes, inputs.device)
        host = _gridencoder._host_offsets(offsets)
        if standalone:
            check(lib.foc_grid_encode_backward_count(ptr(inputs), ptr(offsets), B, D, C, L, float(S), H, gridtype, int(bool(align_corners)), interp, dt,
                                                     host, ptr(ws), ws_bytes, stream_of(inputs)), "grid_encode_backward_count")
            check(lib.foc_grid_encode_forward(ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, float(S), H, None, gridtype,
                                              int(bool(align_corners)), interp, dt, None, stream_of(inputs)), "grid_encode_forward")
        else:
            check(lib.foc_grid_encode_forward_counted(ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, float(S), H, gridtype,
                                                      int(bool(align_corners)), interp, dt, host, ptr(ws), ws_bytes, stream_of(inputs)),
                  "grid_encode_forward_counted")
        idx, st = _gridencoder._pre_state(inputs.device)
        st["ticket"] += 1
        st["key"] = (inputs.data_ptr(), B, L, dt, ws.data_ptr())
        return (idx, st["ticket"], st["key"])

    @staticmethod
    def _precount_valid(ticket, inputs, B, L, dt, ws):
        if ticket is None:
            return False
        idx, number, key = ticket
        st = _gridencoder._pre.get(idx)
        return st is not None and st["ticket"] == number and st["key"] == key == (inputs.data_ptr(), B, L, dt, ws.data_ptr())

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp, grad_bl=False,
                             precount=None):
        _gridencoder._common(inputs, embeddings, offsets)
        require_cuda(grad, grad_embeddings, dy_dx, grad_inputs); _contig(grad, grad_embeddings, dy_dx, grad_inputs)
        dt = dtype_code(grad)                               # the reference dispatches on grad.scalar_type(), :498-499
        if grad_embeddings.dtype != grad.dtype:
            raise RuntimeError("grid_encode_backward: grad_embeddings must share grad's dtype")
        # D=3, C=2 tables: partition + LDS accumulation instead of scattered atomics (FOCNERF_GRID_ATOMIC=1 forces the atomic kernel)
        import os
        ws_bytes = 0 if os.environ.get("FOCNERF_GRID_ATOMIC", "0") == "1" else lib.foc_grid_encode_backward_workspace_bytes(B, D, C, L, dt)
        if ws_bytes and B * 8 * L < 2 ** 32 and _gridencoder._binned_ok(offsets, S, H, L, gridtype):
            # persistent grow-only scratch (2 GB at B = 2M): a fresh torch.empty per call makes the caching allocator
            # re-malloc it whenever the freed block was split in between (measured: 28 ms hiccups per step)
            ws = _scratch.get("grid_bwd", ws_bytes, grad.device)
            counted = _gridencoder._precount_valid(precount, inputs, B, L, dt, ws)
            fn = lib.foc_grid_encode_backward_binned_counted if counted else lib.foc_grid_encode_backward_binned
            check(fn(ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C, L, float(S), H,
                     ptr(dy_dx), ptr(grad_inputs), gridtype, int(bool(align_corners)), interp, dt, int(bool(grad_bl)),
                     _gridencoder._host_offsets(offsets), ptr(ws), ws_bytes, stream_of(inputs)), "grid_encode_backward_binned")
            _gridencoder._invalidate_precount(grad.device)  # the header now belongs to this pass (and a used ticket is spent)
            return
        check(lib.foc_grid_encode_backward(ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C, L, float(S), H,
                                           ptr(dy_dx), ptr(grad_inputs), gridtype, int(bool(align_corners)), interp, dt, int(bool(grad_bl)),
                                           None, stream_of(inputs)), "grid_encode_backward")

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
        require_cuda(inputs, embeddings, grad, offsets); _contig(inputs, embeddings, grad, offsets)
        dt = dtype_code(embeddings)
        if inputs.dtype != embeddings.dtype or grad.dtype != embeddings.dtype:
            raise RuntimeError("grad_total_variation: inputs/grad must share the embeddings dtype")
        check(lib.foc_grad_total_variation(ptr(inputs), ptr(embeddings), ptr(grad), ptr(offsets), weight, B, D, C, L, float(S), H, gridtype,
                                           int(bool(align_corners)), dt, stream_of(inputs)), "grad_total_variation")


class _freqencoder:
    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, C, outputs):
        require_cuda(inputs, outputs); _f32(inputs, outputs); _contig(inputs, outputs)
        check(lib.foc_freq_encode_forward(ptr(inputs), B, D, deg, C, ptr(outputs), stream_of(inputs)), "freq_encode_forward")

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
        require_cuda(grad, outputs, grad_inputs); _f32(grad, outputs, grad_inputs); _contig(grad, outputs, grad_inputs)
        check(lib.foc_freq_encode_backward(ptr(grad), ptr(outputs), B, D, deg, C, ptr(grad_inputs), stream_of(grad)), "freq_encode_backward")


def _half(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float16:
            raise RuntimeError("focnerf_amd: tensor must be a half tensor")   # CHECK_IS_HALF, ffmlp.cu:638


class _ffmlp:
    @staticmethod
    def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer, outputs):
        require_cuda(inputs, weights, forward_buffer, outputs); _half(inputs, weights, forward_buffer, outputs); _contig(inputs, weights, forward_buffer, outputs)
        check(lib.foc_ffmlp_forward(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                                    ptr(forward_buffer), ptr(outputs), stream_of(inputs)), "ffmlp_forward")

    @staticmethod
    def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference_buffer, outputs):
        require_cuda(inputs, weights, outputs); _half(inputs, weights, outputs); _contig(inputs, weights, outputs)
        check(lib.foc_ffmlp_inference(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                                      ptr(inference_buffer), ptr(outputs), stream_of(inputs)), "ffmlp_inference")

    @staticmethod
    def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                       calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
        ts = (grad, inputs, weights, forward_buffer, backward_buffer, grad_inputs, grad_weights)
        require_cuda(*ts); _half(*ts); _contig(*ts)
        ws_bytes = lib.foc_ffmlp_backward_workspace_bytes(input_dim, hidden_dim, num_layers)
        ws = _scratch.get("ffmlp_ws", ws_bytes, grad.device)
        check(lib.foc_ffmlp_backward(ptr(grad), ptr(inputs), ptr(weights), ptr(forward_buffer), B, input_dim, output_dim, hidden_dim, num_layers,
                                     activation, output_activation, int(bool(calc_grad_inputs)), ptr(backward_buffer), ptr(grad_inputs),
                                     ptr(grad_weights), ptr(ws), ws.numel(), stream_of(grad)), "ffmlp_backward")

    @staticmethod
    def ffmlp_forward_planar(inputs_planar, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, outputs):
        """inputs_planar: [input_dim/2, B, 2] half — the encoder's native [L,B,C] output (include/focnerf.h)."""
        require_cuda(inputs_planar, weights, outputs); _half(inputs_planar, weights, outputs); _contig(inputs_planar, weights, outputs)
        check(lib.foc_ffmlp_forward_planar(ptr(inputs_planar), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                           output_activation, ptr(outputs), stream_of(inputs_planar)), "ffmlp_forward_planar")

    @staticmethod
    def ffmlp_backward_planar(grad, inputs_planar, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                              calc_grad_inputs, grad_inputs_planar, grad_weights):
        ts = (grad, inputs_planar, weights, grad_inputs_planar, grad_weights)
        require_cuda(*ts); _half(*ts); _contig(*ts)
        ws = _scratch.get("ffmlp_ws", lib.foc_ffmlp_backward_workspace_bytes(input_dim, hidden_dim, num_layers), grad.device)
        check(lib.foc_ffmlp_backward_planar(ptr(grad), ptr(inputs_planar), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                            activation, output_activation, int(bool(calc_grad_inputs)), ptr(grad_inputs_planar),
                                            ptr(grad_weights), ptr(ws), ws.numel(), stream_of(grad)), "ffmlp_backward_planar")

    @staticmethod
    def allocate_splitk(size):
        check(lib.foc_allocate_splitk(int(size)), "allocate_splitk")

    @staticmethod
    def free_splitk():
        check(lib.foc_free_splitk(), "free_splitk")

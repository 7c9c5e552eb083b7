"""ctypes binding of libfocnerf_hip.so (the C ABI declared in include/focnerf.h).

There is NO fallback: if the shared library is missing or fails to load, importing this
module raises ImportError, and every op raises RuntimeError when the library reports an
error. PyTorch is used only for device memory and streams (tensor.data_ptr(),
torch.cuda.current_stream()).
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: the library then binds to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FOCNERF_LIB_PATH: another build of the same library (A/B timing of kernel variants on one box, tools/ab_libs.sh)
LIB_PATH = os.environ.get("FOCNERF_LIB_PATH") or os.path.join(_HERE, "libfocnerf_hip.so")

c_u8p = ctypes.c_void_p
c_vp = ctypes.c_void_p
u32 = ctypes.c_uint32
u64 = ctypes.c_uint64
f32 = ctypes.c_float
i32 = ctypes.c_int

FOC_F32 = 0
FOC_F16 = 1



class FocOccTrainNode(ctypes.Structure):
    """include/focnerf.h `FocOccTrainNode`, field for field (tests/test_abi.py parses the header and compares names, order and size)."""
    _fields_ = [
        ("struct_bytes", u32),
        ("n_rays", u32), ("max_steps", u32), ("cascade", u32), ("grid_size", u32), ("cap", u32), ("pad_align", u32),
        ("bound", f32), ("dt_gamma", f32), ("min_near", f32),
        ("rays_o", c_vp), ("rays_d", c_vp), ("aabb", c_vp), ("jitter", c_vp),
        ("bitfield", c_vp),
        ("nears", c_vp), ("fars", c_vp), ("enc_in", c_vp), ("deltas", c_vp),
        ("sh_rows", c_vp),
        ("rays", c_vp), ("counter", c_vp), ("march_scratch", c_vp),
        ("levels", u32), ("base_resolution", u32), ("gridtype", u32), ("interp", u32),
        ("align_corners", ctypes.c_int32), ("table_dtype", ctypes.c_int32),
        ("per_level_scale_log2", f32),
        ("embeddings", c_vp),
        ("offsets", c_vp), ("offsets_host", c_vp),
        ("planes", c_vp), ("grid_workspace", c_vp),
        ("grid_workspace_bytes", u64),
        ("sigma_input_dim", u32), ("sigma_hidden", u32), ("sigma_layers", u32), ("sigma_activation", u32), ("sigma_output_activation", u32),
        ("color_hidden", u32), ("color_layers", u32), ("color_activation", u32), ("c_width", u32),
        ("w_sigma", c_vp), ("w_color", c_vp),
        ("h", c_vp), ("c", c_vp),
        ("T_thresh", f32), ("density_scale", f32), ("bg_scalar", f32),
        ("bg_ray", c_vp),
        ("weights_sum", c_vp), ("image_raw", c_vp), ("image", c_vp), ("depth", c_vp),
        ("precounted", ctypes.c_int32),
        ("grad_image", c_vp), ("grad_ws", c_vp),
        ("grad_c", c_vp), ("grad_h0", c_vp), ("grad_h", c_vp), ("grad_planes", c_vp), ("grad_w_color", c_vp), ("grad_w_sigma", c_vp),
        ("grad_embeddings", c_vp), ("mlp_workspace", c_vp),
        ("mlp_workspace_bytes", u64),
    ]


# name -> (restype, [argtypes]) — one entry per declaration in include/focnerf.h
SIGNATURES = {
    "foc_abi_version": (i32, []),
    "foc_last_error": (ctypes.c_char_p, []),
    "foc_arch": (ctypes.c_char_p, []),
    "foc_near_far_from_aabb": (i32, [c_vp, c_vp, c_vp, u32, f32, c_vp, c_vp, c_vp]),
    "foc_sph_from_ray": (i32, [c_vp, c_vp, f32, u32, c_vp, c_vp]),
    "foc_morton3D": (i32, [c_vp, u32, c_vp, c_vp]),
    "foc_morton3D_invert": (i32, [c_vp, u32, c_vp, c_vp]),
    "foc_packbits": (i32, [c_vp, u32, f32, c_vp, c_vp]),
    "foc_march_rays_train": (i32, [c_vp, c_vp, c_vp, f32, f32, u32, u32, u32, u32, u32, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "foc_march_rays_train_field": (i32, [c_vp, c_vp, c_vp, f32, f32, u32, u32, u32, u32, u32, c_vp, c_vp,
                                         c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, c_vp, f32, c_vp]),
    "foc_set_option": (i32, [ctypes.c_char_p, i32]),
    "foc_get_option": (i32, [ctypes.c_char_p, c_vp]),
    "foc_guard_pick_device": (i32, [i32, i32, i32, i32]),
    "foc_grid_forward_index_path": (i32, [u32, u32, u32]),
    "foc_view_tile_order": (i32, [c_vp, u32, u32, u32, c_vp, c_vp, c_vp]),
    "foc_occ_train_forward": (i32, [ctypes.POINTER(FocOccTrainNode), c_vp]),
    "foc_occ_train_backward": (i32, [ctypes.POINTER(FocOccTrainNode), c_vp]),
    "foc_occ_tail_forward": (i32, [c_vp, c_vp, u32, c_vp, c_vp, u32, u32, f32, f32, c_vp, f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "foc_occ_tail_backward": (i32, [c_vp, c_vp, c_vp, c_vp, u32, c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, f32, f32, c_vp, f32, c_vp, c_vp, c_vp]),
    "foc_march_rays_train_scratch_bytes": (u64, [u32, u32]),
    "foc_composite_rays_train_forward": (i32, [c_vp, c_vp, c_vp, c_vp, u32, u32, f32, c_vp, c_vp, c_vp, c_vp]),
    "foc_composite_rays_train_backward": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, f32,
                                                c_vp, c_vp, c_vp]),
    "foc_march_rays": (i32, [u32, u32, c_vp, c_vp, c_vp, c_vp, f32, f32, u32, u32, u32, c_vp, c_vp, c_vp,
                             c_vp, c_vp, c_vp, c_vp, c_vp]),
    "foc_composite_rays": (i32, [u32, u32, f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "foc_compact_alive": (i32, [c_vp, u32, c_vp, c_vp, c_vp, c_vp]),
    "foc_march_rays_two_phase": (i32, [u32, u32, c_vp, c_vp, c_vp, c_vp, f32, f32, u32, u32, u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, i32, c_vp]),
    "foc_composite_compact": (i32, [u32, u32, f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, i32, c_vp]),
    "foc_march_rays_two_phase_fills": (i32, [u32, i32]),
    "foc_march_rays_two_phase_sample_major": (i32, [u32, u32, i32]),
    "foc_occ_render_step_scratch_bytes": (u64, [u32]),
    "foc_occ_render_step": (i32, [u32, u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, f32, f32, u32, u32, u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, u32, f32, u32, c_vp, u32, c_vp, u32, u32, c_vp, f32, c_vp, c_vp, c_vp, c_vp, u32, c_vp, u32, u32, c_vp]),
    "foc_grid_encode_forward": (i32, [c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, f32, u32, c_vp, u32, i32, u32,
                                      i32, c_vp, c_vp]),
    "foc_grid_encode_forward_bl": (i32, [c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, f32, u32, c_vp, u32, i32, u32,
                                         i32, c_vp, c_vp]),
    "foc_grid_encode_backward": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, f32, u32, c_vp, c_vp,
                                       u32, i32, u32, i32, i32, c_vp, c_vp]),
    "foc_grid_encode_backward_workspace_bytes": (u64, [u32, u32, u32, u32, i32]),
    "foc_grid_planes_to_rows": (i32, [c_vp, c_vp, u32, u32, u32, c_vp]),
    "foc_grid_rows_to_planes": (i32, [c_vp, c_vp, u32, u32, u32, c_vp]),
    "foc_grid_encode_backward_binned": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, f32, u32, c_vp, c_vp,
                                              u32, i32, u32, i32, i32, c_vp, c_vp, u64, c_vp]),
    "foc_grid_encode_forward_counted": (i32, [c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, f32, u32, u32, i32, u32, i32, c_vp, c_vp, u64, c_vp]),
    "foc_grid_encode_backward_count": (i32, [c_vp, c_vp, u32, u32, u32, u32, f32, u32, u32, i32, u32, i32, c_vp, c_vp, u64, c_vp]),
    "foc_grid_encode_backward_binned_counted": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, f32, u32, c_vp, c_vp,
                                                      u32, i32, u32, i32, i32, c_vp, c_vp, u64, c_vp]),
    "foc_grad_total_variation": (i32, [c_vp, c_vp, c_vp, c_vp, f32, u32, u32, u32, u32, f32, u32, u32, i32, i32, c_vp]),
    "foc_freq_encode_forward": (i32, [c_vp, u32, u32, u32, u32, c_vp, c_vp]),
    "foc_freq_encode_backward": (i32, [c_vp, c_vp, u32, u32, u32, u32, c_vp, c_vp]),
    "foc_ffmlp_forward": (i32, [c_vp, c_vp, u32, u32, u32, u32, u32, u32, u32, c_vp, c_vp, c_vp]),
    "foc_ffmlp_inference": (i32, [c_vp, c_vp, u32, u32, u32, u32, u32, u32, u32, c_vp, c_vp, c_vp]),
    "foc_ffmlp_backward": (i32, [c_vp, c_vp, c_vp, c_vp, u32, u32, u32, u32, u32, u32, u32, i32,
                                 c_vp, c_vp, c_vp, c_vp, u64, c_vp]),
    "foc_ffmlp_backward_workspace_bytes": (u64, [u32, u32, u32]),
    "foc_ffmlp_forward_planar": (i32, [c_vp, c_vp, u32, u32, u32, u32, u32, u32, u32, c_vp, c_vp]),
    "foc_ffmlp_backward_planar": (i32, [c_vp, c_vp, c_vp, u32, u32, u32, u32, u32, u32, u32, i32, c_vp, c_vp, c_vp, u64, c_vp]),
    "foc_allocate_splitk": (i32, [u64]),
    "foc_free_splitk": (i32, []),
    "foc_combine_select": (i32, [c_vp, c_vp, c_vp, c_vp, u64, c_vp]),
    "foc_combine_pack_keys": (i32, [c_vp, u32, c_vp, u64, c_vp]),
    "foc_combine_unpack": (i32, [c_vp, u32, c_vp, c_vp, c_vp, u64, c_vp]),
    "foc_combine_select_composite": (i32, [c_vp, u32, c_vp, c_vp, u32, u32, c_vp, u32, c_vp, c_vp, c_vp, c_vp]),
    "foc_combine_select4": (i32, [c_vp, c_vp, u64, c_vp]),
    "foc_mo_select": (i32, [c_vp, c_vp, c_vp, c_vp, u64, u32, u32, c_vp]),
    "foc_fixed_field_pack": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, f32, u32, u32, f32, f32, c_vp, c_vp, c_vp, c_vp, u32, c_vp]),
    "foc_composite_fixed_steps": (i32, [c_vp, c_vp, c_vp, c_vp, u32, u32, f32, c_vp, c_vp, c_vp]),
    "foc_fixed_sample": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, f32, c_vp, c_vp, c_vp, u32, c_vp]),
    "foc_fixed_tail_forward": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, f32, u32, u32, f32, f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, c_vp, c_vp]),
    "foc_fixed_tail_backward": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, f32, u32, u32, f32, f32, c_vp, c_vp, u32, c_vp, c_vp]),
    "foc_fixed_head_forward": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, c_vp]),
    "foc_fixed_head_backward": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u32, u32, f32, c_vp, u32, c_vp]),
    "foc_color_head_forward": (i32, [c_vp, c_vp, u32, c_vp, u32, u32, u32, u32, c_vp, u32, c_vp, c_vp]),
    "foc_field_forward_train": (i32, [c_vp, c_vp, u32, c_vp, u32, c_vp, u32, u32, u32, u32, c_vp, c_vp, u32, c_vp, c_vp]),
    "foc_color_head_backward": (i32, [c_vp, c_vp, c_vp, u32, c_vp, c_vp, u32, u32, u32, u32, c_vp, c_vp, c_vp, u64, u32, c_vp, c_vp, c_vp]),
    "foc_fixed_composite_forward": (i32, [c_vp, c_vp, c_vp, f32, u32, u32, f32, c_vp, c_vp]),
    "foc_fixed_composite_backward": (i32, [c_vp, c_vp, c_vp, c_vp, f32, u32, u32, f32, c_vp, c_vp, c_vp]),
    "foc_sample_head_forward": (i32, [c_vp, c_vp, u64, c_vp, c_vp, c_vp, u32, c_vp]),
    "foc_sample_head_backward": (i32, [c_vp, c_vp, c_vp, u64, c_vp, u32, c_vp]),
    "foc_sh_encode": (i32, [c_vp, u64, c_vp, c_vp]),
    "foc_rgb_head_forward": (i32, [c_vp, u64, c_vp, c_vp]),
    "foc_rgb_head_backward": (i32, [c_vp, c_vp, u64, c_vp, c_vp]),
    "foc_fixed_render_inference": (i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, f32, u32, u32, f32, f32, c_vp, c_vp, c_vp, c_vp, u32, c_vp, c_vp]),
    "foc_nerf_field_inference": (i32, [c_vp, i32, c_vp, u32, u32, u32, c_vp, u32, c_vp, u32, u32, u32, u32, c_vp, c_vp, c_vp, c_vp]),
    "foc_mark_untrained_grid": (i32, [c_vp, u32, f32, f32, f32, f32, f32, u32, u32, c_vp, c_vp, c_vp]),
    "foc_grid_cells_xyz": (i32, [u32, u32, f32, c_vp, c_vp, c_vp]),
    "foc_grid_update_sample_workspace_bytes": (u64, [u32, u32]),
    "foc_grid_update_sample": (i32, [c_vp, u32, u32, f32, u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, u64, c_vp]),
    "foc_grid_update_apply_workspace_bytes": (u64, [u32, u32]),
    "foc_grid_update_apply": (i32, [c_vp, u32, u32, c_vp, c_vp, u32, f32, f32, f32, c_vp, c_vp, c_vp, u64, c_vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"focnerf_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C focnerf_amd/csrc`. There is no CPU or PyTorch fallback for these ops.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"focnerf_amd: failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib


class _Stream(ctypes.c_void_p):
    """hipStream_t handle that remembers the device of the tensor it was taken for (`stream_of`)."""
    device_index = None


_multi_device = None        # torch.cuda.device_count() > 1, asked once (counting devices does not initialise the GPU)


def _on_tensor_device(fn):
    """Entry points taking a stream run with the device of THEIR tensors current. The library's own guard (csrc/common.h FocDeviceGuard)
    reads the device from the stream handle, and torch's default stream is the null handle on every device — `cuda:1` tensors on their
    default stream while `cuda:0` is current would launch on device 0 with device-1 pointers. The device therefore comes from the
    tensor `stream_of` was called with. One-GPU processes skip the check."""
    def call(*args):
        global _multi_device
        if _multi_device is None:
            _multi_device = torch.cuda.device_count() > 1
        if _multi_device and args:
            idx = getattr(args[-1], "device_index", None)
            if idx is not None and idx != torch.cuda.current_device():
                with torch.cuda.device(idx):
                    return fn(*args)
        return fn(*args)
    call.__name__ = getattr(fn, "__name__", "foc_call")
    return call


class _Lib:
    """The loaded library with every stream-taking entry point wrapped by `_on_tensor_device`."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name, (_, args) in SIGNATURES.items():
            fn = getattr(cdll, name)
            setattr(self, name, _on_tensor_device(fn) if args and args[-1] is c_vp and not name.endswith("_bytes") else fn)


lib = _Lib(_load())


def check(rc, what=""):
    if rc != 0:
        msg = lib.foc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"focnerf_amd {what}: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (or None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) if os.environ.get("FOC_RAW_STREAM", "1") != "0" else None


def raw_stream(index):
    """hipStream_t (as an int) of torch's current stream on device `index`. `torch.cuda.current_stream(dev).cuda_stream` builds a Stream
    object per call (4.7 us: the largest single item of the occupancy render loop's host time, ten calls per iteration); torch's raw
    accessor returns the handle itself."""
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def stream_of(t=None):
    """hipStream_t of torch's current stream on the tensor's device."""
    dev = t.device if t is not None else None
    index = dev.index if dev is not None and dev.index is not None else torch.cuda.current_device()
    s = _Stream(raw_stream(index))
    s.device_index = index
    return s


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("focnerf_amd: expected a CUDA(HIP) tensor; these ops have no CPU implementation")


def dtype_code(t):
    if t.dtype == torch.float32:
        return FOC_F32
    if t.dtype == torch.float16:
        return FOC_F16
    raise RuntimeError(f"focnerf_amd: unsupported dtype {t.dtype} (float32 or float16 expected)")


# ---------------------------------------------------------------- library options (csrc/common.h FocOpt, include/focnerf.h foc_set_option)
def get_option(name):
    """Current value of a library switch (an int; FOC_OCC_MARCH_FORM: -1 auto, 0 two, 1 row, 2 lane, 3 staged)."""
    v = ctypes.c_int(0)
    check(lib.foc_get_option(name.encode(), ctypes.cast(ctypes.byref(v), ctypes.c_void_p)), "get_option")
    return v.value


def set_option(name, value):
    """Set a library switch for the rest of the process (the environment variable of the same name only sets its INITIAL value)."""
    check(lib.foc_set_option(name.encode(), int(value)), "set_option")


class option:
    """`with option("FOC_GB_FACTORED", 0): ...` — a switch set for the block and put back afterwards (tests, A/B runs)."""

    def __init__(self, name, value):
        self.name, self.value = name, int(value)

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False

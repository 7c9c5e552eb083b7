"""CPU: the C-ABI library loads and exports every symbol include/focnerf.h declares (no compute calls)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "focnerf.h")


def _declared():
    text = open(os.path.join(REPO, "include", "focnerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(foc_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for n in ["foc_near_far_from_aabb", "foc_march_rays_train", "foc_composite_rays_train_forward",
              "foc_composite_rays_train_backward", "foc_march_rays", "foc_composite_rays", "foc_grid_encode_forward",
              "foc_grid_encode_backward", "foc_grad_total_variation", "foc_freq_encode_forward", "foc_freq_encode_backward",
              "foc_ffmlp_forward", "foc_ffmlp_inference", "foc_ffmlp_backward", "foc_combine_select"]:
        assert n in names


def test_library_exports_every_declared_symbol():
    from focnerf_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in _declared():
        assert hasattr(lib, n), f"{n} declared in include/focnerf.h but not exported"
    assert set(_lib.SIGNATURES) == set(_declared()), "focnerf_amd/_lib.py binds a different set than the header declares"


def test_abi_version_and_arch():
    from focnerf_amd import _lib
    assert _lib.lib.foc_abi_version() == 2
    assert _lib.lib.foc_arch() == b"gfx950"
    assert isinstance(_lib.lib.foc_last_error(), bytes)           # (empty in a fresh process; other tests of a session may have left theirs)


def test_argument_validation_needs_no_gpu():
    """Null pointers / bad shapes are rejected on the host before any launch."""
    from focnerf_amd import _lib
    lib = _lib.lib
    assert lib.foc_near_far_from_aabb(None, None, None, 4, 0.2, None, None, None) == 1
    assert b"null pointer" in lib.foc_last_error()
    one = ctypes.c_void_p(8)  # never dereferenced: validation fails first
    rc = lib.foc_grid_encode_forward(one, one, one, one, 4, 3, 3, 16, 1.0, 16, None, 0, 0, 0, 0, None, None)
    assert rc == 1 and b"C must be 1, 2, 4, or 8" in lib.foc_last_error()
    rc = lib.foc_ffmlp_forward(one, one, 128, 32, 16, 48, 2, 0, 6, one, one, None)
    assert rc == 1 and b"hidden_dim" in lib.foc_last_error()
    rc = lib.foc_ffmlp_forward(one, one, 128, 24, 16, 64, 2, 0, 6, one, one, None)
    assert rc == 1 and b"input_dim" in lib.foc_last_error()
    rc = lib.foc_freq_encode_forward(one, 4, 3, 4, 26, one, None)
    assert rc == 1
    # ABI 2: the MLP backward's workspace travels with its size; a buffer sized by version 1's blob formula (~50 KB) is refused, not overrun
    need = lib.foc_ffmlp_backward_workspace_bytes(32, 64, 2)
    assert need > 10 * 1024 * 1024
    rc = lib.foc_ffmlp_backward(one, one, one, None, 128, 32, 16, 64, 2, 0, 6, 1, None, one, one, one, 64 * (32 + 64 + 16) * 4, None)
    assert rc == 1 and b"workspace of" in lib.foc_last_error()
    rc = lib.foc_ffmlp_backward_planar(one, one, one, 128, 32, 16, 64, 2, 0, 6, 1, one, one, one, need - 1, None)
    assert rc == 1 and b"workspace of" in lib.foc_last_error()
    # the colour head with an object feature needs the 48-wide sizing: a buffer sized for 32 inputs is refused
    rc = lib.foc_color_head_backward(one, one, one, 1, None, one, 128, 64, 2, 0, one, one, one, lib.foc_ffmlp_backward_workspace_bytes(32, 64, 2), 4, one, None, None)
    assert rc == 1 and b"workspace of" in lib.foc_last_error()


def test_occ_train_node_struct_matches_the_header_and_is_validated_on_the_host():
    """include/focnerf.h `FocOccTrainNode` against its ctypes mirror (focnerf_amd/_lib.py): same field names in the same order, and the size
    a C compiler gives the header's struct; a null node, a node of another size and an empty node are refused before any launch."""
    import re
    import subprocess
    import tempfile
    from focnerf_amd import _lib
    text = open(HEADER).read()
    body = re.search(r"typedef struct FocOccTrainNode \{(.*?)\} FocOccTrainNode;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"[A-Za-z_0-9]+$", first.strip())[0])
        names += [r.strip().lstrip("*").strip() for r in rest]
    assert names == [f[0] for f in _lib.FocOccTrainNode._fields_]
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "%s"\nint main(void) { printf("%%zu", sizeof(FocOccTrainNode)); return 0; }\n' % HEADER)
        subprocess.run(["gcc", "-o", os.path.join(tmp, "sz"), src], check=True)
        assert int(subprocess.run([os.path.join(tmp, "sz")], capture_output=True, text=True, check=True).stdout) == ctypes.sizeof(_lib.FocOccTrainNode)
    lib = _lib.lib
    for fn in (lib.foc_occ_train_forward, lib.foc_occ_train_backward):
        assert fn(None, None) == 1 and b"null node" in lib.foc_last_error()
        node = _lib.FocOccTrainNode()
        node.struct_bytes = ctypes.sizeof(_lib.FocOccTrainNode) - 8
        assert fn(ctypes.byref(node), None) == 1 and b"bytes" in lib.foc_last_error()
        node.struct_bytes = ctypes.sizeof(_lib.FocOccTrainNode)
        assert fn(ctypes.byref(node), None) == 1 and b"empty node" in lib.foc_last_error()
        node.cap, node.n_rays = 128, 4
        assert fn(ctypes.byref(node), None) == 1 and b"workspace" in lib.foc_last_error()


def test_a_c_caller_links_the_library_and_gets_host_side_refusals(tmp_path):
    """tests/c_abi_consumer.cpp, mode "cpu": a program that is not Python includes include/focnerf.h, links libfocnerf_hip.so and gets the ABI
    version, refusals with messages, the option table — without a device (the device half runs in tests/test_gpu_edge_cases.py)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from util import build_c_abi_consumer
    import oracle                                                    # noqa: F401 — builds oracle/_build/liboracle.so if it is missing
    exe = build_c_abi_consumer(tmp_path)
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C_ABI_CONSUMER_OK cpu" in r.stdout, r.stdout + r.stderr


def test_device_selection_rule_of_the_entry_points():
    """csrc/common.h FocDeviceGuard: a non-null stream decides; the NULL stream (torch's default stream on EVERY device) says nothing, then
    the first pointer argument's device does — a C caller with cuda:1 buffers on the default stream while cuda:0 is current; neither: the
    current device. (The two-device launch itself is tests/test_gpu_edge_cases.py::test_ops_follow_their_tensors_device, skipped below 2 GPUs.)"""
    from focnerf_amd import _lib
    pick = _lib.lib.foc_guard_pick_device
    assert pick(0, 3, 1, 0) == 3            # a real stream: its device, whatever the pointer or the current device say
    assert pick(1, -1, 1, 0) == 1           # NULL stream, device-1 pointer, device 0 current: device 1
    assert pick(1, 2, 1, 0) == 1            # a stream device reported for the NULL handle is ignored
    assert pick(1, -1, -1, 2) == 2          # NULL stream, no device pointer (host memory / NULL): stay where we are
    assert pick(0, -1, 1, 0) == 1           # a stream whose device cannot be read: fall back to the pointer
    assert pick(0, -1, -1, 5) == 5


def test_forward_index_path_admission_needs_no_gpu():
    """Which index arithmetic the fp16 D = 3 C = 2 forward takes per level (csrc/gridencoder.hip ge_forward_level): 24-bit multiplies up to
    2^22 rows, 32-bit byte offsets up to 2^30 rows (hashed levels: power-of-two sizes only), the general form beyond — the guard itself,
    asked on the host (a 2^30-row level is 4 GB of table: the GPU tests stop at 2^23)."""
    from focnerf_amd import _lib
    path = _lib.lib.foc_grid_forward_index_path
    assert path(4920, 16, 0) == 2                              # dense level 0 of the NeRF grid
    assert path(1 << 19, 2048, 373248) == 2                    # a hashed 2^19-row level
    assert path(1 << 22, 2048, 0) == 2 and path((1 << 22) + 8, 160, 0) == 1      # the 24-bit limit (a dense level of 2^22 + 8 rows: generic offsets)
    assert path(1 << 23, 2048, 0) == 1 and path(1 << 30, 4096, 0) == 1
    assert path((1 << 30) + 8, 4096, 0) == 0                   # beyond 2^30 rows: byte offsets no longer fit 32 bits
    assert path(1 << 31, 4096, 0) == 0
    assert path((1 << 19) + 8, 2048, 0) == 0                   # hashed, not a power of two: `%` is no mask
    assert path(1 << 19, 2048, 1) == 0 and path((1 << 19) + 1, 40, 0) == 0       # odd offset / odd size: row pairs not 8-byte aligned


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from focnerf_amd import raymarching
    if torch.cuda.is_available():
        pytest.skip("host-only check")
    with pytest.raises(Exception):   # .cuda() on a GPU-less host, or the explicit CUDA-tensor check
        raymarching.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]), 0.2)

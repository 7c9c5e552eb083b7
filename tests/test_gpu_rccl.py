"""GPU, >= 2 devices: the one-object-per-rank combine over RCCL (`torch.distributed` backend "nccl") — the exchange step of
COMBINED.py:592-618 as `ObjectCombiner.render_view` runs it on the N-GPU node: async `all_to_all_single` of the packed per-sample fields
(double-buffered, overlapped with the next chunk's field evaluation), fused select + composite, one `all_gather_into_tensor` per view.

The gloo tests (tests/test_combine_gloo.py) cover the host logic with injected CPU ops and the single-GPU tests the device kernels; THIS
test is the combination on real links. Skipped on a one-GPU box; on any box with two or more GPUs it starts one fresh process per rank
(tests/rccl_worker.py, the reference's own editable.npz / combined.npz fixtures) before those processes touch a GPU, and requires from
every rank: bit-identical images to the single-device combine of all objects, 1e-4 against the reference's images, overlap on and off."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("world", [2, 4])
def test_render_view_over_rccl_equals_single_device_combine(tmp_path, world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "rccl_worker.py"), str(tmp_path)], env=env, cwd=REPO,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:                                             # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    for r in range(world):
        rep = json.load(open(os.path.join(tmp_path, f"rank{r}.json")))
        assert rep["ok"] and rep["world"] == world and rep["backend"] == "nccl", rep
        assert len(rep["cases"]) >= 5 and all(c["bitwise_equal_to_single_device"] for c in rep["cases"]), rep
        assert any(c["overlap"] for c in rep["cases"]) and any(not c["overlap"] for c in rep["cases"])
        assert all(c["bytes_sent"] > 0 for c in rep["cases"])


def _run_one_rank(script, args, tmp_path):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", script)] + args, env=env, cwd=REPO, capture_output=True, text=True, timeout=420)
    tail = (r.stdout + r.stderr)[-3000:]
    if r.returncode == 77:
        pytest.skip("the nccl (RCCL) process group could not be created on this box: " + tail[-400:])
    assert r.returncode == 0, tail
    return r.stdout, tail


def test_one_rank_fixtures_through_rccl(tmp_path):
    """What a one-GPU box CAN rehearse of the N > 1 path: the worker above as the only rank, with the combiner issuing its collectives anyway
    (`ObjectCombiner(collectives_at_world_1=True)`) — `all_to_all_single` on the double-buffered pieces and `all_gather_into_tensor` really go
    through RCCL (which moves the data on the device), for the reference's editable.npz / combined.npz fixtures and the tie case, overlap on
    and off. Covers the binding (buffers, dtypes, split sizes) and the ordering of RCCL's stream against the kernels on both sides of
    it; says nothing about links."""
    _run_one_rank("rccl_worker.py", [str(tmp_path)], tmp_path)
    rep = json.load(open(os.path.join(tmp_path, "rank0.json")))
    assert rep["ok"] and rep["world"] == 1 and rep["backend"] == "nccl", rep
    assert len(rep["cases"]) >= 13 and all(c["bitwise_equal_to_single_device"] for c in rep["cases"]), rep
    assert any(c["overlap"] for c in rep["cases"]) and any(not c["overlap"] for c in rep["cases"])


def test_one_rank_library_producers_and_every_collective_through_rccl(tmp_path):
    """tests/rccl_one_rank_worker.py: resident objects evaluated by `render_field4` (this library's kernels write the send buffers) through
    `render_view` over RCCL, repeated views, pieces larger / equal / much smaller than the view; plus `select`, `render_chunk` and
    `render_chunk_fast` (all_reduce MAX / SUM, all_gather) — each bit for bit what the exchange-free single-rank combiner returns."""
    out, tail = _run_one_rank("rccl_one_rank_worker.py", [], tmp_path)
    assert "RCCL_ONE_RANK_OK backend nccl" in out, tail

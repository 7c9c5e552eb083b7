"""One rank of tests/test_gpu_rccl.py (NOT a test module): started as a fresh process, one per GPU, before anything in it touches a GPU.

    python tests/rccl_worker.py <out_dir>      with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment

Pushes the reference's own fixtures (tests/golden/editable.npz: editable.py's 8-object loop, two views; combined.npz: COMBINED.py's
4-object loop) through `ObjectCombiner.render_view` over the `nccl` backend (= RCCL): objects split over the ranks in checkpoint order,
all-to-all of the packed per-sample fields by ray, fused select + composite of each rank's ray slices, one all-gather per view
(COMBINED.py:592-618). Every rank compares what it got with the single-device `combine_packed` of all objects, BIT FOR BIT (same kernels,
same operand order: the select's tie rule runs in rank = checkpoint order), and with the reference's images within 1e-4. Overlap on and
off (async all-to-all under the next chunk's evaluation, double-buffered), a ragged last chunk, a chunk smaller than the world's slices.
With WORLD_SIZE=1 the combiner is told to issue its collectives all the same (`collectives_at_world_1`): every case then goes through
RCCL on the one device. Writes <out_dir>/rank<r>.json; exit code 0 = all checks passed, 77 = the process group could not be created."""
import json
import os
import sys
import datetime

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pack(dens, rgb, dev):
    return torch.from_numpy(np.concatenate([dens[..., None], rgb], -1).astype(np.float32)).to(dev).contiguous()


def main():
    out_dir = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    try:
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
        probe = torch.ones(8, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:                                  # noqa: BLE001 — RCCL cannot start on this box: nothing of this repo was reached
        print("RCCL_UNAVAILABLE", repr(e), flush=True)
        sys.exit(77)
    from focnerf_amd.combine import ObjectCombiner, combine_packed
    # started as the only rank (a one-GPU box), the combiner issues its collectives anyway: the same cases then rehearse the RCCL calls
    # themselves — RCCL moves the data on the device — instead of skipping them (a single rank needs no exchange)
    comb = ObjectCombiner(collectives_at_world_1=world == 1)
    report = {"rank": rank, "world": dist.get_world_size(), "backend": dist.get_backend(), "device": torch.cuda.get_device_name(dev), "cases": []}
    assert comb.rank == rank and comb.world == world
    ok_all = True

    def run_case(name, fields, nears, fars, T, ref_white, ref_black, ref_depth, chunk, overlap):
        nonlocal ok_all
        K, N = len(fields), fields[0].shape[0]
        per_rank = K // world
        mine = list(range(rank * per_rank, (rank + 1) * per_rank))
        calls = []

        def make_fn(k):
            def fn(lo, hi, out):
                calls.append((k, lo, hi))
                if out is not None and k % 2 == 0:            # both protocols: fill the offered send buffer, or return a fresh tensor
                    out.copy_(fields[k][lo:hi])
                    return out
                return fields[k][lo:hi].clone()
            return fn
        img, dep = comb.render_view([make_fn(k) for k in mine], N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk, overlap=overlap)
        torch.cuda.synchronize()
        want_img, want_dep = combine_packed(fields, nears, fars, (1.0, 0.0))       # all K objects on this one device
        same = bool(torch.equal(img, want_img) and torch.equal(dep.nan_to_num(), want_dep.nan_to_num()))
        err = None
        if ref_white is not None:
            fin = np.isfinite(ref_depth)
            err = max(float(np.abs(img[0].cpu().numpy() - ref_white).max()), float(np.abs(img[1].cpu().numpy() - ref_black).max()),
                      float(np.abs(dep.cpu().numpy()[fin] - ref_depth[fin]).max()))
        ok = same and (err is None or err <= 1e-4) and len(calls) == per_rank * ((N + chunk - 1) // chunk)
        ok_all = ok_all and ok
        report["cases"].append({"case": name, "K": K, "N": N, "T": T, "chunk": chunk, "overlap": overlap, "bitwise_equal_to_single_device": same,
                                "max_abs_err_vs_reference": err, "bytes_sent": int(comb.bytes_sent), "ok": ok})

    # ---- configs[4]: editable.py, 8 objects, two views (the reference's own per-object fields feed the exchange)
    fx = np.load(os.path.join(GOLDEN, "editable.npz"))
    K, T = int(fx["K"]), int(fx["T"])
    if K % world == 0:
        for v in range(2):
            fields = [pack(fx[f"v{v}_densities"][k], fx[f"v{v}_rgbs"][k], dev) for k in range(K)]
            nears, fars = torch.from_numpy(fx[f"v{v}_nears"]).to(dev), torch.from_numpy(fx[f"v{v}_fars"]).to(dev)
            N = fields[0].shape[0]
            for chunk, overlap in ((int(fx["chunk"]), True), (max(3, N // 3 + 1), False), (N, True), (7, True)):
                run_case(f"editable.npz view {v}", fields, nears, fars, T, fx[f"v{v}_image_white"], fx[f"v{v}_image_black"], fx[f"v{v}_depth_white"], chunk, overlap)
    # ---- configs[3]: COMBINED.py, 4 objects
    g = np.load(os.path.join(GOLDEN, "combined.npz"))
    dens, rgbs = g["densities"][:, 0], g["rgbs"][:, 0]
    K, N, T = dens.shape
    if K % world == 0:
        fields = [pack(dens[k], rgbs[k], dev) for k in range(K)]
        nears, fars = torch.from_numpy(g["nears"]).to(dev), torch.from_numpy(g["fars"]).to(dev)
        for chunk, overlap in ((max(2, N // 4 + 1), True), (N, False)):
            run_case("combined.npz", fields, nears, fars, T, g["image_white"], g["image_black"], g["depth_white"], chunk, overlap)
    # ---- a larger synthetic view with exact ties across ranks: many chunks in flight, the double buffers reused dozens of times
    rng = np.random.default_rng(5)
    K, N, T = 2 * world, 3001, 64
    d = (rng.random((K, N, T)) ** 4 * 40).astype(np.float32)
    d[rng.random((K, N, T)) < 0.5] = 0
    d[K - 1, :, :8] = d[0, :, :8]                                # exact non-zero ties between the first and the last rank's objects
    c = rng.random((K, N, T, 3)).astype(np.float32)
    fields = [pack(d[k], c[k], dev) for k in range(K)]
    nears = torch.from_numpy((rng.random(N) * 0.5 + 0.2).astype(np.float32)).to(dev)
    fars = nears + 1.5
    for chunk, overlap in ((256, True), (256, False), (1000, True)):
        run_case("synthetic ties", fields, nears, fars, T, None, None, None, chunk, overlap)

    report["ok"] = ok_all
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(report, f)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok_all else 1)


if __name__ == "__main__":
    main()

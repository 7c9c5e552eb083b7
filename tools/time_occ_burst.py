"""The occupancy-grid render loop (native, one call per iteration) under different burst lengths and march forms: s/view and whether the image
and depth are the reference schedule's (FOC_RENDER_BURST=1), bit for bit. Run on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic

dev = torch.device("cuda", 0)
m = bench.build_model(2, dev, cuda_ray=True, seed=0).eval()
poses, intr = bench.make_training_rays(dev, 2, 8, seed=0)
o, d = synthetic.get_rays(poses[:1], intr, 800, 800)
kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4, device_compaction=True)


def run(env, reps=4):
    # FOC_RENDER_BURST is read by the Python loop per view; the other two are switches of the library (set through foc_set_option)
    from focnerf_amd import _lib
    os.environ["FOC_RENDER_BURST"] = env.get("FOC_RENDER_BURST", "8")
    _lib.set_option("FOC_OCC_MARCH_FORM", {"": -1, "two": 0, "row": 1, "lane": 2, "staged": 3}[env.get("FOC_OCC_MARCH_FORM", "")])
    _lib.set_option("FOC_OCC_FIELD_PIECE", int(env.get("FOC_OCC_FIELD_PIECE", str(1 << 23))))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = m.render(o, d, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = m.render(o, d, **kw)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t_ref, ref = run({"FOC_RENDER_BURST": "1"})
print(f"burst 1 (reference schedule): {t_ref * 1e3:.2f} ms/view", flush=True)
for env in ({"FOC_RENDER_BURST": "8"}, {"FOC_RENDER_BURST": "8", "FOC_OCC_MARCH_FORM": "staged"}, {"FOC_RENDER_BURST": "16", "FOC_OCC_MARCH_FORM": "staged"},
            {"FOC_RENDER_BURST": "4", "FOC_OCC_MARCH_FORM": "staged"}, {"FOC_RENDER_BURST": "8", "FOC_OCC_MARCH_FORM": "lane"}, {"FOC_RENDER_BURST": "8", "FOC_OCC_MARCH_FORM": "two"},
            {"FOC_RENDER_BURST": "4"}, {"FOC_RENDER_BURST": "16"}, {"FOC_RENDER_BURST": "16", "FOC_OCC_MARCH_FORM": "lane"},
            {"FOC_RENDER_BURST": "8", "FOC_OCC_FIELD_PIECE": str(1 << 20)}, {"FOC_RENDER_BURST": "8", "FOC_OCC_FIELD_PIECE": str(1 << 23)}):
    t, out = run(env)
    same = torch.equal(out["image"], ref["image"]) and torch.equal(out["depth"], ref["depth"])
    print(f"{env}: {t * 1e3:.2f} ms/view  identical={same}", flush=True)

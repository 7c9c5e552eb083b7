"""Sweep of the native occupancy loop: burst lengths x march forms x step caps / transmittance thresholds, each against the Python loop on the
reference's schedule (boolean-mask compaction): prints every combination whose image or depth differs (none expected). Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focnerf_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from focnerf_amd import synthetic
from test_gpu_network import _model
bound = 2
m = _model(bound, True, seed=5).eval()
o, d = synthetic.make_view_rays(48, 48, bound, 1, seed=8, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for burst in ("1", "2", "3", "4", "5", "8"):
        for form in ("", "row", "staged", "lane", "two"):
            os.environ["FOC_RENDER_BURST"] = burst
            _lib.set_option("FOC_OCC_MARCH_FORM", {"": -1, "two": 0, "row": 1, "lane": 2, "staged": 3}[form])
            for max_steps, thresh in ((1024, 1e-4), (100, 1e-4), (37, 1e-4), (1024, 0.3)):
                kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=max_steps, bg_color=1.0, T_thresh=thresh)
                a = m.render(o, d, device_compaction=False, **kw)
                b = m.render(o, d, device_compaction=True, **kw)
                bad = (a["image"] != b["image"]).any(-1).sum().item()
                badd = (a["depth"] != b["depth"]).sum().item()
                if bad or badd:
                    print(f"burst {burst} form {form!r} max_steps {max_steps} thresh {thresh}: image rays differ {bad}, depth {badd}, max diff {(a['image'] - b['image']).abs().max().item():.3e}")
print("done")

"""MONeRFNetwork — one network object whose `density` / `color` are the per-sample maximum-density answer of K object networks
(reference: nerf/multiobjectnetwork.py:10-110, flag `--mo-density-infer` of flags.py:10; nothing in the reference constructs it).

Reference semantics kept: `density(x)` evaluates every object's `density(x)` and keeps, per sample, the sigma and the geo_feat of the
object with the largest sigma; `color(x, d, yolo_details, **kwargs)` evaluates every object's `density(x)` and `color(x, d, yolo_details,
**kwargs)` and returns the colour of that object; both under `no_grad` and `autocast(enabled=fp16)`; the running maximum is
`torch.max(stack([new, best]))`, so a LATER checkpoint takes ties (COMBINED.py's select keeps the earlier one) and NaN stays.

Differences, on purpose:
* the reference's `get_model_with_checkpoint` (:35-38) reads the checkpoint and returns a freshly initialised model WITHOUT applying it,
  once per object per call; here every object's network is loaded once (`checkpoint.load_objects`, weights-only loader) and stays resident;
* the running select is one kernel per object (`foc_mo_select`: decision + row copy) instead of stack / max / take_along_dim;
* `to()` returns the module (the reference's returns None).
"""
import torch

from .network_foc import NeRFNetwork


class MONeRFNetwork(NeRFNetwork):
    def __init__(self, ckpt_list, model_class=None, fp16=False, nw_args=(), nw_kwargs=None, objects=None):
        """ckpt_list: checkpoint paths in the order that decides ties; model_class(*nw_args, **nw_kwargs) builds one object's network
        (default: this package's FOC network); `objects`: already-built networks to use instead of loading `ckpt_list` (tests, callers
        that hold them anyway)."""
        nw_kwargs = dict(nw_kwargs or {})
        super().__init__(*nw_args, **nw_kwargs)
        self.ckpt_list = list(ckpt_list)
        self.fp16 = fp16
        self.model_class = model_class or NeRFNetwork
        self.nw_args, self.nw_kwargs = tuple(nw_args), nw_kwargs
        self.device = None
        self._objects = list(objects) if objects is not None else None

    # ------------------------------------------------------------------ the K resident objects
    def objects(self):
        if self._objects is None:
            from .checkpoint import load_objects
            device = self.device if self.device is not None else next(self.parameters()).device
            self._objects = load_objects(self.ckpt_list, lambda: self.model_class(*self.nw_args, **self.nw_kwargs), device)
        return self._objects

    def to(self, device=None, *args, **kwargs):
        self.device = device
        out = super().to(device, *args, **kwargs)
        if self._objects is not None:
            self._objects = [m.to(device) for m in self._objects]
        return out

    # ------------------------------------------------------------------ multiobjectnetwork.py:40-97
    def color_and_densities(self, *args, **kwargs):
        with torch.no_grad():
            return self._color_and_densities(*args, **kwargs)

    def _color_and_densities(self, x, density_only=True, color_args=(), color_kwargs=None):
        from .combine import HipCombineOps
        color_kwargs = color_kwargs or {}
        best_sigma, best_rows = None, None
        for model in self.objects():
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
                dens = model.density(x)
                sigma = dens['sigma']
                rows = dens['geo_feat'] if density_only else model.color(x, *color_args, **color_kwargs)
            # one element type for the kernel: the masked colour path returns x.dtype rows beside half densities — widen (exactly) then
            work = sigma.dtype if rows.dtype == sigma.dtype else torch.float32
            out_dtypes = (sigma.dtype, rows.dtype)
            sigma, rows = sigma.to(work).contiguous(), rows.to(work).contiguous()
            if best_sigma is None:
                best_sigma, best_rows = sigma.clone(), rows.clone()
            else:
                if best_sigma.dtype != work:                      # (objects of different classes answering in different types)
                    best_sigma, best_rows, sigma, rows = best_sigma.float(), best_rows.float(), sigma.float(), rows.float()
                HipCombineOps.mo_select(sigma, rows, best_sigma, best_rows)
        if best_sigma is None:
            raise RuntimeError("MONeRFNetwork: empty checkpoint list")
        return best_sigma.to(out_dtypes[0]), best_rows.to(out_dtypes[1])

    def color(self, x, d, yolo_details=None, **kwargs):
        return self.color_and_densities(x, density_only=False, color_args=[d, yolo_details], color_kwargs=kwargs)[1]

    def density(self, x, yolo_details=None):
        sigma, geo_feat = self.color_and_densities(x, density_only=True)
        return {'sigma': sigma, 'geo_feat': geo_feat}

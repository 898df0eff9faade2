"""
Expectation-over-Transformation wrapper, same surface as the reference's `EoTWrapper`
(src/defenses/wrappers.py:4-24): `EoTWrapper(model, eot_steps).forward(x: (1,3,H,W)) -> (1, n_classes)`.

When the wrapped model is one of this package's HIP defenders the repeat is folded into the engine (one image read
`eot_steps` times by the image-in kernel, input-gradient summed by its adjoint) instead of materialising
`x.repeat(eot_steps, 1, 1, 1)`; the results are the same numbers.  Any other module takes the reference's literal path.
"""
import torch


class EoTWrapper(torch.nn.Module):

    def __init__(self, model: torch.nn.Module, eot_steps: int):
        super().__init__()
        self.model = model
        self.eot_steps = eot_steps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, 3, h, w) — the reference passes B = 1; B > 1 gives one EoT mean per image, (B, n_classes)."""
        fused = getattr(self.model, 'forward_rows', None)
        if fused is not None:
            preds = fused(x, rep=self.eot_steps)                              # (B * eot, n_classes), image-major
            return preds.view(x.shape[0], self.eot_steps, -1).mean(dim=1)
        if x.shape[0] != 1:
            raise ValueError('the generic EoT path follows the reference and expects a single image')
        x = x.repeat(self.eot_steps, 1, 1, 1)
        preds = self.model(x)
        return torch.mean(preds, dim=0, keepdim=True)

    def class_jacobian(self, x: torch.Tensor, classes=None):
        """ONE forward + ceil(columns / K) backward replays for the per-class input gradients of the EoT-mean logits (the HIP
        engine's K-cotangent plan), or None when the wrapped model has none: attacks.l2_attacks.ClassJacobian then takes one
        autograd backward per class, like the reference (src/attacks/untargeted.py:526-560, :605-635)."""
        fast = getattr(self.model, 'class_jacobian_rows', None)
        return fast(x, self.eot_steps, classes) if fast is not None else None

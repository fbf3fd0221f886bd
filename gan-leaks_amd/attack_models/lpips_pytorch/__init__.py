"""Drop-in for the reference's `attack_models/lpips_pytorch` package surface used on the fbb path:

    import lpips_pytorch as ps
    loss = ps.PerceptualLoss(model='net-lin', net='vgg', use_gpu=True)        # attack_models/utils.py:157
    d = loss.forward(pred, target, normalize=False)                            # -> [N,1,1,1]

(attack_models/lpips_pytorch/__init__.py:9-32 -> DistModel.forward_pair (models/dist_model.py:107-111) -> PNetLin.forward
(models/networks_basic.py:134-181)).  VGG16 + LPIPS v0.1 run in csrc/gl_lpips.hip; weights come from LOCAL files only
(ganleaks_amd.lpips.LpipsModel.from_files; the reference downloads the backbone, pretrained_networks.py:99)."""
from __future__ import annotations

import numpy as np

from ... import lpips as _lp


class PerceptualLoss:
    def __init__(self, model='net-lin', net='vgg', use_gpu=True, lpips_model=None, vgg_path=None, lin_path=None):
        if model != 'net-lin' or net != 'vgg':
            raise NotImplementedError("only model='net-lin', net='vgg' (the configuration attack_models/utils.py:157 builds) is provided")
        print('Setting up Perceptual loss...')
        if lpips_model is None:
            lpips_model = _lp.LpipsModel.from_files(vgg_path, lin_path) if (vgg_path or lin_path) else _lp.default_model()
        self.model = lpips_model
        print('...Done')

    def forward(self, pred, target, normalize=False):
        """pred [N,3,H,W], target [N,3,H,W] or [1,3,H,W]; images in [-1,1] (or [0,1] with normalize=True, as in the reference).
        Returns the LPIPS distances with shape [N,1,1,1]: numpy in -> numpy float32, torch in -> torch tensor on pred's device."""
        is_torch = type(pred).__module__.startswith("torch")
        if normalize:
            target = 2 * target - 1
            pred = 2 * pred - 1
        # forward_pair(target, pred): in0 = target, in1 = pred; the distance is symmetric, the batch is pred's (networks_basic.py:153)
        a = self.model.features(pred)
        g = self.model.features(target)
        lp, _ = _lp.rows_dist(a, g)
        out = np.asarray(lp, np.float32).reshape(-1, 1, 1, 1)
        if is_torch:
            import torch
            return torch.from_numpy(out).to(pred.device)
        return out

    __call__ = forward

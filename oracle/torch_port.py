"""PyTorch-CPU restatement of the reference's per-query search, used ONLY as the timed CPU baseline
(`cpu_baseline.kind = "port"` in bench.py) and as a cross-check of the exact oracle.

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product.

It issues the same torch operations in the same order as the reference, so its run time on the GPU
box's host cores is what `attack_models/fbb.py` would take there:
    custom_knn                    attack_models/fbb.py:73-88
    Loss('l2').forward            attack_models/utils.py:163,169,171-177
    Loss('l2-lpips').forward      attack_models/utils.py:166-176 -> PerceptualLoss.forward (lpips_pytorch/__init__.py:17-32, argument swap)
                                  -> PNetLin.forward (models/networks_basic.py:134-181) on the vgg16 slices (models/pretrained_networks.py:96-134)
The reference file cannot travel to the GPU box; tests/test_oracle.py (build container) checks this
restatement against golden vectors produced by the real custom_knn.
"""
import torch


def l2_loss(x_hat, x_gt):
    # utils.py:163 loss_l2_fn(x, y) = mean((y - x)**2, dim=[1,2,3]); utils.py:176 vec = 0.2 * 0. + l2
    return 0.2 * 0.0 + torch.mean((x_gt - x_hat) ** 2, dim=[1, 2, 3])


def custom_knn(bank, query, loss, batch_size):
    """same op sequence as fbb.py:73-88: per full batch one loss() call and one index tensor built
    from a python range (both are part of what the reference spends its time on), then cat + min."""
    n_batches = len(bank) // batch_size
    x_gt = query.unsqueeze(0)
    dist_parts, index_parts = [], []
    for b in range(n_batches):
        lo, hi = b * batch_size, (b + 1) * batch_size
        dist_parts.append(loss(bank[lo:hi], x_gt))
        index_parts.append(torch.tensor(range(lo, hi)))
    all_dist = torch.cat(dist_parts)          # ValueError when n_batches == 0, as in the reference
    all_index = torch.cat(index_parts)
    best, where = torch.min(all_dist, dim=0)  # first occurrence on ties
    return best.item(), all_index[where].item()


def dequantize(u8_nchw):
    """fbb.py:134-135 on utils.read_image output: float64 2*(u/255)-1 -> .float()"""
    return (2.0 * (torch.from_numpy(u8_nchw).double() / 255.0) - 1.0).float()


_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
_VGG_KEYS = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
_TAP_AFTER = {1, 3, 6, 9, 12}


def make_l2_lpips_loss(vgg_sd, lin):
    """loss(x_hat [B,3,H,W], x_gt [1,3,H,W]) -> [B], the op sequence of the reference in fp32: VGG16 on the query AND on the batch for every
    call (networks_basic.py:147-152), normalize_tensor with eps outside the sqrt (util/util.py:70-73), squared difference broadcast [1] vs [B],
    1x1 lin conv (requires_grad as in the reference, so the autograd graph is built), mean over W then H, sum over the 5 taps, 0.2 * lpips + l2."""
    import torch.nn.functional as F
    w = [torch.from_numpy(vgg_sd["%d.weight" % k]) for k in _VGG_KEYS]
    b = [torch.from_numpy(vgg_sd["%d.bias" % k]) for k in _VGG_KEYS]
    lins = [torch.from_numpy(lin[i].reshape(1, -1, 1, 1).copy()).requires_grad_(True) for i in range(5)]
    shift = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
    scale = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)

    def vgg(x):
        taps, ci = [], 0
        h = x
        for v in _VGG_CFG:
            if v == "M":
                h = F.max_pool2d(h, 2, 2)
                continue
            h = F.relu(F.conv2d(h, w[ci], b[ci], padding=1))
            if ci in _TAP_AFTER:
                taps.append(h)
            ci += 1
        return taps

    def normalize_tensor(f, eps=1e-10):
        nf = torch.sqrt(torch.sum(f ** 2, dim=1)).view(f.size()[0], 1, f.size()[2], f.size()[3])
        return f / (nf.expand_as(f) + eps)

    def lpips(in0, in1):                       # PNetLin.forward(in0 = target / query, in1 = pred / bank batch)
        in0_sc = (in0 - shift.expand_as(in0)) / scale.expand_as(in0)
        in1_sc = (in1 - shift.expand_as(in0)) / scale.expand_as(in0)
        outs0, outs1 = vgg(in0_sc), vgg(in1_sc)
        val = None
        for kk in range(5):
            d = (normalize_tensor(outs0[kk]) - normalize_tensor(outs1[kk])) ** 2
            t = torch.mean(torch.mean(F.conv2d(d, lins[kk]), dim=3), dim=2)
            val = t if val is None else val + t
        return val.view(val.size()[0], val.size()[1], 1, 1)

    def loss(x_hat, x_gt):                     # utils.py:171-177; PerceptualLoss.forward(pred=x_hat, target=x_gt) -> forward_pair(target, pred)
        return 0.2 * lpips(x_gt, x_hat).view(-1) + torch.mean((x_gt - x_hat) ** 2, dim=[1, 2, 3])
    return loss


def make_features_once(vgg_sd, lin):
    """The "features-once" CPU variant BASELINE.md section 3 asks for beside the reference-literal loop: the SAME distance
    0.2 * LPIPS + L2 (utils.py:176) evaluated the way the device path does -- VGG16 once per image, every tap normalised
    (util/util.py:70-73), weighted by sqrt(0.2 * w_lc / (H_l W_l)) and concatenated with the image / sqrt(D) into one row V, so that
    ||V_q - V_n||^2 is the distance and the [Q, N] search is one GEMM.  Needs w >= 0 (true for the vendored vgg.pth).  Returns
    rows(images [n,3,H,W] fp32) -> [n, K] fp32 and search(q_rows, bank_rows) -> (dist [Q], idx [Q]).  Separates the algorithmic speed-up
    (features once per image instead of once per pair: ~1 800 x at configs[2]) from the hardware one."""
    import torch.nn.functional as F
    w = [torch.from_numpy(vgg_sd["%d.weight" % k]) for k in _VGG_KEYS]
    b = [torch.from_numpy(vgg_sd["%d.bias" % k]) for k in _VGG_KEYS]
    lw = [torch.from_numpy(lin[i].reshape(-1).copy()).clamp_min(0.0) for i in range(5)]
    shift = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
    scale = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)

    def rows(x):
        with torch.no_grad():
            n = x.shape[0]
            parts = [x.reshape(n, -1) / float(x[0].numel()) ** 0.5]
            h, ci, tap = (x - shift) / scale, 0, 0
            for v in _VGG_CFG:
                if v == "M":
                    h = F.max_pool2d(h, 2, 2)
                    continue
                h = F.relu(F.conv2d(h, w[ci], b[ci], padding=1))
                if ci in _TAP_AFTER:
                    nf = torch.sqrt(torch.sum(h ** 2, dim=1, keepdim=True))
                    coef = torch.sqrt(0.2 * lw[tap] / float(h.shape[2] * h.shape[3])).view(1, -1, 1, 1)
                    parts.append((h / (nf + 1e-10) * coef).reshape(n, -1))
                    tap += 1
                ci += 1
            return torch.cat(parts, dim=1)

    def search(q_rows, bank_rows):
        with torch.no_grad():
            d = (q_rows ** 2).sum(1, keepdim=True) + (bank_rows ** 2).sum(1)[None, :] - 2.0 * (q_rows @ bank_rows.t())
            best, where = torch.min(d.clamp_min(0.0), dim=1)
            return best, where

    return rows, search

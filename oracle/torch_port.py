"""PyTorch-CPU restatement of the reference's per-query search, used ONLY as the timed CPU baseline
(`cpu_baseline.kind = "port"` in bench.py) and as a cross-check of the exact oracle.

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product.

It issues the same torch operations in the same order as the reference, so its run time on the GPU
box's host cores is what `attack_models/fbb.py` would take there:
    custom_knn                    attack_models/fbb.py:73-88
    Loss('l2').forward            attack_models/utils.py:163,169,171-177
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

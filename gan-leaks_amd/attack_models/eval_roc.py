"""Counterpart of the reference's attack_models/eval_roc.py: ROC / AUROC / AP / precision of the
attack from the saved losses.  Pure host code (O(Q log Q)); no GPU involved.

plot_roc restates the scikit-learn calls of eval_roc.py:14-25 in numpy so the path has no sklearn
dependency; it is pinned to vectors produced by the reference's own plot_roc (tests/golden/roc_*.npz).
"""
from __future__ import annotations

import argparse
import os
import warnings

import numpy as np


def _binary_clf_curve(labels, scores):
    order = np.argsort(scores, kind="mergesort")[::-1]
    s = scores[order]
    y = labels[order]
    thr_idx = np.r_[np.where(np.diff(s))[0], y.size - 1]
    tps = np.cumsum(y)[thr_idx]
    fps = 1 + thr_idx - tps
    return fps, tps, s[thr_idx]


def plot_roc(pos_results, neg_results):
    """eval_roc.py:14-25.  labels: 0 = negative query, 1 = positive query; the caller passes
    score = -distance (eval_roc.py:78).  Returns fpr, tpr, threshold, auc, ap, precision where
    precision is taken at the fixed threshold score > -0.14 (eval_roc.py:21-23)."""
    pos = np.asarray(pos_results, np.float64).reshape(-1)
    neg = np.asarray(neg_results, np.float64).reshape(-1)
    labels = np.concatenate((np.zeros(len(neg)), np.ones(len(pos))))
    results = np.concatenate((neg, pos))
    fps, tps, thr = _binary_clf_curve(labels, results)

    # average_precision_score: sum over distinct thresholds of (R_n - R_{n-1}) * P_n
    denom = tps + fps
    prec_c = np.divide(tps, denom, out=np.zeros_like(tps, dtype=np.float64), where=denom != 0)
    rec_c = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps, dtype=np.float64)
    ap = float(-np.sum(np.diff(np.r_[rec_c[::-1], 0.0]) * np.r_[prec_c[::-1], 1.0][:-1]))

    # roc_auc_score: trapezoid under the full curve (ties get half credit)
    tps_f = np.r_[0, tps].astype(np.float64)
    fps_f = np.r_[0, fps].astype(np.float64)
    if tps_f[-1] <= 0 or fps_f[-1] <= 0:
        raise ValueError("Only one class present; ROC AUC is not defined")
    x_, y_ = fps_f / fps_f[-1], tps_f / tps_f[-1]
    auc = float(np.sum(np.diff(x_) * (y_[1:] + y_[:-1])) * 0.5)      # trapezoid rule, written out (np.trapezoid needs numpy >= 2, the reference pins 1.24)

    # roc_curve(drop_intermediate=True): drop collinear points, then prepend (0, 0, inf)
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    threshold = np.r_[np.inf, thr]
    fpr = fps / fps[-1]
    tpr = tps / tps[-1]

    result_array = results > -0.14
    pp = float(np.sum(result_array))
    precision = float(np.sum(result_array & (labels == 1))) / pp if pp > 0 else 0.0
    return fpr, tpr, threshold, auc, ap, precision


def _pyplot():
    import matplotlib
    matplotlib.use('Agg')
    from matplotlib import pyplot
    return pyplot


def plot_hist(pos_dist, neg_dist, save_file):
    """eval_roc.py:28-37: normalised histograms (100 bins) of the positive and negative distances"""
    plt = _pyplot()
    fig, ax = plt.subplots()
    for values, label in ((np.asarray(pos_dist).reshape(-1), 'positive'), (np.asarray(neg_dist).reshape(-1), 'negative')):
        ax.hist(values, bins=100, alpha=0.5, weights=np.full(values.shape, 1.0 / values.size), label=label)
    ax.legend(loc='upper right')
    ax.set_xlabel('distance')
    ax.set_ylabel('normalized frequency')
    fig.tight_layout()
    fig.savefig(save_file)
    plt.close(fig)


def parse_arguments(argv=None):
    """eval_roc.py:43-54 (same flags)"""
    parser = argparse.ArgumentParser()
    parser.add_argument('--result_load_dir', '-ldir', type=str, default=None, help='directory of the attack result')
    parser.add_argument('--attack_type', type=str, choices=['fbb', 'pbb', 'wb'], help='type of the attack')
    parser.add_argument('--reference_load_dir', '-rdir', default=None, help='directory for the reference model result (optional)')
    parser.add_argument('--save_dir', '-sdir', type=bool, default=True, help='directory for saving the evaluation results (optional)')
    parser.add_argument("--wandb", default=None, help="accepted for compatibility; logging to WandB is not performed")
    parser.add_argument('--local_config', type=str, default=None)
    return parser.parse_args(argv)


def update_args(args, config_dict):
    for name in config_dict:
        setattr(args, name, config_dict[name])


def main(args):
    """eval_roc.py:61-121.  Returns (auc, ap, precision) in addition to printing them; writes roc.png
    next to the losses when matplotlib is available and save_dir is truthy."""
    attack_type = args.attack_type
    result_load_dir = args.result_load_dir
    pos_loss = np.load(os.path.join(result_load_dir, 'pos_loss.npy'))
    neg_loss = np.load(os.path.join(result_load_dir, 'neg_loss.npy'))
    if attack_type != 'fbb':
        pos_loss, neg_loss = pos_loss.flatten(), neg_loss.flatten()
    fpr, tpr, threshold, auc, ap, precision = plot_roc(-pos_loss, -neg_loss)
    print("The AUC ROC value of %s attack is: %.3f " % (attack_type, auc))
    print("The precision of %s attack is: %.3f " % (attack_type, precision))
    curves = [(fpr, tpr, '%s attack, auc=%.3f, ap=%.3f' % (attack_type, auc, ap))]

    if args.reference_load_dir is not None:
        # calibrated attack, eval_roc.py:86-103 (the reference unpacks plot_roc's 6 values into 5
        # names there and would raise; here the sixth is taken)
        pos_ref = np.load(os.path.join(args.reference_load_dir, 'pos_loss.npy'))
        neg_ref = np.load(os.path.join(args.reference_load_dir, 'neg_loss.npy'))
        n_pos = min(len(pos_loss), len(pos_ref))
        n_neg = min(len(neg_loss), len(neg_ref))
        pos_cal = pos_loss[:n_pos].reshape(n_pos, -1)[:, 0] - pos_ref[:n_pos].reshape(n_pos, -1)[:, 0]
        neg_cal = neg_loss[:n_neg].reshape(n_neg, -1)[:, 0] - neg_ref[:n_neg].reshape(n_neg, -1)[:, 0]
        cf, ct, _, cauc, cap, _ = plot_roc(-pos_cal, -neg_cal)
        print("The AUC ROC value of calibrated %s attack is: %.3f " % (attack_type, cauc))
        curves.append((cf, ct, 'calibrated %s attack, auc=%.3f, ap=%.3f' % (attack_type, cauc, cap)))

    if args.save_dir:
        try:
            plt = _pyplot()
        except ImportError:
            warnings.warn("matplotlib not available: roc.png not written")
        else:
            fig, ax = plt.subplots()
            for f, t, label in curves:
                ax.plot(f, t, label=label)
            ax.legend(loc='lower right')
            ax.set(xlabel='false positive', ylabel='true positive', title='ROC curve')
            fig.savefig(os.path.join(result_load_dir, 'roc.png'))
            plt.close(fig)
    return auc, ap, precision


if __name__ == '__main__':
    import yaml
    cli = parse_arguments()
    if cli.local_config is None:
        warnings.warn("No config file was provided. Using default parameters.")
    else:
        with open(str(cli.local_config)) as handle:
            update_args(cli, yaml.safe_load(handle))
    main(cli)

"""Model-selection metrics — mirror of MERBench/toolkit/utils/metric.py:9-100 (label path: bit-exact)."""
import numpy as np


def overall_metric(emo_fscore, val_mse):
    return emo_fscore - val_mse * 0.25


def gain_metric_from_results(eval_results, metric_name='emoval'):
    if metric_name == 'emoval':
        return overall_metric(eval_results['emofscore'], eval_results['valmse'])
    if metric_name == 'emo':
        return eval_results['emofscore']
    if metric_name == 'val':
        return -eval_results['valmse']
    if metric_name == 'loss':
        return -eval_results['loss']
    raise KeyError(metric_name)  # the reference falls through to an UnboundLocalError here


def gain_cv_results(folder_save):
    keys = list(folder_save[0].keys())
    parts = []
    if 'eval_emofscore' in keys:
        parts.append(f"f1:{np.mean([e['eval_emofscore'] for e in folder_save]):.4f}")
    if 'eval_emoacc' in keys:
        parts.append(f"acc:{np.mean([e['eval_emoacc'] for e in folder_save]):.4f}")
    if 'eval_valmse' in keys:
        parts.append(f"val:{np.mean([e['eval_valmse'] for e in folder_save]):.4f}")
    return "_".join(parts)


def average_folder_for_emos(folder_save, testname):
    """Fold-averaged class probabilities per test sample (test loaders are unshuffled, so rows line up)."""
    try:
        labels = folder_save[0][f'{testname}_emolabels']
    except Exception:
        return [], []
    whole_probs = np.array([fold[f'{testname}_emoprobs'] for fold in folder_save])
    avg_preds = [np.mean(whole_probs[:, ii, :], axis=0) for ii in range(len(labels))]
    return labels, avg_preds


def average_folder_for_vals(folder_save, testname):
    try:
        labels = folder_save[0][f'{testname}_vallabels']
    except Exception:
        return [], []
    whole_preds = np.array([fold[f'{testname}_valpreds'] for fold in folder_save])
    return labels, np.mean(whole_preds, axis=0)

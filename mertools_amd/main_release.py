"""Fusion training driver — mirror of MERBench/main-release.py:17-272 (same CLI flags, same 5-fold x epochs x
{train, eval, tests} schedule, same result-file naming) for the models on the hot path.

    python -m mertools_amd.main_release --model attention --feat_type utt --dataset MER2023 \
        --audio_feature A-UTT --text_feature T-UTT --video_feature V-UTT --gpu 0 [--seed 0]

The model forward/backward, both losses and (with --hip_adam) the optimiser step run on the HIP kernels;
sklearn metrics and result bookkeeping stay on the host exactly as in the reference.  `--seed` (declared
but never read by the reference) is actually applied here so that runs are repeatable.
"""
import argparse
import os
import random
import time

import numpy as np
import torch
import torch.optim as optim
import yaml

from . import config
from .fusion_ops import HipAdam
from .toolkit.dataloader import get_dataloaders
from .toolkit.models import get_models
from .toolkit.utils.functions import func_random_select, merge_args_config
from .toolkit.utils.loss import CELoss, MSELoss
from .toolkit.utils.metric import (average_folder_for_emos, average_folder_for_vals, gain_cv_results,
                                   gain_metric_from_results)


def func_update_storage(inputs, prefix, outputs):
    for key in inputs:
        outputs[f'{prefix}_{key}'] = inputs[key]


def train_or_eval_model(args, model, reg_loss, cls_loss, dataloader, epoch, optimizer=None, train=False,
                        dataloader_class=None):
    """One pass over `dataloader`; returns dict(names, loss, emoprobs, emolabels, emoacc, emofscore, valpreds,
    vallabels, valmse) like the reference (main-release.py:17-87)."""
    vidnames, losses = [], []
    val_preds, val_labels, emo_probs, emo_labels = [], [], [], []
    assert not train or optimizer is not None
    config.train = train
    model.train() if train else model.eval()
    fused_clip = isinstance(optimizer, HipAdam)
    for it, data in enumerate(dataloader):
        if train:
            optimizer.zero_grad()
        batch, emos, vals, bnames = data
        vidnames += bnames
        for key in batch:
            batch[key] = batch[key].cuda()
        emos, vals = emos.cuda(), vals.cuda()
        features, emos_out, vals_out, interloss = model(batch)
        loss = interloss
        if args.output_dim1 != 0:
            loss = loss + cls_loss(emos_out, emos)
            emo_probs.append(emos_out.data.cpu().numpy())
            emo_labels.append(emos.data.cpu().numpy())
        if args.output_dim2 != 0:
            loss = loss + reg_loss(vals_out, vals)
            val_preds.append(vals_out.data.cpu().numpy())
            val_labels.append(vals.data.cpu().numpy())
        losses.append(loss.data.cpu().numpy())
        if train:
            loss.backward()
            if model.model.grad_clip != -1 and not fused_clip:
                torch.nn.utils.clip_grad_value_([p for p in model.parameters() if p.requires_grad], model.model.grad_clip)
            optimizer.step()
        if (it + 1) % args.print_iters == 0:
            print(f'process on {it + 1}|{len(dataloader)}, meanloss: {np.mean(losses)}')
    if emo_probs != []:
        emo_probs, emo_labels = np.concatenate(emo_probs), np.concatenate(emo_labels)
    if val_preds != []:
        val_preds, val_labels = np.concatenate(val_preds), np.concatenate(val_labels)
    results, _ = dataloader_class.calculate_results(emo_probs, emo_labels, val_preds, val_labels)
    return dict(names=vidnames, loss=np.mean(losses), **results)


def train_or_eval_graph(args, trainer, dataloader, epoch, train=False, dataloader_class=None):
    """train_or_eval_model without the reference's per-step host round trips (main-release.py:53-59 pulls probabilities, labels
    and the loss to the host 3-5 times per minibatch, ~79k steps per run): every minibatch is ONE replay of a captured HIP
    graph (FusionGraphTrainer: forward, both losses, backward, clipping, Adam — or forward + losses for evaluation), its outputs
    are appended to device-side epoch buffers, and the epoch's probabilities / labels / losses come back in one copy each.
    Same return dict."""
    vidnames = []
    config.train = train
    n = len(dataloader.sampler) if getattr(dataloader, 'sampler', None) is not None else len(dataloader.dataset)
    dev = trainer.flat.device
    probs = torch.empty((n, args.output_dim1), dtype=torch.float32, device=dev)
    vpred = torch.empty((n, args.output_dim2), dtype=torch.float32, device=dev)
    elab = torch.empty((n,), dtype=torch.int64, device=dev)
    vlab = torch.empty((n,), dtype=torch.float32, device=dev)
    losses = torch.empty((len(dataloader),), dtype=torch.float32, device=dev)
    r = 0
    for it, (batch, emos, vals, bnames) in enumerate(dataloader):
        vidnames += bnames
        loss, emos_out, vals_out = (trainer.train_step if train else trainer.eval_step)(batch, emos, vals)
        b = emos.shape[0]
        # the graph's static outputs are overwritten by the next replay: stream-ordered device copies into the epoch buffers
        probs[r:r + b].copy_(emos_out.detach(), non_blocking=True)
        vpred[r:r + b].copy_(vals_out.detach().view(b, -1), non_blocking=True)
        elab[r:r + b].copy_(emos, non_blocking=True)
        vlab[r:r + b].copy_(vals, non_blocking=True)
        losses[it].copy_(loss.detach(), non_blocking=True)
        r += b
        if (it + 1) % args.print_iters == 0:
            print(f'process on {it + 1}|{len(dataloader)}, meanloss: {losses[:it + 1].mean().item()}')
    assert r == n, (r, n)
    emo_probs, emo_labels = probs.cpu().numpy(), elab.cpu().numpy()
    val_preds, val_labels = vpred.cpu().numpy(), vlab.cpu().numpy()
    results, _ = dataloader_class.calculate_results(emo_probs, emo_labels, val_preds, val_labels)
    return dict(names=vidnames, loss=np.mean(losses.cpu().numpy()), **results)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, default=None)
    p.add_argument('--train_dataset', type=str, default=None)
    p.add_argument('--test_dataset', type=str, default=None)
    p.add_argument('--save_root', type=str, default='./saved')
    p.add_argument('--debug', action='store_true', default=False)
    p.add_argument('--savemodel', action='store_true', default=False)
    p.add_argument('--save_iters', type=int, default=1e8)
    p.add_argument('--audio_feature', type=str, default=None)
    p.add_argument('--text_feature', type=str, default=None)
    p.add_argument('--video_feature', type=str, default=None)
    p.add_argument('--feat_type', type=str, default=None)
    p.add_argument('--feat_scale', type=int, default=None)
    p.add_argument('--e2e_name', type=str, default=None)
    p.add_argument('--e2e_dim', type=int, default=None)
    p.add_argument('--n_classes', type=int, default=None)
    p.add_argument('--hyper_path', type=str, default=None)
    p.add_argument('--model', type=str, default=None)
    p.add_argument('--lr', type=float, default=None)
    p.add_argument('--lr_adjust', type=str, default='case1')
    p.add_argument('--l2', type=float, default=0.00001)
    p.add_argument('--batch_size', type=int, default=32)
    p.add_argument('--num_workers', type=int, default=0)
    p.add_argument('--epochs', type=int, default=100)
    p.add_argument('--print_iters', type=int, default=1e8)
    p.add_argument('--gpu', default=0, type=int)
    # MER2024/main-release.py:96-99: noise-robustness and multi-feature fusion studies
    p.add_argument('--train_snr', type=lambda x: None if x == 'None' else str(x), default=None, help='train snr (selects the feature directory)')
    p.add_argument('--test_snr', type=lambda x: None if x == 'None' else str(x), default=None, help='test snr (selects the feature directory)')
    p.add_argument('--fusion_topn', type=int, default=None, help='attention_topn: feature sets per modality slot')
    p.add_argument('--fusion_modality', type=str, default='AVT', help='attention_topn: AVT | AV | AT | VT')
    # additions (not in the reference)
    p.add_argument('--seed', type=int, default=None, help='seed python/numpy/torch RNGs (the reference never seeds)')
    p.add_argument('--hip_adam', action='store_true', default=False, help='optimizer step (and grad clip) in one HIP kernel per tensor')
    p.add_argument('--data_root', type=str, default=None, help='re-point config.PATH_TO_* at this directory')
    p.add_argument('--eager', action='store_true', default=False,
                   help="the reference's loop verbatim (one kernel launch per op, 3-5 host syncs per minibatch) instead of one captured "
                        "HIP graph per minibatch with per-epoch result copies; same results, for bit-comparison")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.cuda.set_device(args.gpu)
    if args.data_root:
        config.set_data_root(args.data_root)
    if args.seed is not None:
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    print('====== Params Pre-analysis =======')
    if args.feat_type == 'utt':
        args.feat_scale = 1
    elif args.feat_type in ('frm_align', 'frm_unalign'):
        for f in (args.audio_feature, args.text_feature, args.video_feature):
            assert f.endswith('FRA')
        args.feat_scale = 6 if args.feat_type == 'frm_align' else 12
    if args.train_dataset is not None:
        args.save_root = f'{args.save_root}-cross'
    whole_features = [f for f in [args.audio_feature, args.text_feature, args.video_feature] if f is not None]
    args.save_root += {0: '-others', 1: '-unimodal', 2: '-bimodal', 3: '-trimodal'}[len(set(whole_features))]
    if args.test_snr is not None:       # MER2024/main-release.py:151-154
        args.save_root = f'{args.save_root}-noise'
    if args.fusion_topn is not None:
        args.save_root = f'{args.save_root}-multitop'
    tune = args.hyper_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'toolkit', 'model-tune.yaml')
    with open(tune) as fh:
        model_config = dict(yaml.safe_load(fh)[args.model])
    if args.hyper_path is None:
        model_config = func_random_select(model_config)
    config.dataset = args.dataset
    args = merge_args_config(args, model_config)
    args.lr = float(args.lr)   # a user-supplied yaml may spell it 1e-3, which PyYAML hands over as a string
    print('args: ', args)
    save_resroot, save_modelroot = os.path.join(args.save_root, 'result'), os.path.join(args.save_root, 'model')
    os.makedirs(save_resroot, exist_ok=True)
    os.makedirs(save_modelroot, exist_ok=True)
    feature_name = "+".join(sorted(set(whole_features)))
    prefix_name = f'features:{feature_name}_dataset:{args.dataset}_model:{args.model}+{args.feat_type}+{args.e2e_name}'
    if args.test_snr is not None:       # MER2024/main-release.py:188-191
        prefix_name += f'_trainsnr:{args.train_snr}_testsnr:{args.test_snr}'
    if args.fusion_topn is not None:
        prefix_name += f'_fusiontopn:{args.fusion_topn}_modality:{args.fusion_modality}'

    print('====== Reading Data =======')
    dataloader_class = get_dataloaders(args)
    train_loaders, eval_loaders, test_loaders = dataloader_class.get_loaders()
    assert len(train_loaders) == len(eval_loaders)
    print(f'train&val folder:{len(train_loaders)}; test sets:{len(test_loaders)}')
    args.audio_dim, args.text_dim, args.video_dim = train_loaders[0].dataset.get_featdim()

    print('====== Training and Evaluation =======')
    folder_save, folder_duration = [], []
    name_time = time.time()
    for ii in range(len(train_loaders)):
        print(f'>>>>> Cross-validation: training on the {ii + 1} folder >>>>>')
        start_time = name_time = time.time()
        model = get_models(args).cuda()
        reg_loss, cls_loss = MSELoss().cuda(), CELoss().cuda()
        assert args.lr_adjust == 'case1', 'lr_adjust=case2 only applies to e2e models (out of scope)'
        # graph path: both heads present (the captured step always adds CE + MSE, as the reference does for these models)
        use_graph = not args.eager and args.output_dim1 != 0 and args.output_dim2 != 0
        trainer = optimizer = None
        if use_graph:
            from .fusion_trainer import FusionGraphTrainer
            trainer = FusionGraphTrainer(model, lr=args.lr, weight_decay=args.l2, grad_clip=model.model.grad_clip)
        elif args.hip_adam:
            optimizer = HipAdam(model.parameters(), lr=args.lr, weight_decay=args.l2, clip_value=model.model.grad_clip)
        else:
            optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.l2)

        def run(loader, train):
            if use_graph:
                return train_or_eval_graph(args, trainer, loader, epoch, train, dataloader_class=dataloader_class)
            return train_or_eval_model(args, model, reg_loss, cls_loss, loader, epoch, optimizer if train else None, train,
                                       dataloader_class=dataloader_class)

        whole_store, whole_metrics = [], []
        for epoch in range(args.epochs):
            epoch_store = {}
            train_results = run(train_loaders[ii], True)
            eval_results = run(eval_loaders[ii], False)
            func_update_storage(eval_results, 'eval', epoch_store)
            train_metric = gain_metric_from_results(train_results, args.metric_name)
            eval_metric = gain_metric_from_results(eval_results, args.metric_name)
            whole_metrics.append(eval_metric)
            print('epoch:%d; metric:%s; train results:%.4f; eval results:%.4f' % (epoch + 1, args.metric_name, train_metric, eval_metric))
            for jj, test_loader in enumerate(test_loaders):
                test_results = run(test_loader, False)
                func_update_storage(test_results, f'test{jj + 1}', epoch_store)
            whole_store.append(epoch_store)
        best_index = np.argmax(np.array(whole_metrics))
        folder_save.append(whole_store[best_index])
        duration = time.time() - start_time
        folder_duration.append(duration)
        print(f'>>>>> Finish: training on the {ii + 1}-th folder, best_index: {best_index}, duration: {duration} >>>>>')
        del model, optimizer, trainer
        torch.cuda.empty_cache()

    print('====== Prediction and Saving =======')
    args.duration = np.sum(folder_duration)
    cv_result = gain_cv_results(folder_save)
    save_path = f'{save_resroot}/cv_{prefix_name}_{cv_result}_{name_time}.npz'
    print(f'save results in {save_path}')
    np.savez_compressed(save_path, args=np.array(args, dtype=object))
    for jj in range(len(test_loaders)):
        emo_labels, emo_probs = average_folder_for_emos(folder_save, f'test{jj + 1}')
        val_labels, val_preds = average_folder_for_vals(folder_save, f'test{jj + 1}')
        _, test_result = dataloader_class.calculate_results(emo_probs, emo_labels, val_preds, val_labels)
        save_path = f'{save_resroot}/test{jj + 1}_{prefix_name}_{test_result}_{name_time}.npz'
        print(f'save results in {save_path}')
        if args.dataset == 'MER2024':   # MER2024/main-release.py:288-290 also stores the fold-averaged test probabilities
            np.savez_compressed(save_path, emo_probs=emo_probs, args=np.array(args, dtype=object))
        else:
            np.savez_compressed(save_path, args=np.array(args, dtype=object))
    return folder_save


if __name__ == '__main__':
    main()

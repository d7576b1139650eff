#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE from /root/reference (read-only).

Only runs in the build container (the GPU box has no /root/reference); the outputs are committed so the
parity tests can use them anywhere.  Nothing from the reference is copied: modules are imported (with
stub modules for dependencies that are absent here: cv2, torchaudio, omegaconf, ...) or individual
functions / classes are exec'd straight from the reference source via `ast`.

    python tests/golden/gen_golden.py
"""
import argparse
import ast
import importlib.machinery
import os
import random
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def exec_defs(path, names, ns):
    """exec only the named top-level functions/classes of a reference source file into ns."""
    src = open(path, encoding="utf-8").read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns


def ref_toolkit():
    import transformers  # noqa: F401  (before any stub lands in sys.modules)
    from transformers import BertTokenizer  # noqa: F401
    for m in ["cv2", "torchaudio", "omegaconf", "thop", "soundfile", "openai", "pytorchvideo", "pytorchvideo.data",
              "pytorchvideo.data.encoded_video", "decord", "timm", "librosa", "tqdm"]:
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                stub = mock.MagicMock()
                stub.__spec__ = importlib.machinery.ModuleSpec(m, None)
                sys.modules[m] = stub
    if REF + "/MERBench" not in sys.path:
        sys.path.insert(0, REF + "/MERBench")
    torch.Tensor.cuda = lambda self, *a, **k: self  # attention.py:55 hard-codes .cuda()
    torch.nn.Module.cuda = lambda self, *a, **k: self


def fusion_goldens():
    ref_toolkit()
    from toolkit.models.attention import Attention
    from toolkit.utils.loss import CELoss, MSELoss
    args = argparse.Namespace(text_dim=80, audio_dim=96, video_dim=64, output_dim1=6, output_dim2=1, dropout=0.0,
                              hidden_dim=64, grad_clip=-1.0, feat_type="utt")
    torch.manual_seed(1237)
    model = Attention(args)
    init = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    B, steps = 32, 5
    xs = dict(audios=torch.randn(steps, B, 96), texts=torch.randn(steps, B, 80), videos=torch.randn(steps, B, 64))
    emos = torch.randint(0, 6, (steps, B))
    vals = torch.randn(steps, B) * 2
    model.eval()
    with torch.no_grad():
        f, e, v, il = model({k: x[0] for k, x in xs.items()})
    out = dict(features=f.numpy(), emos_out=e.numpy(), vals_out=v.numpy(), interloss=il.numpy())
    # 5 Adam steps exactly as main-release.py:50-66,205 (dropout 0 so no RNG enters)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    cls_loss, reg_loss = CELoss(), MSELoss()
    losses, grads0 = [], None
    for s in range(steps):
        opt.zero_grad()
        f, e, v, il = model({k: x[s] for k, x in xs.items()})
        loss = il + cls_loss(e, emos[s]) + reg_loss(v, vals[s])
        loss.backward()
        if s == 0:
            grads0 = {k: p.grad.clone().numpy() for k, p in model.named_parameters()}
        opt.step()
        losses.append(loss.item())
    final = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "fusion_attention.npz"), losses=np.array(losses, dtype=np.float64),
                        emos=emos.numpy(), vals=vals.numpy(), **{f"x_{k}": v.numpy() for k, v in xs.items()},
                        **{f"out_{k}": v for k, v in out.items()}, **{f"init_{k}": v for k, v in init.items()},
                        **{f"final_{k}": v for k, v in final.items()}, **{f"grad0_{k}": v for k, v in grads0.items()})
    # loss functions alone
    pred, tgt = torch.randn(7, 6), torch.randint(0, 6, (7,))
    vp, vt = torch.randn(7, 1), torch.randn(7)
    np.savez(os.path.join(OUT, "losses.npz"), pred=pred.numpy(), tgt=tgt.numpy(), ce=CELoss()(pred, tgt).item(),
             vp=vp.numpy(), vt=vt.numpy(), mse=MSELoss()(vp, vt).item())

    # LF_DNN (MER2024) and MER2023's MLP / Attention
    ns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional}
    from toolkit.models.modules.encoder import MLPEncoder, LSTMEncoder
    ns.update(MLPEncoder=MLPEncoder, LSTMEncoder=LSTMEncoder)
    exec_defs(REF + "/MER2024/toolkit/models/lf_dnn.py", {"LF_DNN"}, ns)
    torch.manual_seed(77)
    lf = ns["LF_DNN"](args).eval()
    batch = {k: x[0] for k, x in xs.items()}
    with torch.no_grad():
        f, e, v, _ = lf(batch)
    np.savez_compressed(os.path.join(OUT, "fusion_lf_dnn.npz"), features=f.numpy(), emos_out=e.numpy(), vals_out=v.numpy(),
                        **{f"init_{k}": v.numpy() for k, v in lf.state_dict().items()})
    ns2 = {"torch": torch, "nn": torch.nn}
    exec_defs(REF + "/MER2023/main-release.py", {"MLP", "Attention"}, ns2)
    torch.manual_seed(78)
    mlp = ns2["MLP"](96 + 80 + 64, 6, 1, layers="64,32", dropout=0.0).eval()
    att = ns2["Attention"](96, 80, 64, 6, 1, layers="64,32", dropout=0.0).eval()
    with torch.no_grad():
        mf, me, mv = mlp(torch.cat([batch["audios"], batch["texts"], batch["videos"]], dim=1))
        af, ae, av = att(batch["audios"], batch["texts"], batch["videos"])
    np.savez_compressed(os.path.join(OUT, "fusion_mer2023.npz"), mlp_features=mf.numpy(), mlp_emos=me.numpy(), mlp_vals=mv.numpy(),
                        att_features=af.numpy(), att_emos=ae.numpy(), att_vals=av.numpy(),
                        **{f"mlp_{k}": v.numpy() for k, v in mlp.state_dict().items()},
                        **{f"att_{k}": v.numpy() for k, v in att.state_dict().items()})


def fusion_frame_goldens():
    """The reference's own frame-level model — toolkit.models.attention.Attention with feat_type='frm_align', i.e. three
    LSTMEncoder branches (modules/encoder.py:45-72) — forward, one full backward and 3 Adam steps on seeded [B, T, D] inputs."""
    ref_toolkit()
    from toolkit.models.attention import Attention
    from toolkit.utils.loss import CELoss, MSELoss
    args = argparse.Namespace(text_dim=40, audio_dim=48, video_dim=32, output_dim1=6, output_dim2=1, dropout=0.0,
                              hidden_dim=64, grad_clip=-1.0, feat_type="frm_align")
    torch.manual_seed(4321)
    model = Attention(args)
    init = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    B, T, steps = 6, 9, 3
    xs = dict(audios=torch.randn(steps, B, T, 48), texts=torch.randn(steps, B, T, 40), videos=torch.randn(steps, B, T, 32))
    emos, vals = torch.randint(0, 6, (steps, B)), torch.randn(steps, B) * 2
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    cls_loss, reg_loss = CELoss(), MSELoss()
    losses, grads0, out0 = [], None, None
    for s in range(steps):
        opt.zero_grad()
        f, e, v, il = model({k: x[s] for k, x in xs.items()})
        loss = il + cls_loss(e, emos[s]) + reg_loss(v, vals[s])
        loss.backward()
        if s == 0:
            grads0 = {k: p.grad.clone().numpy() for k, p in model.named_parameters()}
            out0 = dict(features=f.detach().numpy().copy(), emos_out=e.detach().numpy().copy(), vals_out=v.detach().numpy().copy())
        opt.step()
        losses.append(loss.item())
    final = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "fusion_attention_frm_align.npz"), losses=np.array(losses, dtype=np.float64),
                        emos=emos.numpy(), vals=vals.numpy(), **{f"x_{k}": v.numpy() for k, v in xs.items()},
                        **{f"out_{k}": v for k, v in out0.items()}, **{f"init_{k}": v for k, v in init.items()},
                        **{f"final_{k}": v for k, v in final.items()}, **{f"grad0_{k}": v for k, v in grads0.items()})


def index_goldens():
    ref_toolkit()
    import math
    # --- visual / audio extractor helpers (scripts cannot be imported: argparse + cv2 at import time) ---
    nsv = {"np": np, "math": math}
    exec_defs(REF + "/MERBench/feature_extraction/visual/extract_vision_huggingface.py", {"resample_frames_uniform", "split_into_batch"}, nsv)
    cases = [(1, 16), (5, 16), (16, 16), (17, 16), (37, 16), (100, 16), (250, 64), (63, 64), (3, 8), (1000, 16)]
    res = {}
    for vlen, n in cases:
        res[f"resample_{vlen}_{n}"] = nsv["resample_frames_uniform"](np.arange(vlen), nframe=n)
    res["vsplit_70_32"] = np.array([len(b) for b in nsv["split_into_batch"](np.arange(70), bsize=32)])
    nsa = {"torch": torch, "math": math}
    exec_defs(REF + "/MERBench/feature_extraction/audio/extract_audio_huggingface.py", {"split_into_batch"}, nsa)
    x = torch.arange(1, 26, dtype=torch.float32)[None]
    res["asplit_25_10"] = nsa["split_into_batch"](x, maxlen=10).numpy()
    res["asplit_25_30"] = nsa["split_into_batch"](x, maxlen=30).numpy()
    res["asplit_20_10"] = nsa["split_into_batch"](x[:, :20], maxlen=10).numpy()
    # --- read_data helpers ---
    from toolkit.utils import read_data as rd
    rng = np.random.RandomState(5)
    for (L, dst) in [(7, 7), (3, 8), (12, 4), (13, 4), (100, 17), (5, 1)]:
        f = rng.randn(L, 6).astype(np.float32)
        res[f"mapfeat_in_{L}_{dst}"] = f
        res[f"mapfeat_out_{L}_{dst}"] = rd.func_mapping_feature(f.copy(), dst)
    # --- MER2023 label/index path on the reference's own label file ---
    from toolkit.dataloader.mer2023 import MER2023
    label_path = REF + "/MERBench/dataset/mer2023-dataset-process/label-6way.npz"
    obj = MER2023.__new__(MER2023)
    for split in ["train", "test1", "test2", "test3"]:
        names, labels = obj.read_names_labels(label_path, split)
        res[f"labels_{split}_n"] = np.array(len(names))
        res[f"labels_{split}_first_names"] = np.array(names[:5])
        res[f"labels_{split}_emo"] = np.array([l["emo"] for l in labels], dtype=np.int64)
        res[f"labels_{split}_val"] = np.array([float(l["val"]) for l in labels], dtype=np.float64)
    random.seed(2023)
    folds = obj.random_split_indexes(3373, 5)
    for i, (tr, ev) in enumerate(folds):
        res[f"fold{i}_train"] = np.array(tr, dtype=np.int64)
        res[f"fold{i}_eval"] = np.array(ev, dtype=np.int64)
    rng = np.random.RandomState(6)
    probs, labs = rng.rand(50, 6), rng.randint(0, 6, 50)
    vp, vl = rng.randn(50), rng.randn(50)
    r, s = obj.calculate_results(probs, labs, vp, vl)
    res.update(metric_probs=probs, metric_labs=labs, metric_vp=vp, metric_vl=vl, metric_acc=np.array(r["emoacc"]),
               metric_f1=np.array(r["emofscore"]), metric_mse=np.array(r["valmse"]), metric_str=np.array(s))
    from toolkit.utils import metric as M
    res["metric_emoval"] = np.array(M.gain_metric_from_results(r, "emoval"))
    # --- text: special-token probing with a real (tiny) BERT tokenizer ---
    nst = {"torch": torch}
    exec_defs(REF + "/MERBench/feature_extraction/text/extract_text_huggingface.py", {"find_start_end_pos"}, nst)
    from transformers import BertTokenizer
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("今天气真好你我他是的")
        open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8").write("\n".join(vocab))
        tok = BertTokenizer(os.path.join(d, "vocab.txt"))
        se = nst["find_start_end_pos"](tok)
    res["bert_start_end"] = np.array([se[0], -99 if se[1] is None else se[1]])
    np.savez_compressed(os.path.join(OUT, "index_paths.npz"), **res)


def hf_goldens():
    """Outputs of the HuggingFace classes (the arithmetic the reference calls) on the seeded tiny checkpoints."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from transformers import CLIPConfig, CLIPModel, HubertConfig, HubertModel, RobertaConfig, RobertaModel
    from mertools_amd import synthetic as W
    res = {}
    cfg = W.hubert_config("tiny")
    sd = W.hubert_state_dict(cfg, 1)
    hc = HubertConfig(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, conv_dim=(64,) * 7,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, attn_implementation="eager")
    m = HubertModel(hc).eval()
    m.load_state_dict(sd, strict=False)
    wav = W.synth_audio(3, 8000, seed=5)
    with torch.no_grad():
        hs = m(wav, output_hidden_states=True).hidden_states
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(dim=0)
    res["hubert_hs0"], res["hubert_hs_last"] = hs[0].numpy(), hs[-1].numpy()
    res["hubert_utt"] = np.mean(feat.view(3, -1, 128).numpy(), axis=1)
    cc = W.clip_config("tiny")
    csd = W.clip_state_dict(cc, 3)
    hcc = CLIPConfig(vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                        patch_size=16, image_size=64), projection_dim=64, attn_implementation="eager")
    cm = CLIPModel(hcc).eval()
    cm.load_state_dict(csd, strict=False)
    px = W.synth_frames(5, 64, seed=7)
    with torch.no_grad():
        o = cm.get_image_features(px)
    res["clip_feats"] = (o if torch.is_tensor(o) else o.pooler_output).numpy()
    bc = W.bert_config("tiny")
    bsd = W.bert_state_dict(bc, 4)
    hbc = RobertaConfig(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                        max_position_embeddings=70, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, attn_implementation="eager")
    bm = RobertaModel(hbc, add_pooling_layer=False).eval()
    bm.load_state_dict(bsd, strict=False)
    ids = W.synth_tokens(4, 24, vocab=300, seed=8, bos=3, eos=4)
    with torch.no_grad():
        hs = bm(input_ids=ids, attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states
    f = torch.stack(hs)[[-4, -3, -2, -1]].sum(dim=0)
    res["roberta_frame"], res["roberta_utt"] = f.numpy(), f[:, 1:-1].mean(1).numpy()
    np.savez_compressed(os.path.join(OUT, "encoders_tiny_hf.npz"), **res)


def mer2024_goldens():
    """MER2024's dataloader class and Attention_TOPN, run from the reference's own sources (MER2024/toolkit/dataloader/mer2024.py,
    MER2024/toolkit/models/attention_topn.py) on a small synthetic label file (the reference ships none for MER2024)."""
    ref_toolkit()
    from sklearn.metrics import accuracy_score, f1_score, mean_squared_error
    emos = ['neutral', 'angry', 'happy', 'sad', 'worried', 'surprise']
    rng = np.random.RandomState(2024)
    train = {f"sample_{i:05d}": {"emo": emos[rng.randint(0, 6)], "val": float(rng.randn())} for i in range(137)}
    test1 = {f"test_{i:05d}": {"emo": emos[rng.randint(0, 6)]} for i in range(31)}
    label_path = os.path.join(OUT, "mer2024_label-6way.npz")
    np.savez_compressed(label_path, train_corpus=train, test1_corpus=test1)
    cfg = argparse.Namespace(PATH_TO_LABEL={"MER2024": label_path})
    ns = {"np": np, "random": random, "accuracy_score": accuracy_score, "f1_score": f1_score, "mean_squared_error": mean_squared_error,
          "emo2idx_mer": {e: i for i, e in enumerate(emos)}, "config": cfg}
    exec_defs(REF + "/MER2024/toolkit/dataloader/mer2024.py", {"MER2024"}, ns)
    args = argparse.Namespace(debug=False, batch_size=32, num_workers=0, dataset="MER2024")
    obj = ns["MER2024"](args)
    res = dict(output_dim1=np.array(args.output_dim1), output_dim2=np.array(args.output_dim2), metric_name=np.array(args.metric_name))
    for split in ["train", "test1"]:
        names, labels = obj.read_names_labels(label_path, split)
        res[f"labels_{split}_names"] = np.array(names)
        res[f"labels_{split}_emo"] = np.array([l["emo"] for l in labels], dtype=np.int64)
        res[f"labels_{split}_val"] = np.array([float(l["val"]) for l in labels], dtype=np.float64)
    res["debug_n"] = np.array(len(obj.read_names_labels(label_path, "train", debug=True)[0]))
    random.seed(2024)
    for i, (tr, ev) in enumerate(obj.random_split_indexes(137, 5)):
        res[f"fold{i}_train"] = np.array(tr, dtype=np.int64)
        res[f"fold{i}_eval"] = np.array(ev, dtype=np.int64)
    probs, labs = rng.rand(40, 6), rng.randint(0, 6, 40)
    r, sres = obj.calculate_results(probs, labs, [], [])
    res.update(metric_probs=probs, metric_labs=labs, metric_acc=np.array(r["emoacc"]), metric_f1=np.array(r["emofscore"]),
               metric_str=np.array(sres), metric_keys=np.array(sorted(r)))
    # the model-selection metric under metric_name == 'emo' (MER2024/toolkit/utils/metric.py:22-24)
    nsm = {"np": np}
    exec_defs(REF + "/MER2024/toolkit/utils/metric.py", {"gain_metric_from_results", "overall_metric", "gain_cv_results"}, nsm)
    res["metric_emo"] = np.array(nsm["gain_metric_from_results"](r, "emo"))
    res["cv_str"] = np.array(nsm["gain_cv_results"]([{"eval_emofscore": 0.5, "eval_emoacc": 0.25}, {"eval_emofscore": 0.75, "eval_emoacc": 0.5}]))
    np.savez_compressed(os.path.join(OUT, "index_paths_mer2024.npz"), **res)

    # Attention_TOPN: 2 feature sets per modality slot (6 streams), the emotion-only head of MER2024 (output_dim2 = 0)
    sys.path.insert(0, REF + "/MER2024")
    for k in [k for k in sys.modules if k == "toolkit" or k.startswith("toolkit.")]:
        del sys.modules[k]
    nsa = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional}
    exec_defs(REF + "/MER2024/toolkit/models/modules/encoder.py", {"MLPEncoder"}, nsa)
    exec_defs(REF + "/MER2024/toolkit/models/attention_topn.py", {"Attention_TOPN"}, nsa)
    dims = [96, 48, 80, 64, 32, 72]
    a2 = argparse.Namespace(audio_dim=dims, output_dim1=6, output_dim2=0, dropout=0.0, hidden_dim=64, grad_clip=-1.0)
    torch.manual_seed(2024)
    model = nsa["Attention_TOPN"](a2).eval()
    batch = {f"feat{i}": torch.randn(16, d) for i, d in enumerate(dims)}
    with torch.no_grad():
        f, e, v, il = model(batch)
    np.savez_compressed(os.path.join(OUT, "fusion_attention_topn.npz"), dims=np.array(dims), features=f.numpy(), emos_out=e.numpy(),
                        vals_out=v.numpy(), interloss=il.numpy(), **{f"x_{k}": x.numpy() for k, x in batch.items()},
                        **{f"init_{k}": p.numpy() for k, p in model.state_dict().items()})
    sys.path.remove(REF + "/MER2024")


if __name__ == "__main__":
    if "--mer2024" in sys.argv:      # (re-runs only the MER2024 vectors: the other files stay byte-identical)
        mer2024_goldens()
        sys.exit(0)
    hf_goldens()
    fusion_goldens()
    fusion_frame_goldens()
    index_goldens()
    mer2024_goldens()
    print("wrote:", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))

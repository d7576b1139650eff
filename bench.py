#!/usr/bin/env python3
"""bench.py — clips/sec of the tri-modal (A+V+T) feature-extraction hot path on N MI355X.

One "step" = one batch of B synthetic clips through all three encoders on the HIP path:
  audio  5 s @16 kHz -> HuBERT-base -> last-4 sum -> utterance mean            [B,80000] -> [B,768]
  visual 8 frames x 224^2 -> CLIP-ViT-B/16 get_image_features -> frame mean    [8B,3,224,224] -> [B,512]
  text   64 tokens -> RoBERTa-base -> last-4 sum -> strip specials -> mean     [B,64] -> [B,768]
(BASELINE.json metric; config 4's extraction leg, which is configs 2+3+text on one GPU.)
Inputs are resident in HBM before the timed region.  N>1: one process per GPU (torchrun), clips
shard across ranks with no data-path collective (weak scaling); the timed region is bracketed by
barrier + synchronize and the max over ranks is reported.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (gemm16*: fp16 MFMA GEMM; "_w2" = 2-pass weights-split variant) — algorithmic 2*M*N*K FLOPs of its launches
                divided by their HIP-event durations (measured in a second, instrumented pass over
                the same steps), against the 2.5 PFLOP/s dense fp16 MFMA peak.
  cpu_baseline  the reference's own arithmetic (live HuggingFace classes + the extractor scripts' post-processing, oracle/hf_live.py,
                kind "reference") on the host cores: batch 1 as the reference loops (value) and batch 32; the oracle restatement's
                timing rides along as oracle_port.
  parity        max|x-ref|/max|ref| of the first two clips' features of the LAST timed step (the full-batch kernel selection)
                against the CPU oracle, per modality; the run fails when one exceeds 1e-3.
N > 1 without torchrun: bench.py re-executes itself under torch.distributed.run (one rank per GPU) and fails loudly when
the box has fewer than N GPUs.  Under N > 1 every step also issues the fusion minibatch exchange of BASELINE configs[3]
(distributed.gather_fusion_batch: ONE fused RCCL all-gather of the [B, Da+Dt+Dv] feature rows) on a side stream; its
duration is reported separately ("allgather") and it overlaps the next step's extraction.
"""
import time
_T_PROCESS = time.perf_counter()   # (the cold files -> .npy child reports its import + model-build time from here)
import argparse
import ctypes
import json
import os
import sys

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  An RCCL process group creates streams of its own, after which
# the three modality streams share queues and serialise: 32.4 instead of 30.2 ms per step on ONE GPU with a one-rank group and no
# collective at all (--force-dist; profiles/r03_rccl_hw_queues.txt).  8 queues restore the overlap.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_CLIP = {"hubert-base": 71.67, "clip-vit-b16-8f": 281.0, "roberta-base-64": 11.02,
                  "hubert-large": 185.48, "videomae-large-16f": 1193.7, "roberta-large-64": 39.06}
# the two measured configurations: BASELINE.json configs[3] (base trio, the headline metric) and configs[4] (large trio)
CONFIGS = {
    "base": {"a": ("hubert", "base", "hubert-base"), "v": ("clip", "base16", "clip-vit-b16-8f"), "t": ("bert", "roberta-base", "roberta-base-64"),
             "workload": "tri-modal base extract: HuBERT-base 5s@16kHz + CLIP-ViT-B/16 8x224^2 + RoBERTa-base 64 tok "
                         "(BASELINE.json configs[3] extraction leg = configs[1]+[2]+text on each GPU)"},
    "large": {"a": ("hubert", "large", "hubert-large"), "v": ("videomae", "large", "videomae-large-16f"), "t": ("bert", "roberta-large", "roberta-large-64"),
              "workload": "tri-modal large extract: HuBERT-large 5s@16kHz + VideoMAE-L 16x224^2 + RoBERTa-large 64 tok (BASELINE.json configs[4])"},
}
PEAK_F16_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


MFMA_PASSES = {"gemm16p": 1, "gemm16": 1, "gemm16_w2": 2, "gemm16_x3": 3, "gemm16_mx": 1.5}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--config", default="base", choices=sorted(CONFIGS), help="base = BASELINE configs[3] (the headline metric); large = configs[4]")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="16-bit MFMA operand type; bf16 has no MX kernel: use --precision accurate (3 passes) for parity")
    ap.add_argument("--modalities", default="avt", help="subset of a,v,t (default all three = the headline metric)")
    ap.add_argument("--precision", default="mean", choices=["fast", "balanced", "mx", "mean", "mean_a2", "mean_conv3", "accurate"],
                    help="GEMM passes: fast=1 (fp16), balanced=2 (weights hi+lo f16 planes), mx=1 + MX-fp4 correction of the weight "
                         "residual, mean (default)=1 + the weight residual applied through each sequence's mean activation (a per-clip correction row: "
                         "mer_seq_bias; same parity as balanced / mx), mean_a2 = mean with hi + lo ACTIVATION planes in the blocks (2 passes + the table), "
                         "mean_conv3 = mean with the HuBERT conv stack on three passes (the rungs the load-time self-check tries before accurate), accurate=3 + fp32 attention")
    ap.add_argument("--streams", type=int, default=1, help="1: one HIP stream per modality (default); 0: single stream")
    ap.add_argument("--split", type=int, default=1, help="run each modality's batch as this many sub-batches on their own HIP streams "
                                                           "(kernels of one sub-batch fill the partial last wave of workgroups of the other)")
    ap.add_argument("--split-mods", default="avt", help="modalities --split applies to (the others run their whole batch on one stream)")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path with a one-rank RCCL group (self-test on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the last step's first two clips")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the BASELINE configs[4] (large trio) sub-object of the headline line")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained (>= --sustain-seconds) line")
    ap.add_argument("--no-ladder", action="store_true", help="skip the `ladder` sub-object (the same step under the precision rungs a load-time self-check can move an encoder to)")
    ap.add_argument("--sustain-seconds", type=float, default=20.0)
    ap.add_argument("--e2e-cold-child", default=None, help=argparse.SUPPRESS)   # internal: the fresh process e2e() spawns over its corpus directory
    ap.add_argument("--e2e", type=int, default=1024, help="N > 0: also run N clips files -> .npy through the drop-in drivers (extra key `e2e`, headline line on one GPU only); 0 skips it")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_port_baseline(sample_clips=2):
    """The oracle's fp32 CPU forward (oracle/encoders_ref.py, the restatement the parity tests check against), batch of `sample_clips`."""
    from oracle import encoders_ref as R
    from mertools_amd import synthetic as W
    cores = torch.get_num_threads()
    hc, cc, bc = W.hubert_config("base"), W.clip_config("base16"), W.bert_config("roberta-base")
    hsd, csd, bsd = W.hubert_state_dict(hc, 0), W.clip_state_dict(cc, 0), W.bert_state_dict(bc, 0)
    wav, px, ids = W.synth_audio(sample_clips), W.synth_frames(sample_clips * 8), W.synth_tokens(sample_clips)

    def run():
        with torch.no_grad():
            a = torch.stack(R.hubert_hidden_states(hsd, vars(hc), wav))[[-4, -3, -2, -1]].sum(0).mean(1)
            v = R.clip_image_features(csd, dict(vars(cc.vision_config), projection_dim=cc.projection_dim), px)
            v = v.view(sample_clips, 8, -1).mean(1)
            t = torch.stack(R.bert_hidden_states(bsd, dict(vars(bc), roberta=True), ids, torch.ones_like(ids)))[[-4, -3, -2, -1]].sum(0)[:, 1:-1].mean(1)
        return a, v, t

    run()  # warm-up (thread pools, allocator)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    return {"value": round(sample_clips / sorted(ts)[1], 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"median of 3 x batch of {sample_clips} tri-modal clips, oracle/encoders_ref.py fp32 torch-CPU forward"}


def cpu_baseline():
    """the reference's own arithmetic — the live HuggingFace classes (eager attention, fp32) in the extractor
    scripts' post-processing (oracle/hf_live.py) — on the host cores: (i) batch 1, exactly as the reference loops one clip per
    forward (the headline `value`), (ii) batch 32.  Median of the timed iterations; a bounded sample so the default run stays
    within minutes.  The oracle restatement's own timing is kept as `oracle_port`."""
    from oracle import hf_live as H
    from mertools_amd import synthetic as W
    hub, clip, rob = H.build_base_trio(W)

    def timed(fn, warm, iters, budget_s, min_iters=1):
        t_start = time.perf_counter()
        for _ in range(warm):   # the budget covers the warm-up too: an oversubscribed thread count must not cost minutes
            fn()
            if time.perf_counter() - t_start > 0.3 * budget_s:
                break
        ts = []
        while len(ts) < iters and (len(ts) < min_iters or time.perf_counter() - t_start < budget_s):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    def mode(bs, warm, iters, budget_s, min_iters=1):
        wav, px, ids = W.synth_audio(bs), W.synth_frames(bs * 8), W.synth_tokens(bs)
        per, n_it = {}, {}
        for m, fn in (("a", lambda: H.audio_utt(hub, wav)), ("v", lambda: H.visual_utt(clip, px)), ("t", lambda: H.text_utt(rob, ids))):
            sec, n = timed(fn, warm, iters, budget_s, min_iters)
            per[m], n_it[m] = sec / bs, n
        return {"clips_per_s": round(1.0 / sum(per.values()), 4), "per_modality_clips_per_s": {m: round(1.0 / s, 3) for m, s in per.items()},
                "timed_iterations": n_it, "batch": bs}

    # thread count: measured, not assumed — 32, 64 and one thread per physical core (os.cpu_count() counts SMT siblings on the EPYC
    # boxes: 256 "cpus" = 128 cores; one torch thread per sibling measured 290x slower than 32 threads), the fastest is the baseline
    ncpu = os.cpu_count() or 1
    by_threads = {}
    for th in sorted({min(ncpu, 32), min(ncpu, 64), min(max(ncpu // 2, 1), 128)}):
        torch.set_num_threads(th)
        by_threads[th] = mode(1, 3, 10, 8.0)
    cores = max(by_threads, key=lambda th: by_threads[th]["clips_per_s"])
    torch.set_num_threads(cores)
    b1 = by_threads[cores]
    b32 = mode(32, 0, 3, 10.0, min_iters=3)
    res = {"value": b1["clips_per_s"], "unit": "clips/s", "cores": cores, "cpu_model": _cpu_model(), "kind": "reference",
           "sample": "live HuggingFace HubertModel + CLIPModel.get_image_features + RobertaModel (eager attention, fp32, random-init base "
                     "checkpoints = the HIP path's weights) with the extractor scripts' post-processing; value = tri-modal clips/s at batch 1 "
                     "(the reference's one-clip-per-forward loop): 3 warm-up + median of up to 10 timed forwards per modality (8 s budget each) at "
                     "32 / 64 / one-per-physical-core torch threads (threads_tried), the fastest kept; batch32 = the same at 32 clips per forward (median of 3 forwards per modality)",
           "threads_tried": {str(th): v["clips_per_s"] for th, v in by_threads.items()},
           "batch1": b1, "batch32": b32}
    try:
        res["oracle_port"] = oracle_port_baseline()
    except Exception as e:
        res["oracle_port"] = {"error": repr(e)}
    return res



def kernel_source_sha(root=ROOT):
    """sha256[:16] over every source of the HIP library (mertools_amd/csrc/*.{h,hip,cpp}, sorted by name) — the stamp scripts/pmc_summarize.py,
    scripts/pmc_mfma_summarize.py and scripts/stamp_kernel_stats.py write into a profile: a collection is only quoted for the tree it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "mertools_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# (the one-pass launches of the step: the persistent family gemm16p_kernel<f16, EPI, ACT> — the summarisers pool its instantiations under "gemm16p")
PMC_KERNEL_PREFIX = {"gemm16p": "gemm16p", "gemm16": "gemm16<f16,128,128,32,2,2,1,1,", "gemm16_mx": "gemm16<f16,256,256,32,4,2,1,1,", "gemm16_w2": "gemm16<f16,256,256,32,2,4,1,2,"}


def pmc_traffic(kernel, algorithmic_bytes_per_launch, root=ROOT):
    """HBM-side bytes per launch of `kernel` from the PMC passes (scripts/pmc_traffic.sh -> profiles/*pmc_hbm_traffic*.json).
    A stored figure is only quoted when it was collected on THIS kernel source (the collection's `_source_sha` equals
    kernel_source_sha()); otherwise traffic is None and the detail says which collections were ignored.  -> (traffic, detail)"""
    import glob
    sha = kernel_source_sha(root)
    prefix = PMC_KERNEL_PREFIX.get(kernel)
    stale = []
    for pmc in sorted(glob.glob(os.path.join(root, "profiles", "*pmc_hbm_traffic*.json")), reverse=True):
        d = json.load(open(pmc))
        if d.get("_source_sha") != sha:
            stale.append(os.path.basename(pmc))
            continue
        k = next((v for name, v in d.items() if prefix and name.startswith(prefix)), None)
        if k:
            detail = {"fetch_MB_per_launch_x2_corrected": round(k["fetch_mb_x2"], 1), "write_MB_per_launch": round(k["write_mb"], 1),
                      "algorithmic_MB_per_launch": round(algorithmic_bytes_per_launch / 1e6, 1), "source": "profiles/" + os.path.basename(pmc),
                      "kernel_source_sha": sha}
            return round((k["fetch_mb_x2"] + k["write_mb"]) * 1e6), detail   # FETCH_SIZE x2-corrected + WRITE_SIZE
    return None, {"note": f"no PMC collection matches this kernel source (sha {sha}); older collections ignored: {stale}"}


def pmc_mfma_busy(kernel, root=ROOT):
    """MFMA utilisation of `kernel` from the SQ counter pass (scripts/pmc_mfma.sh -> profiles/*pmc_mfma*.json): SQ_VALU_MFMA_BUSY_CYCLES /
    (SQ_BUSY_CYCLES-derived active CU cycles), quoted — like the traffic — only when the collection is stamped with this kernel source."""
    import glob
    sha = kernel_source_sha(root)
    prefix = PMC_KERNEL_PREFIX.get(kernel)
    for pmc in sorted(glob.glob(os.path.join(root, "profiles", "*pmc_mfma*.json")), reverse=True):
        d = json.load(open(pmc))
        if d.get("_source_sha") != sha:
            continue
        k = next((v for name, v in d.items() if prefix and name.startswith(prefix)), None)
        if k:
            c = k.get("counters_per_launch", {})
            return {"mfma_busy": k.get("mfma_busy"), "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), mean per launch",
                    "SQ_VALU_MFMA_BUSY_CYCLES": c.get("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_INSTS_MFMA": c.get("SQ_INSTS_MFMA"),
                    "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"), "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES"), "SQ_WAIT_INST_ANY": c.get("SQ_WAIT_INST_ANY"),
                    "launches": k.get("launches"), "source": "profiles/" + os.path.basename(pmc), "kernel_source_sha": sha}
    return None


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: one rank per GPU via torch.distributed.run, same flags."""
    n = torch.cuda.device_count()
    if n < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but this box has {n} visible GPU(s); refusing to run fewer ranks than asked "
                 f"(a scaling line must measure what it is labelled with)")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def oracle_features(kind, size, x):
    """The CPU oracle's UTT feature of one modality of CONFIGS (same synthetic weights as the HIP model)."""
    from oracle import encoders_ref as R
    from mertools_amd import synthetic as W
    with torch.no_grad():
        if kind == "hubert":
            hc = W.hubert_config(size)
            return torch.stack(R.hubert_hidden_states(W.hubert_state_dict(hc, 0), vars(hc), x))[[-4, -3, -2, -1]].sum(0).mean(1)
        if kind == "clip":
            cc = W.clip_config(size)
            return R.clip_image_features(W.clip_state_dict(cc, 0), dict(vars(cc.vision_config), projection_dim=cc.projection_dim), x).view(x.shape[0] // 8, 8, -1).mean(1)
        if kind == "videomae":
            vc = W.videomae_config(size)
            return R.videomae_last_hidden_state(W.videomae_state_dict(vc, 0), vars(vc), x).mean(1)   # mean of equal-size segment means
        bc = W.bert_config(size)
        return torch.stack(R.bert_hidden_states(W.bert_state_dict(bc, 0), dict(vars(bc), roberta=True), x, torch.ones_like(x)))[[-4, -3, -2, -1]].sum(0)[:, 1:-1].mean(1)


DETAIL = {}   # filled by parity_check: per-modality companions of the max-norm figure (last call)


def parity_check(feats, inputs, mods, config="base", nclip=2):
    """Features of the first `nclip` clips of the last timed step (computed inside the full batch, i.e. by the kernels the
    timing selected) against the CPU oracle on the same weights and inputs.  north_star tolerance: 1e-3 (max-norm relative).
    Returns (errors per modality, oracle seconds per clip per modality)."""
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    out, secs = {}, {}
    for m in "avt":
        if m not in mods:
            continue
        kind, size, _ = CONFIGS[config][m]
        rows = nclip * (8 if kind == "clip" else 1)
        t0 = time.perf_counter()
        ref = oracle_features(kind, size, inputs[m][:rows].cpu())
        secs[m] = (time.perf_counter() - t0) / nclip
        x, r = feats[m][:nclip].double().cpu(), ref.double()
        out[m] = float((x - r).abs().max() / r.abs().max())
        # the asserted figure is max-norm relative; two norm-free companions so that small-magnitude dimensions are not invisible:
        # the worst clip's cosine distance and the RMS error relative to the RMS feature
        DETAIL[m] = {"one_minus_cosine_max": float(f"{(1.0 - torch.nn.functional.cosine_similarity(x, r, dim=-1)).max().item():.3e}"),
                     "rms_rel": float(f"{((x - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item():.3e}")}
    return {k: float(f"{v:.3e}") for k, v in out.items()}, secs


def build_models(cfgset, mods, B, dev, precision, dtype, rank):
    """-> (models, inputs): every rank the same weights (replicated), its own shard of synthetic clips (seed offset by rank)."""
    from mertools_amd import synthetic as W
    from mertools_amd.encoders import HipBertModel, HipCLIPModel, HipHubertModel, HipVideoMAEModel
    models, inputs = {}, {}
    for m in "avt":
        if m not in mods:
            continue
        kind, size, _ = cfgset[m]
        kw = dict(device=dev, precision=precision, dtype=dtype)
        if kind == "hubert":
            c = W.hubert_config(size)
            models[m], inputs[m] = HipHubertModel(W.hubert_state_dict(c, 0), c, **kw), W.synth_audio(B, seed=1234 + rank).to(dev)
        elif kind == "clip":
            c = W.clip_config(size)
            models[m], inputs[m] = HipCLIPModel(W.clip_state_dict(c, 0), c, **kw), W.synth_frames(B * 8, seed=1235 + rank).to(dev)
        elif kind == "videomae":
            c = W.videomae_config(size)
            models[m], inputs[m] = HipVideoMAEModel(W.videomae_state_dict(c, 0), c, **kw), W.synth_video(B, seed=1235 + rank).to(dev)
        else:
            c = W.bert_config(size)
            models[m], inputs[m] = HipBertModel(W.bert_state_dict(c, 0), c, **kw), W.synth_tokens(B, seed=1236 + rank).to(dev)
    return models, inputs


def measure(args, config, steps, warmup, dev, dist, rank, world, sustain_s=0.0, want_roofline=True, want_parity=True):
    """One configuration (CONFIGS[config]) through the timed protocol: `warmup` untimed steps, barrier + synchronize, `steps` timed steps,
    barrier + synchronize, max over ranks.  -> dict(value, ms_per_step, parity, oracle_secs, roofline, allgather, sustained, feats...)"""
    from mertools_amd import _lib
    B = args.batch
    mods = set(args.modalities)
    cfgset = CONFIGS[config]
    models, inputs = build_models(cfgset, mods, B, dev, args.precision, args.dtype, rank)
    # what each encoder object actually runs (a load-time self-check may have moved it up the precision ladder: ADVICE r4)
    running = {m: {"precision": getattr(models[m], "precision", args.precision), "escalated": getattr(models[m], "escalated", None)} for m in models}
    frames_per_clip = [8] * B
    lengths = [64] * B

    # the three encoders are independent: each runs on its own HIP stream so that the tail of one kernel (a partial
    # last wave of workgroups) and the small text GEMMs overlap with another modality's work; --split S additionally
    # runs every modality's batch as S sub-batches on S streams (same clips per step, same kernels)
    assert B % max(1, args.split) == 0, "--split must divide --batch"
    Sm = {m: (max(1, args.split) if m in args.split_mods else 1) for m in "avt"}   # sub-batches per modality
    S = max(Sm.values())
    # (HIP stream priorities for the short audio / text streams, or for the visual one, were measured in round 3: within 0.3 %)
    streams = {m: [torch.cuda.Stream(device=dev) for _ in range(Sm[m])] for m in "avt"} if (args.streams or S > 1) else None
    per = {"a": 1, "v": 8 if cfgset["v"][0] == "clip" else 1, "t": 1}   # input rows per clip
    parts = {m: [inputs[m][i * (B // Sm[m]) * per[m]:(i + 1) * (B // Sm[m]) * per[m]] for i in range(Sm[m])] for m in mods}

    def run(m, i=None):
        if i is None:   # whole batch (roofline leg)
            xa, xv, xt, n = inputs.get("a"), inputs.get("v"), inputs.get("t"), B
        else:
            xa = xv = xt = parts[m][i]
            n = B // Sm[m]
        if m == "a":
            return models["a"].extract_utterance(xa)
        if m == "v":
            if cfgset["v"][0] == "videomae":
                return models["v"].extract_utterance(xv)
            return models["v"].extract_utterance(xv, frames_per_clip[:n])
        return models["t"].extract_utterance(xt, lengths[:n], 1, -1)

    def step():
        """-> {modality: [features of each sub-batch]}"""
        if streams is None:
            return {m: [run(m)] for m in "avt" if m in mods}
        out = {m: [] for m in mods}
        cur = torch.cuda.current_stream()
        for i in range(S):
            for m in "vat":   # longest first
                if m in mods and i < Sm[m]:
                    streams[m][i].wait_stream(cur)
                    with torch.cuda.stream(streams[m][i]):
                        out[m].append(run(m, i))
        for m in mods:
            for st in streams[m]:
                cur.wait_stream(st)
        return out

    # N > 1: the fusion-minibatch exchange of configs[3] — every rank's [B, Da | Dt | Dv] rows in ONE fused RCCL all-gather
    # (distributed.gather_fusion_batch) on its own stream, so it overlaps the next step's extraction; timed with its own events
    comm = torch.cuda.Stream(device=dev) if (dist is not None and mods == set("avt") and not os.environ.get("MER_BENCH_NO_EXCHANGE")) else None
    ag_events = []

    def exchange(out):
        from mertools_amd import distributed as D
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(comm)
            for m in "atv":
                for x in out[m]:
                    x.record_stream(comm)   # allocated on the modality streams, read here: the allocator must not recycle them early
            a, t, v = (out[m][0] if len(out[m]) == 1 else torch.cat(out[m], 0) for m in "atv")
            full = D.gather_fusion_batch(a, t, v, counts=[B] * world)   # equal blocks: no count exchange, no host sync
            e1.record(comm)
        ag_events.append((e0, e1))
        if len(ag_events) > 64:   # bounded bookkeeping on long (sustained) runs
            del ag_events[:-64]
        return full

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        out = step()
        if comm is not None:
            exchange(out)
    barrier()
    ag_events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
        if comm is not None:
            full = exchange(out)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    feats = {m: torch.cat(out[m], 0) for m in out}
    for o in feats.values():
        assert torch.isfinite(o).all(), "non-finite features"
    res = {"value": B * world * steps / dt, "ms_per_step": dt / steps * 1e3, "dt": dt, "sub_batches": {m: Sm[m] for m in sorted(mods)}}
    if comm is not None:
        assert full[0].shape == (B * world, feats["a"].shape[1]) and torch.equal(full[0][rank * B:(rank + 1) * B], feats["a"]), \
            "fusion minibatch exchange: this rank's rows did not come back in rank order"
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in ag_events)
        res["allgather"] = {"collective": "all_gather_into_tensor (RCCL), one per step, side stream", "ranks": world,
                            "rows_per_rank": B, "bytes_per_rank": int(B * sum(feats[m].shape[1] for m in "atv") * 4),
                            "ms_median": round(ms[len(ms) // 2], 4), "ms_max": round(ms[-1], 4)}

    # sustained line: the same step for >= sustain_s seconds (the 20-step region is < 1 s on a chip whose clocks follow its power
    # budget: DESIGN.md §3) — same barrier / max-over-ranks protocol, reported next to `value`, never instead of it
    if sustain_s > 0:
        n_sus = max(steps, int(sustain_s / (dt / steps)) + 1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            out = step()
            if comm is not None:
                exchange(out)
        barrier()
        ds = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ds = t.item()
        res["sustained"] = {"seconds": round(ds, 2), "steps": n_sus, "clips_per_s": round(B * world * n_sus / ds, 2), "ms_per_step": round(ds / n_sus * 1e3, 3)}

    gflop_clip = sum(GFLOP_PER_CLIP[cfgset[m][2]] for m in mods)
    res["gflop_per_clip"] = gflop_clip
    res["running"] = running
    res["whole_step_tflops"] = gflop_clip * B * world * steps / dt / 1e3
    if rank == 0 and want_parity:
        res["parity"], res["oracle_secs"] = parity_check(feats, inputs, mods, config, nclip=2 if config == "base" else 1)
        res["parity_detail"] = dict(DETAIL)

    if want_roofline:
        lib = _lib.lib()
        lib.mer_prof_enable(1)
        n_it = max(1, min(steps, 3))
        for _ in range(n_it):   # single stream: a kernel's events must bracket only its own execution
            for m in "avt":
                if m in mods:
                    run(m)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.mer_prof_report(buf, len(buf))
        lib.mer_prof_enable(0)
        recs = {r["name"]: r for r in json.loads(buf.value.decode())}
        tot_ms = sum(r["ms"] for r in recs.values())
        dom = max(recs.values(), key=lambda r: r["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic, traffic_detail = pmc_traffic(dom["name"], dom["bytes"] / dom["calls"])
        # FLOPs the kernels actually executed per clip (2*M*N*K of every launch; the CLS-only last block of the CLIP tower computes less
        # than the reference's forward does) next to the reference-algorithmic figure every rate in this line is quoted on
        executed = sum(r["flops"] for r in recs.values()) / (n_it * B) / 1e9
        res["roofline"] = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic, "traffic_detail": traffic_detail,
                           "mfma_busy": pmc_mfma_busy(dom["name"]),
                           # the 2-pass (weights hi+lo) kernel issues 2x the algorithmic MFMA work; 3-pass 3x
                           # (gemm16_mx: one f16 pass + a bf8 x fp4 K=128 correction at the fp8 MFMA rate = 1.5 f16-pass equivalents)
                           "mfma_passes": MFMA_PASSES.get(dom["name"], 1),
                           "mfma_issued_frac": round(ach * MFMA_PASSES.get(dom["name"], 1) / PEAK_F16_TFLOPS, 4),
                           "avg_launch_us": round(dom["ms"] * 1e3 / dom["calls"], 2), "launches": dom["calls"],
                           "share_of_gpu_time": round(dom["ms"] / tot_ms, 4),
                           "other_kernels": {k: {"ms_share": round(v["ms"] / tot_ms, 4),
                                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                                 "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                                             for k, v in recs.items() if k != dom["name"]},
                           "whole_step_tflops": round(res["whole_step_tflops"], 2),
                           "whole_step_frac": round(res["whole_step_tflops"] / PEAK_F16_TFLOPS, 4),
                           "gflop_per_clip_reference": gflop_clip, "gflop_per_clip_executed": round(executed, 2),
                           # every rate above is quoted on the REFERENCE's algorithmic FLOPs per clip; the kernels execute fewer (CLIP's last
                           # block runs for the CLS rows only): the same step on the FLOPs actually issued
                           "whole_step_tflops_executed": round(res["whole_step_tflops"] * executed / gflop_clip, 2),
                           "whole_step_frac_executed": round(res["whole_step_tflops"] * executed / gflop_clip / PEAK_F16_TFLOPS, 4)}
    del models, inputs
    torch.cuda.empty_cache()
    return res


def e2e_cold_child(args):
    """A user's command line: a FRESH process imports the package, builds the three encoders and runs the three drivers once over the
    corpus e2e() prepared — no warm-up pass, first launches of every kernel, cold read-ahead.  Prints one JSON line of wall times."""
    import contextlib
    import glob
    import io
    import transformers as tr
    from mertools_amd import synthetic as W
    from mertools_amd.encoders import HipBertModel, HipCLIPModel, HipHubertModel
    from mertools_amd.extract import audio, text, visual
    root, B = args.e2e_cold_child, args.batch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    t_import = time.perf_counter() - _T_PROCESS
    hc, cc, bc = W.hubert_config("base"), W.clip_config("base16"), W.bert_config("roberta-base")
    kw = dict(device=dev, precision=args.precision, dtype=args.dtype)
    t0 = time.perf_counter()
    ma, mv, mt = HipHubertModel(W.hubert_state_dict(hc, 0), hc, **kw), HipCLIPModel(W.clip_state_dict(cc, 0), cc, **kw), HipBertModel(W.bert_state_dict(bc, 0), bc, **kw)
    tok = tr.BertTokenizer(os.path.join(root, "vocab.txt"))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    wavs = sorted(glob.glob(os.path.join(root, "wav", "*.wav")))
    vids = sorted(os.listdir(os.path.join(root, "face")))
    N = len(wavs)
    secs = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for name, fn in (("a", lambda: audio.extract("hubert-base", wavs, os.path.join(root, "cold_a"), "UTTERANCE", 0, model=ma, batch_rows=B, device_preprocess=True, workers=8, rank=0, world=1)),
                         ("v", lambda: visual.extract(mv, os.path.join(root, "face"), os.path.join(root, "cold_v"), "UTTERANCE", vids=vids, frames_per_batch=8 * B, device_preprocess=True, workers=8, rank=0, world=1)),
                         ("t", lambda: text.extract_embedding("roberta-base", os.path.join(root, f"trans_{N}.csv"), os.path.join(root, "cold_t"), "UTTERANCE", gpu=0, model=mt, tokenizer=tok, batch_size=B, rank=0, world=1))):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            secs[name] = time.perf_counter() - t0
    nfiles = sum(len(os.listdir(os.path.join(root, d))) for d in ("cold_a", "cold_v", "cold_t/roberta-base-UTT"))
    print(json.dumps({"clips": N, "files": nfiles, "import_s": round(t_import, 3), "model_build_s": round(t_build, 3), "driver_s": {k: round(v, 3) for k, v in secs.items()}}))


def e2e_cold(args, root, N):
    """-> the `cold` sub-object of `e2e`: the drivers in a fresh process (e2e_cold_child), no warm-up."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--e2e-cold-child", root, "--batch", str(args.batch), "--precision", args.precision, "--dtype", args.dtype]
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    wall = time.perf_counter() - t0
    if out.returncode != 0:
        return {"error": (out.stderr or out.stdout)[-400:]}
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["files"] == 3 * N, d
    drv = sum(d["driver_s"].values())
    return {"clips_per_s": round(N / drv, 1), "seconds": round(drv, 3), "per_modality_seconds": d["driver_s"],
            "definition": "a fresh process, no warm-up pass: the three drivers one after the other, N / (t_audio + t_visual + t_text); every kernel's first launch, "
                          "the first batches' ramp-up and the cold read-ahead are inside",
            "process_seconds": round(wall, 3), "import_s": d["import_s"], "model_build_s": d["model_build_s"],
            "note": "process_seconds also holds interpreter start, imports and the synthetic checkpoints' construction + weight packing (a real run reads a checkpoint instead)"}


def e2e(args, dev):
    """What a MERTools user gets from the drop-in drivers: files -> .npy.  N synthetic clips on /dev/shm (PCM16 wav, uint8 frame stacks,
    a transcription csv) through extract.audio / visual / text — GPU pre-processing, threaded host read-ahead, pinned asynchronous D2H and
    worker-thread np.save (extract.pipeline) — the three drivers on three host threads, each on its own HIP stream.  Reported next to the
    kernel-only `value`, never instead of it.  The first 32 clips are also run with async_save=False and compared byte for byte."""
    import shutil
    import tempfile
    import threading
    import wave
    import numpy as np
    import pandas as pd
    import transformers as tr
    from mertools_amd import synthetic as W
    from mertools_amd.encoders import HipBertModel, HipCLIPModel, HipHubertModel
    from mertools_amd.extract import audio, text, visual
    N = args.e2e
    root = tempfile.mkdtemp(prefix="mer_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rng = np.random.RandomState(0)
        wavs, vids = [], []
        os.makedirs(os.path.join(root, "wav"))
        for i in range(N):
            pth = os.path.join(root, "wav", f"clip{i:05d}.wav")
            with wave.open(pth, "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
                w.writeframes((np.clip(rng.randn(80000) * 0.1, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())
            wavs.append(pth)
            vid = f"clip{i:05d}"
            os.makedirs(os.path.join(root, "face", vid))
            np.save(os.path.join(root, "face", vid, f"{vid}.npy"), rng.randint(0, 256, (8, 224, 224, 3)).astype(np.uint8))
            vids.append(vid)
        chars = [chr(c) for c in range(0x4E00, 0x4E00 + 3000)]
        for ch in text.PROBE:
            if ch not in chars:
                chars.append(ch)
        vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars
        open(os.path.join(root, "vocab.txt"), "w", encoding="utf-8").write("\n".join(vocab))
        tok = tr.BertTokenizer(os.path.join(root, "vocab.txt"))
        sents = ["".join(chars[j] for j in rng.randint(0, 3000, 62)) for _ in range(N)]   # 62 characters + [CLS] / [SEP] = 64 tokens
        csv = os.path.join(root, "trans.csv")
        pd.DataFrame([dict(name=f"clip{i:05d}", chinese=s_, english="x") for i, s_ in enumerate(sents)]).to_csv(csv, index=False)

        for n in {32, N}:   # the transcription files of the warm-up and of the timed runs (corpus preparation: not timed)
            pd.read_csv(csv).head(n).to_csv(os.path.join(root, f"trans_{n}.csv"), index=False)
        try:      # first, while this process holds no encoder: the same corpus through a fresh process (the number a user sees)
            cold = e2e_cold(args, root, N)
        except Exception as e:
            cold = {"error": repr(e)}

        hc, cc, bc = W.hubert_config("base"), W.clip_config("base16"), W.bert_config("roberta-base")
        kw = dict(device=dev, precision=args.precision, dtype=args.dtype)
        ma, mv, mt = HipHubertModel(W.hubert_state_dict(hc, 0), hc, **kw), HipCLIPModel(W.clip_state_dict(cc, 0), cc, **kw), HipBertModel(W.bert_state_dict(bc, 0), bc, **kw)
        B = args.batch

        def run_a(files, out, asyn=True):
            audio.extract("hubert-base", files, out, "UTTERANCE", dev.index or 0, model=ma, batch_rows=B, device_preprocess=True, workers=8, rank=0, world=1, async_save=asyn)

        def run_v(names, out, asyn=True):
            visual.extract(mv, os.path.join(root, "face"), out, "UTTERANCE", vids=names, frames_per_batch=8 * B, device_preprocess=True, workers=8, rank=0, world=1, async_save=asyn)

        def run_t(n, out, asyn=True, tokenizer=None):
            sub = os.path.join(root, f"trans_{n}.csv")
            text.extract_embedding("roberta-base", sub, out, "UTTERANCE", gpu=dev.index or 0, model=mt, tokenizer=tokenizer or tok, batch_size=B, rank=0, world=1, async_save=asyn)

        # warm-up + byte-identity of the asynchronous path: the first 32 clips with and without it
        import contextlib
        import io
        same = True
        with contextlib.redirect_stdout(io.StringIO()):
            for tag, asyn in (("sync", False), ("async", True)):
                run_a(wavs[:32], os.path.join(root, f"a_{tag}"), asyn)
                run_v(vids[:32], os.path.join(root, f"v_{tag}"), asyn)
                run_t(32, os.path.join(root, f"t_{tag}"), asyn)
        for d0, d1 in (("a_sync", "a_async"), ("v_sync", "v_async"), ("t_sync/roberta-base-UTT", "t_async/roberta-base-UTT")):
            for f in sorted(os.listdir(os.path.join(root, d0))):
                same = same and open(os.path.join(root, d0, f), "rb").read() == open(os.path.join(root, d1, f), "rb").read()
        torch.cuda.synchronize()

        # kernel-only rate of each encoder alone (inputs resident, no files): what its driver is compared with
        xs = {"a": W.synth_audio(B).to(dev), "v": W.synth_frames(B * 8).to(dev), "t": W.synth_tokens(B).to(dev)}
        kern = {}
        for name, fn in (("a", lambda: ma.extract_utterance(xs["a"])), ("v", lambda: mv.extract_utterance(xs["v"], [8] * B)),
                         ("t", lambda: mt.extract_utterance(xs["t"], [64] * B, 1, -1))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
            kern[name] = 8 * B / (time.perf_counter() - t0)
        # the reference runs its three extraction scripts one after the other: each driver alone over the corpus (its own host + GPU time)
        alone, stages = {}, {}
        from mertools_amd.extract import pipeline
        with contextlib.redirect_stdout(io.StringIO()):
            for name, fn in (("a", lambda: run_a(wavs, os.path.join(root, "out_a"))), ("v", lambda: run_v(vids, os.path.join(root, "out_v"))),
                             ("t", lambda: run_t(N, os.path.join(root, "out_t")))):
                torch.cuda.synchronize()
                pipeline.trace_enable(True)   # where the GPU-feeding thread's wall time goes (a handful of perf_counter calls per batch)
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                alone[name] = time.perf_counter() - t0
                stages[name] = {k: round(v * 1e3, 2) for k, v in sorted(pipeline.TRACE.items())}
                pipeline.trace_enable(False)
        nfiles = sum(len(os.listdir(os.path.join(root, d))) for d in ("out_a", "out_v", "out_t/roberta-base-UTT"))
        assert nfiles == 3 * N, f"e2e: {nfiles} feature files for {N} clips x 3 modalities"
        # ... and the three drivers on three host threads at once, each on its own HIP stream (one interpreter: they share the GIL)
        secs = {}

        def timed(name, fn):
            def body():
                with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    secs[name] = time.perf_counter() - t0
            return threading.Thread(target=body, name=f"e2e-{name}")
        ths = [timed("v", lambda: run_v(vids, os.path.join(root, "thr_v"))), timed("a", lambda: run_a(wavs, os.path.join(root, "thr_a"))),
               timed("t", lambda: run_t(N, os.path.join(root, "thr_t")))]
        with contextlib.redirect_stdout(io.StringIO()):   # (process-global: entered once, around the threads — the drivers print their timings)
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            wall = time.perf_counter() - t0
        # ... and as ONE tri-modal pipeline (extract.trimodal.TriModalExtractor: the bench step's three streams per 64-clip batch, the next
        # batch's uploads under the current batch's kernels, one host thread feeding the GPU) over the same files
        tri = None
        try:
            from mertools_amd.extract.audio import read_pcm16
            from mertools_amd.extract.prefetch import prefetch_map
            from mertools_amd.extract.trimodal import TriModalExtractor
            enc = text.batch_encoder(tok, sents[:256])
            names = [f"clip{i:05d}" for i in range(N)]

            def load(i):
                return read_pcm16(wavs[i])[0], np.load(os.path.join(root, "face", names[i], names[i] + ".npy"))

            def batches(ids):
                loaded = prefetch_map(load, range(N), workers=8, chunk=4)
                for b0 in range(0, N, B):
                    n = min(B, N - b0)
                    rows = [next(loaded) for _ in range(n)]
                    tk = ids[b0:b0 + n]
                    T = max(len(x) for x in tk)
                    yield {"names": names[b0:b0 + n], "audio": [r[0] for r in rows], "frames": [r[1] for r in rows], "frames_per_clip": [8] * n,
                           "input_ids": torch.tensor([x + [0] * (T - len(x)) for x in tk], dtype=torch.int64), "lengths": [len(x) for x in tk]}
            tme = TriModalExtractor(audio=ma, visual=mv, text=mt, device=dev)
            dirs = {"audio": os.path.join(root, "tri_a"), "visual": os.path.join(root, "tri_v"), "text": os.path.join(root, "tri_t")}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids = enc(sents)
            ndone = tme.extract_to_dirs(batches(ids), dirs)
            torch.cuda.synchronize()
            t_tri = time.perf_counter() - t0
            # the same features as the three drivers wrote (same kernels, same per-clip bits: a clip's features do not depend on its batch)
            # (audio and text: byte for byte — a clip's features do not depend on its batch; visual: the driver averages the per-frame features
            #  on the host as the reference does, the pipeline on the GPU: the same frames' mean in another summation order)
            pairs = [(m, np.load(os.path.join(dirs[m], n_ + ".npy")), np.load(os.path.join(root, d, n_ + ".npy")))
                     for m, d in (("audio", "out_a"), ("visual", "out_v"), ("text", "out_t/roberta-base-UTT")) for n_ in names[:48] + names[-16:]]
            differ = {m: sum(not np.array_equal(x, y) for mm, x, y in pairs if mm == m) for m in ("audio", "visual", "text")}
            vis_rel = max(float(np.abs(x - y).max() / np.abs(y).max()) for mm, x, y in pairs if mm == "visual")
            agree = differ["audio"] == 0 and differ["text"] == 0 and vis_rel < 1e-6
            tri = {"seconds": round(t_tri, 3), "clips_per_s": round(ndone / t_tri, 1), "frac_of_three_stream_kernel_only": None,
                   "same_files_as_the_three_drivers": bool(agree), "byte_differing_files_of_64_checked": differ, "visual_max_rel_diff": vis_rel,
                   "what": "TriModalExtractor over the same files: 8 read-ahead threads, per-clip arrays gathered into one pinned block per batch and modality (4 copy threads) and sent up on a copy stream, three encoder streams, np.save on 2 worker threads"}
        except Exception as e:
            tri = {"error": repr(e)}
        seq = sum(alone.values())
        kern_seq = 1.0 / sum(1.0 / v for v in kern.values())
        return {"clips": N, "clips_per_s": round(N / seq, 1), "seconds": round(seq, 3), "cold": cold,
                "definition": "WARM process (a 32-clip pass ran first): the three drivers one after the other over the corpus (how the reference's three extraction scripts are run): N / (t_audio + t_visual + t_text)",
                "kernel_only_clips_per_s_same_schedule": round(kern_seq, 1), "frac_of_kernel_only": round(N / seq / kern_seq, 3),
                "per_modality": {m: {"seconds": round(alone[m], 3), "clips_per_s": round(N / alone[m], 1), "kernel_only_clips_per_s": round(kern[m], 1),
                                     "frac": round(N / alone[m] / kern[m], 3), "feeding_thread_ms": stages[m]} for m in "avt"},
                "three_threads_at_once": {"seconds": round(wall, 3), "clips_per_s": round(N / wall, 1), "per_modality_seconds": {k: round(v, 3) for k, v in secs.items()}},
                "trimodal_pipeline": tri,
                "inputs": "PCM16 wav (5 s) + uint8 frame stacks [8,224,224,3] + transcription csv (64 tokens), on /dev/shm", "outputs": f"{nfiles} .npy files (UTT)",
                "drivers": "extract.audio / visual / text: device_preprocess, 8 read-ahead threads (frame stacks read straight into pinned memory), uploads on a side stream, "
                           "pinned async D2H + worker-thread .npy writes; text: the tokenizer the reference loads (AutoTokenizer, use_fast=False), its Rust backend called directly "
                           "when a probe shows the per-sentence ids unchanged (extract.text.batch_encoder), chunk by chunk on a worker thread ahead of the GPU loop",
                "feeding_thread_ms": "wall time of the driver's GPU-feeding thread per stage (extract.pipeline.span): read_wait = blocked on the read-ahead threads, stage = batch "
                                     "assembly + upload, forward = the encoder call (asynchronous launches), submit = hand-over to the writer, drain = waiting for the writer at the end",
                "byte_identical_to_sync_path": bool(same)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    args = parse()
    if args.e2e_cold_child:
        return e2e_cold_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dist = None
    rccl_ranks = None
    json_out = sys.stdout
    if world > 1 or args.force_dist:
        # RCCL prints a version banner on C-level stdout (flushed at exit: it lands BEHIND the JSON line): the line the driver parses keeps
        # the process's real stdout, everything else written to fd 1 goes to stderr
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)   # --force-dist: a one-rank RCCL group on a 1-GPU box, so that the N > 1 code path (exchange on the
        import torch.distributed as dist   # side stream, max-over-ranks reductions) can be exercised where no second GPU exists
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                     # RCCL saw this many ranks (the driver's line carries it)
        rccl_ranks = int(ones.item())
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    mods = set(args.modalities)
    B = args.batch
    cfgset = CONFIGS[args.config]
    headline = args.config == "base" and mods == set("avt")
    r = measure(args, args.config, args.steps, args.warmup, dev, dist, rank, world, sustain_s=0.0 if args.no_sustained else args.sustain_seconds,
                want_roofline=not args.no_roofline, want_parity=not args.no_parity)
    parity = r.get("parity")

    # BASELINE.json configs[4] (the large trio) rides along in the headline line as a sub-object: its own short timed region, roofline
    # fraction and parity.  Served with f16 MFMA operands (same MFMA rate as bf16 on gfx950, three more mantissa bits: DESIGN.md §4).
    large = None
    if headline and not args.no_large:
        try:
            lr = measure(args, "large", max(2, min(args.steps, 4)), 1, dev, dist, rank, world, want_roofline=not args.no_roofline, want_parity=not args.no_parity)
            large = {"workload": CONFIGS["large"]["workload"], "value": round(lr["value"], 2), "unit": "clips/s", "ms_per_step": round(lr["ms_per_step"], 3),
                     "clips_per_gpu_per_step": B, "dtype": args.dtype, "precision": args.precision, "gflop_per_clip": lr["gflop_per_clip"],
                     "whole_step_tflops": round(lr["whole_step_tflops"], 2), "whole_step_frac": round(lr["whole_step_tflops"] / PEAK_F16_TFLOPS, 4),
                     "parity": lr.get("parity"), "parity_detail": lr.get("parity_detail")}
            if lr.get("roofline"):
                rf = lr["roofline"]
                large["roofline"] = {k: rf[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches", "share_of_gpu_time")}
                large["roofline"]["other_kernels"] = {k: v for k, v in rf["other_kernels"].items() if v["ms_share"] >= 0.02}
            if lr.get("oracle_secs"):
                large["cpu_baseline"] = {"value": round(1.0 / sum(lr["oracle_secs"].values()), 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
                                         "sample": "one clip per modality through oracle/encoders_ref.py (fp32 torch-CPU forward, batch 1)"}
        except Exception as e:   # the sub-object is a report, never a reason to lose the headline number
            large = {"error": repr(e)}

    # What the headline step costs on the OTHER rungs of the precision ladder (VERDICT r5 #6a): `from_hf()` runs a load-time self-check and
    # moves an encoder whose checkpoint does not hold 1e-3 under the one-plane default up the ladder (encoders._self_check), so the deployed
    # rate of a real checkpoint lies between `value` (no encoder escalated: what random-init weights give) and the `accurate` figure here.
    ladder = None
    if headline and not args.no_ladder and world == 1 and args.precision == "mean":
        import copy
        ladder = {"what": "the same step with EVERY encoder on that rung (a self-check moves encoders individually: mean_conv3 only exists for the audio encoder)",
                  "steps": 3, "mean": round(r["value"], 2)}
        for prec in ("mean_conv3", "mean_a2", "accurate"):
            try:
                a2 = copy.copy(args)
                a2.precision = prec
                lr = measure(a2, "base", 3, 1, dev, dist, rank, world, want_roofline=False, want_parity=False)
                ladder[prec] = round(lr["value"], 2)
            except Exception as e:
                ladder[prec] = {"error": repr(e)}
            torch.cuda.empty_cache()

    if rank == 0:
        res = {
            "metric": ("clips/sec (A+V+T feature-extract, 5s/8-frame/64-tok)" if args.config == "base" else "clips/sec (A+V+T feature-extract, large trio, 5s/16-frame/64-tok)") if mods == set("avt") else f"clips/sec ({''.join(sorted(mods))} only)",
            "value": round(r["value"], 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": cfgset["workload"],
                       "clips_per_gpu_per_step": B, "modalities": "".join(sorted(mods)), "precision": args.precision,
                       "weights": "random-init (seed 0), HF architectures", "running": r.get("running"), "streams": (3 if args.streams else 1) * max(1, args.split), "sub_batches": r["sub_batches"], "parallelism": f"clip-sharded x{world}, no data-path collective" + ("; one fused fusion-minibatch all-gather per step (side stream)" if "allgather" in r else ""),
                       "gflop_per_clip": r["gflop_per_clip"]},
            "roofline": r.get("roofline"),
            "parity": parity,
            "parity_detail": r.get("parity_detail"),
        }
        if "sustained" in r:
            res["sustained"] = r["sustained"]
        if rccl_ranks is not None:
            res["rccl_ranks"] = rccl_ranks
        if "allgather" in r:
            res["allgather"] = r["allgather"]
        if large is not None:
            res["large"] = large
        if ladder is not None:
            res["ladder"] = ladder
        if args.e2e > 0 and world == 1 and headline:
            try:
                res["e2e"] = e2e(args, dev)
                res["e2e"]["frac_of_three_stream_kernel_only"] = round(res["e2e"]["clips_per_s"] / r["value"], 3)
                if isinstance(res["e2e"].get("trimodal_pipeline"), dict) and "clips_per_s" in res["e2e"]["trimodal_pipeline"]:
                    res["e2e"]["trimodal_pipeline"]["frac_of_three_stream_kernel_only"] = round(res["e2e"]["trimodal_pipeline"]["clips_per_s"] / r["value"], 3)
            except Exception as e:
                res["e2e"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                if args.config == "base":
                    res["cpu_baseline"] = cpu_baseline()
                elif r.get("oracle_secs"):   # large trio: the oracle forward of the parity leg IS the bounded CPU sample (one clip per modality)
                    res["cpu_baseline"] = {"value": round(1.0 / sum(r["oracle_secs"].values()), 4), "unit": "clips/s", "cores": torch.get_num_threads(),
                                           "cpu_model": _cpu_model(), "kind": "port",
                                           "sample": "one clip per modality through oracle/encoders_ref.py (fp32 torch-CPU forward, batch 1), "
                                                     "seconds per clip: " + ", ".join(f"{m}={v:.1f}" for m, v in r["oracle_secs"].items())}
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), file=json_out, flush=True)
        bad = {k: v for k, v in (("base" if args.config == "base" else args.config, parity), ("large", large.get("parity") if isinstance(large, dict) else None)) if v and max(v.values()) > 1e-3}
        if bad:
            sys.exit(f"bench.py: parity vs the CPU oracle exceeds 1e-3: {bad}")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

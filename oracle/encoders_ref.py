"""CPU fp32 restatement of the three encoder forwards on the MERTools feature-extraction path.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker; the product (mertools_amd) never imports this package.

Where the arithmetic lives: the reference calls HuggingFace `transformers` (pinned 4.28.0,
MERBench/environment.yml:44; un-vendored) at
  * MERBench/feature_extraction/audio/extract_audio_huggingface.py:63,97   (HubertModel / Wav2Vec2Model)
  * MERBench/feature_extraction/visual/extract_vision_huggingface.py:85,121 (CLIPModel.get_image_features)
  * MERBench/feature_extraction/text/extract_text_huggingface.py:189,225   (RobertaModel / BertModel)
This file restates the published forward of those classes from a plain state_dict (HF key names)
with torch CPU ops only; `HF:` citations point into transformers/models/ of the installed
5.15.0 package, which tests/test_oracle_pin.py runs side by side with this file (same weights,
same inputs) to pin the restatement.  Parity status: pinned against the live HF classes and the
committed golden vectors (tests/golden/); the reference itself ships no golden vectors for this
path.
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _gelu(x):
    return F.gelu(x)  # exact erf form (HF ACT2FN["gelu"])


def _gelu_new(x):
    """HF ACT2FN["gelu_new"] (NewGELUActivation): the tanh approximation."""
    import math
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)  # HF QuickGELUActivation


def _mhsa(x, wq, bq, wk, bk, wv, bv, wo, bo, heads, key_mask=None, bias=None):
    """softmax((x Wq^T + bq)(x Wk^T + bk)^T / sqrt(d)) (x Wv^T + bv) Wo^T + bo  — HF eager_attention_forward
    (HF:hubert/modeling_hubert.py:236-259; same in clip/roberta).  key_mask: bool [B,T], True = keep."""
    B, T, D = x.shape
    d = D // heads
    q = F.linear(x, wq, bq).view(B, T, heads, d).transpose(1, 2)
    k = F.linear(x, wk, bk).view(B, T, heads, d).transpose(1, 2)
    v = F.linear(x, wv, bv).view(B, T, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5)
    if bias is not None:      # additive score bias [B or 1, H, T, T] (WavLM gated relative position bias, BEiT relative position bias)
        s = s + bias
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, wo, bo)


# ------------------------------------------------------------------------------------------------
# HuBERT / wav2vec2  (HF:hubert/modeling_hubert.py)
# ------------------------------------------------------------------------------------------------
def pos_conv_weight(sd):
    """Fold the weight-norm parametrisation of encoder.pos_conv_embed.conv (dim=2)
    (HF:hubert/modeling_hubert.py:56-80): w = g * v / ||v||_{dims 0,1}."""
    p = "encoder.pos_conv_embed.conv."
    if p + "weight" in sd:
        return sd[p + "weight"]
    if p + "parametrizations.weight.original0" in sd:
        g, v = sd[p + "parametrizations.weight.original0"], sd[p + "parametrizations.weight.original1"]
    else:
        g, v = sd[p + "weight_g"], sd[p + "weight_v"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def wavlm_position_bias(rel_attn_embed, T, num_buckets=320, max_distance=800):
    """WavLMAttention.compute_bias / _relative_positions_bucket (HF:wavlm/modeling_wavlm.py): T5-style bidirectional log
    buckets of (key - query), embedded per head -> [H, T, T].  Same torch ops in the same order as HF (bucket edges are
    float-log comparisons)."""
    import math
    ctx = torch.arange(T, dtype=torch.long)[:, None]
    mem = torch.arange(T, dtype=torch.long)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    buckets = (rel > 0).to(torch.long) * nb
    rel = torch.abs(rel)
    max_exact = nb // 2
    is_small = rel < max_exact
    large = torch.log(rel.float() / max_exact)
    large = large / math.log(max_distance / max_exact)
    large = large * (nb - max_exact)
    large = (max_exact + large).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    buckets = buckets + torch.where(is_small, rel, large)
    return F.embedding(buckets, rel_attn_embed).permute(2, 0, 1)


def wavlm_gate(x, w, b, const, heads):
    """gate[b,h,t] of WavLMAttention.forward steps 1-3: 8-way linear on each head's 64-wide slice of the attention input, two
    groups of four summed, sigmoids, gate_a * (gate_b * const - 1) + 2."""
    B, T, D = x.shape
    g = x.view(B, T, heads, D // heads).permute(0, 2, 1, 3)
    proj = F.linear(g, w, b).view(B, heads, T, 2, 4).sum(-1)
    ga, gb = torch.sigmoid(proj).chunk(2, dim=-1)
    return ga * (gb * const - 1.0) + 2.0          # [B, H, T, 1]


def hubert_hidden_states(sd, cfg, wav):
    """`model(input_values, output_hidden_states=True).hidden_states`
    (extract_audio_huggingface.py:97).  wav: [B, L] fp32.  Returns list of layers+1 tensors [B,T,D]."""
    eps = cfg.get("layer_norm_eps", 1e-5)
    n_conv = len(cfg["conv_kernel"])
    x = wav[:, None, :]
    # feature encoder: HF:...:106-213
    for i in range(n_conv):
        p = f"feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[p + "conv.weight"], sd.get(p + "conv.bias"), stride=cfg["conv_stride"][i])
        if cfg["feat_extract_norm"] == "group" and i == 0:
            x = F.group_norm(x, x.shape[1], sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
        elif cfg["feat_extract_norm"] == "layer":
            x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5).transpose(1, 2)
        x = _gelu(x)
    x = x.transpose(1, 2)  # [B,T,C]
    # feature projection: HF:...:216-231
    if cfg.get("feat_proj_layer_norm", True):
        x = _ln(x, sd, "feature_projection.layer_norm", eps)
    x = F.linear(x, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    # positional conv embedding: HF:...:45-103
    K = cfg["num_conv_pos_embeddings"]
    if cfg.get("model_type") == "data2vec-audio":
        # Data2VecAudioPositionalConvEmbedding (HF:data2vec/modeling_data2vec_audio.py): num_conv_pos_embeddings layers of
        # Conv1d(D, D, conv_pos_kernel_size, padding k//2, groups) -> [drop last if k even] -> LayerNorm(no affine) -> GELU
        Kc = cfg["conv_pos_kernel_size"]
        pc = x.transpose(1, 2)
        for i in range(K):
            q = f"encoder.pos_conv_embed.layers.{i}.conv."
            pc = F.conv1d(pc, sd[q + "weight"], sd[q + "bias"], padding=Kc // 2, groups=cfg["num_conv_pos_embedding_groups"])
            if Kc % 2 == 0:
                pc = pc[:, :, :-1]
            pc = _gelu(F.layer_norm(pc.transpose(1, 2), (pc.shape[1],), None, None, 1e-5)).transpose(1, 2)
        x = x + pc.transpose(1, 2)
    else:
        pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"], padding=K // 2,
                      groups=cfg["num_conv_pos_embedding_groups"])
        if K % 2 == 0:
            pc = pc[:, :, :-1]
        x = x + _gelu(pc).transpose(1, 2)
    stable = cfg.get("do_stable_layer_norm", False)
    if not stable:
        x = _ln(x, sd, "encoder.layer_norm", eps)  # HF:...:439-441
    hs = []
    H = cfg["num_attention_heads"]
    wavlm = cfg.get("model_type") == "wavlm"
    pos_bias = None
    if wavlm:  # layer 0 owns the bucket embedding; every layer gates the same table (HF:wavlm/modeling_wavlm.py WavLMEncoder)
        pos_bias = wavlm_position_bias(sd["encoder.layers.0.attention.rel_attn_embed.weight"], x.shape[1],
                                       cfg.get("num_buckets", 320), cfg.get("max_bucket_distance", 800))
    for l in range(cfg["num_hidden_layers"]):
        hs.append(x)
        p = f"encoder.layers.{l}."
        a = p + "attention."

        def attn(inp):
            bias = None
            if wavlm:
                bias = wavlm_gate(inp, sd[a + "gru_rel_pos_linear.weight"], sd[a + "gru_rel_pos_linear.bias"], sd[a + "gru_rel_pos_const"], H) * pos_bias[None]
            return _mhsa(inp, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"], sd[a + "k_proj.weight"], sd[a + "k_proj.bias"],
                         sd[a + "v_proj.weight"], sd[a + "v_proj.bias"], sd[a + "out_proj.weight"], sd[a + "out_proj.bias"], H, bias=bias)

        def ffn(inp):
            h1 = _gelu(F.linear(inp, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"]))
            return F.linear(h1, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])

        if stable:  # HF:...:504-547 (pre-LN)
            x = x + attn(_ln(x, sd, p + "layer_norm", eps))
            x = x + ffn(_ln(x, sd, p + "final_layer_norm", eps))
        else:       # HF:...:371-404 (post-LN)
            x = _ln(x + attn(x), sd, p + "layer_norm", eps)
            x = _ln(x + ffn(x), sd, p + "final_layer_norm", eps)
    if stable:
        x = _ln(x, sd, "encoder.layer_norm", eps)  # HF:...:612
    hs.append(x)
    return hs


def hubert_cfg_from_hf(c):
    return dict(conv_kernel=list(c.conv_kernel), conv_stride=list(c.conv_stride), conv_dim=list(c.conv_dim),
                feat_extract_norm=c.feat_extract_norm, feat_proj_layer_norm=getattr(c, "feat_proj_layer_norm", True),
                num_conv_pos_embeddings=c.num_conv_pos_embeddings, num_conv_pos_embedding_groups=c.num_conv_pos_embedding_groups,
                do_stable_layer_norm=c.do_stable_layer_norm, num_attention_heads=c.num_attention_heads,
                num_hidden_layers=c.num_hidden_layers, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                layer_norm_eps=c.layer_norm_eps, conv_bias=c.conv_bias)


# ------------------------------------------------------------------------------------------------
# CLIP vision tower + projection  (HF:clip/modeling_clip.py)
# ------------------------------------------------------------------------------------------------
def clip_image_features(sd, cfg, pixel_values):
    """`model.get_image_features(pixel_values)` with transformers-4.28 semantics (a tensor)
    (extract_vision_huggingface.py:121).  pixel_values [N,3,S,S] -> [N, projection_dim]."""
    eps = cfg.get("layer_norm_eps", 1e-5)
    v = "vision_model."
    P = cfg["patch_size"]
    x = F.conv2d(pixel_values, sd[v + "embeddings.patch_embedding.weight"], None, stride=P)  # HF:...:138-217
    N, D = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    cls = sd[v + "embeddings.class_embedding"].expand(N, 1, D)
    x = torch.cat([cls, x], dim=1) + sd[v + "embeddings.position_embedding.weight"][None]
    x = _ln(x, sd, v + "pre_layrnorm", eps)
    H = cfg["num_attention_heads"]
    act = _quick_gelu if cfg.get("hidden_act", "quick_gelu") == "quick_gelu" else _gelu
    for l in range(cfg["num_hidden_layers"]):  # HF:...:353-384 (pre-LN)
        p = f"{v}encoder.layers.{l}."
        a = p + "self_attn."
        h = _ln(x, sd, p + "layer_norm1", eps)
        x = x + _mhsa(h, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"], sd[a + "k_proj.weight"], sd[a + "k_proj.bias"],
                      sd[a + "v_proj.weight"], sd[a + "v_proj.bias"], sd[a + "out_proj.weight"], sd[a + "out_proj.bias"], H)
        h = _ln(x, sd, p + "layer_norm2", eps)
        h = act(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    pooled = _ln(x[:, 0], sd, v + "post_layernorm", eps)  # HF:...:719-748
    return F.linear(pooled, sd["visual_projection.weight"])


def clip_cfg_from_hf(c):
    vc = c.vision_config
    return dict(patch_size=vc.patch_size, image_size=vc.image_size, hidden_size=vc.hidden_size,
                intermediate_size=vc.intermediate_size, num_hidden_layers=vc.num_hidden_layers,
                num_attention_heads=vc.num_attention_heads, layer_norm_eps=vc.layer_norm_eps, hidden_act=vc.hidden_act,
                projection_dim=c.projection_dim, num_channels=vc.num_channels)


# ------------------------------------------------------------------------------------------------
# DINOv2  (HF:dinov2/modeling_dinov2.py; reference branch extract_vision_huggingface.py:133-144)
# ------------------------------------------------------------------------------------------------
def dinov2_pos_embed(sd, cfg, height, width):
    """Dinov2Embeddings.interpolate_pos_encoding (installed HF 5.x form: size-based bicubic, align_corners=False)."""
    pos = sd["embeddings.position_embeddings"]
    P = cfg["patch_size"]
    n_pos = pos.shape[1] - 1
    gh, gw = height // P, width // P
    if gh * gw == n_pos and height == width:
        return pos
    s0 = int(n_pos ** 0.5)
    D = pos.shape[-1]
    pp = pos[:, 1:].reshape(1, s0, s0, D).permute(0, 3, 1, 2)
    pp = F.interpolate(pp.float(), size=(gh, gw), mode="bicubic", align_corners=False)
    return torch.cat([pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, D)], dim=1)


def dinov2_hidden_states(sd, cfg, pixel_values):
    """`model(batch, output_hidden_states=True).hidden_states` (extract_vision_huggingface.py:141): embeddings output + the
    residual stream after every layer (the final `layernorm` is NOT applied to these).  pixel_values [N,3,H,W]."""
    eps = cfg.get("layer_norm_eps", 1e-6)
    P = cfg["patch_size"]
    x = F.conv2d(pixel_values, sd["embeddings.patch_embeddings.projection.weight"], sd["embeddings.patch_embeddings.projection.bias"], stride=P)
    N, D = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["embeddings.cls_token"].expand(N, 1, D), x], dim=1)
    x = x + dinov2_pos_embed(sd, cfg, pixel_values.shape[2], pixel_values.shape[3])
    H = cfg["num_attention_heads"]
    hs = [x]
    for l in range(cfg["num_hidden_layers"]):  # Dinov2Layer: pre-LN, layer scale on both branches
        p = f"encoder.layer.{l}."
        a = p + "attention.attention."
        h = _ln(x, sd, p + "norm1", eps)
        att = _mhsa(h, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], sd[a + "key.bias"], sd[a + "value.weight"],
                    sd[a + "value.bias"], sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"], H)
        x = x + att * sd[p + "layer_scale1.lambda1"]
        if p + "mlp.weights_in.weight" in sd:   # Dinov2SwiGLUFFN (dinov2-giant)
            y1, y2 = F.linear(_ln(x, sd, p + "norm2", eps), sd[p + "mlp.weights_in.weight"], sd[p + "mlp.weights_in.bias"]).chunk(2, dim=-1)
            ff = F.linear(F.silu(y1) * y2, sd[p + "mlp.weights_out.weight"], sd[p + "mlp.weights_out.bias"])
        else:
            h = _gelu(F.linear(_ln(x, sd, p + "norm2", eps), sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
            ff = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + ff * sd[p + "layer_scale2.lambda1"]
        hs.append(x)
    return hs


def dinov2_frame_features(sd, cfg, pixel_values):
    """`torch.stack(hidden_states)[-1].sum(dim=1)` (extract_vision_huggingface.py:142): token SUM (CLS + patches) of the last
    residual stream -> [N, D]."""
    return dinov2_hidden_states(sd, cfg, pixel_values)[-1].sum(dim=1)


# ------------------------------------------------------------------------------------------------
# data2vec-vision / BEiT  (HF:data2vec/modeling_data2vec_vision.py; reference branch extract_vision_huggingface.py:123-131)
# ------------------------------------------------------------------------------------------------
def beit_relative_position_bias(table, window):
    """Data2VecVisionRelativePositionBias at the native window size: [(2w-1)^2 + 3, H] table -> [H, 1+w*w, 1+w*w] (the three
    extra rows are cls->token, token->cls, cls->cls)."""
    w = window
    nrel = (2 * w - 1) * (2 * w - 1) + 3
    coords = torch.stack(torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += w - 1
    rel[:, :, 1] += w - 1
    rel[:, :, 0] *= 2 * w - 1
    idx = torch.zeros((w * w + 1,) * 2, dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = nrel - 3
    idx[0:, 0] = nrel - 2
    idx[0, 0] = nrel - 1
    return table[idx.view(-1)].view(w * w + 1, w * w + 1, -1).permute(2, 0, 1).contiguous()


def data2vec_vision_hidden_states(sd, cfg, pixel_values):
    """`model(batch, output_hidden_states=True).hidden_states` (extract_vision_huggingface.py:130) at the native resolution:
    embeddings output + the residual stream after every layer."""
    eps = cfg.get("layer_norm_eps", 1e-12)
    P = cfg["patch_size"]
    x = F.conv2d(pixel_values, sd["embeddings.patch_embeddings.projection.weight"], sd["embeddings.patch_embeddings.projection.bias"], stride=P)
    N, D = x.shape[0], x.shape[1]
    w = pixel_values.shape[2] // P
    x = torch.cat([sd["embeddings.cls_token"].expand(N, 1, D), x.flatten(2).transpose(1, 2)], dim=1)
    if "embeddings.position_embeddings" in sd:
        x = x + sd["embeddings.position_embeddings"]
    H = cfg["num_attention_heads"]
    shared = sd.get("encoder.relative_position_bias.relative_position_bias_table")
    shared = beit_relative_position_bias(shared, w) if shared is not None else None
    hs = [x]
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        a = p + "attention.attention."
        bias = None
        own = sd.get(a + "relative_position_bias.relative_position_bias_table")
        if own is not None:
            bias = beit_relative_position_bias(own, w)[None]
        if shared is not None:
            bias = shared[None] if bias is None else bias + shared[None]
        h = _ln(x, sd, p + "layernorm_before", eps)
        att = _mhsa(h, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], None, sd[a + "value.weight"], sd[a + "value.bias"],
                    sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"], H, bias=bias)
        x = x + (att * sd[p + "lambda_1"] if p + "lambda_1" in sd else att)
        h = F.linear(_gelu(F.linear(_ln(x, sd, p + "layernorm_after", eps), sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"])),
                     sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = x + (h * sd[p + "lambda_2"] if p + "lambda_2" in sd else h)
        hs.append(x)
    return hs


def data2vec_vision_frame_features(sd, cfg, pixel_values):
    """`torch.stack(hidden_states)[-1].sum(dim=1)` (extract_vision_huggingface.py:131) -> [N, D]."""
    return data2vec_vision_hidden_states(sd, cfg, pixel_values)[-1].sum(dim=1)


# ------------------------------------------------------------------------------------------------
# VideoMAE  (HF:videomae/modeling_videomae.py)
# ------------------------------------------------------------------------------------------------
def videomae_sinusoid(n_position, d_hid):
    """HF:videomae/modeling_videomae.py:80-91 (float64 numpy table cast to float32)."""
    import numpy as np
    tab = np.array([[pos / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)] for pos in range(n_position)])
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.FloatTensor(tab)


def videomae_last_hidden_state(sd, cfg, pixel_values):
    """`model(inputs).last_hidden_state` (extract_vision_huggingface.py:155).  pixel_values [B,F,C,H,W]."""
    eps = cfg.get("layer_norm_eps", 1e-12)
    P, ts = cfg["patch_size"], cfg["tubelet_size"]
    x = F.conv3d(pixel_values.permute(0, 2, 1, 3, 4), sd["embeddings.patch_embeddings.projection.weight"],
                 sd["embeddings.patch_embeddings.projection.bias"], stride=(ts, P, P)).flatten(2).transpose(1, 2)
    x = x + videomae_sinusoid(x.shape[1], x.shape[2])[None]
    H = cfg["num_attention_heads"]
    for l in range(cfg["num_hidden_layers"]):  # HF:...:326-358 (pre-LN)
        p = f"encoder.layer.{l}."
        a = p + "attention.attention."
        h = _ln(x, sd, p + "layernorm_before", eps)
        x = x + _mhsa(h, sd[a + "query.weight"], sd.get(a + "query.bias"), sd[a + "key.weight"], sd.get(a + "key.bias"),
                      sd[a + "value.weight"], sd.get(a + "value.bias"), sd[p + "attention.output.dense.weight"],
                      sd[p + "attention.output.dense.bias"], H)
        h = _gelu(F.linear(_ln(x, sd, p + "layernorm_after", eps), sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        x = x + F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    if "layernorm.weight" in sd:
        x = _ln(x, sd, "layernorm", eps)
    return x


# ------------------------------------------------------------------------------------------------
# BERT / RoBERTa  (HF:roberta/modeling_roberta.py, HF:bert/modeling_bert.py)
# ------------------------------------------------------------------------------------------------
def bert_hidden_states(sd, cfg, input_ids, attention_mask=None, token_type_ids=None):
    """`model(**inputs, output_hidden_states=True).hidden_states` (extract_text_huggingface.py:225)."""
    eps = cfg.get("layer_norm_eps", 1e-12)
    B, T = input_ids.shape
    if cfg.get("roberta", False):  # HF:roberta/modeling_roberta.py:56-155 create_position_ids_from_input_ids
        pad = cfg["pad_token_id"]
        m = (input_ids != pad).long()
        pos_ids = torch.cumsum(m, dim=1) * m + pad
    else:
        pos_ids = torch.arange(T)[None].expand(B, T)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    x = sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.token_type_embeddings.weight"][token_type_ids]
    x = x + sd["embeddings.position_embeddings.weight"][pos_ids]
    x = _ln(x, sd, "embeddings.LayerNorm", eps)
    key_mask = attention_mask.bool() if attention_mask is not None else None
    H = cfg["num_attention_heads"]
    act = _gelu_new if cfg.get("hidden_act", "gelu") == "gelu_new" else _gelu
    if "encoder.embedding_hidden_mapping_in.weight" in sd:
        # ALBERT (HF:albert/modeling_albert.py): factorised embeddings, then ONE block applied num_hidden_layers times;
        # hidden_states[0] is the projected embedding
        x = F.linear(x, sd["encoder.embedding_hidden_mapping_in.weight"], sd["encoder.embedding_hidden_mapping_in.bias"])
        hs = [x]
        p = "encoder.albert_layer_groups.0.albert_layers.0."
        a = p + "attention."
        for _ in range(cfg["num_hidden_layers"]):
            att = _mhsa(x, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], sd[a + "key.bias"], sd[a + "value.weight"],
                        sd[a + "value.bias"], sd[a + "dense.weight"], sd[a + "dense.bias"], H, key_mask)
            x = _ln(x + att, sd, a + "LayerNorm", eps)
            h = act(F.linear(x, sd[p + "ffn.weight"], sd[p + "ffn.bias"]))
            x = _ln(x + F.linear(h, sd[p + "ffn_output.weight"], sd[p + "ffn_output.bias"]), sd, p + "full_layer_layer_norm", eps)
            hs.append(x)
        return hs
    if "embeddings_project.weight" in sd:  # ELECTRA with embedding_size != hidden_size (HF:electra/modeling_electra.py ElectraModel.forward)
        x = F.linear(x, sd["embeddings_project.weight"], sd["embeddings_project.bias"])
    hs = [x]
    for l in range(cfg["num_hidden_layers"]):  # HF:roberta/modeling_roberta.py:186-464 (post-LN)
        p = f"encoder.layer.{l}."
        a = p + "attention.self."
        att = _mhsa(x, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], sd[a + "key.bias"],
                    sd[a + "value.weight"], sd[a + "value.bias"], sd[p + "attention.output.dense.weight"],
                    sd[p + "attention.output.dense.bias"], H, key_mask)
        x = _ln(x + att, sd, p + "attention.output.LayerNorm", eps)
        h = act(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        x = _ln(x + F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]), sd, p + "output.LayerNorm", eps)
        hs.append(x)
    return hs


def bert_cfg_from_hf(c):
    return dict(roberta=(c.model_type in ("roberta", "xlm-roberta")), pad_token_id=c.pad_token_id,
                num_attention_heads=c.num_attention_heads, num_hidden_layers=c.num_hidden_layers,
                hidden_size=c.hidden_size, intermediate_size=c.intermediate_size, layer_norm_eps=c.layer_norm_eps,
                vocab_size=c.vocab_size, max_position_embeddings=c.max_position_embeddings,
                type_vocab_size=c.type_vocab_size, hidden_act=c.hidden_act)


# ------------------------------------------------------------------------------------------------
# Whisper  (HF:whisper/modeling_whisper.py; reference branch extract_audio_huggingface.py:82-90)
# ------------------------------------------------------------------------------------------------
def _whisper_attn(xq, xkv, sd, p, heads, causal=False):
    """WhisperAttention: q/v/out with bias, k without; q scaled by head_dim**-0.5; optional causal mask (decoder self-attention)."""
    B, Tq, D = xq.shape
    Tk = xkv.shape[1]
    d = D // heads
    q = (F.linear(xq, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * d ** -0.5).view(B, Tq, heads, d).transpose(1, 2)
    k = F.linear(xkv, sd[p + "k_proj.weight"]).view(B, Tk, heads, d).transpose(1, 2)
    v = F.linear(xkv, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, Tk, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3))
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), diagonal=1), float("-inf"))
    o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, Tq, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def whisper_encoder(sd, cfg, input_features):
    """WhisperEncoder.forward: input_features [B, n_mels, 2*max_source_positions] -> [B, max_source_positions, D]."""
    x = _gelu(F.conv1d(input_features, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
    x = _gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd["encoder.embed_positions.weight"]
    H = cfg["encoder_attention_heads"]
    for l in range(cfg["encoder_layers"]):
        p = f"encoder.layers.{l}."
        x = x + _whisper_attn(_ln(x, sd, p + "self_attn_layer_norm", 1e-5), _ln(x, sd, p + "self_attn_layer_norm", 1e-5), sd, p + "self_attn.", H)
        h = _gelu(F.linear(_ln(x, sd, p + "final_layer_norm", 1e-5), sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        x = x + F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    return _ln(x, sd, "encoder.layer_norm", 1e-5)


def whisper_last_hidden_state(sd, cfg, input_features, decoder_input_ids):
    """`model(input_features, decoder_input_ids=decoder_input_ids).last_hidden_state` (extract_audio_huggingface.py:87): the
    DECODER's final hidden states [B, T_dec, D] (the reference feeds two start tokens and saves the (2, D) result)."""
    enc = whisper_encoder(sd, cfg, input_features)
    Td = decoder_input_ids.shape[1]
    x = sd["decoder.embed_tokens.weight"][decoder_input_ids] + sd["decoder.embed_positions.weight"][:Td]
    H = cfg["decoder_attention_heads"]
    for l in range(cfg["decoder_layers"]):
        p = f"decoder.layers.{l}."
        h = _ln(x, sd, p + "self_attn_layer_norm", 1e-5)
        x = x + _whisper_attn(h, h, sd, p + "self_attn.", H, causal=True)
        x = x + _whisper_attn(_ln(x, sd, p + "encoder_attn_layer_norm", 1e-5), enc, sd, p + "encoder_attn.", H)
        h = _gelu(F.linear(_ln(x, sd, p + "final_layer_norm", 1e-5), sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        x = x + F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    return _ln(x, sd, "decoder.layer_norm", 1e-5)

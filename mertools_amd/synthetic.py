"""Synthetic checkpoints (HF state_dict key names) and the seeded synthetic inputs of SURVEY.md §8(d).

Used by bench.py, the parity tests and smoke().  There is no network and no pretrained weight on
disk, so throughput and parity are measured on random weights of the exact
architectures.  Unlike the HF default init
(all biases 0, LayerNorm gamma 1 / beta 0) every bias and affine parameter here is non-trivial, so
a kernel that drops a bias or swaps gamma/beta cannot pass.
"""
import math
from types import SimpleNamespace

import torch


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _lin(g, out_f, in_f, std=None):
    std = std if std is not None else 1.0 / math.sqrt(in_f)
    return torch.randn(out_f, in_f, generator=g) * std, torch.randn(out_f, generator=g) * 0.05


def _ln(g, d):
    return 1.0 + 0.1 * torch.randn(d, generator=g), 0.05 * torch.randn(d, generator=g)


def hubert_config(size="base", **over):
    base = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2),
                feat_extract_norm="group", conv_bias=False, feat_proj_layer_norm=True, num_conv_pos_embeddings=128,
                num_conv_pos_embedding_groups=16, do_stable_layer_norm=False, layer_norm_eps=1e-5, model_type="hubert")
    if size == "large":
        base.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                    feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    if size == "tiny":
        base.update(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, conv_dim=(64,) * 7,
                    num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    base.update(over)
    return SimpleNamespace(**base)


def data2vec_audio_config(size="base", **over):
    """Data2VecAudioConfig: every conv layer is followed by LayerNorm, no conv bias, 5 positional conv layers of kernel 19."""
    c = hubert_config(size if size != "base" else "base", feat_extract_norm="layer", conv_bias=False, do_stable_layer_norm=False,
                      num_conv_pos_embeddings=5, conv_pos_kernel_size=19, model_type="data2vec-audio")
    if size == "tiny":
        vars(c).update(num_conv_pos_embedding_groups=4)
    vars(c).update(over)
    return c


def wavlm_config(size="base", **over):
    """WavLMConfig: HuBERT/wav2vec2 wiring + gated relative position bias (320 buckets, max distance 800) in every layer."""
    c = hubert_config(size, model_type="wavlm", num_buckets=320, max_bucket_distance=800)
    vars(c).update(over)
    return c


def hubert_state_dict(cfg, seed=0):
    g = _g(seed)
    sd = {}
    C = cfg.conv_dim[0]
    for i, k in enumerate(cfg.conv_kernel):
        cin = 1 if i == 0 else C
        sd[f"feature_extractor.conv_layers.{i}.conv.weight"] = torch.randn(C, cin, k, generator=g) * math.sqrt(2.0 / (cin * k))
        if cfg.conv_bias:
            sd[f"feature_extractor.conv_layers.{i}.conv.bias"] = torch.randn(C, generator=g) * 0.05
        if (cfg.feat_extract_norm == "group" and i == 0) or cfg.feat_extract_norm == "layer":
            w, b = _ln(g, C)
            sd[f"feature_extractor.conv_layers.{i}.layer_norm.weight"], sd[f"feature_extractor.conv_layers.{i}.layer_norm.bias"] = w, b
    D = cfg.hidden_size
    if cfg.feat_proj_layer_norm:
        sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = _ln(g, C)
    sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"] = _lin(g, D, C)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    if getattr(cfg, "model_type", "hubert") == "data2vec-audio":
        Kc = cfg.conv_pos_kernel_size
        for i in range(K):
            sd[f"encoder.pos_conv_embed.layers.{i}.conv.weight"] = torch.randn(D, D // G, Kc, generator=g) * math.sqrt(2.0 / (Kc * D // G))
            sd[f"encoder.pos_conv_embed.layers.{i}.conv.bias"] = torch.randn(D, generator=g) * 0.05
    else:
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = torch.randn(D, D // G, K, generator=g) * math.sqrt(1.0 / (K * D // G))
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = 0.5 + torch.rand(1, 1, K, generator=g)
        sd["encoder.pos_conv_embed.conv.bias"] = torch.randn(D, generator=g) * 0.05
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = _ln(g, D)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"attention.{n}.weight"], sd[p + f"attention.{n}.bias"] = _lin(g, D, D)
        sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"] = _ln(g, D)
        sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"] = _lin(g, cfg.intermediate_size, D)
        sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"] = _lin(g, D, cfg.intermediate_size)
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = _ln(g, D)
        if getattr(cfg, "model_type", "hubert") == "wavlm":
            dh = D // cfg.num_attention_heads
            sd[p + "attention.gru_rel_pos_const"] = 0.5 + torch.rand(1, cfg.num_attention_heads, 1, 1, generator=g)
            sd[p + "attention.gru_rel_pos_linear.weight"] = torch.randn(8, dh, generator=g) / math.sqrt(dh)
            sd[p + "attention.gru_rel_pos_linear.bias"] = torch.randn(8, generator=g) * 0.1
            if l == 0:
                sd[p + "attention.rel_attn_embed.weight"] = torch.randn(cfg.num_buckets, cfg.num_attention_heads, generator=g) * 0.5
    return sd


def clip_config(size="base16", **over):
    v = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, patch_size=16,
             image_size=224, num_channels=3, layer_norm_eps=1e-5, hidden_act="quick_gelu")
    proj = 512
    if size == "large14":
        v.update(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, patch_size=14)
        proj = 768
    if size == "tiny":
        v.update(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=64)
        proj = 64
    v.update({k: val for k, val in over.items() if k != "projection_dim"})
    return SimpleNamespace(vision_config=SimpleNamespace(**v), projection_dim=over.get("projection_dim", proj), model_type="clip")


def clip_state_dict(cfg, seed=0):
    g = _g(seed)
    vc = cfg.vision_config
    D, P = vc.hidden_size, vc.patch_size
    n = (vc.image_size // P) ** 2
    v = "vision_model."
    sd = {v + "embeddings.class_embedding": torch.randn(D, generator=g) * 0.5,
          v + "embeddings.patch_embedding.weight": torch.randn(D, vc.num_channels, P, P, generator=g) * math.sqrt(1.0 / (vc.num_channels * P * P)),
          v + "embeddings.position_embedding.weight": torch.randn(n + 1, D, generator=g) * 0.3}
    sd[v + "pre_layrnorm.weight"], sd[v + "pre_layrnorm.bias"] = _ln(g, D)
    sd[v + "post_layernorm.weight"], sd[v + "post_layernorm.bias"] = _ln(g, D)
    sd["visual_projection.weight"] = torch.randn(cfg.projection_dim, D, generator=g) / math.sqrt(D)
    for l in range(vc.num_hidden_layers):
        p = f"{v}encoder.layers.{l}."
        for nme in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nme}.weight"], sd[p + f"self_attn.{nme}.bias"] = _lin(g, D, D)
        sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"] = _ln(g, D)
        sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"] = _ln(g, D)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = _lin(g, vc.intermediate_size, D)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = _lin(g, D, vc.intermediate_size, std=0.5 / math.sqrt(vc.intermediate_size))
    return sd


def dinov2_config(size="base", **over):
    """Dinov2Config: image_size is the TRAINING resolution (518 -> 37x37 position grid); inputs are 224x224 crops."""
    base = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, image_size=518, patch_size=14,
                num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True, layerscale_value=1.0, use_swiglu_ffn=False,
                model_type="dinov2")
    if size == "large":
        base.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16)
    if size == "tiny":
        base.update(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=70)
    base.update(over)
    return SimpleNamespace(**base)


def dinov2_state_dict(cfg, seed=0):
    g = _g(seed)
    D, P, Cn = cfg.hidden_size, cfg.patch_size, cfg.num_channels
    n = (cfg.image_size // P) ** 2
    ffn = int(D * cfg.mlp_ratio)
    sd = {"embeddings.cls_token": torch.randn(1, 1, D, generator=g) * 0.5,
          "embeddings.mask_token": torch.zeros(1, D),
          "embeddings.position_embeddings": torch.randn(1, n + 1, D, generator=g) * 0.3,
          "embeddings.patch_embeddings.projection.weight": torch.randn(D, Cn, P, P, generator=g) / math.sqrt(Cn * P * P),
          "embeddings.patch_embeddings.projection.bias": torch.randn(D, generator=g) * 0.05}
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        for nme in ("query", "key", "value"):
            sd[p + f"attention.attention.{nme}.weight"], sd[p + f"attention.attention.{nme}.bias"] = _lin(g, D, D)
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = _lin(g, D, D)
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = _ln(g, D)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = _ln(g, D)
        sd[p + "layer_scale1.lambda1"] = 0.5 + torch.rand(D, generator=g)
        sd[p + "layer_scale2.lambda1"] = 0.5 + torch.rand(D, generator=g)
        if getattr(cfg, "use_swiglu_ffn", False):
            hf = (int(ffn * 2 / 3) + 7) // 8 * 8
            sd[p + "mlp.weights_in.weight"], sd[p + "mlp.weights_in.bias"] = _lin(g, 2 * hf, D)
            sd[p + "mlp.weights_out.weight"], sd[p + "mlp.weights_out.bias"] = _lin(g, D, hf, std=0.5 / math.sqrt(hf))
        else:
            sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = _lin(g, ffn, D)
            sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = _lin(g, D, ffn, std=0.5 / math.sqrt(ffn))
    sd["layernorm.weight"], sd["layernorm.bias"] = _ln(g, D)
    return sd


def data2vec_vision_config(size="base", **over):
    """Data2VecVisionConfig (BEiT wiring): no absolute position table, relative position bias per layer and/or shared, layer scale."""
    base = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224, patch_size=16,
                num_channels=3, layer_norm_eps=1e-12, hidden_act="gelu", use_absolute_position_embeddings=False,
                use_relative_position_bias=False, use_shared_relative_position_bias=True, layer_scale_init_value=0.1,
                use_mean_pooling=True, use_mask_token=False, model_type="data2vec-vision")
    if size == "large":
        base.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    if size == "tiny":
        base.update(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, image_size=64)
    base.update(over)
    return SimpleNamespace(**base)


def data2vec_vision_state_dict(cfg, seed=0):
    g = _g(seed)
    D, P, Cn, H = cfg.hidden_size, cfg.patch_size, cfg.num_channels, cfg.num_attention_heads
    w = cfg.image_size // P
    nrel = (2 * w - 1) * (2 * w - 1) + 3
    sd = {"embeddings.cls_token": torch.randn(1, 1, D, generator=g) * 0.5,
          "embeddings.patch_embeddings.projection.weight": torch.randn(D, Cn, P, P, generator=g) / math.sqrt(Cn * P * P),
          "embeddings.patch_embeddings.projection.bias": torch.randn(D, generator=g) * 0.05}
    if cfg.use_absolute_position_embeddings:
        sd["embeddings.position_embeddings"] = torch.randn(1, w * w + 1, D, generator=g) * 0.3
    if cfg.use_shared_relative_position_bias:
        sd["encoder.relative_position_bias.relative_position_bias_table"] = torch.randn(nrel, H, generator=g) * 0.5
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        a = p + "attention.attention."
        sd[a + "query.weight"], sd[a + "query.bias"] = _lin(g, D, D)
        sd[a + "key.weight"], _ = _lin(g, D, D)
        sd[a + "value.weight"], sd[a + "value.bias"] = _lin(g, D, D)
        if cfg.use_relative_position_bias:
            sd[a + "relative_position_bias.relative_position_bias_table"] = torch.randn(nrel, H, generator=g) * 0.5
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = _lin(g, D, D)
        sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"] = _ln(g, D)
        sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"] = _ln(g, D)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = _lin(g, cfg.intermediate_size, D)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = _lin(g, D, cfg.intermediate_size, std=0.5 / math.sqrt(cfg.intermediate_size))
        if cfg.layer_scale_init_value > 0:
            sd[p + "lambda_1"] = 0.5 + torch.rand(D, generator=g)
            sd[p + "lambda_2"] = 0.5 + torch.rand(D, generator=g)
    if cfg.use_mean_pooling:
        sd["pooler.layernorm.weight"], sd["pooler.layernorm.bias"] = _ln(g, D)
    else:
        sd["layernorm.weight"], sd["layernorm.bias"] = _ln(g, D)
    return sd


def videomae_config(size="base", **over):
    base = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224,
                patch_size=16, num_channels=3, num_frames=16, tubelet_size=2, layer_norm_eps=1e-12, use_mean_pooling=False,
                model_type="videomae")
    if size == "large":
        base.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    if size == "tiny":
        base.update(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, image_size=96, num_frames=8)
    base.update(over)
    return SimpleNamespace(**base)


def videomae_state_dict(cfg, seed=0):
    g = _g(seed)
    D, P, ts, Cn = cfg.hidden_size, cfg.patch_size, cfg.tubelet_size, cfg.num_channels
    sd = {"embeddings.patch_embeddings.projection.weight": torch.randn(D, Cn, ts, P, P, generator=g) / math.sqrt(Cn * ts * P * P),
          "embeddings.patch_embeddings.projection.bias": torch.randn(D, generator=g) * 0.05}
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        for nme in ("query", "key", "value"):
            sd[p + f"attention.attention.{nme}.weight"], sd[p + f"attention.attention.{nme}.bias"] = _lin(g, D, D)
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = _lin(g, D, D)
        sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"] = _ln(g, D)
        sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"] = _ln(g, D)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = _lin(g, cfg.intermediate_size, D)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = _lin(g, D, cfg.intermediate_size, std=0.5 / math.sqrt(cfg.intermediate_size))
    if not cfg.use_mean_pooling:
        sd["layernorm.weight"], sd["layernorm.bias"] = _ln(g, D)
    return sd


def synth_video(B, F=16, S=224, seed=1238):
    px = torch.randint(0, 256, (B, F, S, S, 3), generator=_g(seed), dtype=torch.uint8).float() / 255.0
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    return ((px - mean) / std).permute(0, 1, 4, 2, 3).contiguous()


def whisper_config(size="base", **over):
    """WhisperConfig fields the forward needs (whisper-base: 512 / 6+6 layers / 8 heads; large-v2: 1280 / 32+32 / 20 heads)."""
    base = dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8, decoder_attention_heads=8, encoder_ffn_dim=2048,
                decoder_ffn_dim=2048, num_mel_bins=80, max_source_positions=1500, max_target_positions=448, vocab_size=51865,
                decoder_start_token_id=50258, pad_token_id=50257, model_type="whisper")
    if size == "large":
        base.update(d_model=1280, encoder_layers=32, decoder_layers=32, encoder_attention_heads=20, decoder_attention_heads=20,
                    encoder_ffn_dim=5120, decoder_ffn_dim=5120)
    if size == "tiny":
        base.update(d_model=128, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=256,
                    decoder_ffn_dim=256, max_source_positions=100, max_target_positions=8, vocab_size=64, decoder_start_token_id=3, pad_token_id=2)
    base.update(over)
    return SimpleNamespace(**base)


def whisper_state_dict(cfg, seed=0):
    g = _g(seed)
    D = cfg.d_model
    sd = {"encoder.conv1.weight": torch.randn(D, cfg.num_mel_bins, 3, generator=g) * math.sqrt(1.0 / (3 * cfg.num_mel_bins)),
          "encoder.conv1.bias": torch.randn(D, generator=g) * 0.05,
          "encoder.conv2.weight": torch.randn(D, D, 3, generator=g) * math.sqrt(1.0 / (3 * D)),
          "encoder.conv2.bias": torch.randn(D, generator=g) * 0.05,
          "encoder.embed_positions.weight": torch.randn(cfg.max_source_positions, D, generator=g) * 0.3,
          "decoder.embed_tokens.weight": torch.randn(cfg.vocab_size, D, generator=g) * 0.5,
          "decoder.embed_positions.weight": torch.randn(cfg.max_target_positions, D, generator=g) * 0.3}
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = _ln(g, D)
    sd["decoder.layer_norm.weight"], sd["decoder.layer_norm.bias"] = _ln(g, D)

    def attn(p):
        for nme in ("q_proj", "v_proj", "out_proj"):
            sd[p + nme + ".weight"], sd[p + nme + ".bias"] = _lin(g, D, D)
        sd[p + "k_proj.weight"], _ = _lin(g, D, D)

    for side, nl, ffn in (("encoder", cfg.encoder_layers, cfg.encoder_ffn_dim), ("decoder", cfg.decoder_layers, cfg.decoder_ffn_dim)):
        for l in range(nl):
            p = f"{side}.layers.{l}."
            attn(p + "self_attn.")
            sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"] = _ln(g, D)
            if side == "decoder":
                attn(p + "encoder_attn.")
                sd[p + "encoder_attn_layer_norm.weight"], sd[p + "encoder_attn_layer_norm.bias"] = _ln(g, D)
            sd[p + "fc1.weight"], sd[p + "fc1.bias"] = _lin(g, ffn, D)
            sd[p + "fc2.weight"], sd[p + "fc2.bias"] = _lin(g, D, ffn, std=0.5 / math.sqrt(ffn))
            sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = _ln(g, D)
    return sd


def bert_config(size="roberta-base", **over):
    base = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, vocab_size=50265,
                max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, hidden_act="gelu",
                model_type="roberta")
    if size == "roberta-large":
        base.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    if size == "bert-base":
        base.update(vocab_size=21128, max_position_embeddings=512, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12,
                    model_type="bert")
    if size == "tiny":
        base.update(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                    max_position_embeddings=70)
    base.update(over)
    return SimpleNamespace(**base)


def electra_config(size="tiny", **over):
    """ElectraConfig: BERT blocks; embedding_size != hidden_size (electra-small: 128 -> 256) adds `embeddings_project`."""
    c = bert_config("bert-base", model_type="electra", embedding_size=128, hidden_size=256, num_hidden_layers=12, num_attention_heads=4,
                    intermediate_size=1024, vocab_size=21128)
    if size == "tiny":
        vars(c).update(embedding_size=64, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                       max_position_embeddings=70)
    vars(c).update(over)
    return c


def albert_config(size="tiny", **over):
    """AlbertConfig: 128-wide embeddings mapped to the hidden size, ONE block shared by all layers, gelu_new."""
    c = bert_config("bert-base", model_type="albert", embedding_size=128, hidden_act="gelu_new", vocab_size=30000, type_vocab_size=2,
                    num_hidden_groups=1, inner_group_num=1)
    if size == "tiny":
        vars(c).update(embedding_size=64, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                       max_position_embeddings=70)
    vars(c).update(over)
    return c


def albert_state_dict(cfg, seed=0):
    g = _g(seed)
    D, E = cfg.hidden_size, cfg.embedding_size
    sd = {"embeddings.word_embeddings.weight": torch.randn(cfg.vocab_size, E, generator=g) * 0.5,
          "embeddings.position_embeddings.weight": torch.randn(cfg.max_position_embeddings, E, generator=g) * 0.3,
          "embeddings.token_type_embeddings.weight": torch.randn(cfg.type_vocab_size, E, generator=g) * 0.3}
    sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"] = _ln(g, E)
    sd["encoder.embedding_hidden_mapping_in.weight"], sd["encoder.embedding_hidden_mapping_in.bias"] = _lin(g, D, E)
    p = "encoder.albert_layer_groups.0.albert_layers.0."
    for nme in ("query", "key", "value", "dense"):
        sd[p + f"attention.{nme}.weight"], sd[p + f"attention.{nme}.bias"] = _lin(g, D, D)
    sd[p + "attention.LayerNorm.weight"], sd[p + "attention.LayerNorm.bias"] = _ln(g, D)
    sd[p + "ffn.weight"], sd[p + "ffn.bias"] = _lin(g, cfg.intermediate_size, D)
    sd[p + "ffn_output.weight"], sd[p + "ffn_output.bias"] = _lin(g, D, cfg.intermediate_size)
    sd[p + "full_layer_layer_norm.weight"], sd[p + "full_layer_layer_norm.bias"] = _ln(g, D)
    return sd


def bert_state_dict(cfg, seed=0):
    g = _g(seed)
    D = cfg.hidden_size
    E = getattr(cfg, "embedding_size", D)     # ELECTRA: embedding_size may differ from hidden_size
    sd = {"embeddings.word_embeddings.weight": torch.randn(cfg.vocab_size, E, generator=g) * 0.5,
          "embeddings.position_embeddings.weight": torch.randn(cfg.max_position_embeddings, E, generator=g) * 0.3,
          "embeddings.token_type_embeddings.weight": torch.randn(cfg.type_vocab_size, E, generator=g) * 0.3}
    sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"] = _ln(g, E)
    if E != D:
        sd["embeddings_project.weight"], sd["embeddings_project.bias"] = _lin(g, D, E)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        for nme in ("query", "key", "value"):
            sd[p + f"attention.self.{nme}.weight"], sd[p + f"attention.self.{nme}.bias"] = _lin(g, D, D)
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = _lin(g, D, D)
        sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"] = _ln(g, D)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = _lin(g, cfg.intermediate_size, D)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = _lin(g, D, cfg.intermediate_size)
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = _ln(g, D)
    return sd


def heavy_tailed(sd, sigma=0.7, seed=99):
    """Copy of a synthetic checkpoint whose matrix weights (Linear / Conv kernels) are multiplied elementwise by a log-normal
    factor exp(sigma * n): pretrained checkpoints carry outlier weights / channels that a Gaussian init lacks, and those are
    what stresses 16-bit weight rounding and the MX-fp4 residual plane (one E8M0 scale per 32 k).  Embedding tables, biases
    and LayerNorm parameters are left alone."""
    g = _g(seed)
    out = {}
    for k, v in sd.items():
        if v.dim() >= 2 and k.endswith("weight") and "embeddings." not in k and "embedding" not in k.split(".")[-2]:
            f = torch.exp(sigma * torch.randn(v.shape, generator=g))
            out[k] = v * f / math.sqrt(math.exp(sigma * sigma * 2))   # keep the second moment (activations stay in range)
        else:
            out[k] = v
    return out


def _ln_linear_edges(sd):
    """[(LayerNorm key prefix, [weight keys of the Linear layers that read its output])] of a HuBERT / wav2vec2 (post- or pre-LN),
    RoBERTa / BERT or CLIP-vision checkpoint — the edges along which a LayerNorm channel scale can be compensated exactly.
    LayerNorms whose output is itself a saved feature (the last four hidden states of the post-LN encoders) are left out."""
    keys, edges = set(sd), []
    n = lambda fmt: sum(1 for i in range(1000) if fmt.format(i) in keys)   # noqa: E731
    if "feature_projection.layer_norm.weight" in keys:
        edges.append(("feature_projection.layer_norm", ["feature_projection.projection.weight"]))
    L = n("encoder.layers.{}.layer_norm.weight")
    if L and "encoder.layers.0.attention.q_proj.weight" in keys:   # HuBERT / wav2vec2 / WavLM
        qkv = lambda i: [f"encoder.layers.{i}.attention.{p}_proj.weight" for p in "qkv"]   # noqa: E731
        ffn = lambda i: [f"encoder.layers.{i}.feed_forward.intermediate_dense.weight"]   # noqa: E731
        stable = any(k.startswith("feature_extractor.conv_layers.1.layer_norm") for k in keys)   # the "layer"-norm front end comes with pre-LN blocks
        if stable:
            edges += [(f"encoder.layers.{i}.layer_norm", qkv(i)) for i in range(L)] + [(f"encoder.layers.{i}.final_layer_norm", ffn(i)) for i in range(L)]
        else:
            edges.append(("encoder.layer_norm", qkv(0)))
            edges += [(f"encoder.layers.{i}.layer_norm", ffn(i)) for i in range(L)]
            edges += [(f"encoder.layers.{i}.final_layer_norm", qkv(i + 1)) for i in range(L - 4)]
    L = n("encoder.layer.{}.output.LayerNorm.weight")
    if L:   # BERT / RoBERTa (post-LN)
        qkv = lambda i: [f"encoder.layer.{i}.attention.self.{p}.weight" for p in ("query", "key", "value")]   # noqa: E731
        edges.append(("embeddings.LayerNorm", qkv(0)))
        edges += [(f"encoder.layer.{i}.attention.output.LayerNorm", [f"encoder.layer.{i}.intermediate.dense.weight"]) for i in range(L)]
        edges += [(f"encoder.layer.{i}.output.LayerNorm", qkv(i + 1)) for i in range(L - 4)]
    L = n("vision_model.encoder.layers.{}.layer_norm1.weight")
    for i in range(L):   # CLIP vision tower (pre-LN): every block LayerNorm feeds Linear layers only
        pre = f"vision_model.encoder.layers.{i}."
        edges.append((pre + "layer_norm1", [pre + f"self_attn.{p}_proj.weight" for p in "qkv"]))
        edges.append((pre + "layer_norm2", [pre + "mlp.fc1.weight"]))
    return [(ln, cons) for ln, cons in edges if ln + ".weight" in keys and all(c in keys for c in cons)]


def ln_outliers(sd, channels=3, lo=30.0, hi=100.0, seed=77, compensate=True):
    """Copy of a synthetic checkpoint with ACTIVATION outliers: `channels` channels of the block LayerNorms' affine parameters (gamma
    and beta: the same channels everywhere, as in pretrained checkpoints) are scaled by 30-100x, and — compensate=True — the
    matching input columns of the Linear layers that read the LayerNorm are divided by the same factor.  What pretrained HuBERT /
    RoBERTa checkpoints look like to the kernels: a few massive channels in the 16-bit activation planes against tiny weight columns
    (the rounding residual of those columns is what the MX block scales and the mean-token correction have to get right), with the
    Linear outputs — the attention logits among them — of the unperturbed network; in the post-LN encoders the outlier channels also
    ride the residual stream into the next LayerNorm, as massive activations do.  compensate=False scales gamma only: the attention
    logits then grow by the square of the factor and the softmax turns into an arg-max that no 16-bit Q / K plane can reproduce —
    a different (ill-conditioned) network rather than an outlier test; kept for the record (scratch measurements in DESIGN.md §4)."""
    g = _g(seed)
    out = dict(sd)
    chosen = {}

    def pick(numel):
        if numel not in chosen:   # the same channels in every LayerNorm of a width
            chosen[numel] = (torch.randperm(numel, generator=g)[:channels], lo + (hi - lo) * torch.rand(channels, generator=g))
        return chosen[numel]

    if not compensate:
        for k, v in sd.items():
            if v.dim() == 1 and k.endswith("weight") and ("layer_norm" in k.lower() or "layernorm" in k.lower() or k.endswith("LayerNorm.weight")):
                idx, f = pick(v.numel())
                v = v.clone()
                v[idx] = v[idx] * f
                out[k] = v
        return out
    for ln, consumers in _ln_linear_edges(sd):
        idx, f = pick(sd[ln + ".weight"].numel())
        for suffix in (".weight", ".bias"):
            if ln + suffix in out:
                v = out[ln + suffix].clone()
                v[idx] = v[idx] * f
                out[ln + suffix] = v
        for c in consumers:
            w = out[c].clone()
            w[:, idx] = w[:, idx] / f
            out[c] = w
    return out


# ---- the seeded synthetic inputs of SURVEY.md §8(d) ----
def synth_audio(B, L=80000, seed=1234):
    wav = 0.1 * torch.randn(B, L, generator=_g(seed))
    return (wav - wav.mean(1, keepdim=True)) / torch.sqrt(wav.var(1, unbiased=False, keepdim=True) + 1e-7)


def synth_frames(N, S=224, seed=1235):
    px = torch.randint(0, 256, (N, S, S, 3), generator=_g(seed), dtype=torch.uint8).float() / 255.0
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    return ((px - mean) / std).permute(0, 3, 1, 2).contiguous()


def synth_tokens(B, T=64, vocab=50000, seed=1236, bos=0, eos=2):
    ids = torch.randint(3, vocab, (B, T), generator=_g(seed))
    ids[:, 0] = bos
    ids[:, -1] = eos
    return ids

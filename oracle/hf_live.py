"""TEST / BASELINE INFRASTRUCTURE — never imported by the product.

The reference's own arithmetic for the hot path, live: the HuggingFace classes the extractor scripts call
(`HubertModel`, `CLIPModel.get_image_features`, `RobertaModel`; eager attention, fp32, CPU) wrapped in the scripts'
post-processing —
    audio : hidden_states[-4:] summed, flattened, mean over frames      MERBench/feature_extraction/audio/extract_audio_huggingface.py:97-108
    visual: projected CLS embedding per frame, mean over the clip's frames   .../visual/extract_vision_huggingface.py:118-122,183-189
    text  : hidden_states[-4:] summed, specials stripped, mean over tokens   .../text/extract_text_huggingface.py:225-249
— built from config (the architectures of BASELINE.json's base trio) and loaded with the SAME synthetic state_dicts the
HIP encoders use.  bench.py's `cpu_baseline` leg times these (batch 1 as the reference loops, and batch 32);
tests/test_oracle_pin.py pins oracle/encoders_ref.py against the same classes.
"""
import torch


def build_base_trio(W, seed=0):
    """(hubert, clip, roberta) HF modules in eval mode carrying mertools_amd.synthetic's base checkpoints."""
    from transformers import CLIPConfig, CLIPModel, HubertConfig, HubertModel, RobertaConfig, RobertaModel
    hc, cc, bc = W.hubert_config("base"), W.clip_config("base16"), W.bert_config("roberta-base")
    hub = HubertModel(HubertConfig(attn_implementation="eager")).eval()              # defaults == hubert-base-ls960
    missing, unexpected = hub.load_state_dict(W.hubert_state_dict(hc, seed), strict=False)
    assert not unexpected and all("masked_spec_embed" in k for k in missing), (missing, unexpected)
    vc = cc.vision_config
    clip = CLIPModel(CLIPConfig(vision_config=dict(hidden_size=vc.hidden_size, intermediate_size=vc.intermediate_size,
                                                   num_hidden_layers=vc.num_hidden_layers, num_attention_heads=vc.num_attention_heads,
                                                   patch_size=vc.patch_size, image_size=vc.image_size),
                                projection_dim=cc.projection_dim, attn_implementation="eager")).eval()
    missing, unexpected = clip.load_state_dict(W.clip_state_dict(cc, seed), strict=False)
    assert not unexpected and all(not k.startswith(("vision_model.", "visual_projection.")) or "position_ids" in k for k in missing), missing
    rob = RobertaModel(RobertaConfig(vocab_size=bc.vocab_size, max_position_embeddings=bc.max_position_embeddings, type_vocab_size=bc.type_vocab_size,
                                     pad_token_id=bc.pad_token_id, layer_norm_eps=bc.layer_norm_eps, attn_implementation="eager"),
                       add_pooling_layer=False).eval()
    missing, unexpected = rob.load_state_dict(W.bert_state_dict(bc, seed), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return hub, clip, rob


@torch.no_grad()
def audio_utt(hub, wav):
    """[B, L] -> [B, D]; B = 1 is the reference's loop."""
    hs = hub(wav, output_hidden_states=True).hidden_states
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(dim=0)
    return feat.mean(1)


@torch.no_grad()
def visual_utt(clip, px, frames_per_clip=8):
    o = clip.get_image_features(px)
    o = o if torch.is_tensor(o) else o.pooler_output          # transformers >= 5 returns an output object (SURVEY §8c shim)
    return o.view(-1, frames_per_clip, o.shape[-1]).mean(1)


@torch.no_grad()
def text_utt(rob, ids):
    hs = rob(input_ids=ids, attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states
    return torch.stack(hs)[[-4, -3, -2, -1]].sum(dim=0)[:, 1:-1].mean(1)

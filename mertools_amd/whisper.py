"""Whisper branch of the reference's audio extractor (feature_extraction/audio/extract_audio_huggingface.py:79-89, WHISPER_BASE /
WHISPER_LARGE): `model(input_features, decoder_input_ids=[[start, start]]).last_hidden_state` — the DECODER's two final hidden
states, saved as a (2, D) array per clip (the reference's own quirk: the "UTTERANCE" file of a Whisper model is 2 x D).
HF:whisper/modeling_whisper.py.

Unlike the other encoders this one is orchestrated from Python over the C-ABI operators (no engine handle): the encoder's two
convolutions are implicit-im2col GEMMs over the zero-padded, time-major mel planes (mer_posconv_pack with one group and three
taps; the stride-2 conv is the same GEMM with lda = 2 D), its pre-LN blocks are mer_layernorm / mer_gemm16 / mer_attention
(1500 frames -> the streaming attention kernel).  The decoder has two tokens per clip: its own projections and feed-forward run
in exact fp32 on mer_gemm32 and its attentions on mer_small_attention; the one heavy piece, the cross-attention K/V projection
of the 1500 encoder states in every decoder layer, is a [K|V] mer_gemm16 like the encoder's.  Every FLOP goes through libmer_hip.so; torch only owns the buffers."""
import torch

from . import ops
from .encoders import EncoderOutput, _sd_of

_PASSES = {"fast": 1, "f16": 1, "balanced": 2, "mx": 4, "accurate": 3, "x3": 3}


class _W16:
    """Device planes of one [N, K] weight: f16 hi, optional f16 lo residual, optional MX-fp4 residual (passes = 4)."""

    def __init__(self, t, passes, device):
        t = t.detach().to(torch.float32).contiguous()
        hi, lo = ops.split16_host(t, "f16", passes >= 2)
        self.hi = hi.contiguous().to(device)
        self.lo = lo.contiguous().to(device) if lo is not None else None
        self.mx = None
        if passes == 4:
            packed = ops.mx_pack(t - hi.float())
            self.mx = packed.to(device) if packed is not None else None


class HipWhisperModel:
    def __init__(self, state_dict, config, device="cuda:0", precision="mx"):
        ops._lib.lib()   # fail loudly if the HIP library is missing
        sd = _sd_of(state_dict)
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        self.config, self.device = config, torch.device(device)
        self.passes = _PASSES[precision]
        D = config.d_model
        assert D % 64 == 0 and D // config.encoder_attention_heads == 64 and D // config.decoder_attention_heads == 64, "head_dim must be 64"
        assert config.num_mel_bins % 8 == 0
        dev, P = self.device, self.passes
        f32 = lambda t: t.detach().to(torch.float32).contiguous().to(dev)
        W = lambda t: _W16(t, P, dev)
        # conv weights [D, Cin, 3] -> [D, 3 * Cin] with column = tap * Cin + channel (the im2col row order of the packed planes)
        self.conv1_w, self.conv1_b = W(sd["encoder.conv1.weight"].permute(0, 2, 1).reshape(D, -1)), f32(sd["encoder.conv1.bias"])
        self.conv2_w, self.conv2_b = W(sd["encoder.conv2.weight"].permute(0, 2, 1).reshape(D, -1)), f32(sd["encoder.conv2.bias"])
        self.enc_pos = f32(sd["encoder.embed_positions.weight"])
        z = torch.zeros(D)
        self.enc_layers = []
        for l in range(config.encoder_layers):
            p = f"encoder.layers.{l}."
            a = p + "self_attn."
            self.enc_layers.append(dict(
                ln1=(f32(sd[p + "self_attn_layer_norm.weight"]), f32(sd[p + "self_attn_layer_norm.bias"])),
                wqkv=W(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0)),
                bqkv=f32(torch.cat([sd[a + "q_proj.bias"], z, sd[a + "v_proj.bias"]], 0)),
                wo=W(sd[a + "out_proj.weight"]), bo=f32(sd[a + "out_proj.bias"]),
                ln2=(f32(sd[p + "final_layer_norm.weight"]), f32(sd[p + "final_layer_norm.bias"])),
                w1=W(sd[p + "fc1.weight"]), b1=f32(sd[p + "fc1.bias"]), w2=W(sd[p + "fc2.weight"]), b2=f32(sd[p + "fc2.bias"])))
        self.enc_ln = (f32(sd["encoder.layer_norm.weight"]), f32(sd["encoder.layer_norm.bias"]))
        # decoder: exact fp32 (2 rows per clip; its cost is the cross-attention K/V projections of the encoder states)
        self.dec_tok, self.dec_pos = f32(sd["decoder.embed_tokens.weight"]), f32(sd["decoder.embed_positions.weight"])
        self.dec_layers = []
        for l in range(config.decoder_layers):
            p = f"decoder.layers.{l}."
            lay = {}
            a, c = p + "self_attn.", p + "encoder_attn."
            lay["self"] = dict(wq=f32(sd[a + "q_proj.weight"]), bq=f32(sd[a + "q_proj.bias"]), wk=f32(sd[a + "k_proj.weight"]),
                               wv=f32(sd[a + "v_proj.weight"]), bv=f32(sd[a + "v_proj.bias"]), wo=f32(sd[a + "out_proj.weight"]),
                               bo=f32(sd[a + "out_proj.bias"]))
            lay["cross"] = dict(wq=f32(sd[c + "q_proj.weight"]), bq=f32(sd[c + "q_proj.bias"]), wo=f32(sd[c + "out_proj.weight"]),
                                bo=f32(sd[c + "out_proj.bias"]))
            lay["cross_wkv"] = W(torch.cat([sd[c + "k_proj.weight"], sd[c + "v_proj.weight"]], 0))
            lay["cross_bkv"] = f32(torch.cat([z, sd[c + "v_proj.bias"]], 0))
            lay["ln_self"] = (f32(sd[p + "self_attn_layer_norm.weight"]), f32(sd[p + "self_attn_layer_norm.bias"]))
            lay["ln_cross"] = (f32(sd[p + "encoder_attn_layer_norm.weight"]), f32(sd[p + "encoder_attn_layer_norm.bias"]))
            lay["ln_ffn"] = (f32(sd[p + "final_layer_norm.weight"]), f32(sd[p + "final_layer_norm.bias"]))
            lay.update(w1=f32(sd[p + "fc1.weight"]), b1=f32(sd[p + "fc1.bias"]), w2=f32(sd[p + "fc2.weight"]), b2=f32(sd[p + "fc2.bias"]))
            self.dec_layers.append(lay)
        self.dec_ln = (f32(sd["decoder.layer_norm.weight"]), f32(sd["decoder.layer_norm.bias"]))

    @classmethod
    def from_hf(cls, hf_model, **kw):
        return cls(hf_model.state_dict(), hf_model.config, **kw)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------------------------------------------------
    def _gemm(self, a_hi, a_lo, w, **kw):
        P = self.passes
        return ops.gemm16(a_hi, w.hi, a_lo=a_lo if P == 3 else None, w_lo=w.lo, w_mx=w.mx, passes=P, dtype="f16", **kw)

    def encode(self, input_features):
        """input_features [B, n_mels, 2 * max_source_positions] -> encoder states [B * max_source_positions, D]: (fp32, f16 hi, f16 lo | None)."""
        cfg, P = self.config, self.passes
        lo = P == 3
        B, C_, L = input_features.shape
        T, D, H = cfg.max_source_positions, cfg.d_model, cfg.encoder_attention_heads
        assert L == 2 * T, f"Whisper expects {2 * T} mel frames, got {L}"
        mel = input_features.to(self.device, torch.float32).permute(0, 2, 1).contiguous()            # time-major [B, L, n_mels]
        ph, pl = ops.posconv_pack(mel, 1, 3, lo=lo)                                                   # [B, 1, L + 3, n_mels], one zero row in front
        x, _, _ = self._gemm(ph.view(-1, C_), pl.view(-1, C_) if lo else None, self.conv1_w, bias=self.conv1_b, act="gelu", out32=True,
                             M=B * L, lda=C_, a_rows_per_batch=L, a_batch_stride=(L + 3) * C_)
        ph, pl = ops.posconv_pack(x.view(B, L, D), 1, 3, lo=lo)
        x, _, _ = self._gemm(ph.view(-1, D), pl.view(-1, D) if lo else None, self.conv2_w, bias=self.conv2_b, act="gelu", out32=True,
                             M=B * T, lda=2 * D, a_rows_per_batch=T, a_batch_stride=(L + 3) * D)
        ops.add_pos(x, self.enc_pos)
        for lay in self.enc_layers:
            _, hh, hl = ops.layernorm(x, *lay["ln1"], 1e-5, out32=False, out16=True, out16_lo=lo)
            _, qkv, _ = self._gemm(hh, hl, lay["wqkv"], bias=lay["bqkv"], out16=True)
            ch, cl = ops.attention(qkv, B, T, H, 0.125, out_lo=lo)
            x, _, _ = self._gemm(ch, cl, lay["wo"], bias=lay["bo"], residual=x, out32=True)
            _, hh, hl = ops.layernorm(x, *lay["ln2"], 1e-5, out32=False, out16=True, out16_lo=lo)
            _, fh, fl = self._gemm(hh, hl, lay["w1"], bias=lay["b1"], act="gelu", out16=True, out16_lo=lo)
            x, _, _ = self._gemm(fh, fl, lay["w2"], bias=lay["b2"], residual=x, out32=True)
        return ops.layernorm(x, *self.enc_ln, 1e-5, out16=True, out16_lo=lo)

    def decode(self, enc, decoder_input_ids):
        """enc: the (fp32, hi, lo) planes of encode() -> decoder states fp32 [B, T_dec, D]."""
        cfg = self.config
        B, Td = decoder_input_ids.shape
        T, D, H = cfg.max_source_positions, cfg.d_model, cfg.decoder_attention_heads
        ids = decoder_input_ids.to(self.device)
        x = (self.dec_tok[ids] + self.dec_pos[:Td]).reshape(B * Td, D).contiguous()   # embedding gather: index plumbing, no arithmetic beyond one add
        for lay in self.dec_layers:
            h, _, _ = ops.layernorm(x, *lay["ln_self"], 1e-5)
            w = lay["self"]
            o = ops.small_attention(ops.gemm32(h, w["wq"], w["bq"]), ops.gemm32(h, w["wk"]), ops.gemm32(h, w["wv"], w["bv"]), B, Td, Td, H, 0.125, True)
            ops.gemm32(o, lay["self"]["wo"], lay["self"]["bo"], out=x, accumulate=True)
            h, _, _ = ops.layernorm(x, *lay["ln_cross"], 1e-5)
            kv, _, _ = self._gemm(enc[1], enc[2], lay["cross_wkv"], bias=lay["cross_bkv"], out32=True)       # [B * T, K | V]
            o = ops.small_attention(ops.gemm32(h, lay["cross"]["wq"], lay["cross"]["bq"]), kv[:, :D], kv[:, D:], B, Td, T, H, 0.125, False)
            ops.gemm32(o, lay["cross"]["wo"], lay["cross"]["bo"], out=x, accumulate=True)
            h, _, _ = ops.layernorm(x, *lay["ln_ffn"], 1e-5)
            f = ops.gemm32(h, lay["w1"], lay["b1"], "gelu")
            ops.gemm32(f, lay["w2"], lay["b2"], out=x, accumulate=True)
        out, _, _ = ops.layernorm(x, *self.dec_ln, 1e-5)
        return out.view(B, Td, D)

    def __call__(self, input_features=None, decoder_input_ids=None, **_):
        with torch.cuda.device(self.device):   # the op wrappers launch on the current device's current stream
            enc = self.encode(input_features)
            return EncoderOutput(last_hidden_state=self.decode(enc, decoder_input_ids))

    def extract_utterance(self, input_features):
        """The reference's per-clip output (extract_audio_huggingface.py:84-89): two decoder start tokens -> (2, D) per clip."""
        B = input_features.shape[0]
        ids = torch.full((B, 2), self.config.decoder_start_token_id, dtype=torch.long)
        return self(input_features, decoder_input_ids=ids).last_hidden_state

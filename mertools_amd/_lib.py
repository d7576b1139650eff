"""ctypes binding of libmer_hip.so (declarations mirror include/mer_hip.h one-to-one).

The product path has no CPU fallback: if the shared library is missing or a call fails, this
module raises — nothing here ever routes around the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MER_LIB_PATH: another build of the same ABI — how two library versions are A/B'd on one GPU box, scripts/gpu_ab_lib.sh)
LIB_PATH = os.environ.get("MER_LIB_PATH") or os.path.join(_HERE, "libmer_hip.so")

MER_OK = 0
MER_DT_F16, MER_DT_BF16 = 0, 1
MER_ACT_NONE, MER_ACT_GELU, MER_ACT_QUICK_GELU, MER_ACT_RELU, MER_ACT_GELU_TANH = 0, 1, 2, 3, 4
MER_MAX_CONV = 8
MER_MAX_POS = 8

c_void_p, c_int, c_ll, c_float = C.c_void_p, C.c_int, C.c_longlong, C.c_float


class MerError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int), ("dtype", c_int),
        ("a_hi", c_void_p), ("a_lo", c_void_p), ("lda", c_ll),
        ("a_rows_per_batch", c_int), ("a_batch_stride", c_ll),
        ("w_hi", c_void_p), ("w_lo", c_void_p), ("ldw", c_ll),
        ("bias", c_void_p), ("act", c_int),
        ("residual", c_void_p), ("ldr", c_ll),
        ("c32", c_void_p), ("ldc32", c_ll),
        ("c16_hi", c_void_p), ("c16_lo", c_void_p), ("ldc16", c_ll),
        ("nbatch", c_int), ("nb_inner", c_int),
        ("a_so", c_ll), ("a_si", c_ll), ("w_si", c_ll), ("bias_si", c_ll), ("c_so", c_ll), ("c_si", c_ll),
        ("passes", c_int), ("tile", c_int), ("headmajor_T", c_int), ("headmajor_H", c_int),
        ("w_mx", c_void_p),
        ("w_hi_blk", c_void_p), ("w_lo_blk", c_void_p), ("w_hi_blkp", c_void_p), ("w_hi_blkq", c_void_p),
        ("bias_seg_rows", c_int), ("bias_ld", c_ll),
    ]


class W16(C.Structure):
    _fields_ = [("hi", c_void_p), ("lo", c_void_p), ("mx", c_void_p), ("hi_blk", c_void_p), ("lo_blk", c_void_p), ("hi_blkp", c_void_p), ("hi_blkq", c_void_p)]


class TfLayer(C.Structure):
    _fields_ = [
        ("wqkv", W16), ("bqkv", c_void_p),
        ("wo", W16), ("bo", c_void_p),
        ("ln1_g", c_void_p), ("ln1_b", c_void_p),
        ("w1", W16), ("b1", c_void_p),
        ("w2", W16), ("b2", c_void_p),
        ("ln2_g", c_void_p), ("ln2_b", c_void_p),
        ("attn_bias", c_void_p), ("gru_w", c_void_p), ("gru_b", c_void_p), ("gru_const", c_void_p),
    ]


class TfConfig(C.Structure):
    _fields_ = [("hidden", c_int), ("heads", c_int), ("ffn", c_int), ("layers", c_int), ("pre_ln", c_int),
                ("act", c_int), ("ln_eps", c_float), ("dtype", c_int), ("passes", c_int), ("gated_rel_pos", c_int), ("ffn_swiglu", c_int), ("mx_skip", c_int), ("attn_f32", c_int)]


class HubertConfig(C.Structure):
    _fields_ = [("tf", TfConfig), ("n_conv", c_int), ("conv_dim", c_int),
                ("conv_kernel", c_int * MER_MAX_CONV), ("conv_stride", c_int * MER_MAX_CONV),
                ("feat_norm_group", c_int), ("conv_bias", c_int), ("feat_proj_layer_norm", c_int),
                ("pos_k", c_int), ("pos_groups", c_int), ("stable_layer_norm", c_int), ("conv_passes", c_int),
                ("pos_layers", c_int)]


class HubertWeights(C.Structure):
    _fields_ = [("conv0_w", c_void_p),
                ("conv_norm_g", c_void_p * MER_MAX_CONV), ("conv_norm_b", c_void_p * MER_MAX_CONV),
                ("conv_w", W16 * MER_MAX_CONV), ("conv_b", c_void_p * MER_MAX_CONV),
                ("fp_ln_g", c_void_p), ("fp_ln_b", c_void_p), ("fp_w", W16), ("fp_b", c_void_p),
                ("pos_w", W16), ("pos_b", c_void_p), ("enc_ln_g", c_void_p), ("enc_ln_b", c_void_p),
                ("layers", C.POINTER(TfLayer)), ("pos_ws", W16 * MER_MAX_POS), ("pos_bs", c_void_p * MER_MAX_POS)]


class VitConfig(C.Structure):
    _fields_ = [("tf", TfConfig), ("image_size", c_int), ("patch_size", c_int), ("channels", c_int), ("proj_dim", c_int),
                ("variant", c_int)]


class VitWeights(C.Structure):
    _fields_ = [("patch_w", W16), ("cls", c_void_p), ("pos", c_void_p), ("pre_ln_g", c_void_p), ("pre_ln_b", c_void_p),
                ("post_ln_g", c_void_p), ("post_ln_b", c_void_p), ("proj_w", W16), ("layers", C.POINTER(TfLayer)),
                ("patch_b", c_void_p)]


class VideoMAEConfig(C.Structure):
    _fields_ = [("tf", TfConfig), ("image_size", c_int), ("patch_size", c_int), ("channels", c_int), ("num_frames", c_int),
                ("tubelet_size", c_int), ("final_ln", c_int)]


class VideoMAEWeights(C.Structure):
    _fields_ = [("patch_w", W16), ("patch_b", c_void_p), ("pos", c_void_p), ("final_ln_g", c_void_p), ("final_ln_b", c_void_p),
                ("layers", C.POINTER(TfLayer))]


class BertConfig(C.Structure):
    _fields_ = [("tf", TfConfig), ("vocab", c_int), ("max_pos", c_int), ("type_vocab", c_int), ("pad_id", c_int),
                ("pos_mode", c_int), ("emb_ln_eps", c_float), ("emb_dim", c_int)]


class BertWeights(C.Structure):
    _fields_ = [("word", c_void_p), ("pos", c_void_p), ("type", c_void_p), ("emb_ln_g", c_void_p), ("emb_ln_b", c_void_p),
                ("layers", C.POINTER(TfLayer)), ("emb_proj_w", W16), ("emb_proj_b", c_void_p)]


# name -> (restype, argtypes); every symbol include/mer_hip.h declares must appear here.
_PROTOS = {
    "mer_version": (C.c_char_p, []),
    "mer_last_error": (C.c_char_p, []),
    "mer_target_arch": (C.c_char_p, []),
    "mer_abi_sizeof": (c_int, [C.c_char_p]),
    "mer_set_option": (c_int, [C.c_char_p, c_int]),
    "mer_get_option": (c_int, [C.c_char_p, C.POINTER(c_int)]),
    "mer_set_debug_buffer": (c_int, [c_void_p]),
    "mer_prof_enable": (c_int, [c_int]),
    "mer_prof_report": (c_int, [C.c_char_p, c_int]),
    "mer_gemm16": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "mer_mx_packed_bytes": (c_ll, [c_int, c_int]),
    "mer_mx_pack": (c_int, [c_void_p, c_ll, c_int, c_int, c_void_p]),
    "mer_w_block_bytes": (c_ll, [c_int, c_int]),
    "mer_w_block_pack": (c_int, [c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p]),
    "mer_w_block_pack_p": (c_int, [c_void_p, c_ll, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mer_gemm32": (c_int, [c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_int, c_void_p, c_ll, c_int,
                           c_int, c_int, c_int, c_void_p]),
    "mer_layernorm": (c_int, [c_void_p, c_ll, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_ll,
                              c_void_p, c_void_p, c_ll, c_int, c_void_p]),
    "mer_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int,
                              c_float, c_void_p, c_int, c_void_p]),
    "mer_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int,
                                  c_float, c_void_p, c_int, c_void_p]),
    "mer_attention_hm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_float, c_void_p,
                                 c_int, c_void_p]),
    "mer_split16": (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p]),
    "mer_hubert_conv0_gn": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_hubert_conv0_plain": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mer_posconv_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_vit_patchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_video_patchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_add_pos": (c_int, [c_void_p, c_void_p, c_ll, c_int, c_int, c_void_p]),
    "mer_vit_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_bert_embed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_sum_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p, c_int,
                             c_void_p, c_void_p]),
    "mer_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    "mer_colsum": (c_int, [c_void_p, c_int, c_int, c_ll, c_void_p, c_int, c_void_p]),
    "mer_dropout": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_ll, c_void_p]),
    "mer_fuse_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mer_fuse_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mer_ce_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mer_ce_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mer_mse_loss": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mer_mse_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float, c_float,
                              c_int, c_float, c_void_p]),
    "mer_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float, c_float,
                                  c_void_p, c_float, c_void_p]),
    "mer_inc_i32": (c_int, [c_void_p, c_void_p]),
    "mer_hubert_create": (c_int, [C.POINTER(HubertConfig), C.POINTER(HubertWeights), C.POINTER(c_void_p)]),
    "mer_hubert_destroy": (None, [c_void_p]),
    "mer_hubert_out_frames": (c_int, [c_void_p, c_int]),
    "mer_hubert_workspace_bytes": (c_ll, [c_void_p, c_int, c_int, c_int]),
    "mer_hubert_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p, c_void_p]),
    "mer_hubert_forward_bias": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p]),
    "mer_attention_cls": (c_int, [c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "mer_seq_bias_scratch_bytes": (c_ll, [c_int, c_int]),
    "mer_seq_bias": (c_int, [c_void_p, c_int, c_ll, c_int, c_ll, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_int, c_int,
                             c_void_p, c_void_p, c_ll, c_void_p]),
    "mer_hubert_forward_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p]),
    "mer_hubert_conv0_gn_ragged": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                           c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mer_hubert_valid_frames": (c_int, [c_void_p, c_int, c_int, c_int, C.POINTER(c_int), C.POINTER(c_int), c_void_p, c_void_p, c_void_p]),
    "mer_hubert_valid_frames_all": (c_int, [c_void_p, c_int, c_int, c_int, C.POINTER(c_int), C.POINTER(c_int), c_void_p, c_void_p]),
    "mer_posconv_pack_ragged": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mer_vit_create": (c_int, [C.POINTER(VitConfig), C.POINTER(VitWeights), C.POINTER(c_void_p)]),
    "mer_vit_destroy": (None, [c_void_p]),
    "mer_vit_workspace_bytes": (c_ll, [c_void_p, c_int]),
    "mer_vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_void_p]),
    "mer_vit_forward_tokens": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "mer_attention_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_float,
                                   c_void_p, c_void_p, c_ll, c_void_p, c_int, c_void_p]),
    "mer_wavlm_gate": (c_int, [c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mer_small_attention": (c_int, [c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_ll,
                                    c_void_p]),
    "mer_lstm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mer_lstm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mer_wave_normalize": (c_int, [c_void_p, c_int, c_ll, c_int, c_int, c_int, c_void_p, c_ll, c_void_p]),
    "mer_image_normalize_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mer_image_resize_crop_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mer_swiglu": (c_int, [c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mer_token_reduce": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "mer_videomae_create": (c_int, [C.POINTER(VideoMAEConfig), C.POINTER(VideoMAEWeights), C.POINTER(c_void_p)]),
    "mer_videomae_destroy": (None, [c_void_p]),
    "mer_videomae_workspace_bytes": (c_ll, [c_void_p, c_int]),
    "mer_videomae_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p]),
    "mer_bert_create": (c_int, [C.POINTER(BertConfig), C.POINTER(BertWeights), C.POINTER(c_void_p)]),
    "mer_bert_destroy": (None, [c_void_p]),
    "mer_bert_workspace_bytes": (c_ll, [c_void_p, c_int, c_int, c_int]),
    "mer_bert_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_ll, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises MerError (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MerError(
                f"{LIB_PATH} not found: build it with `python -m mertools_amd.build` "
                "(hipcc --offload-arch=gfx950). mertools_amd has no CPU fallback.")
        # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  It must be in the process first
        # so that libmer_hip.so binds to the runtime that owns torch's device memory and streams;
        # loading ours first leaves two HIP runtimes and ours sees "no ROCm-capable device".
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(h, name, None)
            if fn is None:
                continue  # checked by tests/test_abi.py against the header
            fn.restype = res
            fn.argtypes = args
        # debug switches (include/mer_hip.h: mer_set_option): MER_OPTIONS="gemm_generic_epi=1,gemm_dbg_skip=1" at load time
        for kv in filter(None, os.environ.get("MER_OPTIONS", "").split(",")):
            k, _, v = kv.partition("=")
            if h.mer_set_option(k.strip().encode(), int(v or "1")) != MER_OK:
                raise MerError(f"MER_OPTIONS: unknown option {k!r}")
        _lib = h
    return _lib


def get_option(name):
    """Current value of a mer_set_option switch (include/mer_hip.h: mer_get_option)."""
    v = c_int(0)
    check(lib().mer_get_option(name.encode() if isinstance(name, str) else name, C.byref(v)), "mer_get_option")
    return v.value


def check(rc, what=""):
    if rc != MER_OK:
        msg = lib().mer_last_error()
        raise MerError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def exported_symbols():
    return sorted(_PROTOS)

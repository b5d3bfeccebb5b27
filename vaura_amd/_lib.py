"""ctypes binding of libvaura_hip.so (include/vaura_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load,
``lib()`` raises.  (Building is ``python -m vaura_amd.csrc.build`` / ``__graft_entry__.build()``.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvaura_hip.so")

W_F32, W_BF16, W_FP8, W_H1, W_H2, W_FP8H = 0, 1, 2, 3, 4, 5
EPI_STORE, EPI_RESID, EPI_SWIGLU, EPI_GELU, EPI_LOGITS = 0, 1, 2, 3, 4
KERNEL_KINDS = ("embed", "qkv", "attn", "wo", "w13", "w2", "heads", "sample")

c_float_p = C.c_void_p  # device pointers travel as integers


class Dims(C.Structure):
    _fields_ = [("n_layer", C.c_int32), ("d_model", C.c_int32), ("n_head", C.c_int32), ("ffn_dim", C.c_int32),
                ("n_codebooks", C.c_int32), ("vocab", C.c_int32), ("cond_dim", C.c_int32), ("tok_dim", C.c_int32),
                ("cond_in", C.c_int32), ("codebook_dim", C.c_int32), ("tokens_per_frame", C.c_int32),
                ("eps", C.c_float)]


class LayerWeights(C.Structure):
    _fields_ = [("wqkv", C.c_void_p), ("wo", C.c_void_p), ("w13", C.c_void_p), ("w2", C.c_void_p),
                ("attn_norm", C.c_void_p), ("ffn_norm", C.c_void_p)]


class Sampling(C.Structure):
    _fields_ = [("use_sampling", C.c_int32), ("temp", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float),
                ("cfg_scale", C.c_float), ("seed", C.c_uint64), ("clip_base", C.c_uint64), ("input_is_probs", C.c_int32),
                ("tie_eps", C.c_float)]


class Decoder(C.Structure):
    _fields_ = [("dims", Dims), ("wdtype", C.c_int32), ("batch", C.c_int32), ("rows", C.c_int32),
                ("max_len", C.c_int32), ("timesteps", C.c_int32), ("seq_len", C.c_int32),
                ("n_cond_tokens", C.c_int32), ("prefill_positions", C.c_int32),
                ("plane_shift", C.c_int32), ("kv_dtype", C.c_int32),
                ("layers_host", C.POINTER(LayerWeights)), ("heads", C.c_void_p), ("final_norm", C.c_void_p),
                ("tok_emb", C.c_void_p), ("tok_proj_w", C.c_void_p), ("tok_proj_b", C.c_void_p), ("tok_table", C.c_void_p),
                ("empty_video", C.c_void_p), ("rope", C.c_void_p), ("cond_proj", C.c_void_p),
                ("kcache", C.c_void_p), ("vcache", C.c_void_p), ("seq", C.c_void_p), ("state", C.c_void_p),
                ("noise", C.c_void_p),
                ("ws_h", C.c_void_p), ("ws_qkv", C.c_void_p), ("ws_qkv2", C.c_void_p), ("ws_attn", C.c_void_p), ("ws_ffn", C.c_void_p),
                ("ws_logits", C.c_void_p),
                ("ws_h_split", C.c_void_p), ("ws_attn_split", C.c_void_p), ("ws_ffn_split", C.c_void_p),
                ("ws_ss", C.c_void_p), ("first_norm", C.c_void_p), ("ws_attn_part", C.c_void_p), ("ws_sync", C.c_void_p)]


class Conv(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("wscale", C.c_void_p), ("cin", C.c_int32), ("cout", C.c_int32),
                ("taps", C.c_int32), ("dilation", C.c_int32), ("stride", C.c_int32), ("_pad", C.c_int32)]


class Codec(C.Structure):
    _fields_ = [("n_codebooks", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
                ("latent_dim", C.c_int32), ("n_blocks", C.c_int32), ("n_units", C.c_int32),
                ("rates", C.c_int32 * 4),
                ("codebooks", C.c_void_p), ("out_proj_w", C.c_void_p), ("out_proj_b", C.c_void_p),
                ("conv_in", Conv), ("alpha_up", C.c_void_p * 4), ("up", Conv * 4),
                ("alpha_res", ((C.c_void_p * 2) * 3) * 4), ("res", ((Conv * 2) * 3) * 4),
                ("alpha_out", C.c_void_p), ("conv_out", Conv),
                ("ws", C.c_void_p * 4), ("ws_elems", C.c_size_t), ("precision", C.c_int32), ("_pad1", C.c_int32)]


class CodecEncoder(C.Structure):
    _fields_ = [("n_codebooks", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
                ("latent_dim", C.c_int32), ("n_blocks", C.c_int32), ("n_units", C.c_int32),
                ("rates", C.c_int32 * 4), ("enc_dim", C.c_int32), ("_pad0", C.c_int32),
                ("conv_in_w", C.c_void_p), ("conv_in_b", C.c_void_p),
                ("alpha_res", ((C.c_void_p * 2) * 3) * 4), ("res", ((Conv * 2) * 3) * 4),
                ("alpha_down", C.c_void_p * 4), ("down", Conv * 4),
                ("alpha_out", C.c_void_p), ("conv_out", Conv),
                ("in_proj_w", C.c_void_p), ("in_proj_b", C.c_void_p), ("codebooks", C.c_void_p),
                ("out_proj_w", C.c_void_p), ("out_proj_b", C.c_void_p),
                ("ws", C.c_void_p * 4), ("ws_elems", C.c_size_t)]


class VitAttn(C.Structure):
    _fields_ = [("qkv_w", C.c_void_p), ("qkv_b", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p)]


class VitBlock(C.Structure):
    _fields_ = [("ln1_w", C.c_void_p), ("ln1_b", C.c_void_p), ("ln2_w", C.c_void_p), ("ln2_b", C.c_void_p),
                ("ln3_w", C.c_void_p), ("ln3_b", C.c_void_p), ("space", VitAttn), ("time", VitAttn),
                ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p)]


class Vit(C.Structure):
    _fields_ = [("depth", C.c_int32), ("dim", C.c_int32), ("heads", C.c_int32), ("hidden", C.c_int32),
                ("n_patches", C.c_int32), ("n_frames", C.c_int32),
                ("in_chans", C.c_int32), ("frames", C.c_int32), ("img", C.c_int32), ("patch", C.c_int32), ("patch_t", C.c_int32),
                ("patch_k", C.c_int32), ("eps", C.c_float), ("_pad", C.c_int32),
                ("pe_w", C.c_void_p), ("pe_b", C.c_void_p), ("cls_token", C.c_void_p), ("pos_embed", C.c_void_p),
                ("temp_embed", C.c_void_p), ("blocks_host", C.POINTER(VitBlock)), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
                ("agg_cls", C.c_void_p), ("agg_ln1_w", C.c_void_p), ("agg_ln1_b", C.c_void_p), ("agg_ln2_w", C.c_void_p),
                ("agg_ln2_b", C.c_void_p), ("agg_in_w", C.c_void_p), ("agg_in_b", C.c_void_p), ("agg_out_w", C.c_void_p),
                ("agg_out_b", C.c_void_p), ("agg_l1_w", C.c_void_p), ("agg_l1_b", C.c_void_p), ("agg_l2_w", C.c_void_p),
                ("agg_l2_b", C.c_void_p),
                ("ws_x", C.c_void_p), ("ws_qkv", C.c_void_p), ("ws_a", C.c_void_p), ("ws_h", C.c_void_p), ("ws_p", C.c_void_p),
                ("ws_z", C.c_void_p), ("ws_s", C.c_void_p)]


# name -> (restype, argtypes); every symbol declared in include/vaura_hip.h
SIGNATURES = {
    "vaura_audio_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_void_p, C.c_void_p]),
    "vaura_audio_scratch_elems": (C.c_size_t, [C.c_int]),
    "vaura_audio_loudness": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_float, C.c_void_p,
                                       C.c_void_p]),
    "vaura_audio_loudness_scratch_elems": (C.c_size_t, [C.c_int]),
    "vaura_dac_encode": (C.c_int, [C.POINTER(CodecEncoder), C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vaura_dac_encode_workspace_elems": (C.c_size_t, [C.POINTER(CodecEncoder), C.c_int, C.c_int64]),
    "vaura_avclip_forward": (C.c_int, [C.POINTER(Vit), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vaura_avclip_workspace_bytes": (C.c_size_t, [C.POINTER(Vit), C.c_int, C.c_int]),
    "vaura_version": (C.c_char_p, []),
    "vaura_set_debug_flags": (None, [C.c_uint]),
    "vaura_set_debug_flags2": (None, [C.c_uint]),
    "vaura_debug_counter": (C.c_longlong, [C.c_int]),
    "vaura_struct_size": (C.c_size_t, [C.c_int]),
    "vaura_packed_weight_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int]),
    "vaura_pack_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "vaura_build_token_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]),
    "vaura_pack_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "vaura_unpack_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "vaura_prefill_cond": (C.c_int, [C.POINTER(Dims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_void_p]),
    "vaura_pattern_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vaura_pattern_revert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vaura_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Sampling), C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p]),
    "vaura_decode_step": (C.c_int, [C.POINTER(Decoder), C.POINTER(Sampling), C.c_int, C.c_void_p]),
    "vaura_generate_loop": (C.c_int, [C.POINTER(Decoder), C.POINTER(Sampling), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vaura_step_graph_build": (C.c_int, [C.POINTER(Decoder), C.POINTER(Sampling), C.c_void_p, C.POINTER(C.c_void_p)]),
    "vaura_step_graph_free": (None, [C.c_void_p]),
    "vaura_profile_loop": (C.c_int, [C.POINTER(Decoder), C.POINTER(Sampling), C.c_int, C.c_uint, C.POINTER(C.c_double),
                                     C.POINTER(C.c_int64), C.c_void_p]),
    "vaura_profile_outliers": (None, [C.POINTER(C.c_int64)]),
    "vaura_gemv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                             C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "vaura_gemv_pair": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "vaura_split_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "vaura_attention_step_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vaura_attention_splits": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "vaura_attention_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vaura_dac_decode": (C.c_int, [C.POINTER(Codec), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vaura_dac_workspace_elems": (C.c_size_t, [C.POINTER(Codec), C.c_int, C.c_int]),
    "vaura_dac_conv": (C.c_int, [C.POINTER(Conv), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vaura_snake": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class VauraHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and type the shared library; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VauraHipError(
                f"{LIB_PATH} is missing: the HIP hot path is not built (run `python -m vaura_amd.csrc.build`). "
                "There is no CPU fallback.")
        # torch first: libvaura_hip.so must bind to the HIP runtime torch has already loaded and
        # initialised, otherwise a second runtime without a device context answers (hipErrorNoDevice)
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for which, cls in enumerate([Dims, LayerWeights, Sampling, Decoder, Conv, Codec, CodecEncoder, Vit, VitBlock]):
            if C.sizeof(cls) != handle.vaura_struct_size(which):
                raise VauraHipError(f"{LIB_PATH} was built from a different include/vaura_hip.h: sizeof({cls.__name__}) is "
                                    f"{handle.vaura_struct_size(which)} there, {C.sizeof(cls)} here (rebuild the library)")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        names = {-1: "VAURA_ERR_ARG", -2: "VAURA_ERR_SHAPE", -3: "VAURA_ERR_DTYPE", -4: "VAURA_ERR_STATE"}
        raise VauraHipError(f"{what}: {names.get(rc, rc)}")
    raise VauraHipError(f"{what}: hipError_t {rc}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (0 for None)."""
    return 0 if t is None else int(t.data_ptr())


def current_stream(device=None) -> int:
    """Raw handle of torch's current stream ON ``device`` (default: torch's current device).  Engines pass their own
    device: the current stream of another device would order nothing here."""
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)

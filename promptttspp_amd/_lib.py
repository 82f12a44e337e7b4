"""ctypes binding of libptpp_hip.so (the C ABI declared in include/ptpp.h).

The library is built in-tree by ``promptttspp_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is NO fallback: if the shared object is
missing or a call fails, an exception is raised -- the product path never
silently routes around the HIP kernels.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_int64, c_longlong, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libptpp_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWISH, ACT_TANH, ACT_MISH = 0, 1, 2, 3, 4, 5
ATTN_NEW, ATTN_LEGACY, ATTN_PLAIN = 0, 1, 2


class PtppError(RuntimeError):
    pass


class ConvArgs(Structure):
    """Mirror of ``ptpp_conv1d_args`` (include/ptpp.h)."""

    _fields_ = [
        ("x", c_void_p),
        ("wp", c_void_p),
        ("bias", c_void_p),
        ("res", c_void_p),
        ("y", c_void_p),
        ("lengths", c_void_p),
        ("B", c_int32),
        ("T", c_int32),
        ("Cin", c_int32),
        ("Cout", c_int32),
        ("ks", c_int32),
        ("dil", c_int32),
        ("pad", c_int32),
        ("ldx", c_int32),
        ("ldy", c_int32),
        ("ldr", c_int32),
        ("act", c_int32),
        ("in_mask", c_int32),
        ("out_mask", c_int32),
        ("out_scale", c_float),
        ("dtype", c_int32),
    ]


class TtsLossArgs(Structure):
    """Mirror of ``ptpp_tts_loss_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("pred", "noise", "flen", "pv", "cf0_tgt", "vuv_tgt", "y_dur", "dur", "plen", "y_sty", "sty_tgt",
                                        "total", "comps", "nll_dur", "nll_sty", "scratch")] + \
               [(n, c_int32) for n in ("B", "Tf", "Tp", "M", "G_dur", "G_sty", "D_sty", "dtype")] + \
               [(n, c_float) for n in ("dec_scale", "lp_min", "ls_min")]


class AmpLayerArgs(Structure):
    """Mirror of ``ptpp_amp_layer_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("x", "y", "res2", "w1p", "w2p", "b1", "b2", "log_alpha1", "log_alpha2")] + \
               [(n, c_float * 12) for n in ("up1", "dn1", "up2", "dn2")] + \
               [(n, c_int32) for n in ("B", "T", "C", "ks", "dil")] + \
               [("out_scale", c_float), ("res_scale", c_float), ("dtype", c_int32), ("w1s", c_void_p), ("w2s", c_void_p)]


class SnakeConvArgs(Structure):
    """Mirror of ``ptpp_snake_conv_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("x", "y", "res", "res2", "ws", "bias", "log_alpha")] + \
               [(n, c_float * 12) for n in ("up", "dn")] + \
               [(n, c_int32) for n in ("B", "T", "C", "ks", "dil")] + \
               [("out_scale", c_float), ("res_scale", c_float), ("dtype", c_int32)]


class WgradProblem(Structure):
    """Mirror of ``ptpp_wgrad_problem`` (include/ptpp.h)."""

    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("dbias", c_void_p), ("dil", c_int32), ("pad", c_int32)]


class DiffNetFwdArgs(Structure):
    """Mirror of ``ptpp_diffnet_stack_fwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("h0", "cond_all", "dsteps", "lengths", "skip", "dil_wp", "dil_b", "out_wp", "out_b",
                                        "yin_all", "a_all", "g_all", "x_buf0", "x_buf1", "o_buf")] + \
               [(n, c_int32) for n in ("B", "T", "C", "L", "cycle", "n_slabs", "fused_gate", "dtype")] + \
               [("wstream", c_void_p), ("skip_scaled", c_void_p), ("skip_scale", c_float), ("condx", c_void_p), ("ldcx", c_int32),
                ("yin0", c_void_p)]


class SamplerHeadArgs(Structure):
    """Mirror of ``ptpp_sampler_head_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("s", "ws_p", "ws_b", "wo_p", "wo_b", "x", "noise", "t", "sra", "srm1", "c1", "c2", "logvar", "x_out",
                                        "win_p", "win_b", "ds0", "h0", "yin0")] + \
               [(n, c_int32) for n in ("B", "T", "C", "M", "dtype")]


class DiffNetLayerArgs(Structure):
    """Mirror of ``ptpp_diffnet_layer_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("yin", "x", "cond", "wstream", "dil_b", "out_b", "dnext", "skip", "xn", "yin_next", "a_out",
                                        "g_out", "lengths")] + \
               [(n, c_int32) for n in ("B", "T", "C", "dil", "ldc", "init", "dtype")] + \
               [("skip_scaled", c_void_p), ("skip_scale", c_float), ("condx", c_void_p), ("ldcx", c_int32)]


class DiffNetBwdArgs(Structure):
    """Mirror of ``ptpp_diffnet_stack_bwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("gS", "yin_all", "a_all", "g_all", "lengths", "dil_wpt", "out_wpt", "dw_dil", "db_dil",
                                        "dw_out", "db_out", "gx_all", "do_all", "dg_buf", "dcond_all", "S", "ws_main")] + \
               [("ws_main_bytes", ctypes.c_size_t), ("ws_side", c_void_p), ("ws_side_bytes", ctypes.c_size_t),
                ("side_stream", c_void_p)] + \
               [(n, c_int32) for n in ("B", "T", "C", "L", "cycle", "dtype", "batched_wgrad")] + \
               [("dil_wst", c_void_p), ("out_wst", c_void_p), ("colpart", c_void_p)]


class EncoderLayersFwdArgs(Structure):
    """Mirror of ``ptpp_encoder_layers_fwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("h_in", "h_out", "lengths", "qkv_wp", "qkv_b", "ao_wp", "ao_b", "ln1_g", "ln1_b", "i_wp", "i_b",
                                        "o_wp", "o_b", "ln2_g", "ln2_b", "seeds", "scratch")] + \
               [("scratch_bytes", ctypes.c_size_t), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t), ("eps", c_float), ("p_att", c_float), ("p_hid", c_float)] + \
               [(n, c_int32) for n in ("B", "T", "C", "F", "H", "L", "dtype")]


class ConvLnFwdArgs(Structure):
    """Mirror of ``ptpp_conv_ln_stack_fwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("x0", "lengths", "wp", "bias", "gamma", "beta", "x_all", "z_all", "sum_all", "mean_all",
                                        "rstd_all", "seeds", "ws")] + \
               [("ws_bytes", ctypes.c_size_t), ("eps", c_float), ("drop_in", c_float), ("drop_out", c_float)] + \
               [(n, c_int32) for n in ("B", "T", "C", "n", "ks", "conv_act", "conv_mask", "ln_res", "act_in", "out_mask", "dtype")] + \
               [("wstream", c_void_p)]


class ConvLnBwdArgs(Structure):
    """Mirror of ``ptpp_conv_ln_stack_bwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("gy", "x0", "x_all", "z_all", "sum_all", "mean_all", "rstd_all", "lengths", "wpt", "gamma",
                                        "dw", "db", "dgamma", "dbeta", "gz_all", "tmp", "gx", "seeds", "red_scratch")] + \
               [("red_bytes", ctypes.c_size_t), ("ws_main", c_void_p), ("ws_main_bytes", ctypes.c_size_t), ("ws_side", c_void_p),
                ("ws_side_bytes", ctypes.c_size_t), ("side_stream", c_void_p), ("drop_in", c_float), ("drop_out", c_float)] + \
               [(n, c_int32) for n in ("B", "T", "C", "n", "ks", "conv_act", "conv_mask", "ln_res", "act_in", "out_mask", "dtype",
                                       "batched_wgrad")] + [("wstream_t", c_void_p)]


_CF_W = ["ln_g0", "ln_g1", "ln_g2", "ln_g3", "ln_g4", "ln_b0", "ln_b1", "ln_b2", "ln_b3", "ln_b4",
         "ffm_w1", "ffm_b1", "ffm_w2", "ffm_b2", "ff_w1", "ff_b1", "ff_w2", "ff_b2",
         "qkv_w", "qkv_b", "pos_w", "out_w", "out_b", "bias_u", "bias_v",
         "pw1_w", "pw1_b", "pw2_w", "pw2_b", "dw_w", "dw_b", "bn_g", "bn_b", "bn_rmean", "bn_rvar", "bn_mean_in", "bn_rstd_in"]


class ConformerWeights(Structure):
    """Mirror of ``ptpp_conformer_weights`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in _CF_W]


_CF_G = ["ln_g0", "ln_g1", "ln_g2", "ln_g3", "ln_g4", "ln_b0", "ln_b1", "ln_b2", "ln_b3", "ln_b4",
         "ffm_w1", "ffm_b1", "ffm_w2", "ffm_b2", "ff_w1", "ff_b1", "ff_w2", "ff_b2",
         "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "pos_w", "out_w", "out_b", "bias_u", "bias_v",
         "pw1_w", "pw1_b", "pw2_w", "pw2_b", "dw_w", "dw_b", "bn_sums"]


class ConformerGrads(Structure):
    """Mirror of ``ptpp_conformer_grads`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in _CF_G]


class ConformerFwdArgs(Structure):
    """Mirror of ``ptpp_conformer_block_fwd_args`` (include/ptpp.h)."""

    _fields_ = [("x", c_void_p), ("y", c_void_p), ("pos_emb", c_void_p), ("lengths", c_void_p), ("w", ConformerWeights),
                ("slab", c_void_p), ("slab_bytes", ctypes.c_size_t), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t),
                ("red_scratch", c_void_p), ("red_bytes", ctypes.c_size_t), ("seeds", c_void_p),
                ("p_ffn", c_float), ("p_drop", c_float), ("bn_momentum", c_float), ("bn_eps", c_float)] + \
               [(n, c_int32) for n in ("B", "T", "C", "F", "H", "L", "ks_ffn", "ks_dw", "variant", "bn_train", "save", "dtype")] + \
               [("ffn_ws", c_void_p * 4)]


class ConformerBwdArgs(Structure):
    """Mirror of ``ptpp_conformer_block_bwd_args`` (include/ptpp.h)."""

    _fields_ = [("gy", c_void_p), ("gx", c_void_p), ("x", c_void_p), ("pos_emb", c_void_p), ("lengths", c_void_p),
                ("w", ConformerWeights)] + \
               [(n, c_void_p) for n in ("ffm_w1t", "ffm_w2t", "ff_w1t", "ff_w2t", "qkv_wt", "out_wt", "pw1_wt", "pw2_wt")] + \
               [("g", ConformerGrads), ("slab", c_void_p), ("scratch", c_void_p), ("scratch_bytes", ctypes.c_size_t),
                ("ws_main", c_void_p), ("ws_main_bytes", ctypes.c_size_t), ("ws_side", c_void_p), ("ws_side_bytes", ctypes.c_size_t),
                ("red_scratch", c_void_p), ("red_bytes", ctypes.c_size_t), ("side_stream", c_void_p), ("seeds", c_void_p),
                ("p_ffn", c_float), ("p_drop", c_float)] + \
               [(n, c_int32) for n in ("B", "T", "C", "F", "H", "L", "ks_ffn", "ks_dw", "variant", "bn_train", "dtype")] + \
               [("ffn_wts", c_void_p * 4), ("side_stream2", c_void_p), ("bn_dgamma", c_void_p), ("bn_dbeta", c_void_p)]


class WgradGProblem(Structure):
    """Mirror of ``ptpp_wgrad_gproblem`` (include/ptpp.h)."""

    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("dbias", c_void_p), ("lengths", c_void_p)] + \
               [(n, c_int32) for n in ("B", "T", "Cin", "Cout", "ks", "dil", "pad", "ldx", "lddy")]


class RefEncConvsFwdArgs(Structure):
    """Mirror of ``ptpp_refenc_convs_fwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("x", "y", "cout", "w", "wp_fwd", "wp_bwd", "bn_g", "bn_b", "bn_rmean", "bn_rvar", "slab")] + \
               [("slab_bytes", ctypes.c_size_t), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t), ("red_scratch", c_void_p),
                ("red_bytes", ctypes.c_size_t), ("bn_momentum", c_float), ("bn_eps", c_float)] + \
               [(n, c_int32) for n in ("B", "H", "W", "nlayer", "dtype")]


class RefEncConvsBwdArgs(Structure):
    """Mirror of ``ptpp_refenc_convs_bwd_args`` (include/ptpp.h)."""

    _fields_ = [(n, c_void_p) for n in ("gy", "cout", "wp_bwd", "bn_g", "bn_b", "dwg", "bn_sums", "slab", "scratch")] + \
               [("scratch_bytes", ctypes.c_size_t), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t), ("red_scratch", c_void_p),
                ("red_bytes", ctypes.c_size_t)] + \
               [(n, c_int32) for n in ("B", "H", "W", "nlayer", "dtype")]


P, I, F, U64, I64, SZ = c_void_p, c_int, c_float, c_uint64, c_int64, ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/ptpp.h declares.
SIGNATURES = {
    "ptpp_last_error": (ctypes.c_char_p, []),
    "ptpp_version": (I, []),
    "ptpp_stream_wait": (I, [P, P]),
    "ptpp_conv_cin_padded": (I, [I, I]),
    "ptpp_pack_conv_weight": (I, [P, P, I, I, I, I, I, P]),
    "ptpp_pack_conv2d_3x3": (I, [P, P, P, I, I, I, I, P]),
    "ptpp_pack_conv_weights_batched": (I, [P, I, P, I, P]),
    "ptpp_conv1d_fwd": (I, [POINTER(ConvArgs), P]),
    "ptpp_conv1d_fwd_ex": (I, [POINTER(ConvArgs), P, I, F, F, U64, P]),
    "ptpp_conv1d_fwd_ws": (I, [POINTER(ConvArgs), P, I, F, F, U64, P, SZ, P]),
    "ptpp_conv1d_wgrad": (I, [P, P, P, P, P] + [I] * 11 + [P, SZ, P]),
    "ptpp_conv1d_wgrad_batched": (I, [POINTER(WgradProblem), I, P, I, I, I, I, I, I, I, I, I, P, SZ, P]),
    "ptpp_conv1d_wgrad_grouped": (I, [POINTER(WgradGProblem), I, I, P, SZ, P]),
    "ptpp_conv1d_wgrad_grouped2": (I, [POINTER(WgradGProblem), I, I, P, SZ, P, P]),
    "ptpp_epilogue_bwd": (I, [P, P, P, P, I, I, I, F, I, I, F, U64, I, P]),
    "ptpp_layernorm_fwd": (I, [P] * 9 + [I, I, I, F, I, I, F, U64, F, U64, I, P]),
    "ptpp_layernorm_bwd": (I, [P] * 11 + [I, I, I, I, I, F, U64, F, U64, I, P, SZ, P]),
    "ptpp_layernorm_bwd_add": (I, [P] * 9 + [F, I, P, P, P, I, I, I, I, I, F, U64, F, U64, I, P, SZ, P]),
    "ptpp_attention_fwd": (I, [P] * 9 + [I] * 8 + [F, U64, I, P]),
    "ptpp_attention_bwd": (I, [P] * 16 + [I] * 9 + [F, U64, I, P, SZ, P]),
    "ptpp_attention_win_fwd": (I, [P] * 8 + [I] * 7 + [F, U64, I, P]),
    "ptpp_attention_win_bwd": (I, [P] * 14 + [I] * 8 + [F, U64, I, P]),
    "ptpp_length_regulate_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "ptpp_length_regulate_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "ptpp_posenc_fwd": (I, [P, P, P, I, I, I, F, F, U64, I, P]),
    "ptpp_gate_fwd": (I, [P, P, I64, I, I, P]),
    "ptpp_gate_bwd": (I, [P, P, P, I64, I, I, I, P]),
    "ptpp_diffnet_post_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "ptpp_conv1d_diffnet_post_supported": (I, [I, I, I]),
    "ptpp_conv1d_diffnet_post": (I, [POINTER(ConvArgs), P, P, P, P, P, I, P]),
    "ptpp_ddpm_step": (I, [P] * 10 + [I, I64, I, P]),
    "ptpp_ddpm_step_lp": (I, [P] * 11 + [I, I64, I, P]),
    "ptpp_sampler_head_supported": (I, [I, I, I]),
    "ptpp_sampler_head": (I, [POINTER(SamplerHeadArgs), P]),
    "ptpp_mdn_nll_fwd": (I, [P] * 6 + [I64, I, I, F, F, P]),
    "ptpp_mdn_nll_bwd": (I, [P] * 10 + [I64, I, I, F, F, P]),
    "ptpp_conv1d_gate_bwd_supported": (I, [I, I, I]),
    "ptpp_conv1d_gate_bwd": (I, [POINTER(ConvArgs), P, P, I, P]),
    "ptpp_l1_scratch_bytes": (I64, []),
    "ptpp_l1_masked_mean_fwd": (I, [P, P, P, P, c_float, I64, I, I, P, P, P]),
    "ptpp_l1_masked_mean_bwd": (I, [P, P, P, P, P, c_float, I64, I, I, P, P]),
    "ptpp_conv1d_rt_gate_bwd_supported": (I, [I, I, I]),
    "ptpp_conv1d_rt_gate_bwd": (I, [POINTER(ConvArgs), P, P, P, I, P]),
    "ptpp_conv1d_gate_fwd_save_supported": (I, [I, I, I]),
    "ptpp_conv1d_gate_fwd_save": (I, [POINTER(ConvArgs), P, I, P]),
    "ptpp_diffnet_post_bwd": (I, [P, P, P, P, I, I, I, I, P]),
    "ptpp_diffnet_post_bwd_fill": (I, [P, P, P, I, I, I, I, I, P]),
    "ptpp_colsum_batch": (I, [P, P, I, I, I, I, P]),
    "ptpp_col_reduce": (I, [P, P, P, I64, I, I, P, SZ, P]),
    "ptpp_bn_stats": (I, [P, I64, I, F, F, P, P, P, P, I, P, SZ, P]),
    "ptpp_bn_act_fwd": (I, [P, P, P, P, P, P, I64, I, I, I, P]),
    "ptpp_bn_act_bwd": (I, [P, P, P, P, P, P, P, P, I64, I, I, I, I, P, SZ, P]),
    "ptpp_bn_act_bwd_acc": (I, [P, P, P, P, P, P, P, P, P, P, I64, I, I, I, I, P, SZ, P]),
    "ptpp_glu_bwd_masked": (I, [P, P, P, P, I, I, I, I, P]),
    "ptpp_glu_fwd": (I, [P, P, I64, I, I, P]),
    "ptpp_glu_bwd": (I, [P, P, P, I64, I, I, P]),
    "ptpp_dwconv1d": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "ptpp_dwconv1d_wgrad": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "ptpp_im2col3x3s2": (I, [P, P, I, I, I, I, I, P]),
    "ptpp_im2col3x3s2_c1": (I, [P, P, I, I, I, I, P]),
    "ptpp_col2im3x3s2": (I, [P, P, I, I, I, I, I, P]),
    "ptpp_gru_gate_fwd": (I, [P, I64, P, P, P, I, P, I, I, P]),
    "ptpp_gru_gate_bwd": (I, [P, I64, P, P, P, I, P, P, I64, P, P, I, I, P]),
    "ptpp_gru_seq_supported": (I, [I]),
    "ptpp_gru_seq_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "ptpp_gru_seq_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "ptpp_aa_snake_fwd": (I, [P, P, P, POINTER(c_float), POINTER(c_float), I, I, I, I, P]),
    "ptpp_amp_layer_supported": (I, [I, I]),
    "ptpp_amp_layer_fwd": (I, [POINTER(AmpLayerArgs), P]),
    "ptpp_amp_pack_wstream": (I, [P, P, I, I, I, P]),
    "ptpp_snake_conv_post_supported": (I, [I, I, I]),
    "ptpp_snake_conv_post_tanh": (I, [P, P, POINTER(c_float), POINTER(c_float), P, F, P, I, I, I, I, I, P]),
    "ptpp_snake_conv1d_supported": (I, [I, I]),
    "ptpp_snake_conv1d_fwd": (I, [POINTER(SnakeConvArgs), P]),
    "ptpp_add3_scale": (I, [P, P, P, P, F, I64, I, P]),
    "ptpp_conv_post_tanh": (I, [P, P, F, P, I, I, I, I, I, P]),
    "ptpp_filtfilt": (I, [P, P, P, P, POINTER(ctypes.c_double), POINTER(ctypes.c_double), I, I, I, I, P]),
    "ptpp_bct_to_btc": (I, [P, P, I, I, I, I, P]),
    "ptpp_btc_to_bct": (I, [P, P, I, I, I, I, P]),
    "ptpp_grad_sumsq": (I, [P, I, P, c_longlong, P, P]),
    "ptpp_grad_sumsq_det": (I, [P, I, P, c_longlong, P, P, P]),
    "ptpp_adamw_step": (I, [P, I, P, c_longlong, P, P, F, F, F, F, I, F, P]),
    "ptpp_diffnet_stack_fwd": (I, [POINTER(DiffNetFwdArgs), P]),
    "ptpp_conv1d_rt_supported": (I, [I, I, I, I, I, I]),
    "ptpp_conv1d_rt_ex_supported": (I, [I, I, I, I, I, I]),
    "ptpp_conv_rt_set_min_rows": (I, [c_longlong]),
    "ptpp_conv1d_rt_fwd_ex": (I, [POINTER(ConvArgs), P, ctypes.c_float, ctypes.c_float, U64, P, SZ, P]),
    "ptpp_conv1d_rt_fwd_ex_relu_bwd": (I, [POINTER(ConvArgs), P, P, P, ctypes.c_float, P, SZ, P]),
    "ptpp_conv1d_rt_fwd_ex_partial": (I, [POINTER(ConvArgs), P, P, SZ, POINTER(ctypes.c_int), P]),
    "ptpp_layernorm_bwd_add_splitk": (I, [P, I, P, P, P, P, P, P, P, P, ctypes.c_float, I, P, P, P, I, I, I, I, ctypes.c_float, U64, I, P, SZ, P]),
    "ptpp_conv1d_rt_fwd": (I, [POINTER(ConvArgs), P, ctypes.c_float, P]),
    "ptpp_conv1d_rt_fwd_aux": (I, [POINTER(ConvArgs), P, ctypes.c_float, P, I, ctypes.c_float, P]),
    "ptpp_conv1d_rt_fwd_cs": (I, [POINTER(ConvArgs), P, ctypes.c_float, P, I, ctypes.c_float, P, P]),
    "ptpp_conv1d_rt_colpart_supported": (I, [I, I, I, I, I]),
    "ptpp_nsf_source_supported": (I, [I]),
    "ptpp_nsf_source_scratch_bytes": (SZ, [I, I, I]),
    "ptpp_nsf_source": (I, [P, P, P, P, ctypes.c_float, P, I, I, I, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, P, SZ, P]),
    "ptpp_diffnet_layer_supported": (I, [I, I]),
    "ptpp_diffnet_wstream_bytes": (ctypes.c_int64, [I]),
    "ptpp_diffnet_pack_wstream": (I, [P, P, P, I, I, P]),
    "ptpp_diffnet_wstream_bytes_cond": (ctypes.c_int64, [I]),
    "ptpp_diffnet_pack_wstream_cond": (I, [P, P, P, P, I, I, P]),
    "ptpp_diffnet_layer_fwd": (I, [POINTER(DiffNetLayerArgs), P]),
    "ptpp_diffnet_layer_fwd_dbg": (I, [POINTER(DiffNetLayerArgs), I, P, P]),
    "ptpp_diffnet_stack_bwd": (I, [POINTER(DiffNetBwdArgs), P]),
    "ptpp_conv_ln_stack_fwd": (I, [POINTER(ConvLnFwdArgs), P]),
    "ptpp_conv_ln_stack_bwd": (I, [POINTER(ConvLnBwdArgs), P]),
    "ptpp_cast_from_f32": (I, [P, P, I64, I, P]),
    "ptpp_conformer_block_slab_bytes": (SZ, [I, I, I, I, I, I, I]),
    "ptpp_conformer_block_bwd_scratch_bytes": (SZ, [I, I, I, I, I, I, I]),
    "ptpp_conformer_block_fwd": (I, [POINTER(ConformerFwdArgs), P]),
    "ptpp_conformer_block_bwd": (I, [POINTER(ConformerBwdArgs), P]),
    "ptpp_refenc_convs_slab_bytes": (SZ, [I, I, I, I, P, I]),
    "ptpp_refenc_convs_bwd_scratch_bytes": (SZ, [I, I, I, I, P, I]),
    "ptpp_refenc_convs_fwd": (I, [POINTER(RefEncConvsFwdArgs), P]),
    "ptpp_refenc_convs_bwd": (I, [POINTER(RefEncConvsBwdArgs), P]),
    "ptpp_encoder_layers_fwd": (I, [POINTER(EncoderLayersFwdArgs), P]),
    "ptpp_red_defer": (I, [P, SZ]),
    "ptpp_red_defer_suspend": (I, [I]),
    "ptpp_red_pending": (I, []),
    "ptpp_red_flush": (I, [P]),
    "ptpp_tts_losses_scratch_bytes": (I64, []),
    "ptpp_tts_losses_fwd": (I, [POINTER(TtsLossArgs), P]),
    "ptpp_tts_losses_bwd": (I, [POINTER(TtsLossArgs), P, P, P, P, P, P, P]),
    "ptpp_q_sample_bct": (I, [P, P, P, P, P, I, F, F, F, I, P, I, I, I, I, P]),
    "ptpp_step_sinusoid": (I, [P, I64, F, I, I, P, P]),
    "ptpp_mish_fwd": (I, [P, P, I64, P]),
    "ptpp_mish_bwd": (I, [P, P, P, I64, P]),
    "ptpp_embed_cl_fwd": (I, [P, P, P, F, I, P, I, I, I, I, I, P]),
    "ptpp_embed_cl_bwd": (I, [P, P, P, F, I, P, I, I, I, I, I, I, P]),
    "ptpp_scalar_embed_add": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "ptpp_scalar_embed_bwd": (I, [P, P, P, P, P, I, I, I, I, P, SZ, P]),
    "ptpp_linear_small_fwd": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "ptpp_linear_small_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "ptpp_l2norm_fwd": (I, [P, P, P, I, I, F, P]),
    "ptpp_l2norm_bwd": (I, [P, P, P, P, I, I, F, P]),
    "ptpp_durations_cumsum": (I, [P, I, P, I, I, P]),
    "ptpp_bcast_add_rows": (I, [P, P, P, I, I, I, I, P]),
    "ptpp_rows_sum": (I, [P, P, I, I, I, I, P]),
    "ptpp_comm_unique_id": (I, [P]),
    "ptpp_comm_init": (I, [I, I, P, POINTER(c_void_p)]),
    "ptpp_comm_destroy": (I, [P]),
    "ptpp_allreduce_mean": (I, [P, I64, I, P, P]),
    "ptpp_broadcast": (I, [P, I64, I, I, P, P]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtppError(
            f"{LIB_PATH} not found: build it with `make -C promptttspp_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "promptttspp_amd has no CPU / eager fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ptpp_last_error().decode("utf-8", "replace")
        raise PtppError(f"{what} failed with status {status}: {msg}")

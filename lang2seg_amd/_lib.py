"""ctypes binding of liblang2seg_hip.so (C ABI in include/lang2seg_hip.h).

No CPU fallback exists: if the library is missing or a symbol is absent this module
raises, so a GPU box can never silently run a non-HIP path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'liblang2seg_hip.so')
F32, BF16 = 0, 1

vp, i32, i64, f32, u64, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [('x', vp), ('w', vp), ('y', vp), ('bias', vp), ('add', vp), ('ref', vp),
                ('n_img', i32), ('IH', i32), ('IW', i32), ('Cin', i32), ('OH', i32), ('OW', i32), ('Cout', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('ldx', i32), ('ldy', i32), ('ldadd', i32), ('ldref', i32), ('flags', i32),
                ('out_h', i32), ('out_w', i32), ('out_stride', i32), ('tile', i32), ('split_k', i32), ('xcd_mode', i32), ('algo', i32), ('ws', vp), ('ws_floats', sz), ('prio', i32)]


class RoiBlock0Desc(C.Structure):
    _fields_ = [('feat', vp), ('rois', vp), ('w1', vp), ('b1', vp), ('w2', vp), ('b2', vp), ('pooled', vp), ('y1', vp), ('y2', vp),
                ('H', i32), ('W', i32), ('C', i32), ('R', i32), ('P', i32), ('N1', i32), ('N2', i32), ('spatial_scale', C.c_float), ('debug', i32)]


class WgradDesc(C.Structure):
    _fields_ = [('dy', vp), ('x', vp), ('dw', vp),
                ('n_img', i32), ('IH', i32), ('IW', i32), ('Cin', i32), ('OH', i32), ('OW', i32), ('Cout', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('lddy', i32), ('ldx', i32),
                ('split_k', i32), ('tile', i32), ('ws', vp), ('ws_bytes', sz)]


class WgradProb(C.Structure):
    _fields_ = [('dy', vp * 2), ('x', vp * 2), ('n_img', i32 * 2), ('IH', i32 * 2), ('IW', i32 * 2), ('OH', i32 * 2), ('OW', i32 * 2),
                ('lddy', i32 * 2), ('ldx', i32 * 2), ('dw', vp),
                ('nseg', i32), ('Cin', i32), ('Cout', i32), ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('split', i32), ('ws_off', i64), ('flags', i32), ('reserved', i32)]


class TransposeDesc(C.Structure):
    _fields_ = [('src', vp), ('scale', vp), ('dst', vp), ('Cout', i32), ('taps', i32), ('Cin', i32), ('force_f32', i32)]


class LstmFwdDir(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w_hh', 'b_hh', 'gates_in', 'h_prev', 'c_prev', 'c', 'h', 'act', 'gates_out')]


class LstmBwdDir(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w_hh_T', 'dgates_next', 'dh_ext', 'dc_in', 'act', 'c_prev', 'c', 'dgates', 'dc_prev')]


class Bottleneck64Desc(C.Structure):
    _fields_ = [(n, vp) for n in ('a', 'x', 'w2', 'w3', 'wd', 'w1n', 'b2', 'b3', 'bd', 'b1n', 'y', 'a_next')] + [(n, i32) for n in ('H', 'W', 'Cx')]


class CapRecurFwdArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('w_h2h', 'b_h2h', 'w_h2att', 'b_h2att', 'patt', 'aw', 'ab', 'P', 'b_a2c', 'sums', 'hs', 'cs', 'save', 'tanh_ws', 'wgt', 'state')] + \
               [(n, i32) for n in ('S', 'R', 'AH', 'L')]


class CapRecurBwdArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('w_h2h', 'w_h2att', 'P', 'aw', 'save', 'cs', 'wgt', 'tanh_ws', 'dho', 'dsums', 'da2c', 'ddot', 'datt_h', 'state')] + \
               [(n, i32) for n in ('S', 'R', 'AH', 'L', 'ld_ddot', 'ld_datt_h')]


class SgdSeg(C.Structure):
    _fields_ = [('offset', i64), ('count', i64), ('row_len', i32), ('weight_decay', i32), ('rowscale_off', i64),
                ('lr_mult', f32), ('chunk0', i32), ('flags', i32), ('reserved', i32)]


CONV_RELU, CONV_OUT_F32, CONV_DECONV2X2, CONV_SCATTER = 1, 4, 8, 16
LOSS_RPN_CLS, LOSS_RPN_BOX, LOSS_CLS, LOSS_BOX, LOSS_MASK, LOSS_CAP, LOSS_TOTAL, LOSS_RESPONSE = range(8)

# name -> (restype, argtypes); the stream is always the last argument
SIGS = {
    'l2s_version': (i32, []),
    'l2s_conv_plan_name': (C.c_char_p, [vp, i32]),
    'l2s_roialign_block0_fwd': (i32, [C.POINTER(RoiBlock0Desc), vp]),
    'l2s_conv_igemm': (i32, [C.POINTER(ConvDesc), i32, vp]),
    'l2s_conv_wgrad': (i32, [C.POINTER(WgradDesc), i32, vp]),
    'l2s_wgrad_ws_bytes': (sz, [C.POINTER(WgradDesc), i32]),
    'l2s_wgrad_variant': (i32, [i32, i32, i32, i32, i32, i32, i32, i64, i32]),
    'l2s_wgrad_tiles': (i64, [i32, i32, i32, i32, i32]),
    'l2s_conv_wgrad_grouped': (i32, [vp, vp, i32, i32, i32, vp, sz, vp]),
    'l2s_weight_cast': (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    'l2s_weight_transpose': (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    'l2s_colsum': (i32, [vp, i32, i32, i32, vp, vp, i64, i32, vp]),
    'l2s_weight_transpose_batched': (i32, [vp, i32, i32, i32, vp]),
    'l2s_stem_conv': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'l2s_stem_pack_bytes': (C.c_size_t, []),
    'l2s_stem_pack': (i32, [vp, vp, vp]),
    'l2s_stem_pool_bf16': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'l2s_maxpool3x3s2': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'l2s_fill_f32': (i32, [vp, f32, i64, vp]),
    'l2s_cast': (i32, [vp, i32, vp, i32, i64, vp]),
    'l2s_add3': (i32, [vp, vp, vp, vp, i64, i32, vp]),
    'l2s_avgpool_fwd': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'l2s_avgpool_bwd': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'l2s_adaptive_pool_fwd': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'l2s_adaptive_pool_bwd': (i32, [vp, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'l2s_mask_downsample': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'l2s_dropout_mask': (i32, [vp, i64, f32, vp, u64, vp]),
    'l2s_counter_inc': (i32, [vp, vp]),
    'l2s_stamp': (i32, [vp, vp]),
    'l2s_rpn_decode': (i32, [vp, i32, vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp]),
    'l2s_sort_topk': (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    'l2s_nms_workspace_bytes': (sz, [i32]),
    'l2s_sort_ws_ints': (i64, [i32]),
    'l2s_nms': (i32, [vp, i32, f32, i32, i32, vp, vp, vp, vp]),
    'l2s_gather_rois': (i32, [vp, vp, vp, vp, i32, vp, vp, vp]),
    'l2s_random_keys': (i32, [vp, i64, vp, u64, vp]),
    'l2s_anchor_target_ws_ints': (i64, [i32]),
    'l2s_anchor_target': (i32, [vp, i32, vp, i32, i32, i32, i32, f32, f32, vp, vp, f32, f32, i32, f32, vp, vp, vp, vp, vp, vp]),
    'l2s_proposal_target': (i32, [vp, vp, vp, i32, vp, i32, vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, f32, f32,
                                  vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'l2s_roialign_fwd': (i32, [vp, i32, i32, i32, vp, i32, i32, f32, vp, i32, vp]),
    'l2s_roialign_bwd': (i32, [vp, i32, i32, i32, vp, i32, i32, f32, vp, i32, vp]),
    'l2s_rpn_loss': (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, i32, i32, vp, vp, vp]),
    'l2s_anchor_target_count': (vp, [vp]),
    'l2s_rcnn_loss': (i32, [vp, i32, vp, vp, vp, vp, i32, i32, f32, vp, vp, i32, i32, vp]),
    'l2s_mask_loss': (i32, [vp, i32, vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    'l2s_total_loss': (i32, [vp, f32, vp]),
    'l2s_maskpred_ws_floats': (i64, [i32, i32]),
    'l2s_maskpred_bwd': (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    'l2s_maskpred_bwd_dx': (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    'l2s_maskpred_bwd_reduce': (i32, [vp, vp, vp, i32, i32, vp, vp, vp]),
    'l2s_linear_fwd': (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'l2s_linear_bwd_x': (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, i64, vp]),
    'l2s_linear_bwd_x_ws_floats': (i64, [i32, i32, i32]),
    'l2s_linear_bwd_w': (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp]),
    'l2s_act_bwd': (i32, [vp, vp, i64, i32, vp]),
    'l2s_mask_relu_cast': (i32, [vp, vp, vp, vp, i32, i64, vp]),
    'l2s_embed_fwd': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'l2s_embed_bwd': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    'l2s_lstm_cell_fwd': (i32, [vp, vp, vp, vp, vp, i32, vp]),
    'l2s_lstm_cell_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    'l2s_dynfilter_fwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'l2s_dynfilter_ws_floats': (i64, [i32, i32, i32]),
    'l2s_dynfilter_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    'l2s_dynfilter_bwd_finish': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'l2s_scale_mask': (i32, [vp, vp, vp, vp, C.c_long, i32, vp]),
    'l2s_conv3x3_c3': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'l2s_maxpool2x2_fwd': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'l2s_maxpool2x2_bwd': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'l2s_cropalign_fwd': (i32, [vp, i32, i32, i32, vp, i32, i32, f32, f32, vp, i32, vp]),
    'l2s_cropalign_bwd': (i32, [vp, i32, i32, i32, vp, i32, i32, f32, f32, vp, i32, vp]),
    'l2s_roipool_fwd': (i32, [vp, i32, i32, i32, vp, i32, i32, f32, vp, vp, i32, vp]),
    'l2s_roipool_bwd': (i32, [vp, vp, i32, i32, i32, vp, i32, vp]),
    'l2s_rle_from_string': (i32, [C.c_char_p, vp, i32]),
    'l2s_prep_geometry': (i32, [i32, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(i32)]),
    'l2s_prep_image': (i32, [vp, i32, i32, C.c_double, C.c_double, C.c_double, C.c_double, i32, i32, vp, vp]),
    'l2s_rle_ws_words': (i64, [i32, i32, i32]),
    'l2s_rle_to_mask': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    'l2s_lstm_step_fwd': (i32, [vp, i32, i32, vp]),
    'l2s_lstm_step_bwd': (i32, [vp, i32, i32, vp]),
    'l2s_rcnn_predict': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    'l2s_mask_prob': (i32, [vp, i32, i32, vp, i32, C.c_long, vp, vp]),
    'l2s_response_loss': (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    'l2s_cap_attention_fwd': (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    'l2s_cap_attention_bwd': (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    'l2s_cap_gates_fwd': (i32, [vp, vp, vp, vp, vp, vp, i32, vp]),
    'l2s_cap_gates_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    'l2s_linear2_fwd': (i32, [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp]),
    'l2s_linear_sum2_fwd': (i32, [vp, vp, i32, vp, vp, i32, vp, i32, i32, vp]),
    'l2s_cap_a2c_gates_fwd': (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp]),
    'l2s_cap_attention_bwd_step': (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    'l2s_cap_att_dots_fwd': (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    'l2s_bottleneck64_fwd': (i32, [C.POINTER(Bottleneck64Desc), vp]),
    'l2s_cap_recur_supported': (i32, [i32, i32, i32, i32]),
    'l2s_cap_recur_state_bytes': (sz, [i32]),
    'l2s_cap_recur_fwd': (i32, [C.POINTER(CapRecurFwdArgs), vp]),
    'l2s_cap_recur_bwd': (i32, [C.POINTER(CapRecurBwdArgs), vp]),
    'l2s_cap_apply_gates_fwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    'l2s_cap_gates_bwd_dw': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    'l2s_cap_attention_bwd_step2': (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    'l2s_cap_attention_bwd_batched': (i32, [vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    'l2s_logsoftmax_nll': (i32, [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    'l2s_sgd_momentum': (i32, [vp, vp, vp, vp, i32, vp, f32, f32, f32, f32, vp, i32, i32, vp]),
    'l2s_wgrad_grouped_ws_bytes': (sz, [i32]),
    'l2s_sgd_momentum_range': (i32, [vp, vp, vp, vp, i32, vp, f32, f32, f32, f32, vp, i32, i32, i64, i64, i32, i32, vp]),
    'l2s_sgd_momentum_range_g16': (i32, [vp, vp, i64, vp, vp, i32, vp, f32, f32, f32, f32, vp, i32, i64, i64, i32, i32, vp]),
    'l2s_sgd_chunk': (i32, []),
    'l2s_mul_f32': (i32, [vp, vp, vp, i64, vp]),
    'l2s_add_f32': (i32, [vp, vp, vp, i64, vp]),
    'l2s_stream_fork': (i32, [vp, vp]),
    'l2s_event_record': (i32, [i32, vp]),
    'l2s_event_wait': (i32, [i32, vp]),
    'l2s_memset_async': (i32, [vp, i32, sz, vp]),
    'l2s_memcpy_d2d_async': (i32, [vp, vp, sz, vp]),
    'l2s_tape_begin': (vp, [vp, i32]),
    'l2s_tape_end': (i32, [vp]),
    'l2s_tape_size': (i64, [vp]),
    'l2s_tape_run': (i32, [vp, vp, i32]),
    'l2s_tape_destroy': (i32, [vp]),
    'l2s_tape_mark': (i32, []),
    'l2s_tape_pause': (i32, [i32]),
    'l2s_tape_time_event': (i32, [vp]),
    'l2s_time_event_elapsed': (i32, [i32, i32, vp]),
    'l2s_tape_segments': (i32, [vp]),
    'l2s_tape_run_segment': (i32, [vp, vp, i32, i32]),
}

_lib = None


class L2SError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and bind every symbol the header declares (loud on any miss)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise L2SError('%s not found: run `python __graft_entry__.py` (hipcc build) first; there is no CPU fallback' % LIB_PATH)
    # torch first: its wheel bundles its own libamdhip64, and a process must end up with ONE HIP runtime.  Loaded the other way round
    # (this library's /opt/rocm runtime first, torch's second) every launch from here fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def tools_set(name, value):
    """A/B tools only: set a tunable of the TOOLS build of the library (csrc/knobs.h; tools/build_tools_lib.py builds it, a tool points
    LIB_PATH at it before the first load()).  The product library has no such symbol - its tunables are compile-time constants - and this raises."""
    lib = load()
    fn = getattr(lib, 'l2s_tools_set', None)
    if fn is None:
        raise L2SError('%s is the product library: it has no tunables; build and load the tools library (tools/build_tools_lib.py)' % LIB_PATH)
    fn.restype, fn.argtypes = i32, [C.c_char_p, i32]
    if fn(name.encode(), int(value)) != 0:
        raise L2SError('l2s_tools_set: unknown tunable %r' % name)


def ptr(t):
    """torch tensor (or None) -> raw device/host address."""
    return None if t is None else t.data_ptr()


_FN = {}


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    r = fn(*args)
    if r != 0:
        raise L2SError('%s returned error %d' % (name, r))

"""ctypes binding of libxlstm_hved_hip.so (the C ABI declared in include/xlstm_hved.h).

The library is the product: there is no CPU or PyTorch fallback.  Calling any op without the built
library (or without a GPU) raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libxlstm_hved_hip.so")

XH_F32, XH_BF16, XH_F16 = 0, 1, 2
ARITH_F32_SPLIT, ARITH_K7_VECTOR = 1, 2      # xh_conv_desc.arith bits
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3

c_fp = C.POINTER(C.c_float)
c_dp = C.POINTER(C.c_double)
vp = C.c_void_p
ll = C.c_longlong


MAX_WPTR = 8          # == XH_MAX_WPTR


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("N", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("groups", C.c_int),
        ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Do", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("k", C.c_int), ("stride", C.c_int), ("Ca", C.c_int),
        ("xa_bs", ll), ("xb_bs", ll), ("y_bs", ll),
        ("n_wptr", C.c_int), ("transposed", C.c_int), ("pre", C.c_int), ("pre_slope", C.c_float),
        ("act", C.c_int), ("act_slope", C.c_float), ("epi", C.c_int), ("Cea", C.c_int),
        ("ea_bs", ll), ("eb_bs", ll), ("e_slope", C.c_float),
        ("px_bs", ll), ("pd_bs", ll), ("arith", C.c_int), ("bcast", C.c_int),
    ]


class ConvPtrs(C.Structure):
    _fields_ = [
        ("xa", vp), ("xb", vp), ("w", vp * MAX_WPTR), ("b", vp * MAX_WPTR), ("pre_sc", vp), ("pre_sh", vp), ("y", vp),
        ("ea", vp), ("eb", vp), ("e_sc", vp), ("e_sh", vp), ("red", vp), ("ws", vp), ("ws_bytes", ll),
        ("fin_red", vp), ("fin_mean", vp), ("fin_rstd", vp), ("fin_count", ll), ("ws_packed", C.c_int),
        ("fan", vp), ("fan_bytes", ll),
        ("px", vp), ("pd", vp), ("nb_red", vp), ("nb_mean", vp), ("nb_rstd", vp), ("nb_count", ll),
        ("fin_gamma", vp), ("fin_beta", vp), ("fin_rm", vp), ("fin_rv", vp), ("fin_steps", C.c_int), ("e_ctr", vp),
    ]


class AttenJob(C.Structure):       # == xh_atten_job
    _fields_ = [("p", vp * 8), ("w", vp), ("b", vp), ("g", vp * 8), ("gw", vp), ("gb", vp), ("NS", C.c_int), ("NE", C.c_int),
                ("E", C.c_int), ("K3", C.c_int)]


class PoeJob(C.Structure):         # == xh_poe_job
    _fields_ = [("feat", vp), ("keep", vp), ("eps", vp), ("z", vp), ("mu_stack", vp), ("lv_stack", vp), ("dz", vp), ("dmu_stack", vp),
                ("dlv_stack", vp), ("dfeat", vp), ("dhw", ll), ("N", C.c_int), ("L", C.c_int), ("mask_mu", C.c_int),
                ("rng_used", vp), ("rng_stream", C.c_int)]


class DuseJob(C.Structure):        # == xh_duse_job
    _fields_ = [("p", vp * 10), ("out", vp * 4), ("g", vp * 10), ("gout", vp * 4), ("C", C.c_int)]


class SepJob(C.Structure):         # == xh_sep_job
    _fields_ = [("dw", vp), ("pw", vp), ("w", vp), ("gw", vp), ("g_dw", vp), ("g_pw", vp), ("C", C.c_int), ("K3", C.c_int)]


class HeadJob(C.Structure):        # == xh_head_job
    _fields_ = [("wf", vp), ("bf", vp), ("ws", vp), ("bs", vp), ("w", vp), ("b", vp), ("dwf", vp), ("dbf", vp), ("dws", vp),
                ("dbs", vp), ("gw", vp), ("gb", vp), ("Co", C.c_int), ("Cm", C.c_int), ("Ci", C.c_int)]


MULTI_MAX = 4         # == XH_LEVELS_MAX


class InAffineActArgs(C.Structure):          # == xh_in_affine_act_args
    _fields_ = [("x", vp), ("x_bs", ll), ("y", vp), ("y_bs", ll), ("N", C.c_int), ("C", C.c_int), ("DHW", ll), ("red", vp),
                ("act", C.c_int), ("slope", C.c_float), ("sc", vp), ("sh", vp), ("mean", vp), ("rstd", vp)]


class ActBwdReduceArgs(C.Structure):         # == xh_act_bwd_reduce_args
    _fields_ = [("dy", vp), ("dy_bs", ll), ("x", vp), ("x_bs", ll), ("N", C.c_int), ("C", C.c_int), ("DHW", ll), ("sc", vp), ("sh", vp),
                ("slope", C.c_float), ("red", vp)]


class InBwdApplyArgs(C.Structure):           # == xh_in_bwd_apply_args
    _fields_ = [("dy", vp), ("dy_bs", ll), ("x", vp), ("x_bs", ll), ("dx", vp), ("dx_bs", ll), ("N", C.c_int), ("C", C.c_int), ("DHW", ll),
                ("red", vp), ("mean", vp), ("rstd", vp), ("stat_rs", C.c_int), ("have_g", C.c_int), ("sc", vp), ("sh", vp),
                ("slope", C.c_float), ("accumulate", C.c_int)]


class Upsample2xInActArgs(C.Structure):      # == xh_upsample2x_in_act_args
    _fields_ = [("x", vp), ("x_bs", ll), ("y", vp), ("y_bs", ll), ("N", C.c_int), ("C", C.c_int), ("D", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("red", vp), ("slope", C.c_float), ("sc", vp), ("sh", vp), ("mean", vp), ("rstd", vp)]


class Upsample2xBwdArgs(C.Structure):        # == xh_upsample2x_bwd_act_reduce_args
    _fields_ = [("dy", vp), ("dy_bs", ll), ("dx", vp), ("dx_bs", ll), ("N", C.c_int), ("C", C.c_int), ("D", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("y0", vp), ("y0_bs", ll), ("sc", vp), ("sh", vp), ("slope", C.c_float), ("red", vp)]


VIL_FIELDS = ["norm_w", "proj_up", "conv_w", "conv_b", "q_w", "k_w", "v_w", "ig_w", "ig_b", "fg_w", "fg_b",
              "outnorm_w", "skip", "proj_down"]


class VilParams(C.Structure):
    _fields_ = [(n, vp) for n in VIL_FIELDS]


# name -> (restype, argtypes); mirrors include/xlstm_hved.h one to one
I, F = C.c_int, C.c_float
SIGNATURES = {
    "xh_abi_version": (I, []),
    "xh_set_option": (I, [I, I]),
    "xh_wgrad_plan_minmax": (C.c_double, [I, c_dp, C.POINTER(I), C.POINTER(I), I, C.POINTER(I)]),
    "xh_last_conv_kernel": (C.c_char_p, []),
    "xh_conv3d_fwd": (I, [vp, C.POINTER(ConvDesc), C.POINTER(ConvPtrs)]),
    "xh_conv3d_fwd_pair": (I, [vp, C.POINTER(ConvDesc), C.POINTER(ConvPtrs), C.POINTER(ConvDesc), C.POINTER(ConvPtrs)]),
    "xh_fanin_bytes": (ll, []),
    "xh_conv3d_workspace_bytes": (ll, [C.POINTER(ConvDesc)]),
    "xh_conv3d_fuses_norm_bwd": (I, [C.POINTER(ConvDesc)]),
    "xh_conv3d_supports_bcast": (I, [C.POINTER(ConvDesc)]),
    "xh_init_fold_fwd": (I, [vp, vp, ll, I, I, I, C.POINTER(vp * MAX_WPTR), F, vp, vp, vp, vp]),
    "xh_init_fold_bwd": (I, [vp, vp, ll, I, I, I, C.POINTER(vp * MAX_WPTR), F, vp, C.POINTER(vp * MAX_WPTR)]),
    "xh_conv3d_fuses_bn_finalize": (I, [C.POINTER(ConvDesc)]),
    "xh_conv3d_prepack": (I, [vp, I, vp, vp]),
    "xh_conv3d_prepack_table_bytes": (ll, []),
    "xh_conv3d_prepack_table": (I, [I, vp, vp, vp]),
    "xh_conv3d_prepack_run": (I, [vp, vp, I]),
    "xh_conv3d_wgrad_workspace_bytes": (ll, [C.POINTER(ConvDesc)]),
    "xh_conv3d_dgrad_s2": (I, [vp, C.POINTER(ConvDesc), C.POINTER(ConvPtrs)]),
    "xh_conv3d_wgrad": (I, [vp, C.POINTER(ConvDesc), C.POINTER(ConvPtrs), C.POINTER(vp * MAX_WPTR), C.POINTER(vp * MAX_WPTR)]),
    "xh_conv3d_wgrad_batch": (I, [vp, I, vp, vp, vp, vp]),
    "xh_moments": (I, [vp, I, vp, ll, I, I, ll, vp, ll]),
    "xh_moments2": (I, [vp, I, vp, ll, I, vp, ll, I, I, ll, vp, ll]),
    "xh_norm_finalize": (I, [vp, I, vp, I, I, ll, I, F, vp, vp, vp, vp, I, vp, vp, vp, vp]),
    "xh_affine_act": (I, [vp, I, vp, ll, vp, ll, I, I, ll, vp, vp, I, F]),
    "xh_in_affine_act": (I, [vp, I, vp, ll, vp, ll, I, I, ll, vp, I, F, vp, vp, vp, vp]),
    "xh_bn_affine_act": (I, [vp, I, I, vp, ll, vp, ll, I, I, ll, vp, F, vp, vp, vp, vp, I, I, F, vp, vp, vp, vp]),
    "xh_act_bwd_reduce": (I, [vp, I, vp, ll, vp, ll, I, I, ll, vp, vp, F, vp]),
    "xh_norm_bwd_coef": (I, [vp, I, vp, I, I, ll, I, vp, vp, vp, vp, vp, vp, vp, vp]),
    "xh_in_bwd_apply": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, I, ll, vp, vp, vp, I, I, vp, vp, F, I]),
    "xh_in_bwd_apply2": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, vp, ll, vp, ll, I, I, ll, vp, vp, vp, I]),
    "xh_norm_bwd_apply": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, I, ll, vp, vp, vp, I, vp, vp, F, I]),
    "xh_norm_bwd_fused": (I, [vp, I, I, vp, ll, vp, ll, vp, ll, I, I, ll, vp, I, vp, vp, vp, vp, vp]),
    "xh_dlast_fwd": (I, [vp, I, I, vp, vp, vp, I, I, I, I, I, I, I, I]),
    "xh_dlast_dgrad": (I, [vp, I, I, vp, vp, vp, I, I, I, I, I, I, I, I]),
    "xh_dlast_wgrad": (I, [vp, I, I, vp, vp, vp, F, I, I, I, I, I, I, I, I]),
    "xh_dconv_exact": (I, [vp, I, vp, vp, vp, vp, I, I, I, I, I, I, I, I, I, I, I]),
    "xh_norm_bwd_fused2": (I, [vp, I, I, vp, ll, vp, ll, vp, ll, I, I, I, ll, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "xh_bn_affine_act2": (I, [vp, I, I, vp, ll, vp, ll, I, I, I, ll, vp, F, vp, vp, vp, vp, vp, vp, vp, vp, I, I, F, vp, vp, vp, vp]),
    "xh_maxpool2_fwd": (I, [vp, I, vp, vp, I, I, I, I]),
    "xh_maxpool2_bwd": (I, [vp, I, vp, vp, vp, I, I, I, I, I]),
    "xh_upsample_trilinear_fwd": (I, [vp, I, vp, ll, vp, ll, I, I, I, I, I, I, I, I]),
    "xh_upsample_trilinear_bwd": (I, [vp, I, vp, ll, vp, ll, I, I, I, I, I, I, I, I, I]),
    "xh_upsample2x_in_act_fwd": (I, [vp, I, vp, ll, vp, ll, I, I, I, I, I, vp, F, vp, vp, vp, vp]),
    "xh_upsample2x_bwd_act_reduce": (I, [vp, I, vp, ll, vp, ll, I, I, I, I, I, vp, ll, vp, vp, F, vp]),
    "xh_conv1x1_multi": (I, [vp, I, vp, vp]),
    "xh_in_affine_act_multi": (I, [vp, I, I, vp]),
    "xh_act_bwd_reduce_multi": (I, [vp, I, I, vp]),
    "xh_in_bwd_apply_multi": (I, [vp, I, I, vp]),
    "xh_upsample2x_in_act_multi": (I, [vp, I, I, vp]),
    "xh_upsample2x_bwd_act_reduce_multi": (I, [vp, I, I, vp]),
    "xh_add": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, ll]),
    "xh_act_bwd": (I, [vp, I, vp, vp, vp, ll, I]),
    "xh_poe_fwd": (I, [vp, I, vp, vp, vp, vp, vp, vp, I, I, ll, I]),
    "xh_poe_bwd": (I, [vp, I, vp, vp, vp, vp, vp, vp, vp, I, I, ll, I]),
    "xh_poe_multi": (I, [vp, I, I, I, vp, vp]),
    "xh_philox_normal": (I, [vp, C.c_ulonglong, C.c_ulonglong, I, vp, ll, I]),
    "xh_channel_pool_fwd": (I, [vp, I, vp, ll, vp, ll, I, I, ll]),
    "xh_channel_pool_bwd": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, I, ll, I]),
    "xh_gate_fwd": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, I, ll]),
    "xh_gate_bwd": (I, [vp, I, vp, ll, vp, ll, vp, ll, vp, ll, vp, ll, I, I, ll, I, I]),
    "xh_channel_pool2_fwd": (I, [vp, I, vp, ll, I, vp, ll, I, vp, ll, I, ll]),
    "xh_channel_pool2_bwd": (I, [vp, I, vp, ll, I, vp, ll, I, vp, ll, vp, ll, I, vp, ll, I, I, ll]),
    "xh_gate2_fwd": (I, [vp, I, vp, ll, I, vp, ll, I, vp, ll, vp, ll, I, ll, vp]),
    "xh_gate2_bwd": (I, [vp, I, vp, ll, I, vp, ll, I, vp, ll, vp, ll, vp, ll, I, vp, ll, I, vp, ll, I, ll, I]),
    "xh_gate_maxpool_fwd": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, I, I, I, I, vp, I]),
    "xh_gate_maxpool_bwd": (I, [vp, I, vp, ll, vp, ll, vp, ll, vp, ll, vp, ll, I, I, I, I, I, I, I]),
    "xh_duse_gate_fwd": (I, [vp, I, vp, ll, vp, vp, ll, vp, ll, I, I, ll]),
    "xh_duse_gate_fwd_stats": (I, [vp, I, vp, ll, vp, vp, ll, vp, ll, I, I, ll, vp]),
    "xh_duse_gate_bwd": (I, [vp, I, vp, ll, vp, vp, ll, vp, ll, vp, ll, vp, ll, vp, I, I, ll, I]),
    "xh_duse_gate_bwd_fuses": (I, [I]),
    "xh_rank1_add": (I, [vp, I, vp, ll, vp, ll, vp, vp, I, I, ll]),
    "xh_duse_gate_fc_fwd": (I, [vp, I, vp, ll, vp, ll, vp, ll, I, ll] + [vp] * 11),
    "xh_rank1_add_fc": (I, [vp, I, vp, ll, vp, ll, vp, I, ll] + [vp] * 13),
    "xh_duse_fc_fwd": (I, [vp, vp, vp, ll, I, I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "xh_compose_atten_fwd": (I, [vp] * 9 + [I, I, I, I, vp, vp]),
    "xh_compose_atten_bwd": (I, [vp] * 7 + [I, I, I, I] + [vp] * 10),
    "xh_compose_multi": (I, [vp, I, I, vp, I, vp, I, vp, I, vp, vp, ll]),
    "xh_compose_duse_fwd": (I, [vp, C.POINTER(vp * 10), I, vp, vp, vp, vp]),
    "xh_compose_duse_bwd": (I, [vp, C.POINTER(vp * 10), I, vp, vp, vp, vp, C.POINTER(vp * 10)]),
    "xh_duse_fc_bwd": (I, [vp, vp, vp, ll, I, I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "xh_skr_tail_fwd": (I, [vp, I, vp, vp, vp, vp, vp, vp, I, I, ll]),
    "xh_skr_tail_bn_fwd": (I, [vp, I, vp, vp, vp, vp, vp, vp, vp, I, vp, vp, I, ll, vp, vp, vp, vp]),
    "xh_skr_tail_bwd": (I, [vp, I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, I, I, ll, I, vp]),
    "xh_pair_sums": (I, [vp, I, vp, ll, I, vp, ll, F, I, I, ll, I, F, vp]),
    "xh_lincomb": (I, [vp, I, vp, ll, I, vp, ll, F, vp, ll, I, I, ll, vp, vp, vp, vp, I]),
    "xh_loss_finalize": (I, [vp, I, vp, I, I, C.c_double, C.c_double, vp, vp, vp]),
    "xh_kld_fwd": (I, [vp, I, vp, vp, vp, I, I, ll, vp]),
    "xh_kld_bwd": (I, [vp, I, vp, vp, vp, I, I, ll, F, vp, vp, vp]),
    "xh_nested_weight": (I, [vp, I, vp, ll, vp, ll, I, ll]),
    "xh_fill": (I, [vp, I, vp, ll, F, vp]),
    "xh_scalar_lincomb": (I, [vp, I, vp, vp, vp, vp]),
    "xh_scalar_fanout": (I, [vp, I, vp, vp, vp, vp]),
    "xh_multi_sum": (I, [vp, I, I, vp, vp, vp, vp, vp]),
    "xh_multi_fill": (I, [vp, I, I, vp, vp, vp, vp]),
    "xh_dconv_cl": (I, [vp, I, I, I, I, vp, vp, vp, vp, vp, I, I, I, I, I, I, I, I, I, I, F, vp]),
    "xh_dconv_wgrad_cl": (I, [vp, I, I, I, vp, vp, vp, I, I, I, I, I, I, I, I, I]),
    "xh_dconv_pack": (I, [vp, I, I, I, vp, vp, I, I, I, I]),
    "xh_dconv_unpack_grad": (I, [vp, I, vp, vp, I, I, I, I]),
    "xh_cl_from_ncdhw": (I, [vp, I, vp, ll, I, vp, ll, I, vp, I, I, ll]),
    "xh_cl_to_ncdhw": (I, [vp, I, vp, I, vp, ll, I, vp, ll, I, I, ll]),
    "xh_cl_affine_act": (I, [vp, I, vp, vp, vp, vp, F, I, I, ll]),
    "xh_cl_act_bwd": (I, [vp, I, I, vp, vp, vp, vp, vp, F, vp, vp, vp, vp, I, I, ll]),
    "xh_vil_workspace_floats": (ll, [I, I, I]),
    "xh_vil_fwd": (I, [vp, I, vp, vp, vp, I, I, I, I, I, C.POINTER(VilParams), vp]),
    "xh_vil_bwd": (I, [vp, I, vp, vp, vp, vp, I, I, I, I, C.POINTER(VilParams), C.POINTER(VilParams), vp]),
}

_lib = None


def load():
    """Loads the shared library (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  The HIP library is the only compute path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        # A/B measurements without code changes: XH_SET_OPTIONS="16=0,2=128" applies xh_set_option(key, value) at load time
        for kv in filter(None, os.environ.get("XH_SET_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            check(lib.xh_set_option(int(k), int(v)), f"xh_set_option({kv})")
    return _lib


ERRORS = {-1: "bad argument / unsupported shape combination", -2: "unsupported dtype", -3: "HIP launch failed"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")

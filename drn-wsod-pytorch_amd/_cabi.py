"""ctypes binding of include/drn_wsod.h (libdrn_wsod_hip.so).  There is NO fallback: if the HIP
library is missing or a call fails, this raises — the product path never routes through a CPU or
eager-PyTorch substitute."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdrn_wsod_hip.so")

F32, BF16, FP8 = 0, 1, 2

_T = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float, "Q": ctypes.c_ulonglong}

# signature strings follow include/drn_wsod.h argument for argument
_SIGS = {
    "drn_preprocess_nhwc": "piiipiiippip",
    "drn_resize_bilinear_u8": "piiipiippippiip",
    "drn_conv2d_nhwc": "pppppp" + "iiiiiiiiii" + "lll" + "iip",
    "drn_conv2d_nhwc_q": "pppppp" + "iiiiiiiiii" + "lll" + "iiiifp",
    "drn_conv3x3_pw_nhwc": "pppp" + "i" + "pppp" + "p" + "iii" + "ll" + "f" + "ii" + "p",
    "drn_maxpool2x2_nhwc": "ppiiiiiip",
    "drn_roi_pool_nhwc": "pppppp" + "iiiiii" + "f" + "ll" + "iiiiip",
    "drn_roi_pool_nhwc_t": "pppppp" + "iiiiii" + "f" + "ll" + "iiiiiip",
    "drn_roi_pool_nhwc_ws": "pppppp" + "iiiiii" + "f" + "ll" + "iiiiii" + "plp",
    "drn_tta_accumulate": "ppppllfffiip",
    "drn_pcl_adjacency": "pifpp",
    "drn_pcl_refine": "pipiipipppip" + "ppppppppppi" + "ppp",
    "drn_im2col_t": "pp" + "iiiiiiiiii" + "lip",
    "drn_maxpool2x2_bwd_nhwc": "pppiiiiiip",
    "drn_add": "ppplip",
    "drn_stage_heads_inputs": "ppippip",
    "drn_stage_rois": "ppfpppip",
    "drn_roi_pool_backward_nhwc": "ppppp" + "iiiiii" + "f" + "l" + "iiiip",
    "drn_transpose2d": "ppiilliip",
    "drn_gemm_nt": "pppiiillliiilip",
    "drn_gemm_tn": "ppp" + "iiii" + "lll" + "iilip",
    "drn_gemm_tn_sgd": "ppp" + "iiii" + "lll" + "ppplp" + "fifp",
    "drn_gemm_nt_pair": ("ppp" + "iii" + "lll" + "ili") * 2 + "p",
    "drn_gemm_set_tile": "i",
    "drn_tune": "ii",
    "drn_bias_act_fwd": "pilppQpfplpliiliip",
    "drn_linear_act_fwd": "pppp" + "Qpf" + "plpl" + "iii" + "ll" + "ip",
    "drn_counter_add": "pQp",
    "drn_colsum_reduce": "piipip",
    "drn_bias_act_bwd": "pilppppfplplppiiiip",
    "drn_bias_act_bwd_splits": "pililppppfplplppiiiip",
    "drn_cast2d": "ppiilliip",
    "drn_wsddn_fwd_bwd": "pliiipippppppl" + "pi" + "ifp",
    "drn_oicr_targets": "plpii" + "ppi" + "ppi" + "pi" + "ppi" + "pppppp" + "p",
    "drn_oicr_refine_chain": "plpiipl" + "ppipp" + "ip" + "ppi" + "ppppppp" + "plpp" + "ifp",
    "drn_gemm_nt_act_bwd": "ppiiillppfplplppip",
    "drn_mil_oicr_losses": "pillp" + "Qp" + "pl" + "iii" + "pi" + "pppppp" + "ii" + "pi" + "ppp" + "i" + "pp" + "i" + "ppppppp"
                           + "pp" + "pl" + "ifp",
    "drn_softmax_ce": "pliipppplppifp",
    "drn_mean_softmax": "plpiipiip",
    "drn_box_reg_loss": "pliippppplppifp",
    "drn_apply_deltas": "plppiipfp",
    "drn_sum_small": "pifpp",
    "drn_sgd_step": "pppilpipififp",
    "drn_sgd_step_block": "pppilpip" + "iiiil" + "fifp",
    "drn_detect_topk": "ppiii" + "ffff" + "i" + "pl" + "i" + "ppp",
    "drn_detect_gather": "plippippppp",
    "drn_csc_cpg": "piiiiippp",
    "drn_csc_weights": "piifpipiiifppp",
    "drn_csc_loss": "pliiiippppiiipplp",
    "drn_trunk_shapes": "p" + "iiiiiiii" + "pp",
    "drn_trunk_forward": "piiip" + "iiiii" + "p",
}


class DrnTrunkOp(ctypes.Structure):
    """include/drn_wsod.h `DrnTrunkOp`, field for field"""
    _fields_ = [("kind", ctypes.c_int), ("src", ctypes.c_int), ("dst", ctypes.c_int), ("res", ctypes.c_int),
                ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("ksize", ctypes.c_int), ("stride", ctypes.c_int),
                ("pad", ctypes.c_int), ("dil", ctypes.c_int), ("relu", ctypes.c_int), ("ldw", ctypes.c_long),
                ("dtype", ctypes.c_int), ("out_dtype", ctypes.c_int), ("res_dtype", ctypes.c_int),
                ("res_mult", ctypes.c_float)]


TRUNK_MAX_SLOTS = 16

_lib = None


class DrnError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DrnError(
                "HIP library %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback on the product path)" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = [_T[c] for c in sig]
            fn.restype = ctypes.c_int
        _lib.drn_detect_workspace_bytes.argtypes = [ctypes.c_int]
        _lib.drn_detect_workspace_bytes.restype = ctypes.c_long
        _lib.drn_gemm_nt_main_cols.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.drn_gemm_nt_main_cols.restype = ctypes.c_long
        _lib.drn_roi_pool_workspace_bytes.argtypes = [ctypes.c_int] * 10
        _lib.drn_roi_pool_workspace_bytes.restype = ctypes.c_long
        for kv in filter(None, os.environ.get("DRN_TUNE", "").split(",")):  # A/B runs: DRN_TUNE="5=0,4=1024" (drn_tune knobs)
            k, v = kv.split("=")
            _lib.drn_tune(int(k), int(v))
    return _lib


def exported_symbols():
    return sorted(list(_SIGS) + ["drn_detect_workspace_bytes", "drn_gemm_nt_main_cols", "drn_roi_pool_workspace_bytes"])


_ERR = {-1: "invalid argument", -2: "kernel launch failure", -3: "unsupported"}


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise DrnError("%s failed: %s (%d)" % (name, _ERR.get(rc, "error"), rc))


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "drn ops need device tensors (no CPU path)"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream (the capture stream while a hipGraph is being captured).  The raw getter
    skips building a torch.cuda.Stream object: ~8 us -> < 1 us per launch on the eager step's ~100 launches."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dt(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float8_e4m3fn:
        return FP8
    raise DrnError("unsupported dtype %s" % dtype)


def host_floats(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def host_ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])

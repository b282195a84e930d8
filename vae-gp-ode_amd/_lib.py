"""ctypes binding of libgpode_hip.so (C ABI: include/gpode.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  Build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C vae-gp-ode_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgpode_hip.so')

_c_float_p = ctypes.c_void_p  # device pointers travel as integers
_i = ctypes.c_int
_sz_p = ctypes.POINTER(ctypes.c_size_t)

# name -> (restype, argtypes); mirrors include/gpode.h one to one
SIGNATURES = {
    'gpode_version': (ctypes.c_char_p, []),
    'gpode_last_error': (ctypes.c_char_p, []),
    'gpode_supported': (_i, [_i, _i, _i]),
    'gpode_cache_sizes': (_i, [_i, _i, _i, _i, _i, _sz_p, _sz_p]),
    'gpode_cache_build_fwd': (_i, [_i] * 5 + [_c_float_p] * 20),
    'gpode_cache_bwd_sizes': (_i, [_i, _i, _i, _i, _i, _sz_p]),
    'gpode_cache_build_bwd': (_i, [_i] * 5 + [_c_float_p] * 13 + [_i, ctypes.c_void_p]),
    'gpode_cache_bwd_prepare': (_i, [_i] * 5 + [_c_float_p] * 2 + [ctypes.c_void_p]),
    'gpode_cache_info': (_i, [_c_float_p, ctypes.POINTER(_i), ctypes.c_void_p]),
    'gpode_cache_pivots': (_i, [_c_float_p, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]),
    'gpode_set_backward_solves': (_i, [_i]),
    'gpode_kernel_matrix': (_i, [_i, _i, _i, _c_float_p, _c_float_p, _c_float_p, _i, _c_float_p, _i, _c_float_p, ctypes.c_void_p]),
    'gpode_conditional_ws': (_i, [_i, _i, _i, _i, _sz_p]),
    'gpode_conditional': (_i, [_i, _i, _i, _i] + [_c_float_p] * 5 + [_i, _c_float_p, _i] + [_c_float_p] * 3 + [ctypes.c_void_p]),
    'gpode_svgp_kl_fwd': (_i, [_i, _i, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p]),
    'gpode_svgp_kl_bwd': (_i, [_i, _i] + [_c_float_p] * 5 + [ctypes.c_void_p]),
    'gpode_rhs_fwd': (_i, [_i] * 5 + [_c_float_p, _c_float_p, _i, _c_float_p, _i, ctypes.c_void_p]),
    'gpode_rollout_fwd': (_i, [_i] * 7 + [_c_float_p, _c_float_p, _c_float_p, _i, _i, _c_float_p, _c_float_p, ctypes.c_void_p]),
    'gpode_rollout_bwd': (_i, [_i] * 7 + [_c_float_p] * 4 + [_i, _i, _c_float_p, _c_float_p, ctypes.c_void_p]),
    'gpode_rhs_vjp': (_i, [_i] * 5 + [_c_float_p, _c_float_p, _c_float_p, _i, _c_float_p, ctypes.c_void_p]),
    'gpode_param_grad': (_i, [_i] * 5 + [_c_float_p, _c_float_p, _c_float_p, _i, _c_float_p, _i, _c_float_p, _i, ctypes.c_void_p]),
    # Monte-Carlo draws batched into one call (a leading draw axis on every per-draw operand; `ndraws` follows S)
    'gpode_cache_sizes_n': (_i, [_i] * 6 + [_sz_p, _sz_p]),
    'gpode_cache_build_fwd_n': (_i, [_i] * 6 + [_c_float_p] * 20),
    'gpode_rollout_fwd_n': (_i, [_i] * 8 + [_c_float_p, _c_float_p, _c_float_p, _i, _i, _c_float_p, _c_float_p, ctypes.c_void_p]),
    'gpode_rollout_bwd_n': (_i, [_i] * 8 + [_c_float_p] * 4 + [_i, _i, _c_float_p, _c_float_p, ctypes.c_void_p]),
    'gpode_rollout_bwd_pgrad_chunks': (_i, [_i] * 8),
    'gpode_rollout_bwd_pgrad_n': (_i, [_i] * 8 + [_c_float_p] * 4 + [_i, _i, _c_float_p, _c_float_p, _c_float_p, _i, _c_float_p, ctypes.c_void_p]),
    'gpode_param_grad_n': (_i, [_i] * 6 + [_c_float_p, _c_float_p, _c_float_p, _i, _c_float_p, _i, _c_float_p, _i, ctypes.c_void_p]),
    'gpode_cache_bwd_sizes_n': (_i, [_i] * 6 + [_sz_p]),
    'gpode_cache_build_bwd_n': (_i, [_i] * 6 + [_c_float_p] * 13 + [_i, ctypes.c_void_p]),
    'gpode_cache_bwd_prepare_n': (_i, [_i] * 6 + [_c_float_p] * 2 + [ctypes.c_void_p]),
    # the kernel's own methods (kern.build_cache / sample_freq, kern.compute_nu, kern.f_update)
    'gpode_kern_scratch': (ctypes.c_size_t, [_i] * 5),
    'gpode_kern_cache': (_i, [_i] * 4 + [_c_float_p] * 8 + [ctypes.c_void_p]),
    'gpode_compute_nu_ws': (_i, [_i] * 4 + [_sz_p]),
    'gpode_compute_nu': (_i, [_i] * 4 + [_c_float_p] * 5 + [ctypes.c_void_p]),
    'gpode_f_update': (_i, [_i] * 4 + [_c_float_p] * 5 + [_i, _c_float_p, _c_float_p, ctypes.c_void_p]),
}

_sz = ctypes.c_size_t
_f = ctypes.c_float
_vp = ctypes.c_void_p
SIGNATURES.update({
    'gpode_conv2d_fwd': (_i, [_c_float_p] * 4 + [_i] * 10 + [_vp]),
    'gpode_conv2d_bwd_data': (_i, [_c_float_p] * 4 + [_i] * 10 + [_vp]),
    'gpode_conv2d_fwd_bs': (_i, [_c_float_p, _sz] + [_c_float_p] * 3 + [_i] * 10 + [_vp]),
    'gpode_conv2d_bwd_weight_bs': (_i, [_c_float_p, _sz] + [_c_float_p] * 4 + [_i] * 10 + [_vp]),
    'gpode_convT_fwd_stats_scratch': (_sz, [_i]),
    'gpode_convT_fwd_stats': (_i, [_c_float_p] * 5 + [_i] * 10 + [_c_float_p] * 6 + [_vp, _f, _f, _c_float_p, _c_float_p, _i, _vp]),
    'gpode_conv_wgrad_scratch': (_sz, [_i, _i, _i, _i]),
    'gpode_conv2d_bwd_weight': (_i, [_c_float_p] * 5 + [_i] * 10 + [_vp]),
    'gpode_bn_scratch': (_sz, [_i, _i]),
    'gpode_bn_fwd': (_i, [_c_float_p] * 8 + [_vp, _f, _f, _i, _i, _i, _i, _c_float_p, _vp]),
    'gpode_bn_bwd': (_i, [_c_float_p] * 10 + [_i, _i, _i, _i, _c_float_p, _vp]),
    'gpode_conv2d_bwd_data_bn': (_i, [_c_float_p] * 5 + [_i] * 10 + [_vp]),
    'gpode_conv2d_bwd_weight_bn': (_i, [_c_float_p] * 6 + [_i] * 10 + [_vp]),
    'gpode_bn_stats': (_i, [_c_float_p] * 7 + [_vp, _f, _f, _c_float_p, _i, _i, _i, _c_float_p, _vp]),
    'gpode_bn_moments': (_i, [_c_float_p, _c_float_p, _i, _i, _i, _c_float_p, _vp]),
    'gpode_bn_finalize': (_i, [_c_float_p, _i] + [_c_float_p] * 6 + [_vp, _f, _f, _c_float_p, _i, _vp]),
    'gpode_bn_apply': (_i, [_c_float_p, _c_float_p, _c_float_p, _i, _i, _i, _i, _vp]),
    'gpode_bn_bwd_sums': (_i, [_c_float_p] * 7 + [_i, _i, _i, _i, _c_float_p, _vp]),
    'gpode_bn_bwd_apply': (_i, [_c_float_p] * 8 + [_i, _f] + [_c_float_p] * 4 + [_i, _i, _i, _i, _c_float_p, _vp]),
    'gpode_defer_reductions': (None, [_i]),
    'gpode_flush_reductions': (_i, [_vp]),
    'gpode_dec10_bn_scratch_floats': (_i, []),
    'gpode_dec10_bn_wgrad_scratch_floats': (_i, []),
    'gpode_dec10_bn_bwd_sums_wgrad': (_i, [_c_float_p] * 10 + [_i, _c_float_p, _c_float_p, _vp]),
    'gpode_dec10_bn_bwd_sums': (_i, [_c_float_p] * 8 + [_i, _c_float_p, _vp]),
    'gpode_dec10_bn_bwd_apply': (_i, [_c_float_p] * 9 + [_i, _f] + [_c_float_p] * 4 + [_i, _c_float_p, _vp]),
    'gpode_bn_eval': (_i, [_c_float_p] * 6 + [_f, _c_float_p, _i, _i, _i, _i, _vp]),
    'gpode_chan_sum': (_i, [_c_float_p, _c_float_p, _i, _i, _i, _c_float_p, _vp]),
    'gpode_act_fwd': (_i, [_c_float_p, _c_float_p, _sz, _i, _vp]),
    'gpode_act_bwd': (_i, [_c_float_p, _c_float_p, _c_float_p, _sz, _i, _vp]),
    'gpode_linear_fwd': (_i, [_c_float_p] * 4 + [_i, _i, _i, _vp]),
    'gpode_linear_bwd_scratch': (_sz, [_i, _i, _i]),
    'gpode_linear_bwd': (_i, [_c_float_p] * 6 + [_i, _i, _i, _c_float_p, _vp]),
    'gpode_linear_relu_fwd': (_i, [_c_float_p] * 4 + [_i, _i, _i, _vp]),
    'gpode_linear_relu_bwd': (_i, [_c_float_p] * 6 + [_i, _i, _i, _vp]),
    'gpode_loglik_fwd': (_i, [_c_float_p] * 3 + [_sz, _sz, _vp]),
    'gpode_loglik_bwd': (_i, [_c_float_p] * 4 + [_sz, _sz, _vp]),
    'gpode_loglik_rowsum_fwd': (_i, [_c_float_p] * 3 + [_sz, _sz, _sz, _vp]),
    'gpode_noise_fill': (_i, [_vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_ulonglong, _vp, _vp]),
    'gpode_reparam_kl_fwd': (_i, [_c_float_p, _c_float_p, _i, _c_float_p, _c_float_p, _c_float_p, _i, _i, _vp]),
    'gpode_reparam_kl_bwd': (_i, [_c_float_p] * 4 + [_i] + [_c_float_p] * 3 + [_i, _i, _i, _vp]),
    'gpode_elbo_all_fwd_kl': (_i, [_c_float_p, _i, _i, _c_float_p, _i, _c_float_p, _i, _i, _i, _i, _c_float_p, _c_float_p, _f, _c_float_p, _vp]),
    'gpode_elbo_all_bwd_ll_kl': (_i, [_c_float_p] * 4 + [_i, _i, _i, _i, _c_float_p, _c_float_p, _f, _c_float_p, _c_float_p, _i, _c_float_p, _i] + [_c_float_p] * 5 + [_sz, _sz, _vp]),
    'gpode_reparam_fwd': (_i, [_c_float_p, _c_float_p, _i, _c_float_p, _c_float_p, _i, _i, _vp]),
    'gpode_reparam_bwd': (_i, [_c_float_p, _c_float_p, _i, _c_float_p, _c_float_p, _c_float_p, _i, _i, _i, _vp]),
    'gpode_normal_kl_fwd': (_i, [_c_float_p, _c_float_p, _i, _c_float_p, _i, _i, _vp]),
    'gpode_normal_kl_bwd': (_i, [_c_float_p, _c_float_p, _c_float_p, _i, _c_float_p, _c_float_p, _i, _i, _i, _vp]),
    'gpode_sigmoid_loglik_splits': (_i, [_sz, _sz]),
    'gpode_sigmoid_loglik_fwd': (_i, [_c_float_p] * 4 + [_sz, _sz, _sz, _i, _vp]),
    'gpode_sigmoid_loglik_bwd': (_i, [_c_float_p] * 4 + [_sz, _sz, _sz, _vp]),
    'gpode_elbo_all_fwd': (_i, [_c_float_p, _i, _i, _c_float_p, _c_float_p, _i, _i, _i, _i, _c_float_p, _c_float_p, _f, _c_float_p, _vp]),
    'gpode_elbo_all_bwd': (_i, [_c_float_p] * 4 + [_i, _c_float_p, _c_float_p, _i, _i, _i, _i, _c_float_p, _c_float_p, _f] + [_c_float_p] * 5 + [_vp]),
    'gpode_elbo_all_bwd_ll': (_i, [_c_float_p] * 4 + [_i, _c_float_p, _c_float_p, _i, _i, _i, _i, _c_float_p, _c_float_p, _f] + [_c_float_p] * 8 + [_sz, _sz, _vp]),
    'gpode_elbo_fwd': (_i, [_c_float_p, _i, _c_float_p, _i, _c_float_p, _f, _c_float_p, _vp]),
    'gpode_elbo_bwd': (_i, [_c_float_p, _i, _i, _f, _c_float_p, _c_float_p, _c_float_p, _vp]),
    'gpode_gather_multi': (_i, [_vp, _vp, _i, ctypes.c_longlong, _c_float_p, _vp]),
    'gpode_adam_multi': (_i, [_vp, _vp, _vp, _vp, _vp, _i, ctypes.c_longlong, _f, _f, _f, _f, _i, _vp, _vp]),
    'gpode_loglik_rowsum_bwd': (_i, [_c_float_p] * 4 + [_sz, _sz, _sz, _vp]),
})

_lib = None


class GpodeError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpodeError('libgpode_hip.so not found at %s -- build the HIP extension first '
                         '(__graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point; raise GpodeError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise GpodeError('%s failed (%d): %s' % (name, rc, lib.gpode_last_error().decode()))
    return rc

"""ctypes binding of libhypad_hip.so (the C ABI declared in include/hypad.h).

The product path has no CPU fallback: importing this module without the built
library raises, and every call checks its status code.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HYPAD_DEV_LIB=1 (scripts/diag_*.py only): the development build with in-kernel stamps and the hypad_diag_* entry points
LIB_PATH = os.path.join(_HERE, "lib", "libhypad_hip_dev.so" if os.environ.get("HYPAD_DEV_LIB") == "1" else "libhypad_hip.so")
if os.environ.get("HYPAD_LIB_PATH"):        # A/B timing of two builds of the same library (scripts/ab_libs.sh)
    LIB_PATH = os.environ["HYPAD_LIB_PATH"]


class HypadError(RuntimeError):
    pass


ABI_VERSION = 7            # include/hypad.h: HYPAD_ABI_VERSION -- the struct layouts below are this version's


def _load():
    if not os.path.exists(LIB_PATH):
        raise HypadError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m hypad_amd.build). "
            "hypad_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    # (an override -- HYPAD_LIB_PATH / HYPAD_DEV_LIB, A/B scripts -- may point at a stale build whose structs differ: refuse it
    # before any symbol is bound or called)
    try:
        lib.hypad_abi_version.restype = c_int
        got = lib.hypad_abi_version()
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise HypadError(f"{LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild it (python -m hypad_amd.build --force)")
    return lib


lib = _load()

NET_ENCODER, NET_DECODER, NET_CRITIC_X, NET_CRITIC_Z = 0, 1, 2, 3
ACT_NONE, ACT_TANH, ACT_LEAKY02 = 0, 1, 2
COMB = {"sum": 0, "mult": 1, "uncertainty": 2, "critic": 3, "critic_uncertainty": 4, "sum_uncertainty": 5, "rec": 6,
        "rec_uncertainty": 7, "eucl_mult": 8, "eucl_sum": 9}


class Dropout(Structure):
    _fields_ = [("train_mode", c_int), ("masks", c_void_p), ("seed", c_uint64), ("offset", c_uint64)]


class Dims(Structure):
    _fields_ = [("signal_shape", c_int), ("latent_dim", c_int), ("batch", c_int), ("hyperbolic", c_int), ("n_signals", c_int),
                ("first_signal", c_int)]                       # ABI 5: stream number of model 0 (defaults to 0)


class Nets(Structure):
    _fields_ = [("enc", c_void_p), ("dec", c_void_p), ("cx", c_void_p), ("cz", c_void_p)]


class TrainState(Structure):
    _fields_ = [("params", Nets), ("exp_avg", Nets), ("exp_avg_sq", Nets), ("counters", c_void_p),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("gen_weight_decay", c_float), ("gen_stabilize", c_int)]


class IterIO(Structure):
    _fields_ = [("x", c_void_p), ("x_signal_stride", c_int64), ("x_row_stride", c_int64), ("row_index", c_void_p), ("z", c_void_p), ("alpha", c_void_p),
                ("drop", Dropout), ("losses", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class EpochNoise(Structure):
    _fields_ = [("z_cx", c_void_p), ("alpha_cx", c_void_p), ("z_cz", c_void_p), ("alpha_cz", c_void_p), ("z_gen", c_void_p),
                ("masks_cx", c_void_p), ("masks_cz", c_void_p), ("masks_gen", c_void_p)]


class EpochIO(Structure):
    _fields_ = [("x", c_void_p), ("x_signal_stride", c_int64), ("x_row_stride", c_int64), ("row_index", c_void_p), ("n_batches", c_int),
                ("n_critics", c_int), ("train_mode", c_int), ("seed", c_uint64), ("losses", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("noise", POINTER(EpochNoise)), ("flags", c_int),
                ("aux_streams", POINTER(c_void_p)), ("n_aux_streams", c_int),      # ABI 4: streams for the generator phase's model groups
                ("row_index_signal_stride", c_int64),                              # ABI 5: a row_index plane per signal (0: one shared plane)
                ("enc_table", c_void_p), ("enc_table_rows", c_int64)]              # ABI 6: encoder(x) once per window row (NULL: once per pass)


STATS_WORKSPACE_BYTES = 256 * 5 * 8  # HYPAD_STATS_WORKSPACE_BYTES
EPOCH_PER_ITERATION = 1            # hypad_epoch_io.flags (include/hypad.h: HYPAD_EPOCH_*)
EPOCH_NO_PRODUCERS, EPOCH_ID_ORDER, EPOCH_CLEAR_TILES, EPOCH_PER_MINIBATCH, EPOCH_DW_COLOC, EPOCH_DW_SPREAD = 2, 4, 8, 16, 32, 64
EPOCH_TEST_GIVE_UP_SHIFT = 8


class RecordInfo(Structure):
    _fields_ = [("offset_floats", c_int64), ("record_floats", c_int), ("row_stride", c_int), ("mask_offset_floats", c_int),
                ("mask_row_stride", c_int), ("n_layers", c_int), ("in_dim", c_int)]


P = c_void_p
_SIGS = {
    "hypad_abi_version": (c_int, []),
    "hypad_error_string": (c_char_p, [c_int]),
    "hypad_limits": (None, [POINTER(c_int), POINTER(c_int)]),
    "hypad_param_count": (c_int, [c_int, c_int, c_int, c_int]),
    "hypad_param_tensors": (c_int, [c_int, c_int]),
    "hypad_param_info": (c_int, [c_int, c_int, c_int, c_int, c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "hypad_expmap0_fwd": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_expmap0_bwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_logmap0_fwd": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_logmap0_bwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_mobius_add_fwd": (c_int, [P, P, P, c_int64, c_int, c_int64, P]),
    "hypad_mobius_add_bwd": (c_int, [P, P, P, P, P, c_int64, c_int, c_int64, P]),
    "hypad_project_fwd": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_project_bwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_mobius_head_fwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_mobius_head_bwd": (c_int, [P, P, P, P, P, c_int64, c_int, P]),
    "hypad_mobius_linear_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "hypad_mobius_linear_fwd": (c_int, [P, P, P, P, P, c_int64, c_int, c_int, P]),
    "hypad_mobius_linear_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, c_int64, c_int, c_int, P]),
    "hypad_poincare_rowdist_fwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_poincare_rowdist_bwd": (c_int, [P, P, P, P, P, c_int64, c_int, P]),
    "hypad_hyper_loss_fwd": (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    "hypad_hyper_loss_bwd": (c_int, [P, P, c_float, P, P, c_int64, c_int, c_int, P]),
    "hypad_poincare_pairdist_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "hypad_column_sum": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_linear_act_fwd": (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    "hypad_linear_act_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    "hypad_lstm_bidir_fwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "hypad_lstm_bidir_bwd": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "hypad_lstm_seq_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "hypad_lstm_bidir_seq_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int64, c_int, c_int, c_void_p, c_size_t, P]),
    "hypad_lstm_bidir_seq_fwd_train": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int64, c_int, c_int, c_void_p, c_size_t, P]),
    "hypad_lstm_seq_bwd_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "hypad_lstm_bidir_seq_bwd": (c_int, [P] * 21 + [c_int, c_int64, c_int, c_int, c_void_p, c_size_t, P]),
    "hypad_encoder_fwd": (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    "hypad_decoder_fwd": (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int, POINTER(Dropout), P]),
    "hypad_critic_x_fwd": (c_int, [P, P, P, c_int64, c_int, c_int, POINTER(Dropout), P]),
    "hypad_critic_z_fwd": (c_int, [P, P, P, c_int64, c_int, POINTER(Dropout), P]),
    "hypad_score_forward": (c_int, [P, P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    "hypad_train_workspace_bytes": (c_size_t, [POINTER(Dims)]),
    "hypad_epoch_workspace_bytes": (c_size_t, [POINTER(Dims), c_int, c_int]),
    "hypad_score_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "hypad_score_forward_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "hypad_pack_generator": (c_int, [POINTER(Dims), POINTER(TrainState), c_void_p, c_size_t, c_void_p]),
    "hypad_packed_region": (c_int, [POINTER(Dims), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "hypad_critic_x_iteration": (c_int, [POINTER(Dims), POINTER(TrainState), POINTER(IterIO), P]),
    "hypad_critic_z_iteration": (c_int, [POINTER(Dims), POINTER(TrainState), POINTER(IterIO), P]),
    "hypad_decoder_iteration": (c_int, [POINTER(Dims), POINTER(TrainState), POINTER(IterIO), P]),
    "hypad_train_epoch": (c_int, [POINTER(Dims), POINTER(TrainState), POINTER(EpochIO), P]),
    "hypad_epoch_shuffles": (c_int, [P, c_int, c_int, c_int, c_uint64, P, P]),
    "hypad_epoch_shuffles_signals": (c_int, [P, c_int64, c_int, c_int, P, c_int, c_int, c_uint64, P, P]),
    "hypad_host_mt19937_normal": (c_int, [P, POINTER(c_int), POINTER(c_int), POINTER(c_double), POINTER(c_void_p), c_int, c_int64, c_int64]),
    "hypad_host_torch_mt19937_uniform": (c_int, [P, c_size_t, P, c_int64]),
    "hypad_host_mt19937_normal_mt": (c_int, [P, POINTER(c_int), POINTER(c_int), POINTER(c_double), POINTER(c_void_p), c_int, c_int64, c_int64, c_int]),
    "hypad_epoch_status": (c_int, [POINTER(TrainState), POINTER(c_int), P]),
    "hypad_epoch_restore": (c_int, [POINTER(Dims), POINTER(TrainState), c_void_p, c_size_t, P]),
    "hypad_critic_phase_persistent": (c_int, [POINTER(Dims)]),
    "hypad_critic_phase_producers": (c_int, [POINTER(Dims), c_int]),
    "hypad_epoch_record_info": (c_int, [POINTER(Dims), c_int, c_int, c_int, POINTER(RecordInfo)]),
    "hypad_rng_fill": (c_int, [c_int, c_uint64, c_uint32, c_uint32, c_uint32, c_float, P, c_int64, P]),
    "hypad_critic_z_seed": (c_uint64, [c_uint64]),
    "hypad_profile_iteration": (c_int, [c_int, POINTER(Dims), POINTER(TrainState), POINTER(IterIO), POINTER(c_float), c_int, P]),
    "hypad_adam_step": (c_int, [P, P, P, P, c_int64, c_int, c_float, c_float, c_float, c_float, c_float, P]),
    "hypad_radam_step": (c_int, [P, P, P, P, c_int64, c_int64, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, P]),
    "hypad_unroll_median": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_unroll_true": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_unroll_true_f32": (c_int, [P, c_int64, P, c_int64, c_int, P]),
    "hypad_point_error": (c_int, [P, P, P, c_int64, P]),
    "hypad_area_error": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_dtw_error": (c_int, [P, P, P, c_int64, c_int, P]),
    "hypad_rolling_workspace_bytes": (c_size_t, [c_int64]),
    "hypad_rolling_mean": (c_int, [P, P, P, c_int64, c_int, c_int64, P, c_size_t, P]),
    "hypad_zscore_clip": (c_int, [P, P, c_int64, P, c_size_t, P]),
    "hypad_kde_mode": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_critic_zscore": (c_int, [P, c_double, c_double, P, c_int64, P, c_size_t, P]),
    "hypad_quantile_workspace_bytes": (c_size_t, []),
    "hypad_quantiles": (c_int, [P, c_int64, P, c_int, P, P, c_size_t, P]),
    "hypad_critic_score_workspace_bytes": (c_size_t, []),
    "hypad_critic_score": (c_int, [P, P, c_int64, P, c_size_t, P]),
    "hypad_row_norms": (c_int, [P, P, c_int64, c_int, P]),
    "hypad_combine_scores": (c_int, [c_int, P, P, P, P, c_int64, P]),
}
EXPORTS = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = the library does not match include/hypad.h
    _fn.restype = _res
    _fn.argtypes = _args


def first_order_only(backward):
    """Decorator for the backward of a torch.autograd.Function whose gradient kernels are not themselves differentiable.  Under
    ``create_graph=True`` (grad mode is enabled while the engine runs backward) the returned gradients would be constants: a
    reference-style gradient penalty built from them (train.py:72-93) would train with a zero second-order term and no error
    anywhere.  Raise instead.  (``once_differentiable`` does not: with a constant incoming gradient it returns detached
    tensors silently.)"""
    import functools

    @functools.wraps(backward)
    def guarded(ctx, *grads):
        if torch.is_grad_enabled():
            raise HypadError("double backward (create_graph=True) through hypad_amd's layer functions is not supported: their gradient "
                             "kernels are first-order.  The WGAN-GP iterations with the second-order chain are "
                             "hypad_amd.train.critic_x_iteration / critic_z_iteration")
        return backward(ctx, *grads)
    return guarded


def check(rc, what=""):
    if rc != 0:
        msg = lib.hypad_error_string(int(rc))
        raise HypadError(f"{what or 'hypad call'} failed: {msg.decode() if msg else rc} ({rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    """torch's current stream on the current device as a hipStream_t (the raw C getters: torch.cuda.current_stream() builds a Stream
    object behind three layers of Python per call, ~5 us of the drop-in iteration functions' ~40 us of host time)."""
    try:
        return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:                                   # (a torch without the raw getters)
        return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name="tensor", dtype=torch.float32):
    if not t.is_cuda:
        raise HypadError(f"{name} must live on the GPU (hypad_amd has no CPU path)")
    if t.dtype != dtype:
        raise HypadError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise HypadError(f"{name} must be contiguous")
    return t


_catalogues = {}


def param_catalogue(net, S, L, hyperbolic):
    """[(name, offset, shape)] and the padded float count of one network's arena (a function of the four arguments only: remembered --
    every module construction asks, 0.8 ms of ctypes calls each)."""
    key = (int(net), int(S), int(L), int(bool(hyperbolic)))
    if key not in _catalogues:
        _catalogues[key] = _param_catalogue(*key)
    cat, total = _catalogues[key]
    return list(cat), total


def _param_catalogue(net, S, L, hyperbolic):
    n = lib.hypad_param_tensors(net, int(hyperbolic))
    if n < 0:
        raise HypadError("bad network id")
    out = []
    name = ctypes.create_string_buffer(64)
    off, rows, cols = c_int(), c_int(), c_int()
    for i in range(n):
        check(lib.hypad_param_info(net, S, L, int(hyperbolic), i, name, 64, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols)))
        shape = (rows.value, cols.value) if cols.value > 0 else (rows.value,)
        out.append((name.value.decode(), off.value, shape))
    return out, lib.hypad_param_count(net, S, L, int(hyperbolic))

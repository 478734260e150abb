"""ctypes mirror of include/phantom_amd.h and the loader of libphantom_amd.so.

The product path has no CPU fallback: if the HIP library is missing, ``load_library`` raises.
"""
import ctypes as C
import os

ABI_VERSION = 10
NPI, NPF = 4, 8

# phx_kind
KIND_FACTORY, KIND_SHOP, KIND_CUSTOMER, KIND_SELLER, KIND_BUYER = 1, 2, 3, 4, 5
KIND_HALVER, KIND_CASHBOX, KIND_REQRESP, KIND_FORWARDER = 6, 7, 8, 9
KIND_MOCK_STRAT, KIND_MOCK_AGENT = 10, 11
KIND_PUBLISHER, KIND_ADVERTISER, KIND_ADEXCHANGE = 12, 13, 14
KIND_NAMES = {1: "factory", 2: "shop", 3: "customer", 4: "seller", 5: "buyer", 6: "halver",
              7: "cashbox", 8: "reqresp", 9: "forwarder", 10: "mock", 11: "mock_agent",
              12: "pub", 13: "adv", 14: "adx"}
STRATEGIC_KINDS = (KIND_SHOP, KIND_SELLER, KIND_BUYER, KIND_MOCK_STRAT, KIND_ADVERTISER)
OBS_DIM = {KIND_SHOP: 3, KIND_SELLER: 2, KIND_BUYER: 2, KIND_MOCK_STRAT: 1, KIND_ADVERTISER: 3}

# phx_msg_type
(MSG_STOCK_REQUEST, MSG_STOCK_RESPONSE, MSG_ORDER_REQUEST, MSG_ORDER_RESPONSE, MSG_PRICE,
 MSG_ORDER, MSG_HALVE, MSG_CASH, MSG_REQUEST, MSG_RESPONSE, MSG_PING) = range(1, 12)
(MSG_IMPRESSION_REQ, MSG_BID, MSG_AUCTION_RESULT, MSG_ADS, MSG_IMPRESSION_RES) = range(12, 17)
FLOAT_PAYLOAD_TYPES = (MSG_PRICE, MSG_CASH, MSG_REQUEST, MSG_RESPONSE, MSG_BID, MSG_AUCTION_RESULT)
TAG_PYF, TAG_F32, TAG_F64 = 0, 1, 2      # PHX_TAG_*: numpy scalar kind of an ads-market float

ENV_PLAIN, ENV_FSM, ENV_STACKELBERG = 0, 1, 2
# phx_spec.variant_* (ABI 6)
VR_AUTO, VR_TIME_PARALLEL, VR_LEAN, VR_GENERAL, VR_LAUNCH_LOOP, VR_STORE_WAVES = 0, 1, 2, 3, 4, 5
VB_WHOLE_ENVS = -1
VS_AUTO, VS_FUSED, VS_GENERIC, VS_WIDE, VS_GENERIC_DYNAMIC = 0, 1, 2, 3, 4
SAMPLER_HOST, SAMPLER_UNIFORM = 0, 1
TYPE_NONE, TYPE_CONST = -2, -1
F_IGNORE_CONN_ERRORS, F_NO_PAYLOAD_CHECKS, F_FORCE_GENERIC, F_SHUFFLE_BATCHES, F_MT19937 = 1, 2, 4, 8, 16

(ERR_NONE, ERR_NETWORK, ERR_PAYLOAD, ERR_UNKNOWN_MSG, ERR_ROUND_LIMIT, ERR_QUEUE_FULL,
 ERR_CONTEXT, ERR_FSM_TRANSITION, ERR_HINT) = range(9)
MAX_ROUNDS = 4096            # PHX_MAX_ROUNDS: cap on BatchResolver(round_limit=None) rounds

_u8p, _i32p, _f32p, _f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                             C.POINTER(C.c_double))


class PhxSpec(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_agents", C.c_int32), ("batch", C.c_int32),
        ("num_steps", C.c_int32), ("round_limit", C.c_int32), ("env_type", C.c_int32),
        ("flags", C.c_uint32), ("queue_cap", C.c_int32), ("trace_cap", C.c_int32),
        ("kind", C.c_void_p), ("param_i", C.c_void_p), ("param_f", C.c_void_p),
        ("row_ptr", C.c_void_p), ("col", C.c_void_p),
        ("n_stages", C.c_int32), ("initial_stage", C.c_int32),
        ("stage_act_ptr", C.c_void_p), ("stage_act_idx", C.c_void_p),
        ("stage_rewarded", C.c_void_p), ("stage_rewarded_all", C.c_void_p),
        ("stage_next", C.c_void_p),
        ("n_leaders", C.c_int32), ("n_followers", C.c_int32),
        ("leaders", C.c_void_p), ("followers", C.c_void_p),
        ("seed", C.c_uint64), ("env_offset", C.c_int64),
        ("n_samplers", C.c_int32), ("sampler_kind", C.c_void_p), ("sampler_param", C.c_void_p),
        ("type_src", C.c_void_p),
        ("n_conn", C.c_int32), ("conn_rate", C.c_void_p), ("col_conn", C.c_void_p),
        ("stage_allowed", C.c_void_p),
        ("variant_rollout", C.c_int32), ("variant_block", C.c_int32), ("variant_step", C.c_int32),
        ("variant_reserved", C.c_int32),
        ("stage_tab", C.c_void_p),
        ("n_stage_rules", C.c_int32), ("reserved1", C.c_int32), ("stage_rules", C.c_void_p),
    ]


class PhxField(C.Structure):
    _fields_ = [("field_id", C.c_int32), ("dtype", C.c_int32), ("offset", C.c_int64),
                ("dim0", C.c_int32), ("dim1", C.c_int32), ("dim2", C.c_int32),
                ("kind", C.c_int32), ("name", C.c_char * 32)]


class _Payload(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double)]


class PhxMsgRec(C.Structure):
    _fields_ = [("sender", C.c_uint16), ("receiver", C.c_uint16), ("type", C.c_uint16),
                ("round", C.c_uint16), ("payload", _Payload)]


class PhxStepIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "actions", "action_valid", "exo", "obs", "obs_valid", "reward", "reward_valid",
        "terminated", "truncated", "done_valid", "all_terminated", "all_truncated", "err",
        "msg_log", "msg_count", "shuffle", "next_stage")]


class PhxRolloutIO(C.Structure):
    _fields_ = [("T", C.c_int32), ("hints", C.c_int32)] + [(n, C.c_void_p) for n in (
        "actions", "exo", "obs", "action_out", "reward", "terminated", "truncated", "obs_valid",
        "reward_valid", "last_obs", "err", "msg_log", "msg_count", "reserved_ptr")] + [
        ("n_frag", C.c_int32), ("reserved0", C.c_int32), ("frags", C.c_void_p), ("policy", C.c_void_p)]


class PhxPolicyMLP(C.Structure):
    """phx_policy_mlp (ABI 10): the device-evaluated policy of a rollout"""
    _fields_ = [("n_hidden", C.c_int32), ("width", C.c_int32 * 2), ("activation", C.c_int32),
                ("out_scale", C.c_float), ("out_bias", C.c_float), ("out_lo", C.c_float), ("out_hi", C.c_float),
                ("w", C.c_void_p * 3), ("b", C.c_void_p * 3)]


ACT_RELU, ACT_HARD_TANH = 0, 1
POLICY_MAX_WIDTH = 64


class PhxStageRule(C.Structure):
    """phx_stage_rule (ABI 9): one rule of a device-evaluated state handler"""
    _fields_ = [("stage", C.c_int32), ("agent", C.c_int32), ("cmp", C.c_int32), ("next_stage", C.c_int32),
                ("threshold", C.c_double), ("field", C.c_char * 32)]


CMP = {"<": 0, "<=": 1, ">": 2, ">=": 3, "==": 4, "!=": 5}


class PhxRolloutFrag(C.Structure):
    """one fragment of a fragment-list rollout (phx_rollout_io.frags, ABI 9)"""
    _fields_ = [(n, C.c_void_p) for n in ("obs", "action_out", "reward", "terminated", "truncated", "obs_valid", "reward_valid")]


MAX_FRAGMENTS = 8
RH_ACTIONS_IN_DOMAIN, RH_EXO_IN_DOMAIN = 2, 4      # phx_rollout_io.hints


assert C.sizeof(PhxMsgRec) == 16

_LIB = None
LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")
# PHX_LIB_PATH: development override (A/B runs of experimental builds); the default is the in-tree library
LIB_PATH = os.environ.get("PHX_LIB_PATH") or os.path.join(LIB_DIR, "libphantom_amd.so")

EXPORTS = ("phx_abi_version", "phx_last_error", "phx_last_kernel", "phx_autotune_note", "phx_state_nbytes", "phx_obs_dim",
           "phx_n_strategic", "phx_n_exo", "phx_create", "phx_destroy", "phx_n_fields",
           "phx_field_info", "phx_uses_fused", "phx_sync_fields", "phx_reset", "phx_step", "phx_step_begin", "phx_step_end", "phx_inject",
           "phx_resolve", "phx_rollout", "phx_get_state", "phx_set_state", "phx_trace",
           "phx_pack_flags", "phx_unpack_flags", "phx_mt_seed", "phx_mt_draw")


def built_archs():
    """offload architectures the in-tree library was compiled for (phantom_amd/build.py)."""
    try:
        return [a for a in open(os.path.join(LIB_DIR, "ARCH")).read().strip().split(";") if a]
    except OSError:
        return ["gfx950"]


def _check_arch(torch):
    """Fail at load time, not at the first launch ('invalid device function'), when the visible
    GPU is not one the library holds a code object for."""
    if not torch.cuda.is_available():
        return                         # symbol / spec-size calls work without a GPU; DeviceEnv raises
    name = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
    arch = name.split(":")[0]
    if arch and arch not in built_archs():
        raise RuntimeError(
            f"libphantom_amd.so was built for {built_archs()} but the visible GPU is {arch}: rebuild with "
            f"PHX_OFFLOAD_ARCH='{arch}' (the kernels are tuned for gfx950 / MI355X only)")


def bind_signatures(lib):
    """restype / argtypes of every entry point of include/phantom_amd.h on a loaded library (the ctypes stub of
    INTEGRATION.md section 2).  Used for libphantom_amd.so here and, unchanged, for the CPU restatement behind the same
    symbols in tests (oracle/libphantom_cpu.so: host pointers)."""
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.phx_abi_version.restype = i32
    lib.phx_last_kernel.restype = C.c_char_p
    lib.phx_autotune_note.restype = C.c_char_p
    lib.phx_autotune_note.argtypes = [C.c_void_p]
    lib.phx_last_error.restype = C.c_char_p
    lib.phx_state_nbytes.restype = i64
    lib.phx_state_nbytes.argtypes = [C.POINTER(PhxSpec)]
    for n in ("phx_obs_dim", "phx_n_strategic", "phx_n_exo"):
        getattr(lib, n).restype = i32
        getattr(lib, n).argtypes = [C.POINTER(PhxSpec)]
    lib.phx_create.restype = i32
    lib.phx_create.argtypes = [C.POINTER(PhxSpec), i32, vp, i64, C.POINTER(vp)]
    lib.phx_destroy.restype = None
    lib.phx_destroy.argtypes = [vp]
    lib.phx_n_fields.restype = i32
    lib.phx_n_fields.argtypes = [vp]
    lib.phx_field_info.restype = i32
    lib.phx_field_info.argtypes = [vp, i32, C.POINTER(PhxField)]
    lib.phx_uses_fused.restype = i32
    lib.phx_uses_fused.argtypes = [vp]
    lib.phx_sync_fields.restype = i32
    lib.phx_sync_fields.argtypes = [vp, vp]
    lib.phx_reset.restype = i32
    lib.phx_reset.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    for n in ("phx_step", "phx_step_begin", "phx_step_end"):
        getattr(lib, n).restype = i32
        getattr(lib, n).argtypes = [vp, C.POINTER(PhxStepIO), vp]
    lib.phx_inject.restype = i32
    lib.phx_inject.argtypes = [vp, C.POINTER(PhxMsgRec), i32]
    lib.phx_resolve.restype = i32
    lib.phx_resolve.argtypes = [vp, vp, vp, vp, vp]
    lib.phx_rollout.restype = i32
    lib.phx_rollout.argtypes = [vp, C.POINTER(PhxRolloutIO), vp]
    lib.phx_get_state.restype = i64
    lib.phx_get_state.argtypes = [vp, C.c_char_p, vp, i64, vp]
    lib.phx_set_state.restype = i64
    lib.phx_set_state.argtypes = [vp, C.c_char_p, vp, i64, vp]
    lib.phx_trace.restype = i32
    lib.phx_trace.argtypes = [vp, vp, vp, i32, C.POINTER(PhxMsgRec), i32, vp]
    lib.phx_pack_flags.restype = i32
    lib.phx_pack_flags.argtypes = [vp, vp, i64, vp]
    lib.phx_unpack_flags.restype = i32
    lib.phx_unpack_flags.argtypes = [vp, vp, i64, vp]
    lib.phx_mt_seed.restype = i32
    lib.phx_mt_seed.argtypes = [vp, vp, vp]
    lib.phx_mt_draw.restype = i32
    lib.phx_mt_draw.argtypes = [vp, vp, i32, vp]
    return lib


def load_library():
    """Load libphantom_amd.so (built in-tree by ``phantom_amd.build``); never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback).")
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Import it FIRST so that this
    # library's NEEDED libamdhip64.so.7 resolves to the already-loaded runtime: two HIP runtimes
    # in one process do not share devices/streams ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    bind_signatures(lib)
    if lib.phx_abi_version() != ABI_VERSION:
        raise RuntimeError("libphantom_amd.so ABI version mismatch")
    _check_arch(torch)
    _LIB = lib
    return lib

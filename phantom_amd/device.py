"""DeviceEnv: the thin ctypes layer between the Python host surface and libphantom_amd.so.

PyTorch-ROCm tensors own every buffer (state blob, inputs, outputs); the library receives raw
device pointers plus torch's current HIP stream, so all calls are stream-ordered with the
caller's other torch work and nothing synchronises unless the caller reads values back.
There is NO CPU fallback: without a visible GPU or the built HIP library this raises.
"""
import ctypes as C
from typing import Dict, List, NamedTuple, Optional

import numpy as np

from . import _abi
from .message import Message, payload_to_record, record_to_payload
from .spec import EnvSpec

_DTYPES = None


def _torch():
    import torch
    return torch


class StepTensors(NamedTuple):
    """Device-resident result of one batched step (views of persistent buffers; valid until
    the next step call on the same env)."""
    observations: "object"   # f32 [B, S, D]
    rewards: "object"        # f64 [B, S]
    terminations: "object"   # u8  [B, S]
    truncations: "object"    # u8  [B, S]
    obs_valid: "object"      # u8  [B, S]
    reward_valid: "object"   # u8  [B, S]  0 absent / 1 value / 2 None
    done_valid: "object"     # u8  [B, S]
    all_terminated: "object" # u8  [B]
    all_truncated: "object"  # u8  [B]


class Trajectory(NamedTuple):
    """Device-resident rollout fragment, time-major."""
    observations: "object"   # f32 [T, B, S, D]
    actions: "object"        # f32 [T, B, S]
    rewards: "object"        # f32 [T, B, S]
    terminations: "object"   # u8  [T, B, S]
    truncations: "object"    # u8  [T, B, S]
    last_obs: "object"       # f32 [B, S, D]
    obs_valid: "object" = None      # u8 [T, B, S]  (FSM envs: key present in step.observations)
    reward_valid: "object" = None   # u8 [T, B, S]  (FSM envs: 0 absent / 1 value / 2 None)
    msg_log: "object" = None        # u8 [T, B, trace_cap, 16]  ordered message records per step (tracking on)
    msg_count: "object" = None      # i32 [T, B]
    flat: "object" = None           # u8 [nbytes]: the one buffer every plane above is a section of (alloc_trajectory(flat=True))
    packed_flags: "object" = None   # u8 view of `flat`: bit-packed done flags for rollout collection (pack_done_flags)
    gather_nbytes: int = 0          # prefix of `flat` a learner needs from every rank (distributed.TrajectoryGather)


class DeviceError(RuntimeError):
    pass


class HostStep:
    """A step's outputs in flight to the host (DeviceEnv.pull_step_async)."""

    def __init__(self, dev, buf, event, seq):
        self._dev, self._buf, self._event, self._seq, self._arrays = dev, buf, event, seq, None
        self._consumed = False          # read_into() has copied the arrays out: nothing is left to keep alive

    def read_into(self, dst: Dict[str, "object"]) -> None:
        """the arrays copied straight into ``dst`` (same names, shapes and dtypes): one copy instead of get()'s copy + the caller's"""
        if self._arrays is not None:
            for k, v in self._arrays.items():
                np.copyto(dst[k], v)
            return
        if self._buf is None:                                     # lazy: from the step outputs themselves
            if self._dev._out_seq != self._seq:
                raise DeviceError("a poll() result was first read after a later step / reset had overwritten the step outputs: read it "
                                  "before the next send_actions(), or build the adapter with keep_results=True")
            for k, v in self._dev.pull_step().items():
                np.copyto(dst[k], v)
            return
        if self._dev._host_ring_k - self._seq > 3:
            raise DeviceError("a poll() result was first read more than three steps after it was produced: its host buffer has been reused")
        self._event.synchronize()
        h = self._buf.numpy()
        for name, shape, dtype, off, n in self._dev._out_layout:
            np.copyto(dst[name], h[off:off + n].view(np.dtype(str(dtype).replace("torch.", ""))).reshape(shape))
        if self._dev._host_ring_k - self._seq > 3:
            raise DeviceError("a poll() result was read while its host buffer was being reused")
        self._consumed = True

    def materialise(self) -> None:
        """bring the arrays to the host now if nobody has read them yet (DeviceEnv calls this for a result that is still referenced
        right before its pinned buffer is reused: a poll() result then outlives any number of later steps)"""
        if self._arrays is None and not self._consumed:
            self.get()

    def get(self) -> Dict[str, "object"]:
        if self._arrays is None and self._buf is None:            # lazy: the copy is made now, from the step outputs themselves
            if self._dev._out_seq != self._seq:
                raise DeviceError("a poll() result was first read after a later step / reset had overwritten the step outputs: read it "
                                  "before the next send_actions(), or build the adapter with keep_results=True")
            self._arrays = {k: v.copy() for k, v in self._dev.pull_step().items()}
        if self._arrays is None:
            if self._dev._host_ring_k - self._seq > 3:            # three pinned buffers rotate: this one has been reused since
                raise DeviceError("a poll() result was first read more than three steps after it was produced: its host buffer has been reused")
            self._event.synchronize()
            h = self._buf.numpy()
            self._arrays = {name: h[off:off + n].view(np.dtype(str(dtype).replace("torch.", ""))).reshape(shape).copy()
                            for name, shape, dtype, off, n in self._dev._out_layout}
            if self._dev._host_ring_k - self._seq > 3:            # (reused while it was being read)
                self._arrays = None
                raise DeviceError("a poll() result was read while its host buffer was being reused")
        return self._arrays


class StepGraph:
    """``n`` captured ``phx_step`` launches (DeviceEnv.step_graph); ``replay()`` enqueues them on the
    current stream without per-step host work."""

    def __init__(self, graph, actions, out, n, dev=None):
        self.graph, self.actions, self.out, self.n, self._dev = graph, actions, out, n, dev

    def replay(self):
        if self._dev is not None:
            self._dev._out_seq += 1      # the replay rewrites the step outputs: a lazy poll() result of an earlier step must not read them
        self.graph.replay()
        return self.out


class RolloutGraph:
    """Consecutive ``phx_rollout`` fragments captured ONCE into a hipGraph (DeviceEnv.rollout_graph): ``replay()`` enqueues them on
    the current stream; fragment i of a replay lands in ``trajectories[i]`` (rewritten by every replay)."""

    def __init__(self, graph, trajectories, T):
        self.graph, self.trajectories, self.T = graph, trajectories, T

    def replay(self):
        self.graph.replay()
        return self.trajectories


class DeviceEnv:
    def __init__(self, spec: EnvSpec, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise DeviceError("phantom_amd needs a visible AMD GPU (torch.cuda.is_available() is "
                              "False); the PhantomEnv.step() path has no CPU fallback")
        self.lib = _abi.load_library()
        self.spec = spec
        self.device = torch.device(device if device is not None
                                   else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        self._cspec, self._keep = spec.to_ctypes()
        cs = C.byref(self._cspec)
        self.B, self.S = spec.batch, spec.n_strategic
        self.D = self.lib.phx_obs_dim(cs)
        self.n_exo = self.lib.phx_n_exo(cs)
        assert self.lib.phx_n_strategic(cs) == self.S
        nbytes = self.lib.phx_state_nbytes(cs)
        if nbytes <= 0:
            raise DeviceError("phx_state_nbytes failed: " + self._err())
        self.state = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
        handle = C.c_void_p()
        # phx_create uploads tables and runs the initial reset on the NULL stream; the zero fill above ran
        # on torch's current stream, which need not be ordered against it (non-blocking side streams)
        torch.cuda.current_stream(self.device).synchronize()
        with torch.cuda.device(self.device):
            rc = self.lib.phx_create(cs, self.device.index, self.state.data_ptr(), nbytes,
                                     C.byref(handle))
        if rc != 0:
            raise DeviceError(f"phx_create failed ({rc}): " + self._err())
        self.handle = handle

        self._kind_rank = spec.kind_rank()
        self._fields: Dict[str, "object"] = {}
        dt = {0: torch.int32, 1: torch.float64, 2: torch.uint8, 3: torch.float32}
        for k in range(self.lib.phx_n_fields(handle)):
            f = _abi.PhxField()
            self.lib.phx_field_info(handle, k, C.byref(f))
            n = f.dim0 * f.dim1 * f.dim2
            esz = {0: 4, 1: 8, 2: 1, 3: 4}[f.dtype]
            view = self.state[f.offset:f.offset + n * esz].view(dt[f.dtype])
            shape = [f.dim0, f.dim1] + ([f.dim2] if f.dim2 > 1 else [])
            self._fields[f.name.decode()] = view.view(*shape)
        B, S, D = self.B, max(self.S, 1), self.D
        z = lambda *s, dtype: torch.zeros(*s, dtype=dtype, device=self.device)
        # the step outputs are 256-byte aligned sections of ONE device buffer, so that the dict API
        # (PhantomEnv.step) brings them to the host with a single copy (pull_step)
        layout = [("obs", (B, S, D), torch.float32), ("reward", (B, S), torch.float64),
                  ("obs_valid", (B, S), torch.uint8), ("reward_valid", (B, S), torch.uint8),
                  ("terminated", (B, S), torch.uint8), ("truncated", (B, S), torch.uint8),
                  ("done_valid", (B, S), torch.uint8), ("all_terminated", (B,), torch.uint8),
                  ("all_truncated", (B,), torch.uint8), ("err", (B,), torch.int32)]
        self._out_layout, total = [], 0
        for name, shape, dtype in layout:
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            self._out_layout.append((name, shape, dtype, total, n))
            total += (n + 255) & ~255
        self._out_flat = z(total, dtype=torch.uint8)
        self._out_host = None
        self._rollout_io_cache = {}
        for name, shape, dtype, off, n in self._out_layout:
            setattr(self, name, self._out_flat[off:off + n].view(dtype).view(*shape))
        self.ones_valid = None
        self.msg_log = self.msg_count = None
        if spec.trace_cap > 0:
            self.msg_log = z(B, spec.trace_cap, 16, dtype=torch.uint8)
            self.msg_count = z(B, dtype=torch.int32)
        self.topology_version = 0
        self._step_io = None
        self._out_seq = 0                 # bumped by every call that rewrites the step outputs (HostStep: lazy reads)

    # ---- helpers --------------------------------------------------------------------------
    def _err(self) -> str:
        return (self.lib.phx_last_error() or b"").decode()

    def last_kernel(self) -> str:
        """names of the kernels this thread's last step / rollout / resolve call launched, joined by '+'"""
        return (self.lib.phx_last_kernel() or b"").decode()

    def autotune_note(self) -> str:
        """what PHX_VR_AUTO measured when it last had two kernels for a rollout shape of this env (phx_autotune_note), or ''"""
        return (self.lib.phx_autotune_note(self.handle) or b"").decode()

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise DeviceError(f"{what} failed ({rc}): {self._err()}")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.phx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_fused(self) -> bool:
        """a fused static-schedule kernel serves step/rollout (can turn False: a host-injected
        message moves a fused Stackelberg env onto the generic engine for good)."""
        return bool(self.lib.phx_uses_fused(self.handle))

    def field(self, name: str):
        """torch view [B, n] into the state blob (zero-copy).  ``buyer.prices`` is kept in
        compressed per-seller form by the fused Stackelberg kernel and materialised here."""
        if name == "buyer.prices":
            self._check(self.lib.phx_sync_fields(self.handle, self._stream()), "phx_sync_fields")
        return self._fields[name]

    def field_names(self) -> List[str]:
        return list(self._fields)

    def _read_agent_state(self, agent, field_name: str):
        a = self.spec.index_of(agent.id)
        col = int(self._kind_rank[a])
        v = self._fields[field_name][:, col].cpu().numpy()
        if v.ndim > 1:                      # a per-agent vector (adv.total_*: counts keyed by user id)
            return v[0] if self.B == 1 else v
        return v[0].item() if self.B == 1 else v

    # ---- entry points ---------------------------------------------------------------------
    def reset(self, mask=None, sampler_values=None, conn_on=None):
        """``sampler_values``: f64 [B, n_samplers] (host array or device tensor) = what
        `sampler.sample()` returned for each env (env.py:211-212); ``conn_on``: u8 [B, n_conn] =
        the StochasticNetwork draws (network.py:444-447); None -> device-drawn."""
        torch = _torch()
        mp = vp = cp = None
        if conn_on is not None:
            self._conn_on = torch.as_tensor(conn_on, dtype=torch.uint8).to(self.device).contiguous()
            assert self._conn_on.shape == (self.B, self.spec.n_conn)
            cp = self._conn_on.data_ptr()
        if sampler_values is not None:
            self._sampler_values = torch.as_tensor(sampler_values, dtype=torch.float64).to(
                self.device).contiguous()
            assert self._sampler_values.shape == (self.B, self.spec.n_samplers)
            vp = self._sampler_values.data_ptr()
        if mask is not None:
            mask = torch.as_tensor(mask, dtype=torch.uint8, device=self.device).contiguous()
            mp = mask.data_ptr()
            self.err.masked_fill_(mask.bool(), 0)
        else:
            self.err.zero_()
        self._out_seq += 1
        with torch.cuda.device(self.device):
            self._check(self.lib.phx_reset(self.handle, mp, vp, cp, self.obs.data_ptr(),
                                           self.obs_valid.data_ptr(), self._stream()), "phx_reset")
        return self.obs, self.obs_valid

    def reset_agents(self):
        """Network.reset() without an env: agent.reset() for every agent."""
        self.reset()

    def _fill_step_io(self, io):
        io.obs, io.obs_valid = self.obs.data_ptr(), self.obs_valid.data_ptr()
        io.reward, io.reward_valid = self.reward.data_ptr(), self.reward_valid.data_ptr()
        io.terminated, io.truncated = self.terminated.data_ptr(), self.truncated.data_ptr()
        io.done_valid = self.done_valid.data_ptr()
        io.all_terminated = self.all_terminated.data_ptr()
        io.all_truncated = self.all_truncated.data_ptr()
        io.err = self.err.data_ptr()
        if self.msg_log is not None:
            io.msg_log, io.msg_count = self.msg_log.data_ptr(), self.msg_count.data_ptr()
        self._step_io_ref = C.byref(io)
        self._step_out = StepTensors(self.obs, self.reward, self.terminated, self.truncated,
                                     self.obs_valid, self.reward_valid, self.done_valid,
                                     self.all_terminated, self.all_truncated)

    def _ensure_step_io(self):
        # the output pointers never change: build the struct once, patch the inputs per call
        if self._step_io is None:
            io = self._step_io = _abi.PhxStepIO()
            self._fill_step_io(io)
        return self._step_io

    def step_begin(self, actions, action_valid=None, exo=None, shuffle=None):
        """phx_step_begin: the acting phase and resolve_network() of a step (fsm.py:275-280); agent state is the resolved one
        afterwards, the env's clock words and the step outputs are untouched.  Followed by ``step_end``."""
        return self.step(actions, action_valid, exo, shuffle, None, _entry="phx_step_begin")

    def step_end(self, next_stage=None) -> StepTensors:
        """phx_step_end: the transition to ``next_stage`` (int32 [B]; None: next_stages[0] / the tabulated handler) and the
        step's observations, rewards and done flags (fsm.py:304-380)."""
        torch = _torch()
        io = self._step_io if self._step_io is not None else self._ensure_step_io()
        if next_stage is not None:
            if next_stage.dtype != torch.int32 or tuple(next_stage.shape) != (self.B,) or not next_stage.is_contiguous() \
                    or next_stage.device != self.device:
                raise ValueError(f"next_stage must be a contiguous int32 tensor [{self.B}] on {self.device}")
            io.next_stage = next_stage.data_ptr()
        else:
            io.next_stage = None
        self._out_seq += 1
        rc = self.lib.phx_step_end(self.handle, self._step_io_ref, torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            self._check(rc, "phx_step_end")
        return self._step_out

    def step(self, actions, action_valid=None, exo=None, shuffle=None, next_stage=None, _entry="phx_step") -> StepTensors:
        torch = _torch()
        io = self._step_io
        if io is None:
            io = self._ensure_step_io()
        if self.S > 0:
            if actions.dtype != torch.float32 or not actions.is_contiguous() \
                    or actions.shape != (self.B, self.S) or actions.device != self.device:
                raise ValueError(f"actions must be a contiguous f32 tensor [{self.B}, {self.S}] on {self.device}")
            io.actions = actions.data_ptr()
        io.action_valid = action_valid.data_ptr() if action_valid is not None else None
        if exo is not None:
            if exo.dtype != torch.uint8 or exo.shape != (self.B, self.n_exo) or not exo.is_contiguous():
                raise ValueError(f"exo must be a contiguous u8 tensor [{self.B}, {self.n_exo}]")
            io.exo = exo.data_ptr()
        else:
            io.exo = None
        if shuffle is not None:                   # recorded np.random.shuffle outcomes (BatchResolver(shuffle_batches=True))
            if shuffle.dtype != torch.int16 and shuffle.dtype != torch.uint16:
                raise ValueError("shuffle must be a 16-bit integer tensor")
            if tuple(shuffle.shape) != (self.B, 8 * self.spec.queue_cap) or not shuffle.is_contiguous() \
                    or shuffle.device != self.device:
                raise ValueError(f"shuffle must be a contiguous tensor [{self.B}, {8 * self.spec.queue_cap}] on {self.device}")
            io.shuffle = shuffle.data_ptr()
        else:
            io.shuffle = None
        if next_stage is not None:                # FSM stage handlers' return values, one stage index per env
            if next_stage.dtype != torch.int32 or tuple(next_stage.shape) != (self.B,) or not next_stage.is_contiguous() \
                    or next_stage.device != self.device:
                raise ValueError(f"next_stage must be a contiguous int32 tensor [{self.B}] on {self.device}")
            io.next_stage = next_stage.data_ptr()
        else:
            io.next_stage = None
        self._out_seq += 1
        rc = getattr(self.lib, _entry)(self.handle, self._step_io_ref,
                                       torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            self._check(rc, _entry)
        return self._step_out

    def pull_step(self) -> Dict[str, "object"]:
        """numpy views of the last step's outputs (and err), fetched with ONE device-to-host copy."""
        torch = _torch()
        if self._out_host is None:
            self._out_host = torch.empty(self._out_flat.shape, dtype=torch.uint8, pin_memory=True)
        self._out_host.copy_(self._out_flat, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        h = self._out_host.numpy()
        return {name: h[off:off + n].view(np.dtype(str(dtype).replace("torch.", ""))).reshape(shape)
                for name, shape, dtype, off, n in self._out_layout}

    def pull_step_lazy(self) -> "HostStep":
        """The last step's outputs, brought to the host only if somebody reads them: ``HostStep.get()`` makes the one device-to-host
        copy (and the synchronisation) on its first call -- which has to come before the next launch overwrites the step outputs
        (DeviceError otherwise).  Costs nothing when the result is never read (a tensor-native loop that polls for form's sake)."""
        return HostStep(self, None, None, self._out_seq)

    def pull_step_async(self) -> "HostStep":
        """The last step's outputs on their way to the host WITHOUT a synchronisation: one non-blocking copy of the output buffer
        into one of three rotating pinned buffers with an event behind it.  ``HostStep.get()`` waits for the event (once) and
        returns copies of the arrays, so a result stays valid however many steps follow -- also one that is FIRST read many steps
        later: a result that is still referenced when its buffer comes up for reuse is copied out before the reuse; a HostStep
        nobody kept costs the copy (~25 us of stream time at SC64, B = 4096) and nothing on the host."""
        import weakref
        torch = _torch()
        ring = self.__dict__.setdefault("_host_ring", [])
        owners = self.__dict__.setdefault("_host_ring_owner", [None, None, None])
        if len(ring) < 3:
            ring.append(torch.empty(self._out_flat.shape, dtype=torch.uint8, pin_memory=True))
        k = self.__dict__.get("_host_ring_k", 0)
        slot = k % 3 if len(ring) == 3 else len(ring) - 1
        prev = owners[slot]() if owners[slot] is not None else None
        if prev is not None:
            prev.materialise()           # somebody still holds the result that lives in this buffer and has not read it: copy it out first
        self._host_ring_k = k + 1
        buf = ring[slot]
        buf.copy_(self._out_flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        hs = HostStep(self, buf, ev, k)
        owners[slot] = weakref.ref(hs)
        return hs

    def _needs_valid_planes(self) -> bool:
        # validity masks: stage-masked envs, and kinds whose encode_observation can return None
        return self.spec.env_type != _abi.ENV_PLAIN or bool((self.spec.kind == _abi.KIND_ADVERTISER).any())

    def never_terminates(self) -> bool:
        """no strategic kind of this env can return is_terminated() == True (ShopAgent, the market's
        Seller/Buyer: agents.py:292-323 defaults), so the `terminations` plane of a trajectory is all zero."""
        k = self.spec.kind[self.spec.strategic_idx] if self.S else np.zeros(0, np.uint8)
        return bool(np.isin(k, (_abi.KIND_SHOP, _abi.KIND_SELLER, _abi.KIND_BUYER)).all())

    def alloc_trajectory(self, T: int, record_messages: bool = False, flat: bool = False,
                         terminations: bool = True) -> Trajectory:
        """Uninitialised device buffers for a T-step fragment (time-major).  ``record_messages``:
        also the per-step ordered message log (rollout.py:369-373, needs enable_tracking).
        ``flat``: every plane is a 256-byte aligned section of ONE buffer, ordered so that what a
        learner on another GPU needs is a prefix: obs | actions | rewards | [obs_valid | reward_valid] |
        bit-packed done flags, then (not gathered) the u8 truncations / terminations planes."""
        torch = _torch()
        B, S, D = self.B, self.S, self.D
        e = lambda *s, dtype: torch.empty(*s, dtype=dtype, device=self.device)
        fsm = self._needs_valid_planes()
        if record_messages and self.spec.trace_cap <= 0:
            raise DeviceError("record_messages needs BatchResolver(enable_tracking=True)")
        if flat:
            n = T * B * S
            words = (n + 63) // 64
            planes = 1 if self.never_terminates() else 2
            sections = [("observations", (T, B, S, D), torch.float32), ("actions", (T, B, S), torch.float32),
                        ("rewards", (T, B, S), torch.float32)]
            if fsm:
                sections += [("obs_valid", (T, B, S), torch.uint8), ("reward_valid", (T, B, S), torch.uint8)]
            sections += [("packed_flags", (planes * words * 8,), torch.uint8)]
            tail = [("truncations", (T, B, S), torch.uint8), ("terminations", (T, B, S), torch.uint8),
                    ("last_obs", (B, S, D), torch.float32)]
            offs, total, gather_nbytes = {}, 0, 0
            for name, shape, dtype in sections + tail:
                nb = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
                offs[name] = (total, nb, shape, dtype)
                total += (nb + 255) & ~255
                if name == "packed_flags":
                    gather_nbytes = total
            buf = torch.empty(total, dtype=torch.uint8, device=self.device)
            v = {k: buf[o:o + nb].view(dt).view(*sh) for k, (o, nb, sh, dt) in offs.items()}
            return Trajectory(v["observations"], v["actions"], v["rewards"], v["terminations"], v["truncations"],
                              v["last_obs"], v.get("obs_valid"), v.get("reward_valid"), None, None,
                              buf, v["packed_flags"], gather_nbytes)
        # terminations=False: no `terminations` plane (it is all zero for kinds that never terminate; phx_rollout accepts
        # the omission where the serving kernel can leave the plane out and returns an error otherwise)
        # (truncations directly followed by terminations: where the library zero-fills the flag planes before a rollout kernel
        #  that only stores the non-zero flags, ONE fill covers both)
        if terminations and (T * B * S) % 256 == 0:
            flags = e(2, T, B, S, dtype=torch.uint8)
            trunc, term = flags[0], flags[1]
        else:
            trunc, term = e(T, B, S, dtype=torch.uint8), (e(T, B, S, dtype=torch.uint8) if terminations else None)
        return Trajectory(e(T, B, S, D, dtype=torch.float32), e(T, B, S, dtype=torch.float32),
                          e(T, B, S, dtype=torch.float32), term,
                          trunc, e(B, S, D, dtype=torch.float32),
                          e(T, B, S, dtype=torch.uint8) if fsm else None,
                          e(T, B, S, dtype=torch.uint8) if fsm else None,
                          e(T, B, self.spec.trace_cap, 16, dtype=torch.uint8) if record_messages else None,
                          e(T, B, dtype=torch.int32) if record_messages else None)

    def _check_rollout_buffers(self, T, actions, exo, out: Trajectory):
        """Raw pointers go straight to the kernel: every buffer is checked for dtype, shape,
        contiguity and device here (python -O strips asserts, so these are real errors)."""
        torch = _torch()
        B, S, D = self.B, self.S, self.D

        def need(name, x, dtype, tail, lead=None):
            if x is None:
                raise ValueError(f"rollout: `{name}` is required for this env")
            if x.dtype != dtype or x.device != self.device or not x.is_contiguous():
                raise ValueError(f"rollout: `{name}` must be a contiguous {dtype} tensor on {self.device}")
            if x.data_ptr() % (16 if name.startswith("out.") else x.element_size()):
                raise ValueError(f"rollout: `{name}` must start on a 16-byte boundary (a view at an odd offset?)")
            if tuple(x.shape[1:]) != tuple(tail) or (lead is not None and x.shape[0] != lead) \
                    or (lead is None and x.shape[0] < T):
                want = (lead if lead is not None else f">={T}",) + tuple(tail)
                raise ValueError(f"rollout: `{name}` has shape {tuple(x.shape)}, expected {want}")

        if T < 1:
            raise ValueError("rollout: T must be >= 1")
        if actions is not None:
            need("actions", actions, torch.float32, (B, S), lead=T)
        if exo is not None:
            need("exo", exo, torch.uint8, (B, self.n_exo), lead=T)
        need("out.observations", out.observations, torch.float32, (B, S, D))
        need("out.actions", out.actions, torch.float32, (B, S))
        need("out.rewards", out.rewards, torch.float32, (B, S))
        if out.terminations is not None:                   # None: the all-zero plane left out (the library decides whether it can be)
            need("out.terminations", out.terminations, torch.uint8, (B, S))
        need("out.truncations", out.truncations, torch.uint8, (B, S))
        need("out.last_obs", out.last_obs, torch.float32, (S, D), lead=B)
        if self._needs_valid_planes():
            need("out.obs_valid", out.obs_valid, torch.uint8, (B, S))
            need("out.reward_valid", out.reward_valid, torch.uint8, (B, S))
        elif out.obs_valid is not None or out.reward_valid is not None:
            need("out.obs_valid", out.obs_valid, torch.uint8, (B, S))
            need("out.reward_valid", out.reward_valid, torch.uint8, (B, S))
        if out.msg_log is not None or out.msg_count is not None:
            if self.spec.trace_cap <= 0:
                raise ValueError("rollout: a message log needs BatchResolver(enable_tracking=True)")
            need("out.msg_count", out.msg_count, torch.int32, (B,))
            need("out.msg_log", out.msg_log, torch.uint8, (B, self.spec.trace_cap, 16))

    def rollout(self, T: int, actions=None, exo=None, out: Optional[Trajectory] = None, actions_in_domain: bool = False,
                exo_in_domain: bool = False, policy=None) -> Trajectory:
        """T fused steps into ``out`` (allocated here when None); ``actions`` f32 [T, B, S] / ``exo`` u8 [T, B, n_exo] replay a recorded
        policy / recorded draws (None: the device's random policy / RNG stream).  ``actions_in_domain`` / ``exo_in_domain``: the caller
        vouches that every action rounds to >= 0 (clipped to the action space) / every exo byte is < 5 (``mt_draw`` output): a plain supply
        chain's replay then takes the store-wave kernel without a scan of the inputs (phx_rollout_io.hints).
        ``policy``: a ``phantom_amd.policy.MLPPolicy`` evaluated on the device for every (env, strategic agent) and step from the agent's
        previous observation (phx_rollout_io.policy, ABI 10): T ON-POLICY steps in one launch (plain supply-chain envs)."""
        if policy is not None and actions is not None:
            raise ValueError("rollout: `policy` and replayed `actions` exclude each other")
        owned = out is None
        if owned:
            out = self.alloc_trajectory(T)
        # The argument block of a repeated call into CALLER-owned buffers is built once.  The key is
        # the buffers' addresses and the entry holds no tensor: a fragment allocated here (out=None)
        # is never cached, so repeated env.rollout(T) calls pin nothing (a T=100 SC64 fragment is
        # ~80 MB at B=4096).
        ptr = lambda x: x.data_ptr() if hasattr(x, "data_ptr") else None
        sig = lambda x: (x.data_ptr(), x.numel()) if hasattr(x, "data_ptr") else None     # address AND size: a buffer freed
        key = (T, bool(actions_in_domain), bool(exo_in_domain)) + tuple(sig(x) for x in out[:10]) + (sig(actions), sig(exo), id(policy))    # and reallocated smaller misses
        cached = None if owned else self._rollout_io_cache.get(key)
        if cached is None:
            self._check_rollout_buffers(T, actions, exo, out)
            io = _abi.PhxRolloutIO()
            io.T = T
            io.hints = (_abi.RH_ACTIONS_IN_DOMAIN if actions_in_domain else 0) | (_abi.RH_EXO_IN_DOMAIN if exo_in_domain else 0)
            io.actions, io.exo = ptr(actions), ptr(exo)
            io.obs, io.action_out, io.reward = ptr(out.observations), ptr(out.actions), ptr(out.rewards)
            io.terminated, io.truncated = ptr(out.terminations), ptr(out.truncations)
            io.last_obs = ptr(out.last_obs)
            io.obs_valid, io.reward_valid = ptr(out.obs_valid), ptr(out.reward_valid)
            io.msg_log, io.msg_count = ptr(out.msg_log), ptr(out.msg_count)
            io.err = self.err.data_ptr()
            pol_keep = None
            if policy is not None:
                if policy.obs_dim != self.D:
                    raise ValueError(f"rollout: the policy takes {policy.obs_dim} inputs, the env's observations have {self.D}")
                pol_keep = policy.on(self.device)              # (device weights + the argument struct: kept alive with the cached block)
                io.policy = C.addressof(pol_keep[2])
            cached = (io, C.byref(io), pol_keep, policy)
            if not owned:
                if len(self._rollout_io_cache) >= 4:           # tiny LRU: drop the oldest entry
                    self._rollout_io_cache.pop(next(iter(self._rollout_io_cache)))
                self._rollout_io_cache[key] = cached
        rc = self.lib.phx_rollout(self.handle, cached[1], self._stream())        # the library selects its device itself
        if rc != 0:
            self._check(rc, "phx_rollout")
        return out

    def rollout_fragments(self, T: int, outs, actions=None, exo=None, actions_in_domain: bool = False,
                          exo_in_domain: bool = False) -> List[Trajectory]:
        """``len(outs)`` consecutive T-step fragments from ONE ``phx_rollout`` call (``phx_rollout_io.frags``, ABI 9): the env
        advances ``len(outs) * T`` steps exactly as the same number of ``rollout(T, out=outs[i])`` calls would, fragment i holds
        rows ``[i T, (i + 1) T)``.  The fixed cost of a launch (pipeline fill, placing a 160 KB workgroup on every CU, the kernel
        boundary: ~9 us against 12.5 us of streaming per 100 steps of SC64 at B = 4096) is paid once per call instead of once per
        fragment where the store-wave supply-chain kernel serves the env; other envs run one launch per fragment inside the call.
        ``actions`` / ``exo``: replayed inputs for all ``len(outs) * T`` steps (``*_in_domain``: as in ``rollout``).
        ONE difference from separate calls: only the observation after the LAST step is produced (``phx_rollout_io.last_obs``, written to
        ``outs[-1].last_obs``); the library has no per-fragment boundary observation, so the returned fragments i < k - 1 carry
        ``last_obs=None`` (their ``last_obs`` tensors are left untouched) -- a learner that bootstraps from a fragment's boundary uses the
        next fragment's first row inputs, or asks for one fragment per call."""
        outs = list(outs)
        k = len(outs)
        if k == 1:
            return [self.rollout(T, actions, exo, out=outs[0], actions_in_domain=actions_in_domain, exo_in_domain=exo_in_domain)]
        if not 2 <= k <= _abi.MAX_FRAGMENTS:
            raise ValueError(f"rollout_fragments: 1 .. {_abi.MAX_FRAGMENTS} fragments per call, got {k}")
        ptr = lambda x: x.data_ptr() if hasattr(x, "data_ptr") else None
        sig = lambda x: (x.data_ptr(), x.numel()) if hasattr(x, "data_ptr") else None
        key = ("frags", T, bool(actions_in_domain), bool(exo_in_domain)) + tuple(sig(x) for o in outs for x in o[:8]) + (sig(actions), sig(exo))
        cached = self._rollout_io_cache.get(key)
        if cached is None:
            for o in outs:
                if o.msg_log is not None:
                    raise ValueError("rollout_fragments: plane fragments without message logs (alloc_trajectory(T))")
                self._check_rollout_buffers(T, None, None, o)
            if any((o.terminations is None) != (outs[0].terminations is None) for o in outs):
                raise ValueError("rollout_fragments: `terminations` must be present in every fragment or in none")
            torch = _torch()
            for name, x, tail in (("actions", actions, (self.B, self.S)), ("exo", exo, (self.B, self.n_exo))):
                if x is not None and (tuple(x.shape) != (k * T,) + tail or not x.is_contiguous() or x.device != self.device
                                      or x.dtype != (torch.float32 if name == "actions" else torch.uint8)):
                    raise ValueError(f"rollout_fragments: `{name}` must be a contiguous [{k * T}, {tail[0]}, {tail[1]}] tensor on {self.device}")
            arr = (_abi.PhxRolloutFrag * k)()
            for i, o in enumerate(outs):
                arr[i].obs, arr[i].action_out, arr[i].reward = ptr(o.observations), ptr(o.actions), ptr(o.rewards)
                arr[i].terminated, arr[i].truncated = ptr(o.terminations), ptr(o.truncations)
                arr[i].obs_valid, arr[i].reward_valid = ptr(o.obs_valid), ptr(o.reward_valid)
            io = _abi.PhxRolloutIO()
            io.T, io.n_frag = k * T, k
            io.hints = (_abi.RH_ACTIONS_IN_DOMAIN if actions_in_domain else 0) | (_abi.RH_EXO_IN_DOMAIN if exo_in_domain else 0)
            io.frags = C.cast(arr, C.c_void_p)
            io.actions, io.exo = ptr(actions), ptr(exo)
            io.last_obs = ptr(outs[-1].last_obs)
            io.err = self.err.data_ptr()
            cached = (io, C.byref(io), arr, [o._replace(last_obs=None) for o in outs[:-1]] + [outs[-1]])
            if len(self._rollout_io_cache) >= 4:
                self._rollout_io_cache.pop(next(iter(self._rollout_io_cache)))
            self._rollout_io_cache[key] = cached
        rc = self.lib.phx_rollout(self.handle, cached[1], self._stream())
        if rc != 0:
            self._check(rc, "phx_rollout")
        return cached[3]

    # ---- per-env legacy-numpy MT19937 streams (ABI 7, PHX_F_MT19937) -------------------------------------------------
    def mt_seed(self, seeds):
        """``np.random.seed(seeds[b])`` for the stream of env instance b (32-bit integers)."""
        s = np.ascontiguousarray(np.asarray(seeds).astype(np.uint32)).reshape(-1)
        if s.size != self.B:
            raise ValueError(f"mt_seed: {s.size} seeds for {self.B} env instances")
        self._check(self.lib.phx_mt_seed(self.handle, s.ctypes.data, self._stream()), "phx_mt_seed")

    def mt_draw(self, T: int = 1, out=None):
        """The T * n_exo ``np.random.randint(5)`` calls each instance's reference worker makes in T steps, from the instance's own
        stream: u8 [T, B, n_exo] on the device, the layout phx_step (row t) and phx_rollout replay."""
        torch = _torch()
        if out is None:
            out = torch.empty((T, self.B, self.n_exo), dtype=torch.uint8, device=self.device)
        elif out.dtype != torch.uint8 or tuple(out.shape) != (T, self.B, self.n_exo) or not out.is_contiguous():
            raise ValueError("mt_draw: `out` must be a contiguous u8 [T, B, n_exo] tensor on the env's device")
        self._check(self.lib.phx_mt_draw(self.handle, out.data_ptr(), int(T), self._stream()), "phx_mt_draw")
        return out

    def pack_done_flags(self, traj: Trajectory):
        """bit-pack the done planes of a flat fragment into its ``packed_flags`` section (SURVEY 8e iii):
        word w, bit j = truncations.flat[64 w + j] != 0; a second block of words holds `terminations` unless
        the env's kinds never terminate.  One small launch on the current stream."""
        if traj.flat is None:
            raise ValueError("pack_done_flags needs a fragment from alloc_trajectory(flat=True)")
        n = traj.truncations.numel()
        words = (n + 63) // 64
        pf = traj.packed_flags.view(-1)
        self.pack_flags(traj.truncations, pf)
        if pf.numel() >= 2 * words * 8:
            self.pack_flags(traj.terminations, pf[words * 8:])

    def pack_flags(self, plane, dst):
        """u8 plane [n] (any shape, contiguous) -> ceil(n / 64) little-endian 64-bit words at the start of the u8 tensor ``dst``:
        word w, bit j = plane.flat[64 w + j] != 0.  One small launch on the current stream."""
        n = plane.numel()
        self._check(self.lib.phx_pack_flags(plane.data_ptr(), dst.data_ptr(), n, self._stream()), "phx_pack_flags")

    def unpack_flags(self, packed, n: int, out=None):
        """inverse of the packing above: u8 [n] (0 / 1) from ceil(n / 64) little-endian 64-bit words."""
        torch = _torch()
        if out is None:
            out = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(self.lib.phx_unpack_flags(packed.data_ptr(), out.data_ptr(), n, self._stream()),
                    "phx_unpack_flags")
        return out

    def step_graph(self, actions=None, n: Optional[int] = None, policy=None, action_valid=None):
        """Capture ``n`` consecutive ``phx_step`` launches ONCE into a hipGraph and return a replayable
        ``StepGraph``: per-step launch cadence drops from the host's ~6-9 us to the graph's.

        * ``actions`` f32 [n, B, S]: step i reads ``actions[i]`` -- refill the tensor in place between
          replays (``graph.actions``);
        * or ``policy(step_tensors) -> f32 [B, S]``: a capturable torch callable (no host sync) that maps
          the previous step's device outputs to the next actions -- the whole act/step loop is one graph.
        Outputs of the LAST captured step are in ``graph.out`` (the env's persistent StepTensors)."""
        torch = _torch()
        if actions is None and policy is None:
            raise ValueError("step_graph needs an actions tensor [n, B, S] or a policy callable")
        if actions is not None:
            n = actions.shape[0] if n is None else n
            if actions.dtype != torch.float32 or tuple(actions.shape) != (n, self.B, self.S) \
                    or not actions.is_contiguous() or actions.device != self.device:
                raise ValueError(f"actions must be a contiguous f32 tensor [{n}, {self.B}, {self.S}] on {self.device}")
        elif n is None:
            raise ValueError("step_graph(policy=...) needs n")
        self._ensure_step_io()
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.graph(g, stream=side):          # stream capture records the launches, it does not run them
            out = self._step_out                        # the first policy call sees the env's current outputs
            for i in range(n):                          # (after reset(): the reset observations)
                a = actions[i] if actions is not None else policy(out).contiguous()
                out = self.step(a, action_valid)
        torch.cuda.synchronize(self.device)
        return StepGraph(g, actions, self._step_out, n, self)

    def rollout_graph(self, T: int, trajectories):
        """Capture one ``phx_rollout`` of T steps per buffer of ``trajectories`` (from ``alloc_trajectory(T)``), in order, into a
        hipGraph: a collection loop that consumes fragment after fragment replays the graph instead of paying a host launch per
        fragment (the gap between dependent launches drops from the host's cadence to the graph's).  The env advances
        ``len(trajectories) * T`` steps per replay, exactly as the same sequence of ``rollout`` calls would."""
        torch = _torch()
        trajectories = list(trajectories)
        if not trajectories:
            raise ValueError("rollout_graph needs at least one trajectory buffer")
        for tr in trajectories:
            self._check_rollout_buffers(T, None, None, tr)
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.graph(g, stream=side):          # stream capture records the launches, it does not run them
            for tr in trajectories:
                self.rollout(T, out=tr)
        torch.cuda.synchronize(self.device)
        return RolloutGraph(g, trajectories, T)

    def inject(self, messages: List[Message]):
        if not messages:
            return
        arr = (_abi.PhxMsgRec * len(messages))()
        for k, m in enumerate(messages):
            t, is_f, v = payload_to_record(m.payload)
            arr[k].sender = self.spec.index_of(m.sender_id)
            arr[k].receiver = self.spec.index_of(m.receiver_id)
            arr[k].type = t
            if is_f:
                arr[k].payload.f = float(v)
            else:
                arr[k].payload.i = int(v)
        self._check(self.lib.phx_inject(self.handle, arr, len(messages)), "phx_inject")

    def resolve(self, pending: List[Message], network=None):
        torch = _torch()
        self.inject(pending)
        lp = self.msg_log.data_ptr() if self.msg_log is not None else None
        cp = self.msg_count.data_ptr() if self.msg_count is not None else None
        self.err.zero_()
        with torch.cuda.device(self.device):
            self._check(self.lib.phx_resolve(self.handle, self.err.data_ptr(), lp, cp,
                                             self._stream()), "phx_resolve")
        self.raise_errors(network)
        if network is not None and network.resolver.enable_tracking:
            network.resolver._tracked_messages.extend(self.read_log(0))

    # ---- read-back ----------------------------------------------------------------------------
    def read_log_records(self, b: int = 0) -> np.ndarray:
        """structured array (sender, receiver, type, round, raw i64) of env b's last step."""
        if self.msg_log is None:
            raise DeviceError("tracking is disabled (BatchResolver(enable_tracking=True))")
        n = int(self.msg_count[b].item())
        if n > self.spec.trace_cap:
            raise DeviceError(f"message log overflow: {n} > trace capacity {self.spec.trace_cap}")
        raw = self.msg_log[b, :n].cpu().numpy().tobytes()
        dt = np.dtype([("sender", "<u2"), ("receiver", "<u2"), ("type", "<u2"), ("round", "<u2"),
                       ("raw", "<i8")])
        return np.frombuffer(raw, dtype=dt)

    def read_log(self, b: int = 0) -> List[Message]:
        recs = self.read_log_records(b)
        ids = self.spec.agent_ids
        out = []
        for r in recs:
            raw_i = int(r["raw"])
            raw_f = np.array([raw_i], dtype="<i8").view("<f8")[0].item()
            out.append(Message(ids[int(r["sender"])], ids[int(r["receiver"])],
                               record_to_payload(int(r["type"]), raw_i, raw_f)))
        return out

    def raise_errors(self, network=None, err=None):
        """Re-raise per-env soft error codes as the exceptions the reference raises in step()."""
        if err is None:
            err = self.err.cpu().numpy()
        bad = np.flatnonzero(err)
        if bad.size == 0:
            return
        from .network import NetworkError
        b, code = int(bad[0]), int(err[bad[0]])
        where = f"env instance {b} (+{bad.size - 1} more)" if bad.size > 1 else f"env instance {b}"
        if code == _abi.ERR_NETWORK:
            raise NetworkError(f"No connection between sender and receiver ({where}).")
        if code == _abi.ERR_PAYLOAD:
            raise NetworkError(f"Message payload not allowed between these agent types ({where}).")
        if code == _abi.ERR_UNKNOWN_MSG:
            raise ValueError(f"Unknown message type: receiving agent has no handler ({where}).")
        if code == _abi.ERR_ROUND_LIMIT:
            raise RuntimeError("message(s) still in queue after BatchResolver round limit "
                               f"reached ({where}).")
        if code == _abi.ERR_FSM_TRANSITION:                # fsm.py:304-307
            from .fsm import FSMRuntimeError
            raise FSMRuntimeError(f"FiniteStateMachineEnv attempted invalid transition ({where}).")
        if code == _abi.ERR_CONTEXT:                       # ctx[agent_id] of a non-neighbour, context.py:36-37
            raise KeyError(f"agent view / table entry not present in the handler's context ({where}).")
        raise DeviceError(f"per-round message capacity exceeded ({where}); raise queue capacity")

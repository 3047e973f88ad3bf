"""deepterrainrl_amd -- MI355X-native batched rollout engine for the DeepTerrainRL environments.

Python host-side mirror of the reference's scenario interface for the rollout path, bound over the C ABI of
``include/dtrl.h`` with ctypes (plain pointers and sizes, no torch types cross the boundary).

    cScenarioExp / cScenarioPoliEval (one env)        ->  BatchScenario (N envs, one HIP device)
      ParseArgs + Init                                  ->  BatchScenario(arg_file=..., num_envs=...)
      Update(dt)                                        ->  .Update(dt)
      Reset()                                           ->  .Reset(env_ids=None)
      IsTupleBufferFull/GetTuples/ResetTupleBuffer      ->  .DrainTuples()
      EnableExplore/SetExpRate/SetExpTemp/...           ->  .SetExplore(enable, rate, temp, base_rate)
      SetTerrainParamsLerp                              ->  .SetTerrainParamsLerp(lerp)
      GetCharacter()->BuildPose/BuildVel/SetPose/SetVel ->  .BuildPose() / .BuildVel() / .SetPoseVel()
      GetNNController()->RecordPoliState / LoadNet...   ->  .RecordPoliState() / .SetPolicy(weights, scales)
      GetAvgDist/GetNumEpisodes/GetNumCycles            ->  .EvalStats()

The product path is the HIP library ``lib/libdtrl.so``; importing works anywhere, but constructing a BatchScenario
raises ``DtrlError`` when the library or a HIP device is missing -- there is no CPU fallback.
"""
import os as _os
import sys as _sys
import warnings as _warnings


def configure_hw_queues(n=8):
    """Opt-in: ask HIP for `n` hardware queues (GPU_MAX_HW_QUEUES) unless the user has set the variable. The engine drives its env groups on separate
    HIP streams that must not share a hardware queue: HIP multiplexes all streams of a process onto 4 queues by default, and with RCCL's and a
    framework's streams in the same process the two groups' frame kernels ended up serialised (11.2 M instead of 19.1 M env-steps/s).
    The variable is read when the HIP runtime starts, so call this before the first device call of the PROCESS (bench.py and the training tools do);
    importing the package no longer touches the process environment. Returns the value in effect for a runtime that starts after this call."""
    return int(_os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(n))))


def _warn_hw_queues():
    """Called when a batch is created: with a framework in the process and the default queue count the env-group streams may share a queue."""
    v = _os.environ.get("GPU_MAX_HW_QUEUES")
    if "torch" in _sys.modules and (v is None or int(v) < 8):
        _warnings.warn("GPU_MAX_HW_QUEUES is %s: with torch / RCCL streams in this process the engine's env-group streams may share a HIP hardware queue and "
                       "serialise; call deepterrainrl_amd.configure_hw_queues() (or export GPU_MAX_HW_QUEUES=8) before the HIP runtime starts" % (v or "unset (4)"),
                       RuntimeWarning, stacklevel=3)


import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdtrl.so")
LIB_PATH_F32 = os.path.join(_HERE, "lib", "libdtrl_f32.so")   # the opt-in fp32 build of the same source (-physics_precision= f32): distribution-level parity only

DTRL_OK = 0
FLAG_FALLEN, FLAG_STUMBLED, FLAG_NEW_CYCLE, FLAG_STATE_SHIFT = 1, 2, 4, 8
TUPLE_FAIL, TUPLE_EXP_CRITIC, TUPLE_EXP_ACTOR = 1, 2, 4

# every symbol include/dtrl.h declares (tests check the built library exports all of them)
ABI_SYMBOLS = [
    "dtrl_create", "dtrl_destroy", "dtrl_reset", "dtrl_step", "dtrl_step_begin", "dtrl_step_end", "dtrl_step_updates", "dtrl_run_frames", "dtrl_set_policy",
    "dtrl_policy_num_params", "dtrl_build_output_offset_scale", "dtrl_load_scale_file", "dtrl_write_scale_file", "dtrl_set_explore", "dtrl_set_terrain_lerp", "dtrl_drain_tuples",
    "dtrl_get_pose_vel", "dtrl_set_pose_vel", "dtrl_get_contact_cache", "dtrl_set_contact_cache", "dtrl_get_link_states", "dtrl_add_perturb", "dtrl_apply_rand_force", "dtrl_get_cycle_info", "dtrl_get_action_table", "dtrl_get_poli_state", "dtrl_get_flags", "dtrl_get_torques", "dtrl_get_contacts",
    "dtrl_get_ctrl", "dtrl_sample_ground", "dtrl_eval_stats", "dtrl_dims", "dtrl_kernel_time_ms", "dtrl_last_error", "dtrl_version",
    "dtrl_terrain_build", "dtrl_terrain_load_file", "dtrl_args_parse_string",
    "dtrl_drain_tuples_device", "dtrl_tuple_stats", "dtrl_set_policy_device", "dtrl_get_dist_log", "dtrl_reset_avg_dist", "dtrl_write_dist_log", "dtrl_get_ground_window", "dtrl_drain_tuples_packed", "dtrl_get_policy_output", "dtrl_set_tuple_pipelining", "dtrl_step_end_begin", "dtrl_command_action", "dtrl_side_stream", "dtrl_step_poll", "dtrl_set_policy_device_on", "dtrl_set_policy_device_async",
]


class DtrlError(RuntimeError):
    pass


def _bind(path):
    if not os.path.exists(path):
        raise DtrlError("HIP extension missing: %s (run __graft_entry__.build() / make -C deepterrainrl_amd/csrc)" % path)
    L = C.CDLL(path)
    vp, i32p, u32p, u64p, dp, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.dtrl_create.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.dtrl_destroy.argtypes = [vp]
    L.dtrl_reset.argtypes = [vp, vp, C.c_int, vp]
    L.dtrl_step.argtypes = [vp, C.c_double]
    L.dtrl_step_begin.argtypes = [vp, C.c_double]
    L.dtrl_step_end.argtypes = [vp]
    L.dtrl_step_updates.argtypes = [vp, C.c_int]
    L.dtrl_run_frames.argtypes = [vp, C.c_int, C.c_double]
    L.dtrl_set_policy.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, vp]
    L.dtrl_policy_num_params.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.dtrl_build_output_offset_scale.argtypes = [vp, vp, vp]
    L.dtrl_load_scale_file.argtypes = [vp, C.c_char_p]
    L.dtrl_write_scale_file.argtypes = [vp, C.c_char_p]
    L.dtrl_set_explore.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double]
    L.dtrl_set_terrain_lerp.argtypes = [vp, C.c_double]
    L.dtrl_drain_tuples.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    for name in ("dtrl_get_pose_vel", "dtrl_get_torques"):
        getattr(L, name).argtypes = [vp, vp, C.c_int, vp, vp]
    L.dtrl_set_pose_vel.argtypes = [vp, vp, C.c_int, vp, vp]
    L.dtrl_get_contact_cache.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.dtrl_set_contact_cache.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.dtrl_command_action.argtypes = [vp, vp, C.c_int, vp]
    L.dtrl_side_stream.restype = C.c_void_p; L.dtrl_side_stream.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.dtrl_get_link_states.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.dtrl_add_perturb.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    L.dtrl_apply_rand_force.argtypes = [vp, vp, C.c_int, C.c_uint64]
    L.dtrl_get_cycle_info.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.dtrl_get_action_table.argtypes = [vp, C.POINTER(C.c_int), vp]
    L.dtrl_get_poli_state.argtypes = [vp, vp, C.c_int, vp]
    L.dtrl_get_policy_output.argtypes = [vp, vp, C.c_int, vp]
    L.dtrl_set_tuple_pipelining.argtypes = [vp, C.c_int]
    L.dtrl_step_end_begin.argtypes = [vp, C.c_double]
    L.dtrl_step_poll.argtypes = [vp, C.c_double, C.POINTER(C.c_int)]
    L.dtrl_get_flags.argtypes = [vp, vp, C.c_int, vp]
    L.dtrl_get_contacts.argtypes = [vp, vp, C.c_int, vp]
    L.dtrl_get_ctrl.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.dtrl_sample_ground.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.dtrl_drain_tuples_packed.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.dtrl_get_ground_window.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int64)]
    L.dtrl_eval_stats.argtypes = [vp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dtrl_dims.argtypes = [vp] + [C.POINTER(C.c_int)] * 8
    L.dtrl_kernel_time_ms.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    L.dtrl_terrain_build.argtypes = [C.c_char_p, vp, C.c_uint64, C.c_double, vp, C.c_int, C.POINTER(C.c_int), dp]
    L.dtrl_terrain_load_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.dtrl_args_parse_string.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dtrl_drain_tuples_device.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.dtrl_tuple_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    L.dtrl_set_policy_device.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, vp]
    L.dtrl_set_policy_device_on.argtypes = [vp, vp, C.c_size_t, vp]
    L.dtrl_set_policy_device_async.argtypes = [vp, vp, C.c_size_t, vp]
    L.dtrl_get_dist_log.argtypes = [vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.dtrl_reset_avg_dist.argtypes = [vp]
    L.dtrl_write_dist_log.argtypes = [vp, C.c_char_p]
    L.dtrl_last_error.restype = C.c_char_p
    L.dtrl_last_error.argtypes = [vp]
    L.dtrl_version.restype = C.c_char_p
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class BatchScenario:
    """N reference-shaped scenarios (cScenarioExp / cScenarioPoliEval / cScenarioSimChar) stepped as one batch on one GPU."""

    def _library(self):
        """The HIP library; there is no other backend in the product (tests of the host logic subclass this from tests/conftest.py). `-physics_precision= f32`
        (extra_args) selects the fp32 build of the same source; the library itself refuses a precision it was not built for."""
        return _bind(LIB_PATH_F32 if self._precision == "f32" else LIB_PATH)

    def __init__(self, arg_file=None, num_envs=1, data_root=None, device_id=-1, extra_args=None):
        self._precision = str((extra_args or {}).get("physics_precision", "f64"))
        self._lib = self._library()
        _warn_hw_queues()
        argv = []
        if extra_args:
            for k, v in extra_args.items():
                argv += ["-%s=" % k, str(v)]
        if data_root is not None:
            argv += ["-data_root=", str(data_root)]
        if arg_file is not None:
            argv += ["-arg_file=", str(arg_file)]
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        h = C.c_void_p()
        rc = self._lib.dtrl_create(arr, len(argv), int(num_envs), int(device_id), C.byref(h))
        if rc != DTRL_OK:
            raise DtrlError("dtrl_create failed (%d): %s" % (rc, self._lib.dtrl_last_error(None).decode()))
        self._h = h
        self.num_envs = int(num_envs)
        d = [C.c_int() for _ in range(8)]
        self._lib.dtrl_dims(self._h, *[C.byref(x) for x in d])
        self.L, self.D, self.S, self.A, self.P, self.nn_out, self.num_frags, self.frag_size = (x.value for x in d)
        self.W = 1 + 2 * self.S + self.A

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dtrl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != DTRL_OK:
            raise DtrlError("dtrl call failed (%d): %s" % (rc, self._lib.dtrl_last_error(self._h).decode()))

    def _ids(self, env_ids):
        if env_ids is None:
            return None, self.num_envs
        a = np.ascontiguousarray(env_ids, np.int32)
        return a, len(a)

    # ---- cScenario interface ----
    def Update(self, dt=1.0 / 30.0):
        self._chk(self._lib.dtrl_step(self._h, float(dt)))

    def UpdateBegin(self, dt=1.0 / 30.0):
        """Queue one outer frame on the engine's stream and return (dtrl_step_begin); pair with UpdateEnd()."""
        self._chk(self._lib.dtrl_step_begin(self._h, dt))

    def UpdateEnd(self):
        self._chk(self._lib.dtrl_step_end(self._h))

    def UpdateEndBegin(self, dt=1.0 / 30.0):
        """UpdateEnd() + UpdateBegin(dt) without the barrier between them (dtrl_step_end_begin): each env group is relaunched as soon as its own frame is done."""
        self._chk(self._lib.dtrl_step_end_begin(self._h, float(dt)))

    def UpdatePoll(self, dt=1.0 / 30.0):
        """dtrl_step_poll: relaunch, without blocking, every env group whose frame has already ended (between two UpdateEndBegin calls, after the drain).
        Returns how many groups were relaunched."""
        n = C.c_int(0)
        self._chk(self._lib.dtrl_step_poll(self._h, float(dt), C.byref(n)))
        return n.value

    def StepUpdates(self, n):
        self._chk(self._lib.dtrl_step_updates(self._h, int(n)))

    def RunFrames(self, frames, dt=1.0 / 30.0):
        self._chk(self._lib.dtrl_run_frames(self._h, int(frames), float(dt)))

    def Reset(self, env_ids=None, terrain_seeds=None):
        ids, n = self._ids(env_ids)
        seeds = None if terrain_seeds is None else np.ascontiguousarray(terrain_seeds, np.uint64)
        self._chk(self._lib.dtrl_reset(self._h, _p(ids), n, _p(seeds)))

    def SetPolicy(self, weights, in_off=None, in_scale=None, out_off=None, out_scale=None):
        w = np.ascontiguousarray(weights, np.float32)
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (in_off, in_scale, out_off, out_scale)]
        self._chk(self._lib.dtrl_set_policy(self._h, _p(w), w.size, *[_p(a) for a in arrs]))
        self._policy = (w.copy(), arrs[0], arrs[1])     # kept for the NN-activation recorder (recorders.py)

    def PolicyNumParams(self):
        n = C.c_size_t()
        self._chk(self._lib.dtrl_policy_num_params(self._h, C.byref(n)))
        return n.value

    def BuildNNOutputOffsetScale(self):
        off = np.zeros(self.nn_out); sc = np.zeros(self.nn_out)
        self._chk(self._lib.dtrl_build_output_offset_scale(self._h, _p(off), _p(sc)))
        return off, sc

    def LoadModel(self, model_file):
        """cNeuralNet::LoadModel (learning/NeuralNet.cpp:110-135): Caffe HDF5 weights by layer name, then the normalisers from
        '<model>_scale.txt' next to it when that file exists (GetOffsetScaleFile)."""
        from . import caffe_hdf5
        w = caffe_hdf5.load_mace_weights(model_file, self.num_frags)
        if w.size != self.PolicyNumParams():
            raise DtrlError("%s holds %d parameters, the deploy net needs %d" % (model_file, w.size, self.PolicyNumParams()))
        self.SetPolicy(w)
        scale = os.path.splitext(model_file)[0] + "_scale.txt"
        if os.path.exists(scale):
            self.LoadScale(scale)
        return w

    def LoadScale(self, path):
        """cNeuralNet::LoadScale: install the normaliser vectors of a '<model>_scale.txt' file (weights untouched)."""
        self._chk(self._lib.dtrl_load_scale_file(self._h, os.fsencode(path)))

    def WriteOffsetScale(self, path):
        """cNeuralNet::WriteOffsetScale: write the current normalisers in the reference's file format."""
        self._chk(self._lib.dtrl_write_scale_file(self._h, os.fsencode(path)))

    def SetExplore(self, enable, rate, temp, base_rate):
        self._chk(self._lib.dtrl_set_explore(self._h, int(enable), float(rate), float(temp), float(base_rate)))

    def SetTerrainParamsLerp(self, lerp):
        self._chk(self._lib.dtrl_set_terrain_lerp(self._h, float(lerp)))

    def DrainTuples(self, cap=None):
        cap = cap or max(2 * self.num_envs, 64)
        buf = getattr(self, "_drain_buf", None)
        if buf is None or buf[0].shape[0] < cap:     # (20 MB at 4096 envs: kept, not allocated and zero-filled per call)
            buf = self._drain_buf = (np.empty((cap, self.W), np.float32), np.empty(cap, np.uint32), np.empty(cap, np.int32))
        rows, fl, ids = buf
        n = C.c_int()
        self._chk(self._lib.dtrl_drain_tuples(self._h, _p(rows), _p(fl), _p(ids), cap, C.byref(n)))
        return rows[:n.value].copy(), fl[:n.value].copy(), ids[:n.value].copy()

    def DrainTuplesDevice(self, rows_ptr, flags_ptr, ids_ptr, cap):
        """dtrl_drain_tuples_device: raw DEVICE pointers (e.g. tensor.data_ptr()) of float32 [cap, W] / uint32 [cap] / int32 [cap]; returns n."""
        n = C.c_int()
        self._chk(self._lib.dtrl_drain_tuples_device(self._h, C.c_void_p(rows_ptr), C.c_void_p(flags_ptr) if flags_ptr else None, C.c_void_p(ids_ptr) if ids_ptr else None, int(cap), C.byref(n)))
        return n.value

    def DrainTuplesPacked(self, block_ptr, block_rows, want_count=False):
        """Pending tuples -> ONE device block [block_rows + 1, W + 2] float32 (header row, then rows sorted by env id with the flag word and the GLOBAL
        env id as int32 bit patterns in the two extra columns); the ring is emptied. Returns the row count when want_count (a 4-byte read-back), else None."""
        n = C.c_int(0)
        self._chk(self._lib.dtrl_drain_tuples_packed(self._h, C.c_void_p(int(block_ptr)), int(block_rows), C.byref(n) if want_count else None))
        return n.value if want_count else None

    def TupleStats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64(); cap = C.c_int32()
        self._chk(self._lib.dtrl_tuple_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(cap)))
        return {"pending": a.value, "drained": b.value, "dropped": c.value, "capacity": cap.value}

    def SetPolicyDevice(self, weights_ptr, n, in_off_ptr=None, in_scale_ptr=None, out_off_ptr=None, out_scale_ptr=None):
        """dtrl_set_policy_device: raw DEVICE pointers (float32 weights in Caffe blob order, float64 normalisers or None)."""
        q = [C.c_void_p(x) if x else None for x in (in_off_ptr, in_scale_ptr, out_off_ptr, out_scale_ptr)]
        self._chk(self._lib.dtrl_set_policy_device(self._h, C.c_void_p(weights_ptr), int(n), *q))

    def SetPolicyDeviceOn(self, weights_ptr, n, stream_ptr):
        """dtrl_set_policy_device_on: weights only, the re-layout kernel on the caller's stream (hipStream_t as an int), returns when it has run."""
        self._chk(self._lib.dtrl_set_policy_device_on(self._h, C.c_void_p(weights_ptr), int(n), C.c_void_p(int(stream_ptr)) if stream_ptr else None))

    def SetPolicyDeviceAsync(self, weights_ptr, n, stream_ptr):
        """dtrl_set_policy_device_async: weights only, the re-layout kernel queued on the caller's stream, NO host wait; every env's next launch waits for it on the device."""
        self._chk(self._lib.dtrl_set_policy_device_async(self._h, C.c_void_p(weights_ptr), int(n), C.c_void_p(int(stream_ptr)) if stream_ptr else None))

    def GetDistLog(self):
        """cScenarioPoliEval::GetDistLog over the batch: (distances, env ids), grouped by env, episodes in time order."""
        n = C.c_int()
        self._chk(self._lib.dtrl_get_dist_log(self._h, None, None, 0, C.byref(n)))
        d = np.zeros(n.value); ids = np.zeros(n.value, np.int32)
        if n.value:
            self._chk(self._lib.dtrl_get_dist_log(self._h, _p(d), _p(ids), n.value, C.byref(n)))
        return d, ids

    def ResetAvgDist(self):
        self._chk(self._lib.dtrl_reset_avg_dist(self._h))

    def OutputResults(self, out_file):
        """cOptScenarioPoliEval::OutputResults: append the dist log as one line to out_file."""
        self._chk(self._lib.dtrl_write_dist_log(self._h, os.fsencode(out_file)))

    # ---- character / controller observability ----
    def PoseVel(self, env_ids=None):
        ids, n = self._ids(env_ids)
        q = np.zeros((n, self.D)); qd = np.zeros((n, self.D))
        self._chk(self._lib.dtrl_get_pose_vel(self._h, _p(ids), n, _p(q), _p(qd)))
        return q, qd

    def BuildPose(self, env_ids=None):
        return self.PoseVel(env_ids)[0]

    def BuildVel(self, env_ids=None):
        return self.PoseVel(env_ids)[1]

    def LinkStates(self, env_ids=None):
        """World COM position [n, L, 2], COM velocity [n, L, 2] and body angle [n, L] of every link (GetBodyPart(i)->GetPos() ...)."""
        ids, n = self._ids(env_ids)
        c = np.zeros((n, self.L, 2)); v = np.zeros((n, self.L, 2)); a = np.zeros((n, self.L))
        self._chk(self._lib.dtrl_get_link_states(self._h, _p(ids), n, _p(c), _p(v), _p(a)))
        return c, v, a

    def AddPerturb(self, link, force, duration, local_pos=None, env_ids=None):
        """cScenarioSimChar::AddPerturb with an ePerturbForce: world-frame force [n, 2] on body part link [n] for duration [n] seconds."""
        ids, n = self._ids(env_ids)
        link = np.ascontiguousarray(np.broadcast_to(np.asarray(link, np.int32), (n,)))
        force = np.ascontiguousarray(np.broadcast_to(np.asarray(force, np.float64), (n, 2)))
        duration = np.ascontiguousarray(np.broadcast_to(np.asarray(duration, np.float64), (n,)))
        lp = None if local_pos is None else np.ascontiguousarray(np.broadcast_to(np.asarray(local_pos, np.float64), (n, 2)))
        self._chk(self._lib.dtrl_add_perturb(self._h, _p(ids), n, _p(link), _p(lp) if lp is not None else None, _p(force), _p(duration)))

    def ApplyRandForce(self, seed=0, env_ids=None):
        """cScenarioSimChar::ApplyRandForce(): random body part / direction / magnitude / duration per env."""
        ids, n = self._ids(env_ids)
        self._chk(self._lib.dtrl_apply_rand_force(self._h, _p(ids), n, int(seed)))

    def SetPoseVel(self, q, qd, env_ids=None):
        ids, n = self._ids(env_ids)
        q = np.ascontiguousarray(q, np.float64).reshape(n, self.D); qd = np.ascontiguousarray(qd, np.float64).reshape(n, self.D)
        self._chk(self._lib.dtrl_set_pose_vel(self._h, _p(ids), n, _p(q), _p(qd)))

    def ContactCache(self, env_ids=None):
        """The persistent contact points of Bullet's manifolds (dtrl_get_contact_cache): (count[n], ids[n, 24], lambda[n, 24]) -- with (q, qd) the whole dynamic state."""
        ids, n = self._ids(env_ids)
        cnt = np.zeros(n, np.int32); rid = np.zeros((n, 24), np.int32); lam = np.zeros((n, 24))
        self._chk(self._lib.dtrl_get_contact_cache(self._h, _p(ids), n, _p(cnt), _p(rid), _p(lam)))
        return cnt, rid, lam

    def SetContactCache(self, count, row_ids, lam, env_ids=None):
        ids, n = self._ids(env_ids)
        cnt = np.ascontiguousarray(count, np.int32).reshape(n); rid = np.ascontiguousarray(row_ids, np.int32).reshape(n, 24); lam = np.ascontiguousarray(lam, np.float64).reshape(n, 24)
        self._chk(self._lib.dtrl_set_contact_cache(self._h, _p(ids), n, _p(cnt), _p(rid), _p(lam)))

    def SideStream(self, k=0):
        """(hipStream_t as an int, start delay in us measured at creation) of the k-th side stream: kernels queued there start on the compute units
        `reserve_cus` keeps out of the frame launches, while a frame is in flight. (None, -1.0) without a reservation."""
        d = C.c_double(-1.0)
        p = self._lib.dtrl_side_stream(self._h, int(k), C.byref(d))
        return (int(p) if p else None), float(d.value)

    def CommandAction(self, action_id, env_ids=None):
        """cCharController::CommandAction on the listed envs (all by default): action_id (an int, or one per env) is taken at the next cycle."""
        ids, n = self._ids(env_ids)
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(action_id, np.int32), (n,)))
        self._chk(self._lib.dtrl_command_action(self._h, _p(ids), n, _p(a)))

    def RecordPoliState(self, env_ids=None):
        ids, n = self._ids(env_ids)
        s = np.zeros((n, self.S))
        self._chk(self._lib.dtrl_get_poli_state(self._h, _p(ids), n, _p(s)))
        return s

    def SetTuplePipelining(self, on=True):
        """Two tuple rings, switched by every UpdateBegin: UpdateEnd(f); UpdateBegin(f + 1); DrainTuplesPacked(...) hands out frame f's tuples while
        frame f + 1 runs (the batched form of the reference's env threads feeding the trainer while the others keep stepping, scenarios/ScenarioTrain.cpp:376-410)."""
        self._chk(self._lib.dtrl_set_tuple_pipelining(self._h, 1 if on else 0))

    def PolicyOutput(self, env_ids=None):
        """cNeuralNet::GetLayerState("output") after the controller's last Eval, un-normalised like Eval's out_y (learning/NeuralNet.cpp:352-375, 814-834)."""
        ids, n = self._ids(env_ids)
        y = np.zeros((n, self.nn_out))
        self._chk(self._lib.dtrl_get_policy_output(self._h, _p(ids), n, _p(y)))
        return y

    def Flags(self, env_ids=None):
        ids, n = self._ids(env_ids)
        f = np.zeros(n, np.uint32)
        self._chk(self._lib.dtrl_get_flags(self._h, _p(ids), n, _p(f)))
        return f

    def Torques(self, env_ids=None):
        ids, n = self._ids(env_ids)
        a = np.zeros((n, self.D)); b = np.zeros((n, self.D))
        self._chk(self._lib.dtrl_get_torques(self._h, _p(ids), n, _p(a), _p(b)))
        return a, b

    def Contacts(self, env_ids=None):
        ids, n = self._ids(env_ids)
        f = np.zeros((n, self.L), np.int32)
        self._chk(self._lib.dtrl_get_contacts(self._h, _p(ids), n, _p(f)))
        return f

    def Ctrl(self, env_ids=None):
        ids, n = self._ids(env_ids)
        st = np.zeros(n, np.int32); ph = np.zeros(n); aid = np.zeros(n, np.int32); prm = np.zeros((n, self.P)); tg = np.zeros((n, self.L))
        self._chk(self._lib.dtrl_get_ctrl(self._h, _p(ids), n, _p(st), _p(ph), _p(aid), _p(prm), _p(tg)))
        return st, ph, aid, prm, tg

    def CycleInfo(self, env_ids=None):
        """Per env: cycle counter, reset counter, COM [n, 2] and simulated time at the start of the current cycle, optimisable params of the current action [n, frag_size]."""
        ids, n = self._ids(env_ids)
        nc = np.zeros(n, np.int64); nr = np.zeros(n, np.int64); com = np.zeros((n, 2)); t = np.zeros(n); prm = np.zeros((n, self.frag_size))
        self._chk(self._lib.dtrl_get_cycle_info(self._h, _p(ids), n, _p(nc), _p(nr), _p(com), _p(t), _p(prm)))
        return nc, nr, com, t, prm

    def ActionTable(self):
        """cTerrainRLCharController::BuildActionOptParams for every action: [n_actions, frag_size]."""
        na = C.c_int(0)
        self._chk(self._lib.dtrl_get_action_table(self._h, C.byref(na), None))
        tab = np.zeros((na.value, self.frag_size))
        self._chk(self._lib.dtrl_get_action_table(self._h, C.byref(na), _p(tab)))
        return tab

    def SampleGround(self, env, xs):
        xs = np.ascontiguousarray(xs, np.float64); n = len(xs)
        h = np.zeros(n); seg = np.zeros(n, np.int32); i = np.zeros(n, np.int32); j = np.zeros(n, np.int32)
        self._chk(self._lib.dtrl_sample_ground(self._h, int(env), n, _p(xs), _p(h), _p(seg), _p(i), _p(j)))
        return h, seg, i, j

    def GroundWindow(self, env):
        """One env's two-segment ground window in logical order: [(min_x, max_x, heights float32[w])] * 2 and the number of segments built so far
        (-1 unless -terrain_gen= device)."""
        w = (C.c_int32 * 2)(); mn = (C.c_double * 2)(); mx = (C.c_double * 2)(); nb = C.c_int64(0)
        h0 = np.zeros(512, np.float32); h1 = np.zeros(512, np.float32)
        self._chk(self._lib.dtrl_get_ground_window(self._h, int(env), w, mn, mx, _p(h0), _p(h1), 512, C.byref(nb)))
        return [(mn[0], mx[0], h0[:w[0]].copy()), (mn[1], mx[1], h1[:w[1]].copy())], nb.value

    def EvalStats(self):
        a = C.c_double(); e, c, r = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._lib.dtrl_eval_stats(self._h, C.byref(a), C.byref(e), C.byref(c), C.byref(r)))
        return {"avg_dist": a.value, "episodes": e.value, "cycles": c.value, "resets": r.value}

    def KernelTimeMs(self):
        a = C.c_double(); n = C.c_int64()
        self._chk(self._lib.dtrl_kernel_time_ms(self._h, C.byref(a), C.byref(n)))
        return a.value, n.value


def version():
    return _bind(LIB_PATH).dtrl_version().decode()


# ---- host-side utility entry points (no batch, no device) ----
def terrain_build(type_name, params40, seed, width):
    """cTerrainGen2D::GetTerrainFunc(type)(width, params, cRand(seed), data): (float32 heights, width added)."""
    L = _bind(LIB_PATH)
    p = np.ascontiguousarray(params40, np.float64); buf = np.zeros(8192, np.float32); n = C.c_int(); w = C.c_double()
    rc = L.dtrl_terrain_build(type_name.encode(), _p(p), int(seed), float(width), _p(buf), 8192, C.byref(n), C.byref(w))
    if rc != DTRL_OK:
        raise DtrlError("dtrl_terrain_build failed (%d): %s" % (rc, L.dtrl_last_error(None).decode()))
    return buf[:n.value].copy(), w.value


def terrain_load_file(path, max_sets=8):
    """Terrain file -> (type name, [n_sets, 40] parameter vectors in cTerrainGen2D::eParams order)."""
    L = _bind(LIB_PATH)
    buf = C.create_string_buffer(64); prm = np.zeros((max_sets, 40)); n = C.c_int()
    rc = L.dtrl_terrain_load_file(os.fsencode(path), buf, 64, _p(prm), max_sets, C.byref(n))
    if rc != DTRL_OK:
        raise DtrlError("dtrl_terrain_load_file failed (%d): %s" % (rc, L.dtrl_last_error(None).decode()))
    return buf.value.decode(), prm[:n.value].copy()


def args_parse_string(argv, key):
    """cArgParser(argv) + AppendArgs(-arg_file=) + ParseString(key): (value or None, number of tokens)."""
    L = _bind(LIB_PATH)
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    buf = C.create_string_buffer(4096); found = C.c_int(); nt = C.c_int()
    rc = L.dtrl_args_parse_string(arr, len(argv), key.encode(), buf, 4096, C.byref(found), C.byref(nt))
    if rc != DTRL_OK:
        raise DtrlError("dtrl_args_parse_string failed (%d): %s" % (rc, L.dtrl_last_error(None).decode()))
    return (buf.value.decode() if found.value else None), nt.value

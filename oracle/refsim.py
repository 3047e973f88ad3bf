"""TEST INFRASTRUCTURE -- ctypes binding of oracle/_ref/libref_sim.so: the reference's OWN scenario / character / controller / ground sources
(scenarios/ScenarioSimChar, ScenarioExp(MACE), ScenarioPoliEval; sim/World, SimCharacter, Joint, ContactManager, GroundVar2D, the Dog / Raptor / Goat
controllers with their Q / CACLA / MACE heads, ImpPDController, ...) compiled unchanged from /root/reference by oracle/_ref_build/Makefile against
stand-in third-party headers: Bullet as a state container WITHOUT a physics step, Caffe's network as a forward callback (see
oracle/_ref_build/ref_sim_api.cpp for what is whose). Only tests/ may import this module.

LockStep runs one reference scenario and one oracle env side by side: the reference's cWorld::Update calls a hook instead of Bullet; the hook
advances the ORACLE by one env-step and writes the oracle's post-physics state into the reference's rigid bodies through the reference's own
cSimCharacter::SetPose / SetVel, plus the contact sample distances as manifold points. Everything else of that env-step -- contact flags, ground
update, controller, torque clamp, fall logic, cycle bookkeeping -- is computed by the reference from that state and compared with the oracle's.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref_sim.so")
# two builds of the same sources (oracle/_ref_build/Makefile): "f64" keeps the Bullet stand-in's rigid-body state in double, so a comparison with the fp64
# oracle isolates LOGIC; "f32" is the reference's real configuration (btScalar = float): what its float rounding does to origins, scalings and poses
LIB_PATHS = {"f64": LIB_PATH, "f32": os.path.join(HERE, "_ref", "libref_sim_f32.so")}
_libs = {}
NN_FWD = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double))
STEP_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_double, C.c_int)
POST_SUBSTEP = C.CFUNCTYPE(None, C.c_void_p, C.c_double)
SI_OPTS = ("iterations", "erp", "erp2", "split_impulse", "split_threshold", "warmstarting", "warmstart_factor", "breaking", "max_points", "use_margin", "link_contacts", "safe_margin", "relative_breaking", "vertex_contacts", "friction_warmstart", "friction_skip", "friction_dir", "interleave", "friction_ws_lifted")
SI_DEFAULTS = dict(iterations=10, erp=0.2, erp2=0.8, split_impulse=1, split_threshold=-0.04, warmstarting=1, warmstart_factor=0.85, breaking=0.02, max_points=4, use_margin=1, link_contacts=1, safe_margin=1, relative_breaking=1, vertex_contacts=1, friction_warmstart=1, friction_skip=1, friction_dir=1, interleave=0, friction_ws_lifted=1)


def available():
    return all(os.path.exists(p) for p in LIB_PATHS.values())


def lib(variant="f64"):
    if variant not in _libs:
        if not os.path.exists(LIB_PATHS[variant]):
            raise RuntimeError("reference library missing: run `make -C oracle/_ref_build` where /root/reference exists")
        L = C.CDLL(LIB_PATHS[variant])
        vp = C.c_void_p
        L.ref_nn_config.argtypes = [C.c_int, C.c_int, NN_FWD]
        L.ref_scn_create.restype = vp; L.ref_scn_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_ulong]
        L.ref_scn_free.argtypes = [vp]
        L.ref_scn_seed_ground_and_reset.argtypes = [vp, C.c_ulong]
        L.ref_scn_reset.argtypes = [vp]
        L.ref_scn_set_step_hook.argtypes = [vp, STEP_HOOK, vp]
        L.ref_scn_update.argtypes = [vp, C.c_double]
        L.ref_scn_dims.argtypes = [vp] + [C.POINTER(C.c_int)] * 5
        L.ref_scn_get_pose_vel.argtypes = [vp, vp, vp]
        L.ref_scn_set_pose_vel.argtypes = [vp, vp, vp]
        L.ref_scn_get_bodies.argtypes = [vp, vp, vp, vp, vp]
        L.ref_scn_get_torques.argtypes = [vp, vp, vp]
        L.ref_scn_set_contacts.argtypes = [vp, C.c_int, vp, vp]
        L.ref_scn_get_contact_flags.argtypes = [vp, vp]
        L.ref_scn_get_flags.restype = C.c_uint; L.ref_scn_get_flags.argtypes = [vp]
        L.ref_scn_get_ctrl.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), vp, vp, vp]
        L.ref_scn_get_poli_state.argtypes = [vp, vp]
        L.ref_scn_get_poli_action.argtypes = [vp, vp]
        L.ref_scn_calc_reward.restype = C.c_double; L.ref_scn_calc_reward.argtypes = [vp]
        L.ref_scn_build_output_offset_scale.argtypes = [vp, vp, vp]
        L.ref_scn_set_net_scale.argtypes = [vp, vp, vp, vp, vp]
        L.ref_scn_enable_explore.argtypes = [vp, C.c_int]
        L.ref_scn_command_action.argtypes = [vp, C.c_int]
        L.ref_scn_sample_ground.restype = C.c_double; L.ref_scn_sample_ground.argtypes = [vp, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.ref_scn_ground_segment.argtypes = [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_scn_add_pair_contacts.argtypes = [vp, C.c_int, vp, vp, vp]
        L.ref_scn_time.restype = C.c_double; L.ref_scn_time.argtypes = [vp]
        L.ref_scn_com.argtypes = [vp, vp, vp]
        L.ref_scn_drain_tuples.argtypes = [vp, vp, vp, C.c_int]
        L.ref_scn_eval_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), vp, C.c_int]
        L.ref_scn_use_bullet_si.argtypes = [vp, vp, C.c_int]
        L.ref_scn_si_contacts.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.ref_scn_set_post_substep.argtypes = [vp, POST_SUBSTEP, vp]
        _libs[variant] = L
    return _libs[variant]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_keep = []   # ctypes callbacks must outlive the C side's use of them


def nn_config(in_size, out_size, raw_forward, variant="f64"):
    """raw_forward(x_norm ndarray [in]) -> y_norm ndarray [out]: the network between the two normalisations."""
    def cb(xp, yp):
        x = np.ctypeslib.as_array(xp, shape=(in_size,)).copy()
        y = np.asarray(raw_forward(x), np.float64)
        np.ctypeslib.as_array(yp, shape=(out_size,))[:] = y
    f = NN_FWD(cb); _keep.append(f)
    lib(variant).ref_nn_config(int(in_size), int(out_size), f)


class RefScenario:
    KINDS = {"sim_char": 0, "exp_mace": 1, "poli_eval": 2, "exp": 3}

    def __init__(self, kind, arg_file, cwd, extra_args=None, global_seed=1, variant="f64"):
        self.variant = variant
        argv = []
        for k, v in (extra_args or {}).items():
            argv += ["-%s=" % k, str(v)]
        argv += ["-arg_file=", arg_file]
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        self.h = lib(self.variant).ref_scn_create(self.KINDS[kind], arr, len(argv), os.fsencode(cwd), int(global_seed))
        if not self.h:
            raise RuntimeError("the reference scenario failed to initialise (%s)" % arg_file)
        d = [C.c_int() for _ in range(5)]
        lib(self.variant).ref_scn_dims(self.h, *[C.byref(x) for x in d])
        self.L, self.D, self.S, self.A, self.P = (x.value for x in d)
        self._hook = None

    def __del__(self):
        try:
            lib(self.variant).ref_scn_free(self.h)
        except Exception:
            pass

    def seed_ground_and_reset(self, seed): lib(self.variant).ref_scn_seed_ground_and_reset(self.h, int(seed))
    def reset(self): lib(self.variant).ref_scn_reset(self.h)
    def update(self, dt=1.0 / 30.0): lib(self.variant).ref_scn_update(self.h, float(dt))

    def set_step_hook(self, fn):
        """fn(dt, substeps) is called in place of Bullet's stepSimulation."""
        self._hook = STEP_HOOK(lambda user, dt, n: fn(dt, n))
        lib(self.variant).ref_scn_set_step_hook(self.h, self._hook, None)

    def use_bullet_si(self, **opts):
        """Physics = oracle/or_bullet_si.h (the maximal-coordinate sequential-impulse restatement of Bullet 2.8x's published algorithm) instead of a
        hook: the reference's own stepSimulation call then integrates its rigid bodies. opts override SI_DEFAULTS (Bullet's defaults)."""
        unknown = set(opts) - set(SI_OPTS)
        assert not unknown, unknown
        v = np.array([float(opts.get(k, SI_DEFAULTS[k])) for k in SI_OPTS], np.float64)
        lib(self.variant).ref_scn_use_bullet_si(self.h, _p(v), len(v))

    def si_contacts(self, cap=256):
        la = np.zeros(cap, np.int32); lb = np.zeros(cap, np.int32); d = np.zeros(cap); jn = np.zeros(cap)
        n = lib(self.variant).ref_scn_si_contacts(self.h, _p(la), _p(lb), _p(d), _p(jn), cap)
        return la[:n], lb[:n], d[:n], jn[:n]

    def set_post_substep(self, fn):
        """fn(dt) after every iteration of the env-step loop (scenarios/ScenarioSimChar.cpp:162-173)."""
        self._post = POST_SUBSTEP(lambda user, dt: fn(dt))
        lib(self.variant).ref_scn_set_post_substep(self.h, self._post, None)

    def pose_vel(self):
        q = np.zeros(self.D); qd = np.zeros(self.D); lib(self.variant).ref_scn_get_pose_vel(self.h, _p(q), _p(qd)); return q, qd

    def set_pose_vel(self, q, qd):
        q = np.ascontiguousarray(q, np.float64); qd = np.ascontiguousarray(qd, np.float64); lib(self.variant).ref_scn_set_pose_vel(self.h, _p(q), _p(qd))

    def bodies(self):
        p = np.zeros((self.L, 2)); a = np.zeros(self.L); v = np.zeros((self.L, 2)); w = np.zeros(self.L)
        lib(self.variant).ref_scn_get_bodies(self.h, _p(p), _p(a), _p(v), _p(w)); return p, a, v, w

    def torques(self):
        t = np.zeros(self.L); b = np.zeros(self.L); lib(self.variant).ref_scn_get_torques(self.h, _p(t), _p(b)); return t, b

    def set_contacts(self, links, dists):
        l = np.ascontiguousarray(links, np.int32); d = np.ascontiguousarray(dists, np.float64)
        lib(self.variant).ref_scn_set_contacts(self.h, len(l), _p(l), _p(d))

    def contact_flags(self):
        f = np.zeros(self.L, np.int32); lib(self.variant).ref_scn_get_contact_flags(self.h, _p(f)); return f

    def flags(self): return lib(self.variant).ref_scn_get_flags(self.h)

    def ctrl(self):
        st, aid = C.c_int(), C.c_int(); ph = C.c_double(); prm = np.zeros(max(self.P, 1)); tg = np.zeros(self.L); act = np.zeros(self.L, np.int32)
        lib(self.variant).ref_scn_get_ctrl(self.h, C.byref(st), C.byref(ph), C.byref(aid), _p(prm), _p(tg), _p(act))
        return st.value, ph.value, aid.value, prm, tg, act

    def poli_state(self):
        s = np.zeros(self.S); lib(self.variant).ref_scn_get_poli_state(self.h, _p(s)); return s

    def poli_action(self):
        a = np.zeros(self.A); lib(self.variant).ref_scn_get_poli_action(self.h, _p(a)); return a

    def calc_reward(self): return lib(self.variant).ref_scn_calc_reward(self.h)

    def build_output_offset_scale(self, n_out):
        o = np.zeros(n_out); s = np.zeros(n_out); lib(self.variant).ref_scn_build_output_offset_scale(self.h, _p(o), _p(s)); return o, s

    def set_net_scale(self, io, isc, oo, osc):
        a = [np.ascontiguousarray(x, np.float64) for x in (io, isc, oo, osc)]; lib(self.variant).ref_scn_set_net_scale(self.h, *[_p(x) for x in a])

    def enable_explore(self, on): lib(self.variant).ref_scn_enable_explore(self.h, int(on))
    def command_action(self, a): lib(self.variant).ref_scn_command_action(self.h, int(a))

    def sample_ground(self, x):
        v = C.c_int(); g = C.c_double(); h = lib(self.variant).ref_scn_sample_ground(self.h, float(x), C.byref(v), C.byref(g)); return h, v.value, g.value

    def ground_segment(self, slot):
        buf = np.zeros(1024, np.float32); a, b = C.c_double(), C.c_double()
        n = lib(self.variant).ref_scn_ground_segment(self.h, slot, _p(buf), 1024, C.byref(a), C.byref(b)); return buf[:n].copy(), a.value, b.value

    def add_pair_contacts(self, pairs, dists):
        a = np.ascontiguousarray(pairs[:, 0], np.int32); b = np.ascontiguousarray(pairs[:, 1], np.int32); d = np.ascontiguousarray(dists, np.float64)
        lib(self.variant).ref_scn_add_pair_contacts(self.h, len(d), _p(a), _p(b), _p(d))

    def time(self): return lib(self.variant).ref_scn_time(self.h)

    def com(self):
        p = np.zeros(2); v = np.zeros(2); lib(self.variant).ref_scn_com(self.h, _p(p), _p(v)); return p, v

    def drain_tuples(self, cap=64):
        W = 1 + 2 * self.S + self.A
        rows = np.zeros((cap, W)); fl = np.zeros(cap, np.uint32)
        n = lib(self.variant).ref_scn_drain_tuples(self.h, _p(rows), _p(fl), cap); return rows[:n], fl[:n]

    def eval_stats(self):
        a = C.c_double(); e, c, n = C.c_int(), C.c_int(), C.c_int(); log = np.zeros(256)
        lib(self.variant).ref_scn_eval_stats(self.h, C.byref(a), C.byref(e), C.byref(c), C.byref(n), _p(log), 256)
        return dict(avg_dist=a.value, episodes=e.value, cycles=c.value, dist_log=log[:n.value].copy())


class LockStep:
    """One reference scenario + one oracle env advanced together (see the module docstring). `on_step(k, ref, env)` is called with the reference's
    view of env-step k after the reference finished it (k counts from 0)."""

    def __init__(self, ref, env, contact_band=0.05):
        self.ref, self.env, self.band = ref, env, contact_band
        self.k = 0
        self.snap = None
        self.records = []
        ref.set_step_hook(self._hook)

    def _oracle_snapshot(self):
        e = self.env
        st, ph, aid, prm, tg = e.ctrl()
        tc, ta = e.tau()
        return dict(q=e.pose_vel()[0], qd=e.pose_vel()[1], state=st, phase=ph, action_id=aid, params=prm.copy(), pd_targets=tg.copy(), tau_ctrl=tc.copy(), tau=ta.copy(),
                    contacts=e.contacts().copy(), flags=e.flags())

    def _ref_snapshot(self):
        r = self.ref
        st, ph, aid, prm, tg, act = r.ctrl()
        q, qd = r.pose_vel()
        return dict(q=q, qd=qd, state=st, phase=ph, action_id=aid, params=prm.copy(), pd_targets=tg.copy(), tau=r.torques()[0].copy(), contacts=r.contact_flags().copy(), flags=r.flags(), pd_active=act.copy())

    def _hook(self, dt, substeps):
        if self.snap is not None:
            self.records.append((self.snap, self._ref_snapshot()))   # the env-step that just ended
        e = self.env
        e.step(1)
        q, qd = e.pose_vel()
        self.ref.set_pose_vel(q, qd)
        d = e.contact_distances()
        links, dists = np.nonzero(d < self.band)[0], d[d < self.band]
        self.ref.set_contacts(links, dists)
        pairs, pd = e.pair_distances()                      # link--link manifolds go in as well: the reference's contact manager has to ignore them
        near = pd < self.band
        self.n_pair_manifolds = getattr(self, "n_pair_manifolds", 0) + int((pd < 0.001 / e.m.world_scale).sum())
        if near.any():
            self.ref.add_pair_contacts(pairs[near], pd[near])
        self.snap = self._oracle_snapshot()
        self.k += 1

    def update(self, dt=1.0 / 30.0):
        t0 = self.ref.time()
        self.ref.update(dt)
        rs = self._ref_snapshot()
        # the reference resets INSIDE Update (fall / episode end, after the last env-step of the frame): what it shows now is the fresh episode's initial state
        rs["after_reset"] = self.ref.time() < t0 + 0.5 * dt
        self.records.append((self.snap, rs))
        self.snap = None

"""TEST INFRASTRUCTURE -- ctypes binding of oracle/_ref/libref_core.so: the REFERENCE'S OWN util/Rand, util/ArgParser, sim/TerrainGen2D,
anim/KinTree, sim/SpAlg, sim/RBDModel and sim/RBDUtil translation units compiled unchanged from /root/reference by
oracle/_ref_build/Makefile (stand-in Eigen / jsoncpp headers only; see oracle/_ref_build/ref_api.cpp for what is whose).

Used by tests/test_reference_pin.py to check the restatement in oracle/or_*.h and the product's host code against the reference itself,
and by tests/golden/make_ref_golden.py to freeze reference outputs as fixtures. Only tests/ may import this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref_core.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("reference library missing: run `make -C oracle/_ref_build` where /root/reference exists")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.ref_rand_stream.argtypes = [C.c_ulong, C.c_int, C.c_double, C.c_double, C.c_int, vp]
        L.ref_args_load.restype = vp; L.ref_args_load.argtypes = [C.c_char_p]
        L.ref_args_load_argv.restype = vp; L.ref_args_load_argv.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p]
        L.ref_args_free.argtypes = [vp]
        L.ref_args_count.argtypes = [vp]
        L.ref_args_string.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
        L.ref_args_int.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.ref_args_double.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
        L.ref_args_bool.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.ref_terrain_param_name.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.ref_terrain_default_params.argtypes = [vp]
        L.ref_terrain_vert_spacing.restype = C.c_double
        L.ref_terrain_load_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, vp, C.c_int]
        L.ref_terrain_build.argtypes = [C.c_char_p, vp, C.c_ulong, C.c_double, C.c_int, C.c_float, vp, C.c_int, C.POINTER(C.c_double)]
        L.ref_char_load.restype = vp; L.ref_char_load.argtypes = [C.c_char_p]
        L.ref_char_free.argtypes = [vp]
        L.ref_char_dims.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.ref_char_tables.argtypes = [vp, vp, vp]
        L.ref_rbd.argtypes = [vp] * 10
        L.ref_char_ref_theta.argtypes = [vp, vp]
        L.ref_inv_dyna.argtypes = [vp] * 5
        L.ref_kin_bodies.argtypes = [vp] * 6
        L.ref_kin_world_vel.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rand_stream(seed, kind, a=0.0, b=1.0, n=16):
    """kind: 'double' RandDouble(a, b), 'int' RandInt(a, b), 'norm' RandDoubleNorm(a, b), 'coin' FlipCoin(a), 'sign' RandSign()."""
    out = np.zeros(n)
    lib().ref_rand_stream(int(seed), {"double": 0, "int": 1, "norm": 2, "coin": 3, "sign": 4}[kind], float(a), float(b), n, _p(out))
    return out


class RefArgs:
    """cArgParser over a file (optionally with a command line in front, as optimizer/Main.cpp builds it)."""

    def __init__(self, file, argv=None):
        L = lib()
        if argv:
            arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
            self.h = L.ref_args_load_argv(arr, len(argv), os.fsencode(file) if file else None)
        else:
            self.h = L.ref_args_load(os.fsencode(file))

    def __del__(self):
        try:
            lib().ref_args_free(self.h)
        except Exception:
            pass

    def count(self):
        return lib().ref_args_count(self.h)

    def string(self, key):
        buf = C.create_string_buffer(4096)
        n = lib().ref_args_string(self.h, key.encode(), buf, 4096)
        return None if n < 0 else buf.value.decode()

    def int(self, key):
        v = C.c_int()
        return v.value if lib().ref_args_int(self.h, key.encode(), C.byref(v)) else None

    def double(self, key):
        v = C.c_double()
        return v.value if lib().ref_args_double(self.h, key.encode(), C.byref(v)) else None

    def bool(self, key):
        v = C.c_int()
        return bool(v.value) if lib().ref_args_bool(self.h, key.encode(), C.byref(v)) else None


def terrain_param_names():
    L = lib(); out = []
    for i in range(L.ref_terrain_num_params()):
        buf = C.create_string_buffer(64); L.ref_terrain_param_name(i, buf, 64); out.append(buf.value.decode())
    return out


def terrain_default_params():
    p = np.zeros(lib().ref_terrain_num_params()); lib().ref_terrain_default_params(_p(p)); return p


def terrain_load_file(path, max_sets=8):
    L = lib(); n = L.ref_terrain_num_params()
    buf = C.create_string_buffer(64); params = np.zeros((max_sets, n))
    k = L.ref_terrain_load_file(os.fsencode(path), buf, 64, _p(params), max_sets)
    if k < 0:
        raise IOError("cannot read terrain file %s (%d)" % (path, k))
    return buf.value.decode(), params[:k].copy()


def terrain_build(type_name, params40, seed, width, prefix=0, prefix_h=0.0):
    """One strip by the reference's terrain function of that name; returns (heights float32, width added)."""
    buf = np.zeros(8192, np.float32); p = np.ascontiguousarray(params40, np.float64); w = C.c_double()
    n = lib().ref_terrain_build(type_name.encode(), _p(p), int(seed), float(width), int(prefix), float(prefix_h), _p(buf), 8192, C.byref(w))
    assert n <= 8192
    return buf[:n].copy(), w.value


class RefChar:
    """cKinTree tables + cRBDModel of one character file, by the reference's own code."""

    def __init__(self, char_file):
        self.h = lib().ref_char_load(os.fsencode(char_file))
        if not self.h:
            raise IOError("reference loader rejected %s" % char_file)
        a, b, m = C.c_int(), C.c_int(), C.c_double()
        lib().ref_char_dims(self.h, C.byref(a), C.byref(b), C.byref(m))
        self.L, self.D, self.total_mass = a.value, b.value, m.value

    def __del__(self):
        try:
            lib().ref_char_free(self.h)
        except Exception:
            pass

    def tables(self):
        j = np.zeros((self.L, 9)); b = np.zeros((self.L, 8))
        lib().ref_char_tables(self.h, _p(j), _p(b)); return j, b

    def ref_theta(self):
        out = np.zeros(self.L); lib().ref_char_ref_theta(self.h, _p(out)); return out

    def rbd(self, q, qd):
        """dict(H, C, grav, J, com, com_vel, joint_pos) after cRBDModel::Update(q, qd)."""
        D, L = self.D, self.L
        q = np.ascontiguousarray(q, np.float64); qd = np.ascontiguousarray(qd, np.float64)
        o = dict(H=np.zeros((D, D)), C=np.zeros(D), grav=np.zeros(D), J=np.zeros((6, D)), com=np.zeros(3), com_vel=np.zeros(3), joint_pos=np.zeros((L, 3)))
        lib().ref_rbd(self.h, _p(q), _p(qd), _p(o["H"]), _p(o["C"]), _p(o["grav"]), _p(o["J"]), _p(o["com"]), _p(o["com_vel"]), _p(o["joint_pos"]))
        return o

    def inv_dyna(self, q, qd, acc):
        q, qd, acc = (np.ascontiguousarray(x, np.float64) for x in (q, qd, acc)); tau = np.zeros(self.D)
        lib().ref_inv_dyna(self.h, _p(q), _p(qd), _p(acc), _p(tau)); return tau

    def kin_bodies(self, q):
        q = np.ascontiguousarray(q, np.float64)
        bp = np.zeros((self.L, 3)); bt = np.zeros(self.L); jp = np.zeros((self.L, 3)); jt = np.zeros(self.L)
        lib().ref_kin_bodies(self.h, _p(q), _p(bp), _p(bt), _p(jp), _p(jt)); return bp, bt, jp, jt

    def world_vel(self, q, qd, parent_id, attach):
        q, qd = (np.ascontiguousarray(x, np.float64) for x in (q, qd)); a = np.ascontiguousarray(attach, np.float64); out = np.zeros(3)
        lib().ref_kin_world_vel(self.h, _p(q), _p(qd), int(parent_id), _p(a), _p(out)); return out

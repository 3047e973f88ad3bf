"""ORACLE (test infrastructure, NOT product code).

Python-side loader + ctypes binding for oracle/_ref/libdtrl_oracle.so.

Reads the reference's own input files (args/*.txt, data/characters/*.txt, data/controllers/**, data/states/*.txt,
data/terrain/*.txt, data/policies/*/nets/*deploy.prototxt, *_scale.txt) with Python's json module -- deliberately
independent of the product's C++ loader so the two cross-check each other -- and flattens them into the OrcModel
struct of or_model.h.

Restates (file:line relative to /root/reference):
  util/ArgParser.cpp:42-108      arg-file tokenisation ('-key= value', '//' comments, first match wins)
  anim/KinTree.cpp:409-457,990-1023  joint table (defaults LimLow=1, LimHigh=0; root attach zeroed)
  anim/KinTree.cpp:160-183       body defs        sim/PDController.cpp:19-78  PD params
  sim/DogController.cpp:399-454  ReadParams (MiscParams + StateParams order), :629-700 Controllers block
  sim/SimDog.cpp:5-33            collision groups (tail collides with nothing)
  sim/TerrainGen2D.cpp:8-81      terrain param names/defaults, ParseType
  scenarios/ScenarioSimChar.cpp:76-108, 670-706   scenario args, terrain file
  scenarios/ScenarioExp.cpp:31-45                  exploration args
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import json
import math
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libdtrl_oracle.so")

MAXL, MAXD, MAXP, MAXSETS, MAXACT, MAXTP, MAXCP = 24, 26, 40, 8, 16, 4, 64


class OrcModel(C.Structure):
    _fields_ = [
        ("char_type", C.c_int32), ("ctrl_type", C.c_int32), ("L", C.c_int32), ("D", C.c_int32),
        ("parent", C.c_int32 * MAXL), ("joint_type", C.c_int32 * MAXL),
        ("attach", (C.c_double * 3) * MAXL),
        ("lim_lo", C.c_double * MAXL), ("lim_hi", C.c_double * MAXL), ("ref_theta", C.c_double * MAXL),
        ("body_attach", (C.c_double * 3) * MAXL), ("body_theta", C.c_double * MAXL),
        ("body_size", (C.c_double * 3) * MAXL), ("body_mass", C.c_double * MAXL),
        ("col_group", C.c_int32 * MAXL),
        ("kp", C.c_double * MAXL), ("kd", C.c_double * MAXL), ("torque_lim", C.c_double * MAXL), ("target_theta", C.c_double * MAXL),
        ("use_world", C.c_int32 * MAXL),
        ("n_sets", C.c_int32), ("n_params", C.c_int32),
        ("ctrl_params", (C.c_double * MAXP) * MAXSETS),
        ("opt_mask", C.c_int32 * MAXP), ("exp_noise", C.c_double),
        ("n_actions", C.c_int32),
        ("act_idx0", C.c_int32 * MAXACT), ("act_idx1", C.c_int32 * MAXACT),
        ("act_blend", C.c_double * MAXACT), ("act_cyclic", C.c_int32 * MAXACT),
        ("default_action", C.c_int32), ("enable_grav_comp", C.c_int32), ("enable_virtual_forces", C.c_int32),
        ("target_vel_x", C.c_double),
        ("pose0", C.c_double * MAXD), ("vel0", C.c_double * MAXD),
        ("valid_init_pos_x", C.c_int32), ("init_pos_x", C.c_double),
        ("num_update_steps", C.c_int32), ("num_sim_substeps", C.c_int32), ("world_scale", C.c_double),
        ("terrain_type", C.c_int32), ("n_terrain_sets", C.c_int32),
        ("terrain_params", (C.c_double * 40) * MAXTP), ("terrain_blend", C.c_double),
        ("scenario", C.c_int32), ("tuple_buffer_size", C.c_int32), ("enable_explore", C.c_int32),
        ("exp_rate", C.c_double), ("exp_temp", C.c_double), ("exp_base_rate", C.c_double),
        ("link_contacts", C.c_int32), ("n_cpairs", C.c_int32), ("cpair_a", C.c_int32 * MAXCP), ("cpair_b", C.c_int32 * MAXCP),
        ("contact_margin", C.c_double), ("link_margin", C.c_double * MAXL), ("warm_start", C.c_int32), ("link_brk", C.c_double * MAXL), ("mass_matrix_every", C.c_int32),
    ]


class OrcNetDesc(C.Structure):
    _fields_ = [
        ("n_terrain", C.c_int32), ("n_char", C.c_int32), ("conv_ch", C.c_int32 * 3), ("conv_k", C.c_int32 * 3),
        ("fc_terr", C.c_int32), ("fc_trunk", C.c_int32), ("fc_head", C.c_int32), ("n_frags", C.c_int32), ("frag_size", C.c_int32),
    ]


# ---------------------------------------------------------------------------------------------------------
def parse_arg_file(path):
    """util/ArgParser.cpp:42-108: whitespace tokens, '//' starts a comment until end of line."""
    toks = []
    with open(path, "r") as f:
        for line in f:
            line = line.split("//", 1)[0]
            toks.extend(line.split())
    return toks


def args_to_dict(tokens):
    """first match wins (FindKeyIndex scans from the front); a key is '-name=' (len >= 3)."""
    def is_key(t):
        return len(t) >= 3 and t[0] == "-" and t[-1] == "="
    out = {}
    for i, t in enumerate(tokens):
        if is_key(t) and i + 1 < len(tokens) and not is_key(tokens[i + 1]):
            out.setdefault(t[1:-1], tokens[i + 1])
    return out


TERRAIN_TYPES = ["flat", "gaps", "steps", "walls", "bumps", "mixed", "narrow_gaps", "slopes", "slopes_gaps", "slopes_steps",
                 "slopes_walls", "slopes_mixed", "slopes_narrow_gaps", "cliffs"]
TERRAIN_PARAMS = [
    ("GapSpacingMin", 4), ("GapSpacingMax", 7), ("GapWMin", 0.5), ("GapWMax", 2), ("GapHMin", -2), ("GapHMax", -2),
    ("WallSpacingMin", 6), ("WallSpacingMax", 8), ("WallWMin", 0.2), ("WallWMax", 0.2), ("WallHMin", 0.25), ("WallHMax", 0.5),
    ("StepSpacingMin", 5), ("StepSpacingMax", 7), ("StepH0Min", 0.1), ("StepH0Max", 0.4), ("StepH1Min", -0.4), ("StepH1Max", -0.1),
    ("BumpHMin", 0), ("BumpHMax", 0.03),
    ("NarrowGapSpacingMin", 3), ("NarrowGapSpacingMax", 6), ("NarrowGapDistMin", 0.1), ("NarrowGapDistMax", 0.4),
    ("NarrowGapWMin", 0.15), ("NarrowGapWMax", 0.5), ("NarrowGapDepthMin", -2), ("NarrowGapDepthMax", -2),
    ("NarrowGapCountMin", 1), ("NarrowGapCountMax", 4),
    ("CliffSpacingMin", 5), ("CliffSpacingMax", 7), ("CliffH0Min", 0.1), ("CliffH0Max", 0.4), ("CliffH1Min", -0.4), ("CliffH1Max", -0.1),
    ("CliffMiniCountMax", 0), ("SlopeDeltaRange", 0.25), ("SlopeDeltaMin", -0.35), ("SlopeDeltaMax", 0.35),
]
DOG_MISC = ["TransTime", "Cv", "BackForceX", "BackForceY", "FrontForceX", "FrontForceY"]
DOG_STATES = ["BackStance", "Extend", "FrontStance", "Gather"]
DOG_STATE_PARAMS = ["SpineCurve", "Shoulder", "Elbow", "Hip", "Knee", "Ankle"]
# sim/SimDog.cpp:5-33: body=2, front leg=4, back leg=8, tail=0
DOG_COL = [2] * 9 + [0] * 4 + [4] * 4 + [8] * 4
# sim/SimRaptor.cpp:5-29: body (incl. tail) = 2, legs = 4
RAPTOR_COL = [2] * 11 + [4] * 8
RAPTOR_MISC = ["TransTime", "Cv", "Cd", "ForceX", "ForceY"]
RAPTOR_STATES = ["Contact", "Down", "Passing", "Up"]
RAPTOR_STATE_PARAMS = ["RootPitch", "SpineCurve", "StanceHip", "StanceKnee", "StanceAnkle", "SwingHip", "SwingKnee", "SwingAnkle"]
# sim/RaptorController.cpp:71-114 gOptParamsMasks
RAPTOR_OPT_MASK = [0, 1, 1, 0, 0] + [1, 0, 1, 1, 1, 1, 1, 1] * 2 + [0, 0, 1, 1, 1, 1, 1, 1] * 2
CTRL_NAMES = {"dog": ("dog", 0), "dog_mace": ("dog", 1), "goat_mace": ("dog", 1), "raptor": ("raptor", 0), "raptor_mace": ("raptor", 1), "dog_cacla": ("dog", 2),
              "raptor_cacla": ("raptor", 2)}   # sim/RaptorControllerCacla.cpp (mExpNoise 0.15 by character type below)
SCENARIOS = {"sim_char": 0, "train": 1, "train_mace": 1, "exp": 1, "exp_mace": 1, "train_cacla": 1, "exp_cacla": 1, "poli_eval": 2}


def load_json(path):
    with open(path, "r") as f:
        return json.load(f)


def read_dog_ctrl_params(path):
    """sim/DogController.cpp:399-454"""
    d = load_json(path)
    v = [float(d["MiscParams"][k]) for k in DOG_MISC]
    for s in DOG_STATES:
        v += [float(d["StateParams"][s][k]) for k in DOG_STATE_PARAMS]
    v[0] = abs(v[0]); v[1] = abs(v[1])  # PostProcessParams
    return v


def read_raptor_ctrl_params(path):
    """sim/RaptorController.cpp:442-495"""
    d = load_json(path)
    v = [float(d["MiscParams"][k]) for k in RAPTOR_MISC]
    for s in RAPTOR_STATES:
        v += [float(d["StateParams"][s][k]) for k in RAPTOR_STATE_PARAMS]
    v[0] = abs(v[0]); v[1] = abs(v[1]); v[2] = abs(v[2])  # PostProcessParams
    return v


def terrain_params_from_json(obj):
    return [float(obj.get(name, dflt)) for name, dflt in TERRAIN_PARAMS]


def build_model(arg_file, root, overrides=None):
    """Flatten an args/*.txt scenario into (OrcModel, info dict). `root` is the directory the relative paths in the
    arg file resolve against (the reference is run with its repo root as cwd)."""
    args = args_to_dict(parse_arg_file(os.path.join(root, arg_file)))
    if overrides:
        args.update({k: str(v) for k, v in overrides.items()})
    m = OrcModel()
    char = load_json(os.path.join(root, args["character_file"]))
    joints = char["Skeleton"]["Joints"]
    bodies = char["BodyDefs"]
    L = len(joints)
    assert L == len(bodies) and L <= MAXL
    char_name, ctrl_type = CTRL_NAMES[args.get("char_ctrl", "dog")]
    assert char_name == args.get("char_type", char_name)
    m.char_type = 0 if char_name == "dog" else 1
    m.ctrl_type = ctrl_type
    m.L = L
    D = 0
    for j, jd in enumerate(joints):
        m.parent[j] = int(jd.get("Parent", -1))
        assert m.parent[j] < j
        m.joint_type[j] = int(jd.get("Type", 0))
        is_root = m.parent[j] < 0
        for k, key in enumerate(("AttachX", "AttachY", "AttachZ")):
            m.attach[j][k] = 0.0 if is_root else float(jd.get(key, 0))
        m.lim_lo[j] = float(jd.get("LimLow", 1))
        m.lim_hi[j] = float(jd.get("LimHigh", 0))
        D += 3 if m.joint_type[j] == 1 else 1
    m.D = D
    for j, bd in enumerate(bodies):
        assert bd["Shape"] == "box"
        m.body_mass[j] = float(bd["Mass"])
        m.body_theta[j] = float(bd.get("Theta", 0))
        for k, key in enumerate(("AttachX", "AttachY", "AttachZ")):
            m.body_attach[j][k] = float(bd.get(key, 0))
        for k, key in enumerate(("Param0", "Param1", "Param2")):
            m.body_size[j][k] = float(bd.get(key, 0))
    # cSimCharacter::BuildConstraints (sim/SimCharacter.cpp:846-853): ref_theta = -angle(RotMatToAxisAngle(BodyJointTrans(parent) * ParentChildTrans(zero pose)
    # * BodyJointTrans(child))): the rotation parts multiply to R_z(theta_parent + theta_child), RotMatToAxisAngle returns acos((trace - 1) / 2) in [0, pi]
    # (util/MathUtil.cpp:128-149). The hinge limits apply to theta + ref_theta (sim/World.cpp:543-553, 624-626): only the dog's / goat's root body is rotated (0.61),
    # so spine0, tail0 and hip carry ref_theta = -0.61 and everything else 0
    for j in range(L):
        p = m.parent[j]
        if p < 0:
            m.ref_theta[j] = 0.0
        else:
            c = math.cos(m.body_theta[p] + m.body_theta[j])
            m.ref_theta[j] = -math.acos(max(-1.0, min(1.0, c)))
    pds = char["PDControllers"]
    assert len(pds) == L
    for j, pd in enumerate(pds):
        m.kp[j] = float(pd.get("Kp", 0)); m.kd[j] = float(pd.get("Kd", 0))
        m.torque_lim[j] = float(pd.get("TorqueLim", 0)); m.target_theta[j] = float(pd.get("TargetTheta", 0))
        m.use_world[j] = int(pd.get("UseWorldCoord", 0) != 0)
    if m.char_type == 0:
        assert L == 21
        for j in range(L):
            m.col_group[j] = DOG_COL[j]
    else:
        assert L == 19
        for j in range(L):
            m.col_group[j] = RAPTOR_COL[j]
    # link--link collision pairs (sim/SimDog.cpp:73-81, sim/SimRaptor.cpp; sim/World.cpp:626): same non-zero collision group, not joined by a hinge, and
    # overlapping in z (joint attach z accumulated down the chain + body attach z, box depth Param2): the raptor's two legs share a group but sit
    # 0.16 m apart in z with 0.065 m deep boxes, so they pass each other freely
    zpos = [0.0] * L
    for j in range(L):
        p = m.parent[j]
        zpos[j] = (zpos[p] if p >= 0 else 0.0) + float(joints[j].get("AttachZ", 0))
    zc = [zpos[j] + float(bodies[j].get("AttachZ", 0)) for j in range(L)]
    n = 0
    for a in range(L):
        for b in range(a + 1, L):
            if m.col_group[a] == 0 or m.col_group[a] != m.col_group[b] or m.parent[b] == a or m.parent[a] == b:
                continue
            if abs(zc[a] - zc[b]) >= 0.5 * (m.body_size[a][2] + m.body_size[b][2]):
                continue
            assert n < MAXCP
            m.cpair_a[n] = a; m.cpair_b[n] = b; n += 1
    m.n_cpairs = n
    m.link_contacts = int(args.get("link_contacts", 1))
    ctrl = char["Controllers"]
    files = ctrl["Files"]
    m.n_sets = len(files)
    m.n_params = 30 if m.char_type == 0 else 37
    mask = [0] + [1] * 29 if m.char_type == 0 else RAPTOR_OPT_MASK
    for i, b in enumerate(mask):
        m.opt_mask[i] = b
    m.exp_noise = 0.2 if m.char_type == 0 else 0.15
    for s, fpath in enumerate(files):
        v = (read_dog_ctrl_params if m.char_type == 0 else read_raptor_ctrl_params)(os.path.join(root, fpath))
        for i, x in enumerate(v):
            m.ctrl_params[s][i] = x
    acts = ctrl["Actions"]
    m.n_actions = len(acts)
    for a, ad in enumerate(acts):
        m.act_idx0[a] = int(ad["ParamIdx0"]); m.act_idx1[a] = int(ad["ParamIdx1"])
        m.act_blend[a] = float(ad["Blend"]); m.act_cyclic[a] = int(bool(ad["Cyclic"]))
    m.default_action = int(ctrl.get("DefaultAction", 0))
    m.enable_grav_comp = int(bool(ctrl.get("EnableGravityCompensation", True)))
    m.enable_virtual_forces = int(bool(ctrl.get("EnableVirtualForces", True)))   # only cRaptorController reads this key; dog VF is always on
    m.target_vel_x = 2.0 if args.get("char_ctrl") == "goat_mace" else 4.0
    state = load_json(os.path.join(root, args["state_file"]))
    assert len(state["Pose"]) == D and len(state["Vel"]) == D
    for i in range(D):
        m.pose0[i] = float(state["Pose"][i]); m.vel0[i] = float(state["Vel"][i])
    m.valid_init_pos_x = int("char_init_pos_x" in args)
    m.init_pos_x = float(args.get("char_init_pos_x", 0))
    m.num_update_steps = int(args.get("num_update_steps", 20))
    m.num_sim_substeps = int(args.get("num_sim_substeps", 1))
    m.world_scale = float(args.get("world_scale", 1))
    m.contact_margin = float(args.get("collision_margin", 0.04)) / m.world_scale   # CONVEX_DISTANCE_MARGIN, world-scaled units -> metres
    # btBoxShape::btBoxShape -> setSafeMargin(halfExtents, 0.1): margin = min(CONVEX_DISTANCE_MARGIN, 0.1 * smallest half extent), in the world-scaled units
    # cWorld::BuildBoxShape hands to Bullet (sim/World.cpp:475-482: scale * size / 2)
    safe = int(args.get("safe_margin", 1))
    for j in range(L):
        he = min(0.5 * m.body_size[j][k] for k in range(3))
        m.link_margin[j] = min(m.contact_margin, 0.1 * he) if safe else m.contact_margin
    m.warm_start = int(args.get("warm_start", 1))
    brk = float(args.get("contact_breaking", 0.02))   # gContactBreakingThreshold x the box's angular-motion disc |half extents| (the world scale cancels)
    for j in range(L):
        hx, hy, hz = (0.5 * m.body_size[j][k] for k in range(3))
        m.link_brk[j] = brk * math.sqrt(hx * hx + hy * hy + hz * hz)
    m.mass_matrix_every = max(1, int(args.get("mass_matrix_every", 1)))
    m.terrain_type = 0
    m.n_terrain_sets = 1
    dflt = [d for _, d in TERRAIN_PARAMS]
    for k in range(40):
        m.terrain_params[0][k] = dflt[k]
    tf = args.get("terrain_file", "")
    if tf:
        t = load_json(os.path.join(root, tf))
        m.terrain_type = TERRAIN_TYPES.index(t.get("Type", "flat") or "flat")
        sets = t.get("Params", [])
        if sets:
            m.n_terrain_sets = len(sets)
            for s, obj in enumerate(sets):
                for k, x in enumerate(terrain_params_from_json(obj)):
                    m.terrain_params[s][k] = x
    m.terrain_blend = float(args.get("terrain_blend", 0))
    m.scenario = SCENARIOS.get(args.get("scenario", "sim_char"), 0)
    m.tuple_buffer_size = int(args.get("tuple_buffer_size", 16))
    m.enable_explore = int(m.scenario == 1)
    m.exp_rate = float(args.get("exp_rate", 0.1))
    m.exp_temp = float(args.get("exp_temp", 1))
    m.exp_base_rate = float(args.get("exp_base_rate", 0.01))
    if m.char_type == 0:
        m.enable_virtual_forces = 1
    info = {"args": args, "char": char, "S": 200 + (2 * L - 1) + 2 * L, "n_opt": int(sum(mask)), "root": root}
    return m, info


def parse_deploy_prototxt(path):
    """Extract the MACE-family topology from a Caffe deploy prototxt (text scan; no protobuf schema needed)."""
    txt = open(path).read()
    dims = [int(x) for x in re.findall(r"input_dim:\s*(\d+)", txt)]
    layers = []
    for blk in re.split(r"\blayer\s*\{", txt)[1:]:
        name = re.search(r'name:\s*"([^"]+)"', blk).group(1)
        typ = re.search(r'type:\s*"([^"]+)"', blk).group(1)
        no = re.search(r"num_output:\s*(\d+)", blk)
        kw = re.search(r"kernel_w:\s*(\d+)", blk)
        sp = re.search(r"slice_point:\s*(\d+)", blk)
        layers.append({"name": name, "type": typ, "num_output": int(no.group(1)) if no else None,
                       "kernel_w": int(kw.group(1)) if kw else None, "slice_point": int(sp.group(1)) if sp else None})
    d = OrcNetDesc()
    d.n_terrain = [l for l in layers if l["type"] == "Slice"][0]["slice_point"]
    d.n_char = dims[-1] - d.n_terrain
    convs = [l for l in layers if l["type"] == "Convolution"]
    assert len(convs) == 3
    for i, l in enumerate(convs):
        d.conv_ch[i] = l["num_output"]; d.conv_k[i] = l["kernel_w"]
    ips = {l["name"]: l["num_output"] for l in layers if l["type"] == "InnerProduct"}
    if "ip0" not in ips and {"ip1", "ip2", "output"} <= set(ips):
        # the CACLA actor (dog_actor_deploy.prototxt): one head ip1 -> ip2 -> output; held as a one-fragment MACE-family net whose critic head is zero
        d.fc_terr = ips["terr_ip0"]; d.fc_trunk = ips["ip1"]; d.fc_head = ips["ip2"]; d.n_frags = 1; d.frag_size = ips["output"]
        d.actor_only = True
        return d
    d.fc_terr = ips["terr_ip0"]; d.fc_trunk = ips["ip0"]; d.fc_head = ips["val_ip0"]
    d.n_frags = ips["val_ip1"]; d.frag_size = ips["a0_ip1"]
    assert all(ips["a%d_ip0" % f] == d.fc_head and ips["a%d_ip1" % f] == d.frag_size for f in range(d.n_frags))
    return d


def actor_policy_to_mace(desc, w_actor, out_off, out_scale):
    """Actor blobs (conv0..2, terr_ip0, ip1, ip2, output) and its output normalisers -> the one-fragment MACE-family form the oracle's
    PolicyNet evaluates: an all-zero critic head (val_ip0, val_ip1) behind the trunk, one neutral critic slot in front of the outputs."""
    head = desc.fc_head * desc.fc_trunk + desc.fc_head + desc.fc_head + 1
    tail = desc.fc_head * desc.fc_trunk + desc.fc_head + desc.frag_size * desc.fc_head + desc.frag_size
    w = np.concatenate([w_actor[:len(w_actor) - tail], np.zeros(head, np.float32), w_actor[len(w_actor) - tail:]]).astype(np.float32)
    return w, np.concatenate([[0.0], out_off]), np.concatenate([[1.0], out_scale])


def actor_xavier_weights(desc, seed=1234):
    """Synthetic actor weights in the actor's own blob order (conv0..2, terr_ip0, ip1, ip2, output)."""
    rng = np.random.RandomState(seed)
    out = []

    def blob(nout, fan_in):
        s = np.sqrt(3.0 / fan_in)
        out.append(rng.uniform(-s, s, size=nout * fan_in).astype(np.float32)); out.append(np.zeros(nout, np.float32))
    cin, w = 1, desc.n_terrain
    for l in range(3):
        blob(desc.conv_ch[l], cin * desc.conv_k[l]); cin = desc.conv_ch[l]; w = w - desc.conv_k[l] + 1
    blob(desc.fc_terr, cin * w); blob(desc.fc_trunk, desc.fc_terr + desc.n_char); blob(desc.fc_head, desc.fc_trunk); blob(desc.frag_size, desc.fc_head)
    return np.concatenate(out)


def xavier_weights(desc, seed=1234):
    """Synthetic weights in the flat layout: Caffe 'xavier' filler (uniform +-sqrt(3/fan_in)) and constant-0 biases, in
    blob order conv0..2, terr_ip0, ip0, val_ip0, val_ip1, a{f}_ip0, a{f}_ip1 (the trained *.h5 blobs are absent:
    /root/reference/.MISSING_LARGE_BLOBS)."""
    rng = np.random.RandomState(seed)
    out = []

    def blob(nout, fan_in):
        s = np.sqrt(3.0 / fan_in)
        out.append(rng.uniform(-s, s, size=nout * fan_in).astype(np.float32))
        out.append(np.zeros(nout, np.float32))
    cin, w = 1, desc.n_terrain
    for l in range(3):
        blob(desc.conv_ch[l], cin * desc.conv_k[l]); cin = desc.conv_ch[l]; w = w - desc.conv_k[l] + 1
    blob(desc.fc_terr, cin * w)
    blob(desc.fc_trunk, desc.fc_terr + desc.n_char)
    blob(desc.fc_head, desc.fc_trunk); blob(desc.n_frags, desc.fc_head)
    for _ in range(desc.n_frags):
        blob(desc.fc_head, desc.fc_trunk); blob(desc.frag_size, desc.fc_head)
    return np.concatenate(out)


def load_scale_file(path):
    d = load_json(path)
    return tuple(np.asarray(d[k], np.float64) for k in ("InputOffset", "InputScale", "OutputOffset", "OutputScale"))


def build_output_offset_scale(m, n_frags):
    """sim/BaseControllerMACE.cpp:75-168 + sim/DogControllerMACE.cpp:93-99: NN output offset/scale from controller files."""
    P = m.n_params
    opt = [i for i in range(P) if m.opt_mask[i]]

    def action_opt(a):
        p0 = np.array(m.ctrl_params[m.act_idx0[a]][:P]); p1 = np.array(m.ctrl_params[m.act_idx1[a]][:P])
        b = m.act_blend[a]
        return ((1 - b) * p0 + b * p1)[opt]
    frag = len(opt)
    da = m.default_action
    f_off = -action_opt(da)
    f_scale = np.ones(frag)
    if m.n_actions > 1:
        f_scale = np.zeros(frag)
        for a in range(m.n_actions):
            if a != da:
                f_scale = np.maximum(f_scale, np.abs(action_opt(a) + f_off))
        f_scale = 1.0 / f_scale
    off = np.zeros(n_frags + n_frags * frag); scale = np.ones_like(off)
    for f in range(n_frags):
        bias = np.array(m.ctrl_params[f % m.n_sets][:P])[opt]
        off[f] = -0.5; scale[f] = 2
        off[n_frags + f * frag: n_frags + (f + 1) * frag] = -bias
        scale[n_frags + f * frag: n_frags + (f + 1) * frag] = f_scale
    return off, scale


# ---------------------------------------------------------------------------------------------------------
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle library missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcModel), C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_create_with_policy.restype = C.c_void_p
        L.orc_create_with_policy.argtypes = [C.POINTER(OrcModel), C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcNetDesc)] + [C.c_void_p] * 5
        L.orc_net_num_params.restype = C.c_uint64
        L.orc_net_num_params.argtypes = [C.POINTER(OrcNetDesc)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_update.argtypes = [C.c_void_p, C.c_double]
        L.orc_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
        L.orc_get_pose_vel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_pose_vel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_add_perturb.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 5
        L.orc_get_warm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; L.orc_get_warm.restype = C.c_int
        L.orc_set_warm.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_get_tau.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_contacts.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_flags.restype = C.c_uint32
        L.orc_get_flags.argtypes = [C.c_void_p]
        L.orc_get_ctrl.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.orc_get_poli_state.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.orc_frame_end.argtypes = [C.c_void_p]
        L.orc_command_action.argtypes = [C.c_void_p, C.c_int]
        L.orc_contact_distances.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pair_distances.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_time.restype = C.c_double; L.orc_time.argtypes = [C.c_void_p]
        L.orc_dist_log.restype = C.c_int
        L.orc_dist_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_drain_tuples.restype = C.c_int
        L.orc_drain_tuples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_drain_tuples_f64.restype = C.c_int
        L.orc_drain_tuples_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_rbd.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_ctrl_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_terrain_build.restype = C.c_int
        L.orc_terrain_build.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_int]
        L.orc_sample_ground.restype = C.c_double
        L.orc_sample_ground.argtypes = [C.c_void_p, C.c_double] + [C.POINTER(C.c_int32)] * 4
        L.orc_ground_segment.restype = C.c_int
        L.orc_ground_segment.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.POINTER(C.c_double)] * 3
        L.orc_nn_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_batch_run.restype = C.c_double
        L.orc_batch_run.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(OrcNetDesc)] + [C.c_void_p] * 5 + [C.POINTER(C.c_int64)] * 2
        L.orc_batch_eval.restype = C.c_double
        L.orc_batch_eval.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcNetDesc)] + [C.c_void_p] * 6
        L.orc_batch_trace.restype = C.c_double
        L.orc_batch_trace.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcNetDesc)] + [C.c_void_p] * 12 + [C.c_double]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """One reference-shaped environment (cScenarioSimChar / cScenarioExp / cScenarioPoliEval) on the CPU oracle."""

    def __init__(self, model, terrain_seed=0, rng_seed=0, env_id=0, policy=None):
        self.L_ = lib()
        self.m = model
        self._keep = policy
        if policy is None:
            self.h = self.L_.orc_create(C.byref(model), terrain_seed, rng_seed, env_id)
        else:
            desc, w, io, isc, oo, osc = policy
            w = np.ascontiguousarray(w, np.float32)
            io, isc, oo, osc = (np.ascontiguousarray(x, np.float64) for x in (io, isc, oo, osc))
            self._keep = (desc, w, io, isc, oo, osc)
            self.h = self.L_.orc_create_with_policy(C.byref(model), terrain_seed, rng_seed, env_id, C.byref(desc), _p(w), _p(io), _p(isc), _p(oo), _p(osc))
        d = [C.c_int() for _ in range(5)]
        self.L_.orc_dims(self.h, *[C.byref(x) for x in d])
        self.L, self.D, self.S, self.A, self.P = (x.value for x in d)

    def __del__(self):
        try:
            self.L_.orc_destroy(self.h)
        except Exception:
            pass

    def reset(self): self.L_.orc_reset(self.h)
    def update(self, dt=1.0 / 30.0): self.L_.orc_update(self.h, dt)
    def step(self, n=1): self.L_.orc_step(self.h, n)

    def pose_vel(self):
        q = np.zeros(self.D); qd = np.zeros(self.D)
        self.L_.orc_get_pose_vel(self.h, _p(q), _p(qd)); return q, qd

    def set_pose_vel(self, q, qd):
        q = np.ascontiguousarray(q, np.float64); qd = np.ascontiguousarray(qd, np.float64)
        self.L_.orc_set_pose_vel(self.h, _p(q), _p(qd))

    def warm_cache(self):
        """(count, ids[24], lambda[24]): the integrator's persistent contact rows -- with (q, qd) the whole dynamic state (the product: dtrl_get_contact_cache)"""
        ids = np.zeros(24, np.int32); lam = np.zeros(24)
        n = self.L_.orc_get_warm(self.h, _p(ids), _p(lam))
        return n, ids, lam

    def set_warm_cache(self, count, ids, lam):
        ids = np.ascontiguousarray(ids, np.int32); lam = np.ascontiguousarray(lam, np.float64)
        self.L_.orc_set_warm(self.h, int(count), _p(ids), _p(lam))

    def add_perturb(self, link, local_pos, force, duration):
        self.L_.orc_add_perturb(self.h, int(link), float(local_pos[0]), float(local_pos[1]), float(force[0]), float(force[1]), float(duration))

    def tau(self):
        a = np.zeros(self.D); b = np.zeros(self.D)
        self.L_.orc_get_tau(self.h, _p(a), _p(b)); return a, b

    def contacts(self):
        f = np.zeros(self.L, np.int32); self.L_.orc_get_contacts(self.h, _p(f)); return f

    def flags(self): return self.L_.orc_get_flags(self.h)

    def ctrl(self):
        st = C.c_int(); ph = C.c_double(); aid = C.c_int(); prm = np.zeros(self.P); tg = np.zeros(self.L)
        self.L_.orc_get_ctrl(self.h, C.byref(st), C.byref(ph), C.byref(aid), _p(prm), _p(tg))
        return st.value, ph.value, aid.value, prm, tg

    def poli_state(self):
        s = np.zeros(self.S); self.L_.orc_get_poli_state(self.h, _p(s)); return s

    def bodies(self):
        c = np.zeros((self.L, 2)); v = np.zeros((self.L, 2)); psi = np.zeros(self.L)
        self.L_.orc_get_bodies(self.h, _p(c), _p(v), _p(psi)); return c, v, psi

    def stats(self):
        r, c, e, t = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(); a = C.c_double()
        self.L_.orc_stats(self.h, C.byref(r), C.byref(c), C.byref(e), C.byref(a), C.byref(t))
        return {"resets": r.value, "cycles": c.value, "episodes": e.value, "avg_dist": a.value, "terrain_builds": t.value}

    def frame_end(self): self.L_.orc_frame_end(self.h)
    def command_action(self, a): self.L_.orc_command_action(self.h, int(a))

    def contact_distances(self):
        d = np.zeros((self.L, 6)); self.L_.orc_contact_distances(self.h, _p(d)); return d

    def pair_distances(self):
        """(pairs [n, 2], smallest separation of each link--link collision pair [n]; negative = overlapping)"""
        n = self.m.n_cpairs
        d = np.zeros(max(n, 1)); self.L_.orc_pair_distances(self.h, _p(d))
        return np.array([(self.m.cpair_a[i], self.m.cpair_b[i]) for i in range(n)], np.int32).reshape(n, 2), d[:n]

    def dist_log(self):
        buf = np.zeros(4096); n = self.L_.orc_dist_log(self.h, _p(buf), 4096); return buf[:n].copy()

    def drain_tuples(self, cap=1024, f64=False):
        W = 1 + 2 * self.S + self.A
        rows = np.zeros((cap, W), np.float64 if f64 else np.float32); fl = np.zeros(cap, np.uint32)
        n = (self.L_.orc_drain_tuples_f64 if f64 else self.L_.orc_drain_tuples)(self.h, _p(rows), _p(fl), cap)
        return rows[:n], fl[:n]

    def rbd(self, q, qd):
        D = self.D
        q = np.ascontiguousarray(q, np.float64); qd = np.ascontiguousarray(qd, np.float64)
        H = np.zeros((D, D)); Cq = np.zeros(D); Ct = np.zeros(D); g = np.zeros(D)
        self.L_.orc_rbd(self.h, _p(q), _p(qd), _p(H), _p(Cq), _p(Ct), _p(g)); return H, Cq, Ct, g

    def ctrl_eval(self, q, qd, contacts, state, phase):
        q = np.ascontiguousarray(q, np.float64); qd = np.ascontiguousarray(qd, np.float64)
        c = np.ascontiguousarray(contacts, np.int32); a = np.zeros(self.D); b = np.zeros(self.D)
        self.L_.orc_ctrl_eval(self.h, _p(q), _p(qd), _p(c), state, phase, _p(a), _p(b)); return a, b

    def sample_ground(self, x):
        v, s, i, j = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        h = self.L_.orc_sample_ground(self.h, float(x), C.byref(v), C.byref(s), C.byref(i), C.byref(j))
        return h, v.value, s.value, i.value, j.value

    def ground_segment(self, slot):
        buf = np.zeros(1024, np.float32); a, b, c = C.c_double(), C.c_double(), C.c_double()
        n = self.L_.orc_ground_segment(self.h, slot, _p(buf), 1024, C.byref(a), C.byref(b), C.byref(c))
        return buf[:n].copy(), a.value, b.value, c.value

    def nn_eval(self, x):
        x = np.ascontiguousarray(x, np.float64); y = np.zeros(self._keep[0].n_frags * (1 + self._keep[0].frag_size))
        self.L_.orc_nn_eval(self.h, _p(x), _p(y)); return y


def terrain_build(ttype, params40, seed, width):
    buf = np.zeros(4096, np.float32); p = np.ascontiguousarray(params40, np.float64)
    n = lib().orc_terrain_build(int(ttype), _p(p), int(seed), float(width), _p(buf), 4096)
    return buf[:n].copy()


def rng_draw(seed, env_id, n):
    out = np.zeros(n); lib().orc_rng_draw(seed, env_id, n, _p(out)); return out


def batch_run(model, n_envs, n_threads, n_frames, terrain_seed0=0, rng_seed=0, policy=None):
    r, c = C.c_int64(), C.c_int64()
    if policy is None:
        v = lib().orc_batch_run(C.byref(model), n_envs, n_threads, n_frames, terrain_seed0, rng_seed, None, None, None, None, None, None, C.byref(r), C.byref(c))
    else:
        desc, w, io, isc, oo, osc = policy
        w = np.ascontiguousarray(w, np.float32)
        io, isc, oo, osc = (np.ascontiguousarray(x, np.float64) for x in (io, isc, oo, osc))
        v = lib().orc_batch_run(C.byref(model), n_envs, n_threads, n_frames, terrain_seed0, rng_seed, C.byref(desc), _p(w), _p(io), _p(isc), _p(oo), _p(osc), C.byref(r), C.byref(c))
    return v, r.value, c.value


def batch_eval(model, n_envs, n_threads, n_frames, terrain_seed0=0, rng_seed=0, env_id0=0, policy=None):
    """Distribution-level statistics of n_envs oracle envs: dict(resets, cycles, episodes, dist_sum, dist_sq_sum, env_steps, seconds)."""
    out = np.zeros(6)
    desc, w, io, isc, oo, osc = policy
    w = np.ascontiguousarray(w, np.float32)
    io, isc, oo, osc = (np.ascontiguousarray(x, np.float64) for x in (io, isc, oo, osc))
    sec = lib().orc_batch_eval(C.byref(model), n_envs, n_threads, n_frames, terrain_seed0, rng_seed, env_id0, C.byref(desc), _p(w), _p(io), _p(isc), _p(oo), _p(osc), _p(out))
    return dict(resets=out[0], cycles=out[1], episodes=out[2], dist_sum=out[3], dist_sq_sum=out[4], env_steps=out[5], seconds=sec)


DIAG_KEYS = ("link_cap_substeps", "row_cap_substeps", "rows_ge_16_substeps", "pair_row_substeps", "substeps", "max_rows", "sum_rows", "resets")


def batch_trace(model, n_envs, n_threads, n_frames, terrain_seed0=0, rng_seed=0, env_id0=0, policy=None, contact_cache=False, nudge=0.0):
    """n_envs free-running oracle envs on n_threads host threads: dict(q, qd [frames, envs, D], diag [envs, 8] per DIAG_KEYS, resets [frames, envs], seconds) and, with
    contact_cache, ws_n [frames, envs], ws_id / ws_lam [frames, envs, 24]: the persistent contact rows after every frame. nudge: every env starts with its first joint
    angle moved by that much (sensitivity probe)."""
    D = int(model.D)
    q = np.zeros((n_frames, n_envs, D)); qd = np.zeros((n_frames, n_envs, D)); diag = np.zeros((n_envs, 8), np.int64); resets = np.zeros((n_frames, n_envs), np.int32)
    ws_n = np.zeros((n_frames, n_envs), np.int32) if contact_cache else None
    ws_id = np.zeros((n_frames, n_envs, 24), np.int32) if contact_cache else None
    ws_lam = np.zeros((n_frames, n_envs, 24)) if contact_cache else None
    pa = [None] * 6
    if policy is not None:
        desc, w, io, isc, oo, osc = policy
        w = np.ascontiguousarray(w, np.float32)
        io, isc, oo, osc = (np.ascontiguousarray(x, np.float64) for x in (io, isc, oo, osc))
        pa = [C.byref(desc), _p(w), _p(io), _p(isc), _p(oo), _p(osc)]
    sec = lib().orc_batch_trace(C.byref(model), n_envs, n_threads, n_frames, terrain_seed0, rng_seed, env_id0, *pa, _p(q), _p(qd), _p(diag),
                                _p(ws_n) if contact_cache else None, _p(ws_id) if contact_cache else None, _p(ws_lam) if contact_cache else None, _p(resets), float(nudge))
    return dict(q=q, qd=qd, diag=diag, resets=resets, ws_n=ws_n, ws_id=ws_id, ws_lam=ws_lam, seconds=sec)

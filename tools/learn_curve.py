#!/usr/bin/env python3
"""Does the training loop LEARN through this engine? (VERDICT r4 #2/#3.)  GPU:  python tools/learn_curve.py --char dog --iters 60000 --out profiles/r05_learning_curve_dog.txt

The reference's MACE training (scenarios/ScenarioTrain.cpp:340-410 with args/opt_args_train[_raptor]_mace.txt: 50 000 initial samples, replay memory 500 000, one
Train() per 32 new tuples, exploration annealed over 50 000 iterations, target net frozen for 500) runs UNCHANGED except for two things that are stated here:
  * -terrain_file= is the terrain of the BASELINE scene the policy is evaluated on (dog: slopes_mixed instead of the arg file's mixed; raptor: narrow_gaps instead of
    mixed_raptor) -- the shipped dog_mace3_slopes_mixed / raptor_mace3_narrow_gaps models carry those terrains in their names;
  * the run stops at --iters trainer iterations (-trainer_max_iter= 1e9 in the arg file).
4096 (8192) lock-stepped envs feed the one trainer (the reference: 4 threads), the native HIP trainer step trains while the next frame rolls out (overlap).
Every --eval-every iterations (trainer_int_iter = 2000: where cScenarioTrain writes its intermediate model) the CURRENT weights + normalisers are evaluated GREEDILY
(poli_eval: no exploration) on a separate batch of --eval-envs envs with FIXED terrain seeds for --eval-frames outer frames (optimizer/scenarios/OptScenarioPoliEval.cpp:170-211):
  speed        metres per second of simulated time, all envs (finished episodes + the running ones)        falls_k   falls per 1000 env-steps
  avg_dist     cScenarioPoliEval's mean distance per finished episode (nan: nobody fell)                   alive     fraction of envs that never fell during the evaluation
The first line (iteration 0) is the xavier-initialised net: the baseline every earlier number of this repository was measured on."""
import argparse, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
from deepterrainrl_amd import train_loop

CHARS = {"dog": dict(train="args/opt_args_train_mace.txt", evalf="args/dog_slopes_mixed_args.txt", terrain="data/terrain/slopes_mixed.txt", envs=4096),
         "raptor": dict(train="args/opt_args_train_raptor_mace.txt", evalf="args/raptor_narrow_gaps_args.txt", terrain="data/terrain/narrow_gaps.txt", envs=8192),
         # BASELINE configs[4]'s loop as the reference ships it (its own terrain: cliffs_rugged; one substep of 1/600 s per env-step, world scale 1)
         "goat": dict(train="args/opt_args_train_goat_mace.txt", evalf="args/goat_cliffs_args.txt", terrain="data/terrain/cliffs_rugged.txt", envs=8192)}


SCENARIO = da.BatchScenario


def evaluate(arg_file, root, weights, norm, n, frames, seed=777001):
    b = SCENARIO(arg_file, n, data_root=root, extra_args={"terrain_seed": seed})
    b.SetPolicy(weights, *norm)
    x0 = b.PoseVel()[0][:, 0].copy()
    for _ in range(frames):
        b.Update(1.0 / 30.0)
    st = b.EvalStats()
    d, ids = b.GetDistLog()
    x1 = b.PoseVel()[0][:, 0]
    total = float(d.sum()) + float((x1 - x0).sum())        # every reset puts the character back on x0 (cScenarioSimChar::InitCharacterPos)
    fell = np.zeros(n, bool); fell[np.asarray(ids, np.int64)] = True
    T = frames / 30.0
    return dict(speed=total / (n * T), falls_k=1000.0 * st["resets"] / (n * frames * 20.0), avg_dist=float(d.mean()) if len(d) else float("nan"), alive=float(1.0 - fell.mean()),
                episodes=int(st["episodes"]), cycles=int(st["cycles"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--char", choices=sorted(CHARS), default="dog")
    ap.add_argument("--iters", type=int, default=60000)
    ap.add_argument("--envs", type=int, default=None)
    ap.add_argument("--eval-every", type=int, default=2000)
    ap.add_argument("--eval-envs", type=int, default=512)
    ap.add_argument("--eval-frames", type=int, default=300)
    ap.add_argument("--data-root", default=os.path.join(REPO, "tests", "golden", "refdata"))
    ap.add_argument("--out", default="")
    ap.add_argument("--save", default="", help="directory for <char>_mace3_<terrain>_model.h5 + _scale.txt (tests/golden/policies)")
    ap.add_argument("--trainer", choices=["hip", "torch"], default="hip")
    ap.add_argument("--init-samples", type=int, default=None, help="(smoke runs only) override -trainer_num_init_samples=")
    ap.add_argument("--lib", default="", help="(CPU smoke runs only) bind the scenario to this build of the engine, e.g. tests/emul/libdtrl_emul.so")
    ap.add_argument("--sequential", action="store_true", help="no overlap: train on frame f's tuples before frame f+1 is launched")
    a = ap.parse_args()
    c = CHARS[a.char]; envs = a.envs or c["envs"]
    global SCENARIO
    if a.lib:
        class LibScenario(da.BatchScenario):
            def _library(self):
                return da._bind(os.path.abspath(a.lib))
        SCENARIO = LibScenario
    lines = ["# tools/learn_curve.py --char %s --iters %d: %s with -terrain_file= %s, %d envs, native trainer, %s; greedy evaluation of the current net every %d iterations on %d envs x %d frames of %s (fixed terrain seeds)"
             % (a.char, a.iters, c["train"], c["terrain"], envs, "sequential" if a.sequential else "overlapped", a.eval_every, a.eval_envs, a.eval_frames, c["evalf"]),
             "# %8s %9s %9s %8s %8s %9s %7s %9s %9s %8s" % ("iter", "tuples", "wall_s", "speed", "falls_k", "avg_dist", "alive", "episodes", "critic", "exp_rate")]
    t0 = time.time()
    curve = []

    def eval_fn(it, t, b):
        norm = t.GetOffsetScale()
        r = evaluate(c["evalf"], a.data_root, t.GetWeights(), norm, a.eval_envs, a.eval_frames)
        r.update(iter=it, tuples=t.GetNumTuples(), wall=time.time() - t0, loss=float(t.last_loss) if t.last_loss is not None else float("nan"))
        curve.append(r)
        line = "  %8d %9d %9.1f %8.3f %8.3f %9.3f %7.3f %9d %9.4g" % (it, r["tuples"], r["wall"], r["speed"], r["falls_k"], r["avg_dist"], r["alive"], r["episodes"], r["loss"])
        lines.append(line); print(line, flush=True)

    stem = None
    if a.save:
        os.makedirs(a.save, exist_ok=True)
        stem = os.path.join(a.save, "%s_mace3_%s_model" % (a.char, os.path.splitext(os.path.basename(c["terrain"]))[0].replace("cliffs_rugged", "cliffs")))
    st = train_loop.train(c["train"], a.data_root, envs, max_iters=a.iters, overlap=not a.sequential, trainer=a.trainer, scenario_cls=SCENARIO,
                          extra_args=dict({"terrain_file": c["terrain"]}, **({"trainer_num_init_samples": a.init_samples} if a.init_samples is not None else {})),
                          eval_every=a.eval_every, eval_fn=eval_fn, out_model_file=(stem + ".h5") if stem else None, out_scale_file=(stem + "_scale.txt") if stem else None)
    # the final net once more, through the files just written when there are any (policy-file row f2: caffe_hdf5 writer -> reader -> dtrl_set_policy)
    if stem:
        b = SCENARIO(c["evalf"], 8, data_root=a.data_root)
        w = b.LoadModel(stem + ".h5")
        assert np.array_equal(w, st["weights"].astype(np.float32)), "the model file does not round-trip"
    base, last = curve[0], curve[-1]
    best = max(curve, key=lambda r: r["speed"])
    lines.append("# frames %d, trainer iterations %d, tuples %d, %.1f s wall (evaluations included): %.2f M env-steps/s while training, %.0f Train()/s" % (
        st["frames"], st["iters"], st["tuples"], st["seconds"], st["env_steps_per_s"] / 1e6, st["trainer_iters_per_s"]))
    lines.append("# xavier baseline: speed %.3f m/s, %.3f falls / 1000 env-steps, %.0f %% never fell; last: speed %.3f, falls %.3f, %.0f %%; best speed %.3f at iteration %d" % (
        base["speed"], base["falls_k"], 100 * base["alive"], last["speed"], last["falls_k"], 100 * last["alive"], best["speed"], best["iter"]))
    print("\n".join(lines[-2:]))
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

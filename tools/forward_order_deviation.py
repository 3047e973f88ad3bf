#!/usr/bin/env python3
"""What the summation order of `terr_ip0` (the 5984-wide inner product of the dog_mace3 net) costs, and what it does to the Boltzmann actor choice at
T = 0.025. CPU only, numpy + the torch peer net; nothing of the engine is involved.

The reference evaluates the net with Caffe instantiated for DOUBLE (learning/NeuralNet.h:13 `typedef double tNNData`, `caffe::Net<tNNData>`), i.e. a cblas_dgemv
whose accumulation order is the BLAS library's business. The product's frame kernel sums tile-major in fp64, the oracle channel-major in fp64.
Three forwards of the same float32-valued weights on the same normalised states:
   A  fp64, strict channel-major sequence (k = channel * 187 + position ascending: the oracle's loop)
   B  fp64, tile-major sequence (tiles of 10 conv2 positions; inside a tile channel, then position: the frame kernel's order)
   C  every layer in float32 (what a Caffe built for `float` would compute: NOT the reference's configuration, shown for scale)
Outputs are un-normalised with the SHIPPED scale file (dog_mace3_slopes_mixed_model_scale.txt: value heads y / 2 + 0.5) and the actor probabilities are
cBaseControllerMACE::BoltzmannSelectActor's p_i ~ exp((q_i - q_max) / T) (sim/BaseControllerMACE.cpp:350-371), T = -exp_temp= 0.025 (args/opt_args_train_mace.txt:19).
Weights: Caffe's xavier fillers (no trained .h5 ships with the reference: Q spread over the actors ~0.4, one actor dominates), plus the same net with the value head's
last layer scaled by 0.1 so that the actors are nearly tied (spread ~0.04 = 1.6 T), which is where a Q difference moves the probabilities most."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from deepterrainrl_amd import trainer as tr
NETS = os.path.join(REPO, "tests", "golden", "refdata", "data/policies/dog/nets")
SCALE = os.path.join(REPO, "tests", "golden", "refdata", "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt")
S, A, NF, T = 283, 30, 3, 0.025
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000

t = tr.MACETrainer(os.path.join(NETS, "dog_mace3_train.prototxt"), os.path.join(NETS, "dog_mace3_solver.prototxt"), S, A, mem_size=64, device="cpu", dtype=torch.float64, seed=7, use_graphs=False)
net = t.net
sc = json.load(open(SCALE))
o_off, o_scale = np.array(sc["OutputOffset"][:NF]), np.array(sc["OutputScale"][:NF])
rng = np.random.RandomState(3)
X = torch.as_tensor(rng.normal(0, 1, (B, S)))


def tail(net, tpre, x):
    relu = torch.nn.functional.relu
    h = relu(net.mods[4](torch.cat([relu(tpre), x[:, net.n_terrain:]], 1)))
    return net.mods[6](relu(net.mods[5](h)))    # the value head: one Q per actor


def study(label, gain):
    with torch.no_grad():
        w32 = net.get_flat()                       # float32-valued weights, as the engine receives them
        net.set_flat(w32.astype(np.float64))
        net.mods[6].weight.mul_(gain)
        relu = torch.nn.functional.relu
        c = X[:, :net.n_terrain].unsqueeze(1)
        for m in net.mods[:3]:
            c = relu(m(c))
        C, P = c.shape[1], c.shape[2]               # 32 x 187
        flat = c.flatten(1).numpy()                 # k = channel * P + position
        W = net.mods[3].weight.numpy(); bias = net.mods[3].bias.numpy()
        order_a = np.arange(C * P)
        order_b = np.array([ch * P + p for t0 in range(0, P, 10) for ch in range(C) for p in range(t0, min(t0 + 10, P))])
        pre = {}
        for key, order in (("A", order_a), ("B", order_b)):
            acc = np.zeros((B, W.shape[0]))
            for k in order:                          # a strict left-to-right sum (numpy's own dot would pairwise / block it)
                acc += flat[:, k:k + 1] * W[:, k][None, :]
            pre[key] = acc + bias
        qa = tail(net, torch.as_tensor(pre["A"]), X).numpy()
        qb = tail(net, torch.as_tensor(pre["B"]), X).numpy()
        n32 = tr.MACETrainer(os.path.join(NETS, "dog_mace3_train.prototxt"), os.path.join(NETS, "dog_mace3_solver.prototxt"), S, A, mem_size=64, device="cpu", dtype=torch.float32, seed=7, use_graphs=False).net
        n32.set_flat(net.get_flat())
        qc = n32(X.float())[:, :NF].double().numpy()
        net.mods[6].weight.div_(gain)
    out = {}
    for key, q in (("A", qa), ("B", qb), ("C", qc)):
        qq = q / o_scale - o_off                    # un-normalised Q
        e = np.exp((qq - qq.max(1, keepdims=True)) / T)
        out[key] = (qq, e / e.sum(1, keepdims=True))
    qa_, pa = out["A"]
    print("## %s: un-normalised Q spread over actors (max - min): median %.3g, max %.3g; p_max median %.3f" % (label, np.median(qa_.max(1) - qa_.min(1)), (qa_.max(1) - qa_.min(1)).max(), np.median(pa.max(1))))
    for key, name in (("B", "fp64 tile-major (frame kernel)  vs fp64 channel-major"), ("C", "all-float32 forward             vs fp64 channel-major")):
        q, p = out[key]
        tv = 0.5 * np.abs(p - pa).sum(1)            # total variation = probability that the two selectors pick different actors from the same uniform draw
        print("   %s:  max |dQ| %.3g   max |dp| %.3g   P(different actor) mean %.3g max %.3g   argmax flips %d / %d" % (
            name, np.abs(q - qa_).max(), np.abs(p - pa).max(), tv.mean(), tv.max(), int((q.argmax(1) != qa_.argmax(1)).sum()), B))


print("# terr_ip0 summation order and the Boltzmann actor choice, %d normalised states, T = %g" % (B, T))
study("xavier-initialised net", 1.0)
study("value head scaled x0.1 (actors nearly tied: the selector at its most sensitive, p_max ~ 0.5)", 0.1)

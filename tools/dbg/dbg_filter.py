import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_trainer as TT, test_hip_trainer as TH
rng = np.random.RandomState(9)
rows, flags = TT.random_rows(rng, 200, p_actor=0.5)
t = TH.make_native(None, "cuda"); p = TH.make_peer("cuda")
w0 = p.GetWeights().copy(); t.SetWeights(w0)
t.AddTuples(rows, flags); p.AddTuples(rows, flags)
ids = list(range(40, 72))
n = len(ids)
t.nt.idx[32:32 + n] = ids
t.nt.actor_filter(n); t.nt.sync()
better = t.nt.better[:n].copy()
idx = p._idx(ids); r = p.mem[idx]
curr = p._eval(p._target_net(), r[:, 1:1 + 283])[:, :3].max(1).values
new = p._new_q(r, idx)
print("native ", better.tolist())
print("peer   ", (new > curr).int().tolist())
print("flags host", flags[ids].tolist())
print("flags dev ", t.flags_dev[torch.as_tensor(ids, device="cuda")].tolist())
# native eval of the same rows
Y = t._eval(t.net, r[:, 1:1 + 283].contiguous()); t.nt.sync()
print("eval diff s ", (Y - p._eval(p.net, r[:, 1:1+283])).abs().max().item())
print("curr", curr[:6].tolist(), "new", new[:6].tolist())

xin = t.nt.debug_get(10, 64 * 283).reshape(64, 283)
out = t.nt.debug_get(11, 64 * 90).reshape(64, 90)
nq = t.nt.debug_get(12, 32)
xs = ((r[:, 1:1+283] + p.in_off) * p.in_scale).cpu().numpy(); xe = ((r[:, 314:] + p.in_off) * p.in_scale).cpu().numpy()
print("xin s diff", np.abs(xin[:32] - xs).max(), "xin s' diff", np.abs(xin[32:] - xe).max())
po = p.net(torch.as_tensor(np.concatenate([xs, xe]), device="cuda")).detach().cpu().numpy()
print("out diff", np.abs(out - po).max(), "newq native", nq[:6], "peer", new[:6].tolist())
print("idx window", t.nt.idx[32:40], "better raw", t.nt.better[:8])

"""cScenarioPoliEval's per-cycle recorders for a batch (reference: scenarios/ScenarioPoliEval.cpp:234-404).

The reference appends one line per VALID gait cycle (mCycleCount >= 1) of its single env to up to three text files:
  RecordAction          "<action id>,\\t<opt param>,\\t..."              (header by InitActionRecord: one "%i, %.5f, ..." line per action)
  RecordVel             "<COM x velocity over the span since the last record>"
  RecordActionIDState   "<action id>,\\t<policy state>,\\t..."
  RecordNNActivation    "<action id>,\\t<blob value>,\\t..."           (cNeuralNet::GetLayerState(layer): the named blob of the policy net after the forward
                        pass that chose the cycle's action, learning/NeuralNet.cpp:814-833)
Everything they read is constant over a cycle (the action, its policy state, the COM / time at the cycle's start), so the batch engine
does not need a per-env-step hook: `PoliEvalRecorder.Poll()` after every `Update()` / `RunFrames()` looks at the cycle counters, and writes
the lines of the envs that started a new cycle, in the reference's formats (`std::to_string` = "%f"). One file set per recorded env.
A gait cycle lasts ~13 outer frames, so polling once per frame (or every few frames) never misses one; Poll() raises if it ever did.

Two details of the reference are kept: its clock mTime advances by the whole outer frame BEFORE the frame's env-steps run
(scenarios/ScenarioSimChar.cpp:153-154), so the time span in RecordVel is a whole number of frames; and mCycleCount is cleared by Init / Clear
only, not by Reset. One difference: a cycle that starts in the very frame whose end detects a fall has lost its state by the time the frame's
host work (which applies the reset) returns; the reference would still write its line, here it is counted in `lost` instead.
"""
import numpy as np


def _to_string(x):
    return "%f" % x          # std::to_string(double)


class PoliEvalRecorder:
    def __init__(self, batch, env_ids, action_file=None, vel_file=None, action_id_state_file=None, frame_dt=1.0 / 30.0,
                 nn_activation_file=None, nn_activation_layer=None, policy_net=None):
        """File names are templates with an `{env}` field (e.g. "out/actions_{env}.txt"); None disables that recorder.
        nn_activation_file + nn_activation_layer (-record_nn_activation / -nn_activation_output_file / -nn_activation_layer): the frame kernel keeps no
        per-layer blobs (activations are transient), so the recorder re-runs the recorded policy state of the cycle through the same network on the
        batch's GPU (trainer.MaceNet in fp64, the weights and input normaliser last passed to SetPolicy) and writes the named blob. policy_net: path of the
        deploy prototxt (the arg file's -policy_net=)."""
        self.b = batch
        self.nn_layer = nn_activation_layer if nn_activation_file else None
        self._net = None
        if self.nn_layer:
            import torch
            from . import trainer
            if policy_net is None or getattr(batch, "_policy", None) is None:
                raise ValueError("the NN-activation recorder needs policy_net= and a policy installed with SetPolicy")
            w, io, isc = batch._policy
            dev = "cuda" if torch.cuda.is_available() else "cpu"
            self._net = trainer.MaceNet(trainer.parse_net(policy_net)).to(dev, torch.float64)
            self._net.set_flat(w)
            self._io = torch.as_tensor(io if io is not None else np.zeros(batch.S), device=dev)
            self._isc = torch.as_tensor(isc if isc is not None else np.ones(batch.S), device=dev)
        self.frames_per_s = 1.0 / frame_dt
        self.lost = 0
        self.ids = np.ascontiguousarray(env_ids, np.int32)
        self.files = {"action": action_file, "vel": vel_file, "ids": action_id_state_file, "nn": nn_activation_file if self.nn_layer else None}
        nc, nr, com, t, _ = batch.CycleInfo(self.ids)
        self.cycles = nc.copy(); self.resets = nr.copy()
        # cScenarioPoliEval::Init / Reset: mPrevCOMPos = CalcCOM(), mPrevTime = mTime. Only the x component is ever written, and every reset
        # restores the same pose at the same root x, so the COM x of the freshly initialised env (construct the recorder before stepping, or
        # right after Reset()) is also the COM x after every later reset
        self.init_com_x = com[:, 0].copy()
        self.prev_com_x = com[:, 0].copy(); self.prev_time = t.copy()
        for e in self.ids:
            if action_file:   # InitActionRecord
                with open(action_file.format(env=int(e)), "w") as f:
                    for a, row in enumerate(batch.ActionTable()):
                        f.write("%i" % a + "".join(", %.5f" % v for v in row) + "\n")
            for k in ("vel", "ids", "nn"):   # cFileUtil::ClearFile
                if self.files[k]:
                    open(self.files[k].format(env=int(e)), "w").close()
        self.lines = 0

    def _append(self, kind, env, text):
        with open(self.files[kind].format(env=int(env)), "a") as f:
            f.write(text)

    def Poll(self):
        """Call after stepping. Returns the number of cycles recorded by this call."""
        nc, nr, com, t, prm = self.b.CycleInfo(self.ids)
        _, _, aid, _, _ = self.b.Ctrl(self.ids)
        ps = self.b.RecordPoliState(self.ids) if (self.files["ids"] or self.files["nn"]) else None
        n = 0
        for i, e in enumerate(self.ids):
            if nr[i] != self.resets[i]:         # cScenarioPoliEval::Reset since the last poll (mCycleCount is NOT reset, as in the reference)
                self.resets[i] = nr[i]; self.prev_com_x[i] = self.init_com_x[i]; self.prev_time[i] = 0.0
                if t[i] == 0.0 and nc[i] != self.cycles[i]:   # the reset has been applied and no env-step ran since: the new cycle predates it
                    self.lost += int(nc[i] - self.cycles[i]); self.cycles[i] = nc[i]
            if nc[i] == self.cycles[i]:
                continue
            if nc[i] > self.cycles[i] + 1:
                raise RuntimeError("env %d completed %d cycles between two polls; poll at least once per cycle" % (e, nc[i] - self.cycles[i]))
            started = self.cycles[i]            # value of mCycleCount when NewCycleUpdate ran
            self.cycles[i] = nc[i]
            if started < 1:                     # IsValidCycle(): mCycleCount >= gNumWarmupCycles (= 1)
                continue
            if self.files["nn"]:   # RecordNNActivation comes first in cScenarioPoliEval::NewCycleUpdate
                import torch
                with torch.no_grad():
                    x = (torch.as_tensor(ps[i:i + 1], device=self._io.device) + self._io) * self._isc
                    blob = self._net.named_blobs(x)[self.nn_layer][0].cpu().numpy()
                self._append("nn", e, str(int(aid[i])) + "".join(",\t" + _to_string(v) for v in blob) + "\n")
            if self.files["action"]:
                self._append("action", e, str(int(aid[i])) + "".join(",\t" + _to_string(v) for v in prm[i]) + "\n")
            if self.files["vel"]:
                m_time = np.ceil(t[i] * self.frames_per_s - 1e-6) / self.frames_per_s     # mTime when NewCycleUpdate ran: the end of that outer frame
                dt = m_time - self.prev_time[i]
                self._append("vel", e, _to_string((com[i, 0] - self.prev_com_x[i]) / dt) + "\n")
                self.prev_com_x[i] = com[i, 0]; self.prev_time[i] = m_time
            if self.files["ids"]:
                self._append("ids", e, str(int(aid[i])) + "".join(",\t" + _to_string(v) for v in ps[i]) + "\n")
            n += 1
        self.lines += n
        return n

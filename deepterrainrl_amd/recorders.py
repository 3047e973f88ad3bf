"""cScenarioPoliEval's per-cycle recorders for a batch (reference: scenarios/ScenarioPoliEval.cpp:234-404).

The reference appends one line per VALID gait cycle (mCycleCount >= 1) of its single env to up to three text files:
  RecordAction          "<action id>,\\t<opt param>,\\t..."              (header by InitActionRecord: one "%i, %.5f, ..." line per action)
  RecordVel             "<COM x velocity over the span since the last record>"
  RecordActionIDState   "<action id>,\\t<policy state>,\\t..."
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
    def __init__(self, batch, env_ids, action_file=None, vel_file=None, action_id_state_file=None, frame_dt=1.0 / 30.0):
        """File names are templates with an `{env}` field (e.g. "out/actions_{env}.txt"); None disables that recorder."""
        self.b = batch
        self.frames_per_s = 1.0 / frame_dt
        self.lost = 0
        self.ids = np.ascontiguousarray(env_ids, np.int32)
        self.files = {"action": action_file, "vel": vel_file, "ids": action_id_state_file}
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
            for k in ("vel", "ids"):   # cFileUtil::ClearFile
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
        ps = self.b.RecordPoliState(self.ids) if self.files["ids"] else None
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

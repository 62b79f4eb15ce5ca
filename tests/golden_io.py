"""Load tests/golden/*.npz (written by oracle/harness/make_golden.py from runs of
the reference itself) into the structures the oracle / HIP library consume."""
from __future__ import annotations

import glob
import os

import numpy as np

from cygym_amd import abi
from cygym_amd import spec as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STATE_KEYS = ["flags", "busy", "wl", "comp_by", "st_flags", "st_busy", "st_wl", "st_comp_by",
              "blocked", "ring", "ienv", "fenv"]


def fixture_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


class Fixture:
    def __init__(self, name, path=None):
        z = np.load(path if path is not None else os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.z = z
        st = {k[len("static_"):]: z[k] for k in z.files if k.startswith("static_")}
        self.topo = abi.TopologyArrays(
            M=int(st["M"]), X=int(st["X"]), dstatic=st["dstatic"], vuln=st["vuln"], napps=st["napps"],
            os_val=st["os_val"], version=st["version"], anomaly=st["anomaly"], out_ptr=st["out_ptr"],
            out_col=st["out_col"], in_ptr=st["in_ptr"], in_col=st["in_col"], in_eid=st["in_eid"],
            max_extra=int(st.get("max_extra", 0)), det_apl=st["det_apl"]).normalised()
        self.K = self.topo.max_extra   # > 0: the fixture follows the edges evolve_network adds (extra-edge list)
        cfg = dict(zip([str(k) for k in z["config_keys"]], z["config_vals"]))
        inv = {v: k for k, v in abi.BASELINES.items()}
        self.cfg = abi.EnvConfig(
            seed=int(z["seed"]), env_id_base=int(z["env_id_base"]),
            num_of_device=int(cfg["num_of_device"]), min_network_size=int(cfg["min_network_size"]),
            max_exploits=int(cfg["max_exploits"]), evolve_period=int(cfg["evolve_period"]),
            workload_cap=int(cfg["workload_cap"]), workload_period_base=int(cfg["workload_period_base"]),
            workload_period_max=int(cfg["workload_period_max"]),
            scaling_vulnerability=int(cfg["scaling_vulnerability"]), fast_scan=int(cfg["fast_scan"]),
            n_att_actions=int(cfg["n_att_actions"]), n_def_actions=int(cfg["n_def_actions"]),
            zero_day=int(cfg["zero_day"]), zero_day_owned_mask=int(cfg["zero_day_owned_mask"]),
            default_high=int(cfg["default_high"]), baseline=inv[int(cfg["baseline"])],
            work_scale=cfg["work_scale"], comp_scale=cfg["comp_scale"], def_scale=cfg["def_scale"],
            gamma=cfg["gamma"], lambda_events=cfg["lambda_events"], p_add=cfg["p_add"],
            p_attacker=cfg["p_attacker"], turbo=int(cfg["turbo"]),
            turbo_fraction_clients=cfg["turbo_fraction_clients"], turbo_fraction_servers=cfg["turbo_fraction_servers"],
            turbo_max_clients=int(cfg["turbo_max_clients"]), turbo_max_servers=int(cfg["turbo_max_servers"]),
            turbo_ramp_steps=int(cfg["turbo_ramp_steps"]), turbo_train_max_logs=int(cfg["turbo_train_max_logs"]),
            turbo_train_stride=int(cfg["turbo_train_stride"]))
        self.N = z["init_flags"].shape[0]
        self.T = z["exp_flags"].shape[1]
        self.M = self.topo.M
        self.G = z["act_atype"].shape[2]
        self.init = {k: z["init_" + k] for k in STATE_KEYS}
        self.pre = {k: z["pre_" + k] for k in STATE_KEYS} if "pre_flags" in z.files else None
        self.exp = {k: z["exp_" + k] for k in STATE_KEYS}
        if self.K > 0:
            self.init["extra"] = z["init_extra"]
            self.exp["extra"] = z["exp_extra"]
            if self.pre is not None:
                self.pre["extra"] = z["pre_extra"]
        # trained-detector mode: history ring the fits read, forests the reference fitted (one event per
        # Detector.train(non-empty logs)), the rows each was fitted on
        self.init["hist"] = z["init_hist"]
        self.init["forest"] = z["init_forest"]
        if self.pre is not None:
            self.pre["hist"] = z["pre_hist"]
        self.fin_hist = z["fin_hist"]
        self.det_events = {}
        ptr = z["det_rows_ptr"]
        for i, (e, t) in enumerate(zip(z["det_env"], z["det_tick"])):
            self.det_events.setdefault(int(t), []).append(
                dict(env=int(e), rng_tick=int(z["det_rng_tick"][i]), forest=z["det_forest"][i], rows=z["det_rows"][ptr[i]:ptr[i + 1]],
                     n_fits=int(z["det_n_fits"][i])))
        self.sklearn_version = str(z["det_sklearn_version"])
        for k in ("raw", "shaped", "done", "obs", "obs_def", "obs_att", "topo_same"):
            self.exp[k] = z["exp_" + k]
        # per-env flat device lists -> padded [N][T][L]
        ptr = z["act_dev_ptr"]
        cnt = z["act_dev_cnt"]            # [N][T][G]
        self.L = max(1, int(cnt.sum(axis=2).max()))
        if "act_is_none" in z.files and z["act_is_none"].any():
            self.L = max(self.L, self.M)   # default actions may list every device (:852-870)
        self.dev = np.zeros((self.N, self.T, self.L), np.int16)
        flat = z["act_dev_flat"]
        for e in range(self.N):
            row = flat[ptr[e]:ptr[e + 1]]
            per_t = cnt[e].sum(axis=1)
            off = np.concatenate([[0], np.cumsum(per_t)])
            for t in range(self.T):
                self.dev[e, t, :per_t[t]] = row[off[t]:off[t + 1]]

    def is_none(self, e, t) -> bool:
        return "act_is_none" in self.z.files and bool(self.z["act_is_none"][e, t])

    def python_action(self, e, t):
        """The reference-style action of env e at tick t: None, a 4-tuple, or a list of 4-tuples."""
        z = self.z
        if self.is_none(e, t):
            return None
        ng = int(z["act_n_groups"][e, t])
        groups = []
        off = 0
        for g in range(max(1, ng)):
            n = int(z["act_dev_cnt"][e, t, g])
            ne = int(z["act_n_exploit"][e, t, g])
            app = int(z["act_app"][e, t, g])
            groups.append((int(z["act_atype"][e, t, g]), np.array(z["act_exploit"][e, t, g, :ne], dtype=int),
                           [int(x) for x in self.dev[e, t, off:off + n]], app if app >= 0 else None))
            off += n
        return groups if ng > 0 else groups[0]

    def mode_name(self, e, t) -> str:
        return "defender" if (int(self.z["act_mode"][e, t]) & 0xFF) == S.MODE_DEFENDER else "attacker"

    def is_partial(self, e, t) -> bool:
        return bool(int(self.z["act_mode"][e, t]) & S.MODE_PARTIAL)

    def actions(self, t, alloc, flags=None):
        """Fill an action dict (oracle.driver.alloc_actions / torch mirror) for tick t.
        `flags` ([N][M], state before the tick) is needed where the fixture holds action=None:
        the default is substituted by the build's host logic (cygym_amd/host_logic.py)."""
        z = self.z
        alloc["mode"][:] = z["act_mode"][:, t]
        alloc["n_groups"][:] = z["act_n_groups"][:, t]
        alloc["atype"][:] = z["act_atype"][:, t]
        alloc["n_exploit"][:] = z["act_n_exploit"][:, t]
        alloc["exploit"][:] = z["act_exploit"][:, t]
        alloc["app"][:] = z["act_app"][:, t]
        alloc["dev_cnt"][:] = z["act_dev_cnt"][:, t]
        alloc["dev_idx"][:] = self.dev[:, t]
        for e in range(self.N):
            if self.is_none(e, t):
                from cygym_amd import host_logic as HL
                a = HL.default_action(self.mode_name(e, t), self.cfg.baseline, np.asarray(flags)[e])
                HL.encode_into(alloc, e, self.mode_name(e, t), [a], False, self.M)
        return alloc

    def expected_state(self, t):
        out = {}
        for k in STATE_KEYS:
            v = self.exp[k][:, t]
            if k == "blocked":
                v = abi.pack_blocked(v, self.topo.EW)
            if k == "ring":
                v = np.where(v < 0, 0xFFFF, v).astype(np.uint16)
            out[k] = v
        if self.K > 0:
            out["extra"] = self.exp["extra"][:, t]
        return out

    def follows_topology(self) -> bool:
        return self.K > 0

    def service_detectors(self, t, state, install):
        """Play the host's part of Detector.train after tick t: for every training the reference did at this tick,
        check what the tick recorded (request header, CG_E_DET_PENDING, the history ring the fit would read) against
        the reference's own training rows, then hand the reference-fitted forest to `install(env, words)`.
        `state`: numpy views of ienv / forest / hist after the tick."""
        from cygym_amd import detector as D
        for ev in self.det_events.get(t, []):
            e = ev["env"]
            fl = int(state["ienv"][e, S.I_FLAGS])
            assert fl & S.E_DET_PENDING and fl & S.E_DET_TRAIN, f"{self.name} t={t} env {e}: no pending training"
            hdr = np.asarray(state["forest"][e][:S.FOREST_HDR]).astype(np.int64)
            lt = int(state["ienv"][e, S.I_LOG_TOTAL])
            assert hdr[3] == ev["rng_tick"] and hdr[4] == lt and hdr[6] == ev["n_fits"], \
                f"{self.name} t={t} env {e}: request header {hdr[3:7]} vs ({ev['rng_tick']}, {lt}, {ev['n_fits']} fits)"
            rows = D.training_window(np.asarray(state["hist"][e]), lt, bool(self.cfg.turbo), self.cfg.turbo_train_max_logs, self.cfg.turbo_train_stride)
            np.testing.assert_array_equal(rows, ev["rows"], err_msg=f"{self.name} t={t} env {e}: training rows")
            install(e, ev["forest"])

    def check_final_hist(self, got_hist, log_total):
        for e in range(self.N):
            n = min(int(log_total[e]), S.HIST_RING)
            idx = np.arange(int(log_total[e]) - n, int(log_total[e])) % S.HIST_RING
            np.testing.assert_array_equal(np.asarray(got_hist[e]).reshape(S.HIST_RING, 2).astype(np.int64)[idx],
                                          self.fin_hist[e].astype(np.int64)[idx], err_msg=f"{self.name}: hist env {e}")


def compare_state(got: dict, exp: dict, label: str, ring_total=None):
    """Bit-exact comparison of the integer planes; returns list of mismatch strings."""
    bad = []
    for k in ["flags", "busy", "wl", "comp_by", "st_flags", "st_busy", "st_wl", "st_comp_by", "blocked"]:
        g = np.asarray(got[k]).astype(np.int64)
        x = np.asarray(exp[k]).astype(np.int64)
        if not np.array_equal(g, x):
            idx = np.argwhere(g != x)[0]
            bad.append(f"{label}: {k} differs at {tuple(idx)}: got {g[tuple(idx)]} exp {x[tuple(idx)]}")
    gi = np.asarray(got["ienv"]).astype(np.int64)
    xi = np.asarray(exp["ienv"]).astype(np.int64)
    cols = [c for c in range(S.I_COUNT) if c not in (S.I_LAST_NCOMP,)]
    # TOPO_OVF / BUSY_SAT are build-side diagnostics the reference does not have
    mask = ~(S.E_TOPO_OVF | S.E_BUSY_SAT)
    gi2, xi2 = gi.copy(), xi.copy()
    gi2[:, S.I_FLAGS] &= mask
    xi2[:, S.I_FLAGS] &= mask
    for c in cols:
        if not np.array_equal(gi2[:, c], xi2[:, c]):
            e = int(np.argwhere(gi2[:, c] != xi2[:, c])[0][0])
            bad.append(f"{label}: ienv[{c}] env {e}: got {gi2[e, c]} exp {xi2[e, c]}")
    # ring: only the valid window (last min(total, R) entries) is defined
    gr = np.asarray(got["ring"]).astype(np.int64).reshape(gi.shape[0], S.LOG_RING, 2)
    xr = np.asarray(exp["ring"]).astype(np.int64).reshape(gi.shape[0], S.LOG_RING, 2)
    for e in range(gi.shape[0]):
        tot = int(xi[e, S.I_LOG_TOTAL])
        n = min(tot, S.LOG_RING)
        for j in range(tot - n, tot):
            if not np.array_equal(gr[e, j % S.LOG_RING], xr[e, j % S.LOG_RING]):
                bad.append(f"{label}: ring env {e} slot {j % S.LOG_RING}: got {gr[e, j % S.LOG_RING]} exp {xr[e, j % S.LOG_RING]}")
                break
    if "extra" in exp and np.asarray(exp["extra"]).size:   # live entries of the extra-edge list + their blocked bits
        gx = np.asarray(got["extra"]).astype(np.int64)
        xx = np.asarray(exp["extra"]).astype(np.int64)
        K = (xx.shape[1] * 32) // 33
        while K + (K + 31) // 32 < xx.shape[1]:
            K += 1
        for e in range(gi.shape[0]):
            n = int(xi[e, S.I_FLAGS]) >> S.E_NX_SHIFT
            if not np.array_equal(gx[e, :n], xx[e, :n]):
                bad.append(f"{label}: extra edges env {e}: got {[(k >> 16, k & 0xFFFF) for k in gx[e, :n]]} exp {[(k >> 16, k & 0xFFFF) for k in xx[e, :n]]}")
                continue
            gb = [(gx[e, K + (j >> 5)] >> (j & 31)) & 1 for j in range(n)]
            xb = [(xx[e, K + (j >> 5)] >> (j & 31)) & 1 for j in range(n)]
            if gb != xb:
                bad.append(f"{label}: extra-edge blocked bits env {e}: got {gb} exp {xb}")
    gf = np.asarray(got["fenv"], np.float64)
    xf = np.asarray(exp["fenv"], np.float64)
    # cumulative f64 cost accumulators: 1e-9 absolute + 1e-12 relative (the north star asks for 1e-6 on float rewards).  The
    # relative part is for the per-log scan path (fast_scan = False): the reference adds 0.5 * def_scale once per scanned log
    # entry -- thousands of equal terms per tick -- where the kernel adds their product; after a hundred ticks at 2048 devices
    # the two sums differ by ~1e-9 at 5e4 (tools/fuzz.py case 41011, same on every build since the path exists).
    if not np.allclose(gf, xf, rtol=1e-12, atol=1e-9):
        idx = np.argwhere(~np.isclose(gf, xf, rtol=1e-12, atol=1e-9))[0]
        bad.append(f"{label}: fenv differs at {tuple(idx)}: got {gf[tuple(idx)]} exp {xf[tuple(idx)]}")
    return bad


def assert_obs_equal(got, exp, slow_scan: bool, err_msg: str = ""):
    """Observation views against the reference's: exact -- except, on the per-log scan path (fast_scan = False), the
    anomaly column, whose values are scikit-learn's decision_function floats (0.5 - 2^(-s/c): one exp2 / pow apart
    between numpy, libm and the device): within the north star's 1e-6."""
    got, exp = np.asarray(got), np.asarray(exp)
    if not slow_scan:
        np.testing.assert_array_equal(got, exp, err_msg=err_msg)
        return
    g = got.reshape(got.shape[0], -1, 6).copy()
    x = exp.reshape(exp.shape[0], -1, 6).copy()
    np.testing.assert_allclose(g[:, :, 3], x[:, :, 3], rtol=0, atol=1e-6, err_msg=err_msg + " (anomaly column)")
    g[:, :, 3] = x[:, :, 3]
    np.testing.assert_array_equal(g, x, err_msg=err_msg)


def check_oracle_against_fixture(fx: "Fixture") -> int:
    """Step the CPU oracle through a fixture (outputs of the reference itself) and compare every tick:
    integer planes / counters / ring / extra edges bit-exact, rewards within 1e-9, the three observation
    views exact.  Returns the number of ticks compared.  Used by tests/test_oracle_golden.py and by
    oracle/harness/fuzz_reference.py."""
    from oracle import driver as od
    name = fx.name
    ob = od.OracleBatch(fx.topo, fx.cfg, fx.N, detector=True)
    ob.load_state(fx.init)
    act = od.alloc_actions(fx.N, fx.G, fx.L)
    alive = np.ones(fx.N, bool)   # without an extra-edge list parity is defined while the topology is unchanged
    checked = 0
    for t in range(fx.T):
        fx.actions(t, act, flags=ob.state["flags"])
        obs, raw, shaped, done = ob.step(act)
        fx.service_detectors(t, ob.state, ob.install_forest)
        assert not (ob.state["ienv"][:, S.I_FLAGS] & S.E_UNPINNED).any(), f"{name} t={t}: a scan ran without a current forest"
        same = fx.exp["topo_same"][:, t].astype(bool)
        # where the reference ADDED edges (evolve star / PA), the build must have flagged it
        ovf = (ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0
        if fx.follows_topology():   # the added edges are part of the compared state: parity never ends
            assert not ovf.any(), f"{name} t={t}: extra-edge list overflowed"
        else:
            assert np.array_equal(ovf[alive], ~same[alive]), f"{name} t={t}: TOPO_OVF {ovf} vs topo_same {same}"
            alive &= same
        if not alive.any():
            break
        exp = fx.expected_state(t)
        sel = np.where(alive)[0]
        got = {k: v[sel] for k, v in ob.state.items()}
        bad = compare_state(got, {k: v[sel] for k, v in exp.items()}, f"{name} t={t}")
        assert not bad, "\n".join(bad[:8])
        slow = not fx.cfg.fast_scan
        assert_obs_equal(obs[sel].reshape(len(sel), -1), fx.exp["obs"][sel, t].reshape(len(sel), -1), slow, f"{name} obs t={t}")
        np.testing.assert_allclose(raw[sel], fx.exp["raw"][sel, t], rtol=0, atol=1e-9, err_msg=f"{name} raw t={t}")
        np.testing.assert_allclose(shaped[sel], fx.exp["shaped"][sel, t], rtol=0, atol=1e-9, err_msg=f"{name} shaped t={t}")
        np.testing.assert_array_equal(done[sel], fx.exp["done"][sel, t], err_msg=f"{name} done t={t}")
        assert_obs_equal(ob.observe(1)[sel], fx.exp["obs_def"][sel, t], slow, f"{name} obs_def t={t}")
        np.testing.assert_array_equal(ob.observe(2)[sel], fx.exp["obs_att"][sel, t], err_msg=f"{name} obs_att t={t}")
        checked += 1
    if alive.all():
        fx.check_final_hist(ob.state["hist"], ob.state["ienv"][:, S.I_LOG_TOTAL])
    return checked

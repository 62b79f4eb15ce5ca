"""Oracle harness: run the UNMODIFIED reference environment in this container and
export flat golden fixtures for the batched tick.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build
container); nothing here travels to the GPU box except the .npz fixtures it wrote
under tests/golden/.  No reference source is copied: the reference modules are
imported from where they lie, with
  * stand-ins for the three absent third-party packages (oracle/harness/standins:
    gym, igraph, pymetis -- this repo's own code),
  * a synthetic CVE.csv (the real one is a Kaggle download, unavailable offline),
  * the reference's RNG call sites (`random.*`, `numpy.random.*` module attributes)
    fed from the build's Philox stream (cygym_amd/rng.py), addressed by
    (env, tick, site, a, b) -- see include/cygym_spec.h.  The call site is
    recovered from the caller's frame; reference files are untouched.

Canonicalisations applied by the injected functions (documented in DESIGN.md):
  * choice() over a *set* (evolve's pick_from, zero-day owned indices) picks from
    the sorted elements -- CPython set order is an implementation artefact;
  * sample(pop, k) returns the k elements with the smallest (philox key, id);
  * shuffle(list of devices) orders by (philox key, id).
"""
from __future__ import annotations

import copy
import math
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("CYGYM_REFERENCE", "/root/reference")

sys.dont_write_bytecode = True  # /root/reference must stay untouched (no __pycache__)

if REPO not in sys.path:
    sys.path.insert(0, REPO)

from cygym_amd import rng as R          # noqa: E402
from cygym_amd import spec as S         # noqa: E402
from cygym_amd import abi               # noqa: E402


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE, "volt_typhoon_env.py"))


# --------------------------------------------------------------------------
# environment bootstrap
# --------------------------------------------------------------------------
_SCRATCH = None
_MODS = None


def _write_cve_csv(path: str):
    """Synthetic CVE table with the columns the path reads (parse_json.py:37-53)."""
    import random as _r
    rr = _r.Random(1234)
    rows = ["matchCriteriaId,exploitabilityScore,impactScore,baseSeverity"]
    rows.append("ED3A999C-9184-4D27-A62E-3D8A3F0D4F27,3.9,5.9,CRITICAL")
    rows.append("0A5713AE-B7C5-4599-8E4F-9C235E73E5F6,6.0,5.9,HIGH")
    for i in range(200):
        rid = "%08X-%04X-%04X-%04X-%012X" % (rr.getrandbits(32), rr.getrandbits(16), rr.getrandbits(16),
                                            rr.getrandbits(16), rr.getrandbits(48))
        rows.append(f"{rid},{rr.choice([1.2, 1.8, 2.8, 3.9])},{rr.choice([1.4, 3.6, 5.9])},MEDIUM")
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


def load_reference():
    """Import the reference modules (once). Returns the module namespace dict."""
    global _SCRATCH, _MODS
    if _MODS is not None:
        return _MODS
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE}")
    _SCRATCH = tempfile.mkdtemp(prefix="cygym_oracle_")
    _write_cve_csv(os.path.join(_SCRATCH, "CVE.csv"))
    os.chdir(_SCRATCH)  # the reference opens its log + CVE.csv + snapshots in CWD
    sys.path.insert(0, os.path.join(HERE, "standins"))
    sys.path.insert(0, REFERENCE)
    import logging
    import random
    import volt_typhoon_env as vte
    import CyberDefenseEnv as cde
    import CDSimulator as cds
    import CDSimulatorComponents as cdc
    logging.disable(logging.CRITICAL)
    _hash_by_creation_order(cdc.App, cdc.Vulnerability)
    _MODS = dict(vte=vte, cde=cde, cds=cds, cdc=cdc, random=random, np=np)
    return _MODS


def _hash_by_creation_order(*classes):
    """The reference's network generator keeps `App` and `Vulnerability` objects -- which define no __hash__ -- in
    Python sets (CDSimulator.py:26, :30) and picks from `list(that_set)` (`changeVulTarget`, `_attach_extra`,
    `randomSampleGenerator`): with the default address-based hash the generated network depends on where the
    allocator happened to put those objects, i.e. on everything the process did before (round 1's fixtures changed
    with the set of scenarios generated earlier in the same interpreter; a refactoring of this harness moved them
    again).  Give both classes a hash that counts creations instead (class attributes swapped at harness start, like
    the RNG call sites; reference files untouched).  Only network GENERATION iterates these sets -- it is outside
    the parity path, the harness exports its result -- the tick never does."""
    import itertools
    for cls in classes:
        counter = itertools.count(1)
        orig_init = cls.__init__

        def init(self, *a, _orig=orig_init, _counter=counter, **k):
            self._cg_serial = next(_counter)
            _orig(self, *a, **k)
        cls.__init__ = init
        cls.__hash__ = lambda self: self._cg_serial


# --------------------------------------------------------------------------
# RNG injection
# --------------------------------------------------------------------------
class DrawContext:
    """Addresses draws: set .seed/.env/.tick before each reference call."""

    def __init__(self):
        self.seed = 0
        self.env = 0
        self.tick = 0
        self.active = False
        self.occ = {}            # (site, a) -> next occurrence number
        self.scan_base = 0       # env.scan_cnt at tick start
        self.poisson_tab = None
        self.trace = []          # (site, a, b, u32) of every draw this tick

    def begin(self, seed, env_id, tick, env_obj=None):
        self.seed, self.env, self.tick = seed, env_id, tick
        self.occ = {}
        self.trace = []
        self.scan_base = getattr(env_obj, "scan_cnt", 0) if env_obj is not None else 0
        self.active = True

    def end(self):
        self.active = False

    def next_occ(self, site, a):
        k = (site, a)
        v = self.occ.get(k, 0)
        self.occ[k] = v + 1
        return v

    def draw(self, site, a=0, b=0):
        u = R.draw(self.seed, self.env, self.tick, site, a, b)
        self.trace.append((site, a, b, u))
        return u


CTX = DrawContext()
_ORIG = {}


def _frames(depth=8):
    f = sys._getframe(2)
    out = []
    while f is not None and len(out) < depth:
        out.append(f)
        f = f.f_back
    return out


def _find(frames, name):
    for f in frames:
        if f.f_code.co_name == name:
            return f
    return None


def _inj_randint(lo, hi):
    if not CTX.active:
        return _ORIG["randint"](lo, hi)
    fr = _frames()
    if fr[0].f_code.co_name != "_stall":
        raise RuntimeError(f"unmapped randint site: {fr[0].f_code.co_name}:{fr[0].f_lineno}")
    caller = fr[1]
    loc = caller.f_locals
    at = loc.get("action_type")
    env = loc.get("self")
    if at == 3:
        site, a = S.SITE_STALL_REVERT, loc["device"].id
        b = 0
    elif at == 1:
        site, a = S.SITE_STALL_CLEAN, loc["device"].id
        b = CTX.next_occ(site, a)
    elif at == 4:
        site, a = S.SITE_STALL_PATCH, loc["device"].id
        b = CTX.next_occ(site, a)
    elif at == 5:
        site = S.SITE_STALL_SCAN
        a = loc["dev_id"] if "dev_id" in loc and getattr(env, "fast_scan", True) else loc["dsrc"].id
        b = int(env.scan_cnt) - CTX.scan_base - 1
    elif at == 13:
        site, a = S.SITE_STALL_ISOLATE, int(loc["dev_id"])
        b = CTX.next_occ(site, a)
    else:
        raise RuntimeError(f"unmapped _stall caller: action_type={at} at {caller.f_code.co_name}:{caller.f_lineno}")
    return R.randint(CTX.draw(site, a, b), lo, hi)


def _inj_choice(seq):
    if not CTX.active:
        return _ORIG["choice"](seq)
    fr = _frames()
    name = fr[0].f_code.co_name
    if name == "_random_incident_unblocked_edge":
        a = fr[0].f_locals["device_name"]
        site = S.SITE_PICK_BLOCK
        b = CTX.next_occ(site, a)
        return seq[R.index(CTX.draw(site, a, b), len(seq))]
    if name == "_random_incident_blocked_edge":
        a = fr[0].f_locals["device_name"]
        site = S.SITE_PICK_UNBLOCK
        b = CTX.next_occ(site, a)
        return seq[R.index(CTX.draw(site, a, b), len(seq))]
    if name == "pick_from":
        evo = _find(fr, "evolve_network")
        env = evo.f_locals["self"]
        s = fr[0].f_locals["s"]
        site = S.SITE_EVO_PICK_IN if s is env._inactive_ids else S.SITE_EVO_PICK_ACT
        ev = evo.f_locals["_"]
        canon = sorted(seq)
        return canon[R.index(CTX.draw(site, ev, 0), len(canon))]
    if name == "step":
        loc = fr[0].f_locals
        if "probe_from_device_id" not in loc and loc.get("action_type") == 1:
            site = S.SITE_ZERODAY
            a = CTX.next_occ(site, 0)
            canon = sorted(seq)
            return canon[R.index(CTX.draw(site, a, 0), len(canon))]
        if loc.get("action_type") == 2:
            return seq[R.index(CTX.draw(S.SITE_PROBE_SRC, 0, 0), len(seq))]
    if name == "predict":   # Detector.predict in random-detection mode (CDSimulator.py:697-700): the per-log scan path
        step = _find(fr, "step")   # (volt_typhoon_env.py:1030-1035); a = position of the log in this scan, b = scan ordinal
        env = step.f_locals["self"]
        b = int(env.scan_cnt) - CTX.scan_base - 1
        a = CTX.next_occ(S.SITE_DET_COIN, b)
        return seq[R.index(CTX.draw(S.SITE_DET_COIN, a, b), len(seq))]
    if name == "<listcomp>" and fr[1].f_code.co_name == "batch_predict":
        step = _find(fr, "step")
        env = step.f_locals["self"]
        b = int(env.scan_cnt) - CTX.scan_base - 1
        a = CTX.next_occ(S.SITE_DET_COIN, b)
        return seq[R.index(CTX.draw(S.SITE_DET_COIN, a, b), len(seq))]
    raise RuntimeError(f"unmapped choice site: {name}:{fr[0].f_lineno}")


def _inj_random():
    if not CTX.active:
        return _ORIG["random"]()
    fr = _frames()
    name = fr[0].f_code.co_name
    if name == "evolve_network":
        ev = fr[0].f_locals["_"]
        k = CTX.next_occ(S.SITE_EVO_COIN, ev)
        site = S.SITE_EVO_COIN if k == 0 else S.SITE_EVO_ATT
        return CTX.draw(site, ev, 0) / 4294967296.0
    if name == "generate_workloads":
        return CTX.draw(S.SITE_LAZY, fr[0].f_locals["did"], 0) / 4294967296.0
    raise RuntimeError(f"unmapped random() site: {name}:{fr[0].f_lineno}")


def _keyed_order(site, ids):
    keyed = sorted((CTX.draw(site, int(i), 0), int(i)) for i in ids)
    return [i for _, i in keyed]


def _inj_sample(population, k):
    if not CTX.active:
        return _ORIG["sample"](population, k)
    fr = _frames()
    if fr[0].f_code.co_name != "generate_workloads":
        raise RuntimeError(f"unmapped sample site: {fr[0].f_code.co_name}:{fr[0].f_lineno}")
    wtype = fr[0].f_locals["wtype"]
    site = S.SITE_ARR_CLIENT if wtype == "client" else S.SITE_ARR_SERVER
    if k > len(population):
        raise ValueError("sample larger than population")
    return _keyed_order(site, population)[:k]


def _inj_shuffle(lst):
    if not CTX.active:
        return _ORIG["shuffle"](lst)
    fr = _frames()
    if fr[0].f_code.co_name != "randomize_compromise_and_ownership":
        raise RuntimeError(f"unmapped shuffle site: {fr[0].f_code.co_name}:{fr[0].f_lineno}")
    by_id = {d.id: d for d in lst}
    order = _keyed_order(S.SITE_SHUFFLE, by_id.keys())
    lst[:] = [by_id[i] for i in order]


def _inj_uniform(a, b):
    if not CTX.active:
        return _ORIG["uniform"](a, b)
    fr = _frames()
    evo = _find(fr, "evolve_network")
    if evo is None:
        raise RuntimeError(f"unmapped uniform site: {fr[0].f_code.co_name}:{fr[0].f_lineno}")
    nid = evo.f_locals["nid"]
    return a + (b - a) * (CTX.draw(S.SITE_EVO_PA, int(nid), 0) / 4294967296.0)


def _inj_poisson(lam=1.0, size=None):
    if not CTX.active:
        return _ORIG["poisson"](lam, size)
    tab = R.poisson_table(float(lam), S.POISSON_TABLE)
    return R.cdf_lookup(CTX.draw(S.SITE_EVO_POISSON, 0, 0), tab)


def _inj_triangular(left, mode, right, size=None):
    if not CTX.active:
        return _ORIG["triangular"](left, mode, right, size)
    fr = _frames()
    if fr[0].f_code.co_name != "generate_workloads" or left != 0:
        raise RuntimeError(f"unmapped triangular site: {fr[0].f_code.co_name}:{fr[0].f_lineno}")
    did = fr[0].f_locals["did"]
    tab = R.triangular_ceil_table(float(mode), float(right), S.TRI_TABLE)
    val = 1 + R.cdf_lookup(CTX.draw(S.SITE_ARR_TIME, int(did), 0), tab)
    return np.array([float(val)])


_ORIG_NP_SEED = np.random.seed


def install_rng():
    mods = load_reference()
    rnd = mods["random"]
    if _ORIG:
        return
    _ORIG.update(randint=rnd.randint, choice=rnd.choice, random=rnd.random, sample=rnd.sample,
                 shuffle=rnd.shuffle, uniform=rnd.uniform,
                 poisson=np.random.poisson, triangular=np.random.triangular)
    rnd.randint = _inj_randint
    rnd.choice = _inj_choice
    rnd.random = _inj_random
    rnd.sample = _inj_sample
    rnd.shuffle = _inj_shuffle
    rnd.uniform = _inj_uniform
    np.random.poisson = _inj_poisson
    np.random.triangular = _inj_triangular


# --------------------------------------------------------------------------
# building an environment and flattening it
# --------------------------------------------------------------------------
def build_env(M, n_active, *, init_seed=1, overrides=None, strip_vuln_frac=0.0,
              extra_reachable=0, prewarm_star=False):
    """Construct and initialise a reference environment (init uses plain seeded MT;
    initialisation is outside the parity path -- the harness exports its result)."""
    mods = load_reference()
    rnd = mods["random"]
    rnd.seed(init_seed)
    np.random.seed(init_seed)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        env = mods["vte"].Volt_Typhoon_CyberDefenseEnv()
        env.numOfDevice = int(n_active)
        env.Max_network_size = int(M)
        env.base_line = "Nash"
        env.tech = "DO"
        env.mode = "defender"
        env.zero_day = False
        env.its = 0
        for k, v in (overrides or {}).items():
            setattr(env, k, v)
        env.initialize_environment()
        env._rebuild_graph_cache()      # what DoubleOracle.restore does (do_agent.py:893)
    # scenario diversification through the reference's own object API
    rr = np.random.RandomState(init_seed + 77)
    devs = list(env.simulator.subnet.net.values())
    if strip_vuln_frac > 0:
        for d in devs:
            if d.attacker_owned:
                continue
            if rr.rand() < strip_vuln_frac:
                for app in d.apps.values():
                    app.vulnerabilities.clear()
    for _ in range(extra_reachable):
        d = devs[rr.randint(len(devs))]
        d.reachable_by_attacker = True
    if prewarm_star:
        # let the reference's own evolve_network() materialise the attacker star edges
        # (CyberDefenseEnv.py:738-774) once, with no Poisson events, before the export
        lam = env.lambda_events
        env.lambda_events = 0.0
        env.evolve_network()
        env.lambda_events = lam
        env._rebuild_graph_cache()
    return env


from cygym_amd.interchange import (exploit_index_map, flatten_static, flatten_config, zero_day_mask,   # noqa: E402,F401
                                    flatten_dynamic, extra_edges)


def export_forest(env, rs_check=None):
    """Flatten the forest the reference's Detector holds (cygym_amd/detector.py, layout in cygym_spec.h) and
    check the flat walk against the reference's own batch_predict before it becomes a fixture."""
    from cygym_amd import detector as D
    det = env.simulator.detector
    words = D.flatten_forest(det.model)
    M = len(env.simulator.subnet.net)
    if M <= 48:
        pts = [(a, b) for a in range(M) for b in range(M)]
    else:
        rr = np.random.RandomState(M) if rs_check is None else rs_check
        pts = [(int(a), int(b)) for a, b in rr.randint(0, M, size=(3000, 2))]
    pts += [(int(l["from_device"]), int(l["to_device"])) for l in env.simulator.logger.logs[-64:]]
    ref = np.array([p == "A" for p in det.batch_predict(pts)])
    got = D.predict_flat(words, pts)
    assert np.array_equal(ref, got), "flattened forest disagrees with the reference's batch_predict"
    return words


class AscendingSet(set):
    """A set that iterates in ascending order.  evolve_network's preferential attachment walks
    `self._active_ids` (CyberDefenseEnv.py:795) in CPython hash-table order, which depends on the whole
    insertion / deletion history of the set (a re-added id lands behind its own tombstone).  The flat
    restatement defines that walk as ascending device id; scenarios that reach the PA branch install this
    class for env._active_ids / _inactive_ids so that the reference run follows the same definition."""
    def __iter__(self):
        return iter(sorted(set.__iter__(self)))


def install_ascending_sets(env):
    act, inact = AscendingSet(), AscendingSet()
    for d in env.simulator.subnet.net.values():          # what evolve_network's first call builds (:654-659)
        (act if not d.Not_yet_added else inact).add(d.id)
    env._active_ids, env._inactive_ids = act, inact


def topology_signature(env):
    return tuple((u, tuple(v)) for u, v in sorted(env._outnbrs.items()))


# --------------------------------------------------------------------------
# scenario runner
# --------------------------------------------------------------------------
def run_scenario(env0, n_envs, n_ticks, action_fn, *, seed=0, env_id_base=0, pre_fn=None,
                 record_draws=False, max_extra=0):
    """Deep-copy `env0` n_envs times, drive each with `action_fn(e, t, env, rs)` ->
    (mode, action) and record the flattened state after every tick.

    `action` is whatever the reference's step() accepts (None, a 4-tuple, or a list of
    4-tuples for step_grouped).  Returns a dict of stacked arrays."""
    install_rng()
    if getattr(env0, "zero_day", False):   # the product's restatement of the zero-day bookkeeping (:1504-1563) vs the reference's sets
        from cygym_amd.interchange import zero_day_bookkeeping
        zb = zero_day_bookkeeping(len(env0.simulator.exploits), env0.k_known, env0.j_private, sorted(env0.private_exploit_indices))
        assert set(zb["common"]) == set(env0.common_exploit_indices) and set(zb["private"]) == set(env0.private_exploit_indices)
        assert zb["owned_mask"] == zero_day_mask(env0) and len(zb["pool"]) == len(env0.unknown_pool_ids)
    static = flatten_static(env0)
    static["max_extra"] = int(max_extra)   # capacity of the per-env extra-edge list in this fixture
    config = flatten_config(env0)
    sig0 = topology_signature(env0)
    M = static["M"]
    per_env = []
    for e in range(n_envs):
        env = copy.deepcopy(env0)
        env._rebuild_graph_cache()
        rs = np.random.RandomState(10007 * (seed + 1) + e)
        env_id = env_id_base + e
        rng_tick = 0
        pre_dyn = flatten_dynamic(env, static)
        if pre_fn is not None:
            CTX.begin(seed, env_id, rng_tick, env)
            try:
                if pre_fn(e, env, rs):
                    rng_tick += 1
            finally:
                CTX.end()
        init_dyn = flatten_dynamic(env, static)
        init_dyn["ienv"][S.I_RNG_TICK] = rng_tick
        det = env.simulator.detector
        init_dyn["forest"] = np.zeros(S.FOREST_WORDS, np.uint32)
        if det.trained:
            init_dyn["forest"] = export_forest(env)
        fit_obj = getattr(det.model, "estimators_", None)   # held, not just its id(): a freed list's address gets reused
        det_events = []     # (tick, forest words, training rows) of every Detector.train(non-empty) of this env
        ticks = []
        acts = []
        for t in range(n_ticks):
            mode, action = action_fn(e, t, env, rs)
            agent_cnt = None
            if mode & S.MODE_PARTIAL:        # drive step(action, agent_cnt=<something else than len(net)>)
                agent_cnt = len(env.simulator.subnet.net) + 1
            env.mode = "defender" if (mode & 0xFF) == S.MODE_DEFENDER else "attacker"
            CTX.begin(seed, env_id, rng_tick, env)
            # IsolationForest(random_state=None).fit draws from the process-global numpy stream (CDSimulator.py:683,
            # :694): seed it with the draw addressed (env, tick, CG_SITE_DET_FIT) -- nothing else on the step path
            # reads that stream once poisson / triangular are injected
            _ORIG_NP_SEED(int(R.draw(seed, env_id, rng_tick, S.SITE_DET_FIT, 0, 0)))
            err = None
            try:
                state, raw, shaped, done, info, logs = env.step(action) if agent_cnt is None else env.step(action, agent_cnt)
            except ValueError as ex:   # malformed 11/12/13 actions raise in the reference
                err = str(ex)
            finally:
                CTX.end()
            if err is not None:
                raise RuntimeError(f"scenario produced a reference exception at env {e} tick {t}: {err}")
            det = env.simulator.detector
            if det.trained and getattr(det.model, "estimators_", None) is not fit_obj:   # refitted during this tick
                fit_obj = det.model.estimators_
                rows = [(int(l["from_device"]), int(l["to_device"])) for l in env.simulator.logger.logs[-S.TRAIN_WINDOW:]]
                if env.turbo:    # _train_detector clips and down-samples in turbo mode (volt_typhoon_env.py:165-169)
                    rows = rows[-int(env.turbo_train_max_logs):][:: max(1, int(env.turbo_train_stride))]
                grouped = isinstance(action, (list, tuple)) and action and isinstance(action[0], (list, tuple))
                n_fits = sum(1 for g in action if int(g[0]) == 10) if grouped else 1   # every action-10 group refits
                det_events.append((t, rng_tick, export_forest(env), np.asarray(rows, np.int32).reshape(-1, 2), n_fits))
            rng_tick += 1
            dyn = flatten_dynamic(env, static)
            dyn["ienv"][S.I_RNG_TICK] = rng_tick
            dyn["ienv"][S.I_LAST_ATYPE] = int(info.get("executed_atype", -1)) if isinstance(info, dict) else -1
            dyn["raw"] = float(raw)
            dyn["shaped"] = float(shaped)
            dyn["done"] = int(bool(done))
            dyn["obs"] = np.asarray(state, np.float64).reshape(M, 6).astype(np.float32)
            dyn["obs_def"] = np.asarray(env._get_defender_state(), np.float64).astype(np.float32)
            dyn["obs_att"] = np.asarray(env._get_attacker_state(), np.float32)
            dyn["topo_same"] = int(topology_signature(env) == sig0)
            if record_draws:
                dyn["draws"] = list(CTX.trace)
            ticks.append(dyn)
            acts.append((mode, action))
        pre_dyn["forest"] = np.zeros(S.FOREST_WORDS, np.uint32)
        per_env.append(dict(init=init_dyn, pre=pre_dyn, ticks=ticks, acts=acts, det_events=det_events))
    return dict(static=static, config=config, envs=per_env, seed=seed, env_id_base=env_id_base)


def encode_actions(acts, M, max_groups=1):
    """Encode reference-style actions into the flat form of the C ABI.
    Returns dict(mode[T], n_groups[T], atype[T,G], exploit[T,G,X], n_exploit[T,G],
    dev_ptr / dev_idx (ragged), app[T,G])."""
    T = len(acts)
    G = max_groups
    mode = np.zeros(T, np.int32)
    ng = np.zeros(T, np.int32)
    atype = np.zeros((T, G), np.int32)
    nexp = np.zeros((T, G), np.int32)
    expl = np.full((T, G, S.MAX_EXPLOITS), -1, np.int32)
    app = np.full((T, G), -1, np.int32)
    dev_cnt = np.zeros((T, G), np.int32)
    is_none = np.zeros(T, np.int32)
    dev_flat = []
    for t, (m, a) in enumerate(acts):
        mode[t] = m
        if a is None:                     # the reference substitutes a default (:847-874); the build's
            is_none[t] = 1                # host logic must do the same -- only the marker is stored
            atype[t, 0] = 8
            continue
        groups = a if (isinstance(a, (list, tuple)) and a and isinstance(a[0], (list, tuple))) else [a]
        grouped = groups is a
        ng[t] = len(groups) if grouped else 0      # 0 => single-action step()
        assert len(groups) <= G
        for g, (at, ex, dv, ap) in enumerate(groups):
            atype[t, g] = int(at)
            ex = [] if ex is None else list(np.asarray(ex).reshape(-1))
            assert len(ex) <= S.MAX_EXPLOITS
            nexp[t, g] = len(ex)
            for j, x in enumerate(ex):
                expl[t, g, j] = int(x)
            dv = [] if dv is None else [int(x) for x in np.asarray(dv).reshape(-1)]
            dev_cnt[t, g] = len(dv)
            dev_flat.extend(dv)
            app[t, g] = int(ap) if isinstance(ap, int) and not isinstance(ap, bool) else -1
    return dict(mode=mode, n_groups=ng, atype=atype, n_exploit=nexp, exploit=expl, app=app,
                dev_cnt=dev_cnt, is_none=is_none, dev_flat=np.asarray(dev_flat, np.int32))


def save_fixture(path, result, max_groups=1):
    """Write one scenario as a compressed .npz of plain arrays (no pickles)."""
    st, cfg = result["static"], result["config"]
    out = {f"static_{k}": np.asarray(v) for k, v in st.items()}
    out["config_keys"] = np.array(sorted(cfg.keys()))
    out["config_vals"] = np.array([float(cfg[k]) for k in sorted(cfg.keys())], np.float64)
    out["seed"] = np.int64(result["seed"])
    out["env_id_base"] = np.int64(result["env_id_base"])
    envs = result["envs"]
    N = len(envs)
    T = len(envs[0]["ticks"])
    keys = ["flags", "busy", "wl", "comp_by", "st_flags", "st_busy", "st_wl", "st_comp_by",
            "blocked", "ring", "ienv", "fenv", "extra"]    # `extra` has zero width when max_extra == 0
    for k in keys:
        out[f"init_{k}"] = np.stack([e["init"][k] for e in envs])
        out[f"pre_{k}"] = np.stack([e["pre"][k] for e in envs])
        out[f"exp_{k}"] = np.stack([np.stack([tk[k] for tk in e["ticks"]]) for e in envs])
    for k in ["raw", "shaped", "done", "obs", "obs_def", "obs_att", "topo_same"]:
        out[f"exp_{k}"] = np.stack([np.stack([np.asarray(tk[k]) for tk in e["ticks"]]) for e in envs])
    # comm-log history (what Detector.train fits on): before the first tick and after the last one
    out["init_hist"] = np.stack([e["init"]["hist"] for e in envs])
    out["pre_hist"] = np.stack([e["pre"]["hist"] for e in envs])
    out["fin_hist"] = np.stack([e["ticks"][-1]["hist"] for e in envs])
    out["init_forest"] = np.stack([e["init"]["forest"] for e in envs])
    # every Detector.train(non-empty logs) of the run (action 10): env, tick, rng tick, the forest the reference
    # fitted (flattened: cygym_spec.h) and the rows it was fitted on
    ev = [(i, t, rt, w, x, nf) for i, e in enumerate(envs) for (t, rt, w, x, nf) in e["det_events"]]
    out["det_env"] = np.asarray([v[0] for v in ev], np.int32)
    out["det_tick"] = np.asarray([v[1] for v in ev], np.int32)
    out["det_rng_tick"] = np.asarray([v[2] for v in ev], np.int32)
    out["det_n_fits"] = np.asarray([v[5] for v in ev], np.int32)    # fits the reference did in that tick (grouped ticks)
    out["det_forest"] = np.stack([v[3] for v in ev]) if ev else np.zeros((0, S.FOREST_WORDS), np.uint32)
    out["det_rows_ptr"] = np.cumsum([0] + [len(v[4]) for v in ev]).astype(np.int64)
    out["det_rows"] = np.concatenate([v[4] for v in ev]).astype(np.int32) if ev else np.zeros((0, 2), np.int32)
    import sklearn
    from cygym_amd import detector as D
    out["det_sklearn_version"] = np.array(sklearn.__version__)
    out["static_det_apl"] = D.apl_table()
    enc = [encode_actions(e["acts"], st["M"], max_groups) for e in envs]
    for k in ["mode", "n_groups", "atype", "n_exploit", "exploit", "app", "dev_cnt", "is_none"]:
        out[f"act_{k}"] = np.stack([a[k] for a in enc])
    out["act_dev_ptr"] = np.cumsum([0] + [len(a["dev_flat"]) for a in enc]).astype(np.int64)
    out["act_dev_flat"] = np.concatenate([a["dev_flat"] for a in enc]) if enc else np.zeros(0, np.int32)
    # narrow dtypes for size
    for k in list(out.keys()):
        v = out[k]
        if isinstance(v, np.ndarray) and v.dtype == np.int64 and k not in ("act_dev_ptr",):
            if v.size and np.abs(v).max() < 2 ** 31:
                out[k] = v.astype(np.int32)
    np.savez_compressed(path, **out)
    return N, T

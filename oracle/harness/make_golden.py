"""Generate tests/golden/*.npz by running the reference itself (see ref_harness.py).

Run in the build container only:
    python oracle/harness/make_golden.py [name ...]            (re)write tests/golden/
    python oracle/harness/make_golden.py --check [name ...]    regenerate into a temp dir, fail on any difference
The fixtures are plain arrays: exported topology, initial struct-of-arrays state,
the action script, and the expected state / rewards / observations after every tick.

Reproducibility.  The reference's network generator keeps `App` / `Vulnerability` objects -- which define no
__hash__ -- in Python sets (CDSimulator.py:26, :30) and then does `random.choice(list(that_set))` (`_attach_extra`,
`changeVulTarget`), so the network it builds depends on object ADDRESSES, i.e. on everything the process allocated
before: generated back to back in one interpreter, a scenario's `static_vuln` depended on which scenarios ran
before it (round-1 finding), and even in fresh interpreters it moved when this harness's own imports changed.
Two measures: (1) ref_harness gives those two classes a creation-order hash (class attributes swapped at start-up,
reference files untouched), which removes the address dependence at its root; (2) every scenario is generated in
its OWN fresh interpreter with PYTHONHASHSEED=0 (string-keyed sets) and, belt and braces, address-space
randomisation off (`setarch -R`; `--aslr` leaves it on, to show that the result no longer depends on it).
Initialisation is outside the parity path (the harness exports its result); the tick itself has no such dependence.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H          # noqa: E402
from cygym_amd import spec as S  # noqa: E402

GOLDEN = os.path.join(H.REPO, "tests", "golden")

DEF, ATT = S.MODE_DEFENDER, S.MODE_ATTACKER


def dev_list(rs, M, kmax, unique=True):
    k = int(rs.randint(1, max(2, kmax + 1)))
    if unique:
        return [int(x) for x in rs.choice(M, size=min(k, M), replace=False)]
    return [int(x) for x in rs.randint(0, M, size=k)]


def mixed_actions(M, def_types, att_types, kmax, X=2, unique=True):
    def fn(e, t, env, rs):
        mode = DEF if (t % 2 == 0) else ATT
        if mode == DEF:
            at = int(rs.choice(def_types))
            dv = dev_list(rs, M, kmax, unique)
            if at in (2, 3, 8) and rs.rand() < 0.3:
                dv = []
            if at == 10 and rs.rand() < 0.5:
                dv = []
            app = int(rs.randint(-1, 9))
            return mode, (at, np.array([int(rs.randint(0, X))]), dv, app)
        at = int(rs.choice(att_types))
        ne = 1 if rs.rand() < 0.8 else 2
        ex = np.array([int(rs.randint(0, X + 1)) for _ in range(ne)])
        return mode, (at, ex, dev_list(rs, M, 3), 0)
    return fn


ALL_DEF = [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13]
ALL_ATT = [1, 1, 2, 3]

SCENARIOS = {}


def scenario(name):
    def deco(f):
        SCENARIOS[name] = f
        return f
    return deco


@scenario("s16_mixed")
def s16_mixed():
    env0 = H.build_env(16, 12, init_seed=3, strip_vuln_frac=0.4, extra_reachable=1)
    return H.run_scenario(env0, 3, 300, mixed_actions(16, ALL_DEF, ALL_ATT, 5), seed=11), 1


@scenario("s64_mixed")
def s64_mixed():
    env0 = H.build_env(64, 48, init_seed=5, strip_vuln_frac=0.5, extra_reachable=3)
    return H.run_scenario(env0, 3, 300, mixed_actions(64, ALL_DEF, ALL_ATT, 8), seed=12, env_id_base=1000), 1


@scenario("s256_mixed")
def s256_mixed():
    env0 = H.build_env(256, 200, init_seed=7, strip_vuln_frac=0.5, extra_reachable=6)
    return H.run_scenario(env0, 2, 90, mixed_actions(256, ALL_DEF, ALL_ATT, 32), seed=13, env_id_base=7), 1


@scenario("s16_train")
def s16_train():
    """Action 10 (detector training) present; later scans run the trained detector."""
    env0 = H.build_env(16, 14, init_seed=9, strip_vuln_frac=0.3)
    return H.run_scenario(env0, 2, 120, mixed_actions(16, ALL_DEF + [10, 10], ALL_ATT, 4), seed=14), 1


@scenario("s32_grouped")
def s32_grouped():
    """step_grouped (IPPO/MAPPO style) interleaved with single-action steps."""
    M = 32
    env0 = H.build_env(M, 24, init_seed=21, strip_vuln_frac=0.4, extra_reachable=2)
    single = mixed_actions(M, ALL_DEF, ALL_ATT, 6)

    def fn(e, t, env, rs):
        if (t // 3) % 2 == 0 or rs.rand() < 0.3:
            mode = DEF if (t % 2 == 0) else ATT
            ng = int(rs.randint(1, 5))
            groups = []
            for _ in range(ng):
                if mode == DEF:
                    at = int(rs.choice([0, 1, 1, 1, 2, 3, 4, 5, 7, 8, 11, 13]))
                    dv = dev_list(rs, M, 6, unique=(rs.rand() < 0.7))
                else:
                    at = int(rs.choice([0, 1, 2, 3]))
                    dv = dev_list(rs, M, 3)
                groups.append((at, np.array([0]), dv, 0))
            return mode, groups
        return single(e, t, env, rs)
    return H.run_scenario(env0, 2, 240, fn, seed=15, env_id_base=50), 4


@scenario("s24_norng")
def s24_norng():
    """RNG-free: lambda_events = 0, workload_cap = 0, no stalling / picking actions."""
    env0 = H.build_env(24, 20, init_seed=31, strip_vuln_frac=0.4, extra_reachable=2,
                       overrides=dict(lambda_events=0.0, workload_cap=0))
    return H.run_scenario(env0, 2, 150, mixed_actions(24, [2, 7, 8, 10, 11, 12], [1, 1, 3], 4), seed=16), 1


@scenario("s16_dups")
def s16_dups():
    """Duplicate / unsorted device lists and out-of-range action types."""
    M = 16
    env0 = H.build_env(M, 13, init_seed=41, strip_vuln_frac=0.3, extra_reachable=1)
    base = mixed_actions(M, ALL_DEF, ALL_ATT, 10, unique=False)

    def fn(e, t, env, rs):
        mode, (at, ex, dv, app) = base(e, t, env, rs)
        if rs.rand() < 0.08:
            at = int(rs.choice([-1, 14, 99])) if mode == DEF else int(rs.choice([-2, 0, 4, 5, 9]))
        return mode, (at, ex, dv, app)
    return H.run_scenario(env0, 3, 300, fn, seed=17, env_id_base=3), 1


@scenario("s16_baselines")
def s16_baselines():
    """base_line != "Nash": defender forced to no-op (:913), "No Attack" gates the attacker (:1130)."""
    out = None
    M = 16
    res = []
    for i, bl in enumerate(["No Defense", "No Attack", "Preset"]):
        env0 = H.build_env(M, 12, init_seed=51, strip_vuln_frac=0.3, overrides=dict(base_line=bl))
        r = H.run_scenario(env0, 1, 80, mixed_actions(M, ALL_DEF, ALL_ATT, 4), seed=18, env_id_base=i)
        res.append((bl, r))
    return res, 1


@scenario("s16_none")
def s16_none():
    """action=None: the reference's baseline defaults by mode x base_line (:847-874)."""
    M = 16
    res = []
    for i, bl in enumerate(["Nash", "No Defense", "No Attack", "Preset"]):
        env0 = H.build_env(M, 12, init_seed=55, strip_vuln_frac=0.3, overrides=dict(base_line=bl))
        base = mixed_actions(M, ALL_DEF, ALL_ATT, 4)

        def fn(e, t, env, rs, base=base):
            mode, a = base(e, t, env, rs)
            return mode, (None if rs.rand() < 0.6 else a)
        r = H.run_scenario(env0, 1, 80, fn, seed=24, env_id_base=10 + i)
        res.append((bl, r))
    return res, 1


@scenario("s16_partial")
def s16_partial():
    """step(action, agent_cnt != len(net)): no workload advance / arrivals / step counters (:1207, :1307)."""
    M = 16
    env0 = H.build_env(M, 13, init_seed=111, strip_vuln_frac=0.3, extra_reachable=1)
    base = mixed_actions(M, ALL_DEF, ALL_ATT, 4)

    def fn(e, t, env, rs):
        mode, a = base(e, t, env, rs)
        if rs.rand() < 0.35:
            mode |= S.MODE_PARTIAL
        return mode, a
    return H.run_scenario(env0, 2, 160, fn, seed=25, env_id_base=20), 1


@scenario("s600_sparse")
def s600_sparse():
    """len(net) > 500: sparse attacker connect (:1344) and the lazy workload path (CDSimulator.py:325)."""
    M = 600
    env0 = H.build_env(M, 520, init_seed=61, strip_vuln_frac=0.5, extra_reachable=5, prewarm_star=True)
    # a removal event can deactivate the hub: the star then re-forms around the next owned device
    return H.run_scenario(env0, 1, 70, mixed_actions(M, ALL_DEF, ALL_ATT, 40), seed=19, max_extra=128), 1


@scenario("s16_randomize")
def s16_randomize():
    """randomize_compromise_and_ownership() before the episode (do_agent.py:189): with several
    attacker-owned devices the next evolve star-connects the reshuffled set (CyberDefenseEnv.py:738-774);
    the added edges live in the env's extra-edge list."""
    M = 16
    env0 = H.build_env(M, 13, init_seed=71, strip_vuln_frac=0.3, overrides=dict(sv_attacker_fraction=0.25))

    def pre(e, env, rs):
        env.randomize_compromise_and_ownership()
        return True
    return H.run_scenario(env0, 3, 120, mixed_actions(M, ALL_DEF, ALL_ATT, 4), seed=20, pre_fn=pre, max_extra=32), 1


def edge_heavy_actions(M, kmax, X=2):
    """Block / unblock / spread / probe dominate, so that the added edges are walked, picked and flipped."""
    def fn(e, t, env, rs):
        if t % 2 == 0:
            at = int(rs.choice([6, 6, 6, 9, 9, 1, 7, 5, 8, 13]))
            owned = [d.id for d in env.simulator.subnet.net.values() if d.attacker_owned]
            dv = dev_list(rs, M, kmax, unique=rs.rand() < 0.7)
            if owned and rs.rand() < 0.7:       # aim at the star's endpoints
                dv = [int(x) for x in rs.choice(owned, size=min(len(owned), 1 + int(rs.randint(0, 4))), replace=True)] + dv[:2]
            return DEF, (at, np.array([0]), dv, 0)
        at = int(rs.choice([1, 1, 1, 2, 2, 3]))
        ne = 1 if rs.rand() < 0.7 else 2
        return ATT, (at, np.array([int(rs.randint(0, X)) for _ in range(ne)]), [], 0)
    return fn


@scenario("s24_star")
def s24_star():
    """Attacker-owned devices keep appearing (p_attacker > 0) and the hub can be removed: the star grows
    (CyberDefenseEnv.py:690-694, 738-774) while block / unblock / spread / probe run over the merged rows."""
    M = 24
    env0 = H.build_env(M, 12, init_seed=111, strip_vuln_frac=0.4, extra_reachable=1,
                       overrides=dict(lambda_events=1.5, p_add=0.45, p_attacker=0.4, Min_network_size=4))
    return H.run_scenario(env0, 3, 240, edge_heavy_actions(M, 5), seed=31, max_extra=256), 1


@scenario("s20_pa")
def s20_pa():
    """Inactive devices with no edges at all: once activated they are attached by preferential attachment
    (CyberDefenseEnv.py:776-843, random.uniform :817)."""
    M = 20
    env0 = H.build_env(M, 10, init_seed=121, strip_vuln_frac=0.3,
                       overrides=dict(lambda_events=2.0, p_add=0.7, p_attacker=0.15, Min_network_size=4))
    g = env0.simulator.subnet.graph
    lonely = [d.id for d in env0.simulator.subnet.net.values() if d.Not_yet_added][:6]
    kill = [e.index for e in g.es if e.source in lonely or e.target in lonely]
    g.delete_edges(kill)
    env0._rebuild_graph_cache()
    H.install_ascending_sets(env0)   # the PA walk over env._active_ids is defined as ascending id (see ref_harness)
    return H.run_scenario(env0, 3, 200, edge_heavy_actions(M, 4), seed=32, max_extra=64), 1


@scenario("s16_coin")
def s16_coin():
    """Detector in random-detection mode (Detector.train([]) CDSimulator.py:688-690)."""
    M = 16
    env0 = H.build_env(M, 14, init_seed=81, strip_vuln_frac=0.2, extra_reachable=1)

    def pre(e, env, rs):
        env.simulator.detector.train([])
        return False
    return H.run_scenario(env0, 2, 200, mixed_actions(M, [5, 5, 5, 1, 8, 6], [1, 1, 2], 5), seed=21, pre_fn=pre), 1


@scenario("s12_episode")
def s12_episode():
    """Crosses the 1000-tick episode cap (CyberDefenseEnv.py:549)."""
    M = 12
    env0 = H.build_env(M, 10, init_seed=91, strip_vuln_frac=0.3)
    return H.run_scenario(env0, 1, 1004, mixed_actions(M, ALL_DEF, ALL_ATT, 3), seed=22), 1


@scenario("s16_zeroday")
def s16_zeroday():
    M = 16
    env0 = H.build_env(M, 12, init_seed=101, strip_vuln_frac=0.2,
                       overrides=dict(zero_day=True, k_known=1, j_private=1))
    return H.run_scenario(env0, 2, 120, mixed_actions(M, ALL_DEF, ALL_ATT, 4, X=3), seed=23), 1


def trained_actions(M, kmax, X=2, grouped=False):
    """Spread-heavy attacker (fills the comm log), defender dominated by scans (5) and detector training (10)."""
    def fn(e, t, env, rs):
        if t % 2 == 1 or t < 3:
            at = int(rs.choice([1, 1, 1, 2]))
            return ATT, (at, np.array([int(rs.randint(0, X))]), [], 0)
        if grouped and rs.rand() < 0.25:      # step_grouped honours action 10 too (volt_typhoon_env.py:654-664)
            groups = [(int(rs.choice([10, 1, 2, 8])), np.array([0]), dev_list(rs, M, kmax), 0) for _ in range(int(rs.randint(1, 4)))]
            return DEF, groups
        at = int(rs.choice([5, 5, 5, 5, 10, 10, 1, 6, 7, 8, 13]))
        dv = dev_list(rs, M, kmax, unique=rs.rand() < 0.8)
        if at == 10 and rs.rand() < 0.5:
            dv = []
        return DEF, (at, np.array([0]), dv, 0)
    return fn


@scenario("s16_trained")
def s16_trained():
    """Trained-detector mode (CDSimulator.py:688-695, :721-723): action 10 fits the IsolationForest on the last
    <= 2000 logs, every later scan (action 5) goes through its predictions, flagged senders are un-compromised
    and stalled (volt_typhoon_env.py:1051-1069).  Retraining several times per episode."""
    M = 16
    env0 = H.build_env(M, 14, init_seed=131, strip_vuln_frac=0.2, extra_reachable=1)
    return H.run_scenario(env0, 3, 260, trained_actions(M, 5), seed=41), 1


@scenario("s64_trained")
def s64_trained():
    """The same at 64 devices with longer device lists, grouped ticks carrying action 10, and a log that grows
    past the 2000-entry training window and the 2048-entry history ring."""
    M = 64
    env0 = H.build_env(M, 56, init_seed=141, strip_vuln_frac=0.3, extra_reachable=3)
    return H.run_scenario(env0, 2, 220, trained_actions(M, 12, grouped=True), seed=42, env_id_base=300), 4


@scenario("s64_turbo")
def s64_turbo():
    """env.turbo = True (volt_typhoon_env.py:92; no reference driver sets it): arrivals capped and ramped with step_num
    (:219-231; a short arrival period so that many ticks hit it), scans skip the detector (:1055), trainings fit on
    the clipped, strided log (:165-169)."""
    M = 64
    env0 = H.build_env(M, 56, init_seed=151, strip_vuln_frac=0.3, extra_reachable=2,
                       overrides=dict(turbo=True, workload_period_base=3, turbo_ramp_steps=60, turbo_fraction_clients=0.13))
    return H.run_scenario(env0, 2, 260, mixed_actions(M, ALL_DEF + [5, 5, 10], ALL_ATT, 8), seed=43, env_id_base=400), 1


def slow_scan_actions(M, kmax, X=2, train=True):
    """Spread-heavy attacker (fills the comm log); defender dominated by per-log scans (5), with trainings (10), cleans,
    duplicate scan lists and the occasional no-op."""
    def fn(e, t, env, rs):
        if t % 2 == 1 or t < 3:
            at = int(rs.choice([1, 1, 1, 2]))
            return ATT, (at, np.array([int(rs.randint(0, X))]), [], 0)
        at = int(rs.choice([5, 5, 5, 5, 5, 1, 6, 7, 8, 13] + ([10, 10] if train else [])))
        dv = dev_list(rs, M, kmax, unique=rs.rand() < 0.7)
        if at == 10 and rs.rand() < 0.5:
            dv = []
        return DEF, (at, np.array([0]), dv, 0)
    return fn


@scenario("s16_slowscan")
def s16_slowscan():
    """fast_scan = False (volt_typhoon_env.py:1030-1050): every scan predicts the last <= 256 log entries one by one with
    Detector.predict, 0.5 * def_scale per entry; "A" entries discover and clean their SENDER; the scanned device's
    anomaly_score becomes the decision_function value of the last entry (observation column 3).  Untrained first
    (scores None), then trained and retrained by action 10; the log outgrows the 256-entry window."""
    M = 16
    env0 = H.build_env(M, 14, init_seed=161, strip_vuln_frac=0.2, extra_reachable=1, overrides=dict(fast_scan=False))
    return H.run_scenario(env0, 3, 220, slow_scan_actions(M, 4), seed=51), 1


@scenario("s16_slowcoin")
def s16_slowcoin():
    """The per-log scan path with the detector in random-detection mode (Detector.train([]), CDSimulator.py:688-690,
    :697-700): one coin per log entry and scan; a flagged sender keeps the stall of the last scan that flagged it."""
    M = 16
    env0 = H.build_env(M, 14, init_seed=171, strip_vuln_frac=0.2, extra_reachable=1, overrides=dict(fast_scan=False))

    def pre(e, env, rs):
        env.simulator.detector.train([])
        return False
    return H.run_scenario(env0, 2, 160, slow_scan_actions(M, 4, train=False), seed=52, pre_fn=pre), 1


@scenario("s32_slowturbo")
def s32_slowturbo():
    """fast_scan = False with env.turbo = True (:1036-1038): scans still pay per entry, predict nothing, and set the scanned
    devices' anomaly_score to 0.0."""
    M = 32
    env0 = H.build_env(M, 28, init_seed=181, strip_vuln_frac=0.3, extra_reachable=2,
                       overrides=dict(fast_scan=False, turbo=True, workload_period_base=3, turbo_ramp_steps=40))
    return H.run_scenario(env0, 2, 120, slow_scan_actions(M, 6), seed=53, env_id_base=500), 1


def outputs_of(name):
    """File stems a scenario writes."""
    if name == "s16_baselines":
        return [f"{name}_{b}" for b in ("no_defense", "no_attack", "preset")]
    if name == "s16_none":
        return [f"{name}_{b}" for b in ("nash", "no_defense", "no_attack", "preset")]
    return [name]


def generate_one(name, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    res, G = SCENARIOS[name]()
    if isinstance(res, list):
        for bl, r in res:
            path = os.path.join(out_dir, f"{name}_{bl.replace(' ', '_').lower()}.npz")
            n, t = H.save_fixture(path, r, G)
            print(f"{path}: N={n} T={t} {os.path.getsize(path) / 1024:.1f} KiB", flush=True)
    else:
        path = os.path.join(out_dir, f"{name}.npz")
        n, t = H.save_fixture(path, res, G)
        print(f"{path}: N={n} T={t} {os.path.getsize(path) / 1024:.1f} KiB", flush=True)


ASLR = False


def spawn(name, out_dir):
    """One scenario in its own fresh interpreter: no ASLR, fixed hash seed, no bytecode written (see module doc)."""
    import platform
    import shutil
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--one", name, "--out", out_dir]
    if ASLR:
        pass
    elif shutil.which("setarch"):
        cmd = ["setarch", platform.machine(), "-R"] + cmd
    else:
        print("[make_golden] warning: setarch not found, address-space randomisation stays on", file=sys.stderr)
    env = dict(os.environ, PYTHONHASHSEED="0", PYTHONDONTWRITEBYTECODE="1")
    subprocess.run(cmd, check=True, env=env)


def compare_dirs(new_dir, old_dir, stems):
    """Every array of every fixture must be identical (same key set, dtype, shape, bytes)."""
    bad = []
    for stem in stems:
        a_path, b_path = os.path.join(new_dir, stem + ".npz"), os.path.join(old_dir, stem + ".npz")
        if not os.path.exists(b_path):
            bad.append(f"{stem}: not committed under {old_dir}")
            continue
        a, b = np.load(a_path), np.load(b_path)
        if set(a.files) != set(b.files):
            bad.append(f"{stem}: key sets differ: {sorted(set(a.files) ^ set(b.files))}")
            continue
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype != y.dtype or x.shape != y.shape or not np.array_equal(x, y):
                bad.append(f"{stem}: {k} differs")
    return bad


def main(argv):
    import argparse
    import tempfile
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--one", help="(internal) generate this scenario in the current process")
    ap.add_argument("--out", default=GOLDEN)
    ap.add_argument("--check", action="store_true", help="regenerate into a temp dir and compare with tests/golden")
    ap.add_argument("--aslr", action="store_true", help="leave address-space randomisation on in the child interpreters")
    args = ap.parse_args(argv)
    global ASLR
    ASLR = args.aslr
    if args.one:
        generate_one(args.one, args.out)
        return 0
    names = args.names or list(SCENARIOS)
    if not args.check:
        for name in names:
            spawn(name, GOLDEN)
        keysets = {stem: frozenset(np.load(os.path.join(GOLDEN, stem + ".npz")).files) for n in SCENARIOS for stem in outputs_of(n)
                   if os.path.exists(os.path.join(GOLDEN, stem + ".npz"))}
        if len(set(keysets.values())) > 1:
            print("[make_golden] warning: fixtures do not share one key set (regenerate all of them)", file=sys.stderr)
        return 0
    with tempfile.TemporaryDirectory(prefix="cygym_golden_check_") as tmp:
        for name in names:
            spawn(name, tmp)
        bad = compare_dirs(tmp, GOLDEN, [s for n in names for s in outputs_of(n)])
    for line in bad:
        print("DIFF", line)
    print(f"[make_golden --check] {len(names)} scenario(s): {'clean' if not bad else str(len(bad)) + ' difference(s)'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

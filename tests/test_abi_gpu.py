"""Error behaviour of the C ABI on the GPU (include/cygym_abi.h): every misuse comes back as a negative
CYGYM_E* code with a message; nothing aborts, nothing silently falls back."""
import ctypes as C

import numpy as np
import pytest

from cygym_amd import abi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

EINVAL, EHIP, EUNSUP, ENOTBOUND = -1, -2, -3, -4


def test_misuse_returns_error_codes():
    from cygym_amd import _lib
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    lib = _lib.load()
    topo, init, ck = make_topology(16, 2, seed=2)
    cfg = abi.EnvConfig(seed=2, **ck)
    t, c = topo.to_c(), cfg.to_c()
    h = C.c_void_p()
    assert lib.cygym_create(C.byref(t), C.byref(c), 8, 0, C.byref(h)) == 0
    a, o = abi.Actions(), abi.Outputs()
    assert lib.cygym_step(h, C.byref(a), C.byref(o), None) == ENOTBOUND      # no cygym_bind yet
    assert b"not bound" in lib.cygym_last_error(h)
    b = abi.Buffers()
    assert lib.cygym_bind(h, C.byref(b)) == EINVAL                            # null planes
    lib.cygym_destroy(h)

    env = BatchedCyberDefenseEnv(topo, cfg, 8, init, device="cuda:0", max_groups=1, max_devs=4)
    assert env.lib.cygym_step(env._h, C.byref(abi.Actions()), C.byref(env._out), None) == EINVAL   # null action arrays
    assert env.lib.cygym_observe(env._h, 7, C.c_void_p(env.obs.data_ptr()), None) == EINVAL          # unknown role
    assert env.lib.cygym_rollout(env._h, 0, C.byref(abi.Actions()), C.byref(env._out), None) == EINVAL
    # the env still works after the rejected calls
    env.gen_actions(0)
    obs, raw, shaped, done = env.step()
    assert np.isfinite(raw.cpu().numpy()).all()
    env.close()

    # auto_reset needs a snapshot: BatchedCyberDefenseEnv always sets one, the raw ABI must insist on it
    cfg2 = abi.EnvConfig(seed=2, auto_reset=1, **ck)
    c2 = cfg2.to_c()
    h2 = C.c_void_p()
    assert lib.cygym_create(C.byref(t), C.byref(c2), 8, 0, C.byref(h2)) == 0
    st = {k: torch.zeros_like(v) for k, v in env.state.items() if k in abi.BUFFER_FIELDS}
    bb = abi.Buffers()
    for k in abi.BUFFER_FIELDS:
        setattr(bb, k, st[k].data_ptr() if st[k].numel() else None)
    bb.n_envs = 8
    assert lib.cygym_bind(h2, C.byref(bb)) == 0
    aa = abi.Actions()
    for k, v in env.act.items():
        setattr(aa, k, v.data_ptr())
    aa.max_groups, aa.max_devs = 1, 4
    assert lib.cygym_step(h2, C.byref(aa), C.byref(env._out), None) == EINVAL
    assert b"snapshot" in lib.cygym_last_error(h2)
    lib.cygym_destroy(h2)


def test_launches_follow_the_callers_stream():
    """Calls are stream-ordered on torch's CURRENT stream (include/cygym_abi.h conventions): the same ticks issued
    inside a side stream, interleaved with other work on it, give the same state as on the default stream."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    topo, init, ck = make_topology(64, 4, seed=4)
    cfg = abi.EnvConfig(seed=4, **ck)
    a = BatchedCyberDefenseEnv(topo, cfg, 256, init, device="cuda:0", max_groups=1, max_devs=8)
    b = BatchedCyberDefenseEnv(topo, cfg, 256, init, device="cuda:0", max_groups=1, max_devs=8)
    side = torch.cuda.Stream(device="cuda:0")
    junk = torch.zeros(1 << 22, device="cuda:0")
    for t in range(40):
        a.gen_actions(t)
        a.step()
    act, out = b.alloc_rollout(10)
    with torch.cuda.stream(side):
        for t in range(30):
            junk.add_(1.0)          # unrelated work queued on the same side stream
            b.gen_actions(t)
            b.step()
        b.gen_actions_rollout(30, act)
        b.rollout(act, out)         # the last 10 ticks as one launch, still on the side stream
    side.synchronize()
    sa, sb = a.state_numpy(), b.state_numpy()
    for k in ("live", "stash", "blocked", "ring", "ienv", "fenv"):
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    np.testing.assert_array_equal(a.raw.cpu().numpy(), out["raw"][-1].cpu().numpy())
    a.close(); b.close()


def test_sub_batches_on_streams_equal_full_batch_steps():
    """cygym_step_range: the batch stepped as 4 sub-batches, each on its own stream (the pipelined closed-loop
    driver of bench.py), gives the same state, rewards and observations as full-batch cygym_step."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    topo, init, ck = make_topology(64, 4, seed=6)
    cfg = abi.EnvConfig(seed=6, **ck)
    N, S_ = 300, 4     # ragged: the last sub-batch is shorter
    a = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    b = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    scripts = []
    for t in range(30):
        act = {k: torch.empty_like(v) for k, v in a.act.items()}
        a.gen_actions(t, act)
        scripts.append(act)
    torch.cuda.synchronize()
    for t in range(30):
        a.step(scripts[t])
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(S_)]
    per = (N + S_ - 1) // S_
    for j, st in enumerate(streams):
        lo, n = j * per, max(0, min(per, N - j * per))
        with torch.cuda.stream(st):
            for t in range(30):
                b.step_range(lo, n, scripts[t])
    torch.cuda.synchronize()
    sa, sb = a.state_numpy(), b.state_numpy()
    for k in ("live", "stash", "blocked", "ring", "ienv", "fenv"):
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    np.testing.assert_array_equal(a.raw.cpu().numpy(), b.raw.cpu().numpy())
    np.testing.assert_array_equal(a.obs.cpu().numpy(), b.obs.cpu().numpy())
    # ranges outside the batch are refused, an empty one is a no-op
    assert b.lib.cygym_step_range(b._h, N - 2, 3, C.byref(b.actions_struct()), C.byref(b._out), None) == EINVAL
    assert b.lib.cygym_step_range(b._h, 5, 0, C.byref(b.actions_struct()), C.byref(b._out), None) == 0
    a.close(); b.close()


def test_malformed_action_tensors_are_python_errors():
    """The kernels index the action tensors by raw pointer: wrong dtype / shape / device must be rejected on the
    host (a short tensor would otherwise be an out-of-bounds read on the GPU)."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    topo, init, ck = make_topology(16, 2, seed=2)
    env = BatchedCyberDefenseEnv(topo, abi.EnvConfig(seed=2, **ck), 8, init, device="cuda:0", max_groups=2, max_devs=4)
    good = {k: v.clone() for k, v in env.act.items()}
    env.step(good)
    for k, bad in (("mode", good["mode"].to(torch.int64)), ("atype", good["atype"][:4].contiguous()),
                   ("exploit", good["exploit"][:, :, :3].contiguous()), ("dev_cnt", good["dev_cnt"][:, :1].contiguous()),
                   ("dev_idx", good["dev_idx"].cpu()), ("app", good["app"].t())):
        with pytest.raises(ValueError):
            env.step({**good, k: bad})
    act, out = env.alloc_rollout(3)
    env.rollout(act, out)
    with pytest.raises(ValueError):
        env.rollout({**act, "n_groups": act["n_groups"][:2].contiguous()}, out)
    with pytest.raises(ValueError):
        env.rollout(act, {**out, "obs": out["obs"][:, :4].contiguous()})
    # raw ABI: more ids than envs, missing scratch
    ids = torch.zeros(9, dtype=torch.int32, device="cuda:0")
    assert env.lib.cygym_reset(env._h, None, C.c_void_p(ids.data_ptr()), 9, None) == EINVAL
    assert env.lib.cygym_randomize(env._h, None, 8, None, None) == EINVAL
    env.close()


def test_slow_scan_needs_its_planes_and_rolls_out_tick_by_tick():
    """fast_scan = False: cygym_bind refuses a buffer set without the history / anomaly planes; cygym_rollout on such a
    handle (the per-log scan path lives in the per-tick kernels) gives the reference's trajectory of fixture
    s16_slowcoin -- rewards of every tick, final state, final anomaly scores."""
    import golden_io as gio
    from cygym_amd import _lib
    from cygym_amd import spec as S
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from oracle import driver as od
    fx = gio.Fixture("s16_slowcoin")
    assert not fx.cfg.fast_scan
    lib = _lib.load()
    t, c = fx.topo.to_c(), fx.cfg.to_c()
    h = C.c_void_p()
    assert lib.cygym_create(C.byref(t), C.byref(c), fx.N, 0, C.byref(h)) == 0
    env = BatchedCyberDefenseEnv(fx.topo, fx.cfg, fx.N, fx.init, device="cuda:0", max_groups=fx.G, max_devs=fx.L)
    assert env.detector and env.state["anomaly"].shape == (fx.N, fx.M)       # the planes come with the flag
    bb = abi.Buffers()
    for k in abi.BUFFER_FIELDS:
        setattr(bb, k, env.state[k].data_ptr() if (env.state[k].numel() and k != "anomaly") else None)
    bb.n_envs = fx.N
    assert lib.cygym_bind(h, C.byref(bb)) == EINVAL and b"anomaly" in lib.cygym_last_error(h)
    lib.cygym_destroy(h)
    act, out = env.alloc_rollout(fx.T)
    one = od.alloc_actions(fx.N, fx.G, fx.L)
    for t_ in range(fx.T):
        fx.actions(t_, one)                       # (no action=None ticks in this fixture: the state is not needed)
        for k in act:
            act[k][t_].copy_(torch.from_numpy(one[k]).reshape(act[k][t_].shape))
    env.rollout(act, out)
    np.testing.assert_allclose(out["raw"].cpu().numpy().T, fx.exp["raw"], rtol=0, atol=1e-9)
    got = env.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, fx.expected_state(fx.T - 1), "s16_slowcoin rollout")
    np.testing.assert_allclose(got["anomaly"], fx.exp["obs"][:, fx.T - 1, :, 3], rtol=0, atol=1e-6)
    gio.assert_obs_equal(out["obs"][fx.T - 1].cpu().numpy().reshape(fx.N, -1), fx.exp["obs"][:, fx.T - 1].reshape(fx.N, -1), True, "last obs")
    env.close()

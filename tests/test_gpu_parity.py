"""-m gpu: the HIP kernels (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (north_star): trajectories within 1e-4 rel-L2 of the fp64 oracle for the fp32 RK4
kernel; reward/done exact except within 1e-6 of a threshold.  The fp64 RK45 kernel is held to 1e-6.
"""
import numpy as np
import pytest
import torch

from helpers import assert_discrete_rewards, both_events_spec, counter_tables, rel_l2
from oracle import pvder_model as pm
from oracle.bridge_env import BridgeVecEnv

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4       # north_star: "trajectories within 1e-4 rel-L2 of the SciPy reference"
FP64_TOL = 1e-6


def make_vec(N, device, **kw):
    from gym_PVDER_amd.vec_env import PVDERVec
    return PVDERVec(N, device=device, **kw)


def test_reset_observation_matches_oracle(gpu_device):
    env = make_vec(256, gpu_device)
    obs = env.reset().cpu().numpy().astype(np.float64)
    m = pm.build_model("50")
    y0 = pm.steady_state(m)
    o = pm.algebraic(m, y0, 0.0, 1.0, 100.0, 0.0, m.Vdc_ref0)
    want = np.array([o.i[0].real, o.i[0].imag, o.v[0].real, o.v[0].imag, o.S_PCC.real, o.S_PCC.imag, o.Vdc,
                     o.Ppv, m.Vdc_ref0, 0.0, 0.0])
    assert obs.shape == (256, 11)
    assert np.abs(obs - want[None, :]).max() < 2e-6
    assert int(env.istate[2].min()) == 1 and int(env.istate[2].max()) == 1       # episode counter


def test_device_event_generator_matches_oracle(gpu_device):
    N = 64
    env = make_vec(N, gpu_device, seed=2024, env_id_base=1000, events_spec=both_events_spec())
    env.reset()
    vg, si = counter_tables(N, 2024, episode=1, env_id_base=1000, spec=both_events_spec())
    assert np.array_equal(env.ev_vg.cpu().numpy(), vg.astype(np.float32))
    assert np.array_equal(env.ev_si.cpu().numpy(), si.astype(np.float32))
    assert (vg[1:] != 1.0).any() and (si[1:] != 100.0).any()


@pytest.mark.parametrize("n_sub,steps", [(4, 20), (15, 6)])
def test_rk4_f32_trajectory_vs_fp64_bridge(gpu_device, n_sub, steps):
    """fp32 HIP RK4 vs the fp64 C bridge running the SAME scheme (micro-steps, event tables)."""
    N, micro = 64, 4096
    rng = np.random.default_rng(7)
    actions = rng.integers(0, 5, size=(steps, N))
    spec = both_events_spec()
    spec["voltage"]["t_events_start"] = spec["insolation"]["t_events_start"] = 0.2   # events inside the run
    spec["voltage"]["t_events_step"] = 0.2
    env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, integrator="rk4", micro_steps=micro,
                   seed=11, events_spec=spec, auto_reset=False)
    env.reset()
    vg, si = env.ev_vg.cpu().numpy().astype(np.float64), env.ev_si.cpu().numpy().astype(np.float64)
    ref = BridgeVecEnv(N, n=n_sub, method="rk4", micro=micro)
    ref.reset(vg, si, ev_s0=env.ev_s0, ev_ds=env.ev_ds)
    got_obs, want_obs = [], []
    for t in range(steps):
        obs, rew, done, _ = env.step(torch.as_tensor(actions[t], device=gpu_device))
        o2, r2, d2 = ref.step(actions[t])
        got_obs.append(obs.cpu().numpy().astype(np.float64))
        want_obs.append(o2)
        # reward exact unless the oracle sits within 1e-6 of a threshold (discrete ladder)
        assert np.array_equal(done.cpu().numpy(), d2)
        assert_discrete_rewards(rew.cpu().numpy(), r2, o2, what="step %d" % t)
    got, want = np.stack(got_obs), np.stack(want_obs)            # [T, N, 11]
    err = rel_l2(got.transpose(1, 0, 2).reshape(N, -1), want.transpose(1, 0, 2).reshape(N, -1), axis=1)
    assert err.max() < FP32_TOL, "max rel-L2 over envs %.3e" % err.max()
    st = env.step_stats()
    assert st["n_nonfinite"] == 0 and st["n_bad_action"] == 0


def test_rk45_f64_trajectory_vs_bridge(gpu_device):
    N, n_sub, steps = 64, 4, 3
    rng = np.random.default_rng(3)
    actions = rng.integers(0, 5, size=(steps, N))
    env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, integrator="rk45", rtol=1e-8,
                   atol=1e-10, seed=5, auto_reset=False)
    env.reset()
    vg, si = env.ev_vg.cpu().numpy().astype(np.float64), env.ev_si.cpu().numpy().astype(np.float64)
    ref = BridgeVecEnv(N, n=n_sub, method="rk45", rtol=1e-8, atol=1e-10)
    ref.reset(vg, si, ev_s0=env.ev_s0, ev_ds=env.ev_ds)
    for t in range(steps):
        obs, rew, done, _ = env.step(torch.as_tensor(actions[t], device=gpu_device))
        o2, r2, d2 = ref.step(actions[t])
        err = rel_l2(obs.cpu().numpy(), o2, axis=1)
        assert err.max() < FP64_TOL, "step %d rel-L2 %.3e" % (t, err.max())
        assert np.array_equal(rew.cpu().numpy(), r2)


@pytest.mark.parametrize("n_sub,steps,micro,coarse,graded", [(4, 20, 32, 1, True), (15, 6, 32, 1, True), (15, 6, 8, 0, False),
                                                               (15, 6, 8, 1, True), (15, 6, 8, 2, False), (15, 6, 4, 1, False)])
def test_etdrk4_f32_trajectory_vs_converged_bridge(gpu_device, n_sub, steps, micro, coarse, graded):
    """fp32 ETDRK4 vs the converged fp64 solution (C bridge RK45, rtol 1e-8): the default schedule (a
    sim time step that starts at an input jump takes a first step of 1/(60*32) s and one over the
    remainder, every other one a single step), uniform 8 (coarse = 0), and other variants."""
    N = 64
    rng = np.random.default_rng(17)
    actions = rng.integers(0, 5, size=(steps, N))
    spec = both_events_spec()
    spec["voltage"]["t_events_start"] = spec["insolation"]["t_events_start"] = 0.2
    spec["voltage"]["t_events_step"] = 0.2
    env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, integrator="etdrk4", micro_steps=micro,
                   micro_coarse=coarse, graded=graded, seed=23, events_spec=spec, auto_reset=False)
    env.reset()
    vg, si = env.ev_vg.cpu().numpy().astype(np.float64), env.ev_si.cpu().numpy().astype(np.float64)
    ref = BridgeVecEnv(N, n=n_sub, method="rk45", rtol=1e-8, atol=1e-11)
    ref.reset(vg, si, ev_s0=env.ev_s0, ev_ds=env.ev_ds)
    got_obs, want_obs = [], []
    for t in range(steps):
        obs, rew, done, _ = env.step(torch.as_tensor(actions[t], device=gpu_device))
        o2, r2, d2 = ref.step(actions[t])
        got_obs.append(obs.cpu().numpy().astype(np.float64))
        want_obs.append(o2)
        assert np.array_equal(done.cpu().numpy(), d2)
        assert_discrete_rewards(rew.cpu().numpy(), r2, o2, what="step %d" % t)
    got, want = np.stack(got_obs), np.stack(want_obs)
    err = rel_l2(got.transpose(1, 0, 2).reshape(N, -1), want.transpose(1, 0, 2).reshape(N, -1), axis=1)
    assert err.max() < FP32_TOL, "max rel-L2 over envs %.3e" % err.max()
    assert env.step_stats()["n_nonfinite"] == 0


def test_duty_cycle_saturation_anti_windup(gpu_device):
    """State pushed past |m| = m_limit (low DC-link voltage): the conditional-integration
    anti-windup branches run on the device and ETDRK4 takes its steps on the switched operator sets there.

    With a duty cycle AT its limit the right-hand side is discontinuous (component-wise freezing by
    sign tests) and the trajectory slides/chatters along |m| = m_limit: the result then depends on
    the integrator at the 1e-2 level -- on the CPU, fp64 RK4 at h and h/2 differ by 1.7e-2 and RK45
    by 1.4e-1 (DESIGN.md "Saturation"); on the device two builds of the same kernel that round
    differently moved the worst env between 4e-2 and 5.3e-2.  So: agreement to 1e-1 all round -- each device
    integrator against fp64 RK4, and the two against each other."""
    N, n_sub, steps = 64, 4, 3
    vdc = np.linspace(0.975, 0.99, N)
    m = pm.build_model("50")
    envs = {}
    for integrator, micro in (("rk4", 4096), ("etdrk4", 32)):
        env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, integrator=integrator,
                       micro_steps=micro, seed=5, auto_reset=False)
        env.reset()
        # perturbed start: Vdc = Vdc_ref ~ 0.98 pu  ->  the loop drives |m| = |vt|/(Vdc/2) to ~1.02
        env.fstate[18].copy_(torch.as_tensor(vdc, dtype=env.fstate.dtype, device=gpu_device))
        env.fstate[24].copy_(torch.as_tensor(vdc, dtype=env.fstate.dtype, device=gpu_device))
        envs[integrator] = env
    ref = BridgeVecEnv(N, n=n_sub, method="rk4", micro=4096)
    e0 = envs["rk4"]
    ref.reset(e0.ev_vg.cpu().numpy().astype(np.float64), e0.ev_si.cpu().numpy().astype(np.float64),
              ev_s0=e0.ev_s0, ev_ds=e0.ev_ds)
    ref.Y[:, 18] = vdc
    ref.vdcref[:] = vdc
    m_peak = 0.0
    a = np.zeros(N, dtype=np.int64)
    for t in range(steps):
        o2, r2, d2 = ref.step(a)
        outs = {}
        for name, env in envs.items():
            obs, rew, done, _ = env.step(torch.as_tensor(a, device=gpu_device))
            outs[name] = obs.cpu().numpy().astype(np.float64)
            assert np.isfinite(outs[name]).all() and env.step_stats()["n_nonfinite"] == 0
            assert rel_l2(outs[name], o2, axis=1).max() < 1e-1, (name, t)
        # ETDRK4 takes the steps at the limit on the switched operator sets of the anti-windup modes, classic RK4
        # chatters through them 4 us at a time: both inside the spread of the integrators of this regime (this scenario
        # after ONE env step: fp64 RK4 at h vs h/2 2e-2, RK4 vs converged RK45 1.6e-1, device ETDRK4 vs RK4 4.4e-2)
        assert rel_l2(outs["etdrk4"], outs["rk4"], axis=1).max() < 1e-1, t
        m_peak = max(m_peak, np.abs(m.Kp_GCC * (ref.Y[:, 4] + 1j * ref.Y[:, 5]) + ref.Y[:, 2] + 1j * ref.Y[:, 3]).max())
    assert m_peak > m.m_limit, "the scenario must drive |m| past the limit (got %.4f)" % m_peak
    # on the device too the duty cycle sits at/above the limit (the freeze logic is active)
    st = envs["rk4"].fstate[:23].t().cpu().numpy().astype(np.float64)
    assert np.abs(m.Kp_GCC * (st[:, 4] + 1j * st[:, 5]) + st[:, 2] + 1j * st[:, 3]).max() > m.m_limit


def test_mixed_batch_deferral_to_23_state_path(gpu_device):
    """Half of the envs start unbalanced / near saturation: the balanced fast path must hand exactly
    those to the 23-state path, the others keep the fast path; both halves must match a run that
    sends every env down the 23-state path, and the wave-reduced statistics must cover every env once."""
    N, n_sub, steps = 512, 4, 4
    runs = {}
    for knob in (0, 1):                       # 0: fast path + 23-state tail for deferred envs, 1: every env through the tail
        env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, integrator="etdrk4", seed=41,
                       auto_reset=True, max_sim_time=1.0)
        env._debug_knob = knob
        env.reset()
        # every second env: unbalanced current (phase b scaled) -> must be deferred
        env.fstate[6, 1::2] *= 1.001
        # every fourth env: low DC-link voltage -> duty cycle driven to its limit
        env.fstate[18, 3::4] = 0.985
        env.fstate[24, 3::4] = 0.985
        gen = torch.Generator(device=gpu_device); gen.manual_seed(3)
        outs = []
        for t in range(steps):
            a = torch.randint(0, 5, (N,), generator=gen, device=gpu_device, dtype=torch.uint8)
            obs, rew, done, _ = env.step(a)
            st = env.step_stats()
            assert st["reward_sum"] == float(rew.sum()) and st["n_done"] == int(done.sum()) and st["n_nonfinite"] == 0
            assert int(env.istate[3].min()) == 0 and int(env.istate[3].max()) == 0     # no kDeferred marker left behind
            outs.append(obs.clone())
        runs[knob] = torch.stack(outs).cpu().numpy().astype(np.float64)
        assert int(env.steps.min()) == steps and int(env.steps.max()) == steps
    fast, full = runs[0], runs[1]
    # step 1: the perturbed envs were deferred and ran the very same kernel in both runs: bit-identical
    assert np.array_equal(fast[0, 1::2], full[0, 1::2])
    # afterwards the current controller has re-balanced the merely unbalanced ones (they return to
    # the fast path); everything stays within the same-scheme tolerance of the 23-state path,
    # the saturated ones (chattering regime, see test_duty_cycle_saturation_anti_windup) loosely
    sat = np.zeros(N, dtype=bool); sat[3::4] = True
    assert rel_l2(fast[:, ~sat].reshape(-1, 11), full[:, ~sat].reshape(-1, 11)) < 1e-5
    assert rel_l2(fast[:, sat].reshape(-1, 11), full[:, sat].reshape(-1, 11)) < 2e-3


def test_current_limit_defers_to_23_state_path(gpu_device):
    """A reactive-power reference far beyond the inverter rating drives |iref| past iref_limit, where
    the conditional-integration anti-windup of the outer PI loops switches on.  The balanced fast
    path carries no anti-windup logic: it must hand exactly those envs to the 23-state path (so they
    come out bit-identical to a run of the 23-state kernel alone) and both must follow the converged
    fp64 solution of the oracle, which has the same logic (oracle/pvder_model.py rhs)."""
    N, n_sub, steps = 128, 4, 6
    m = pm.build_model("50")
    qref = np.zeros(N)
    qref[::2] = -8.0                              # every second env: iref_I = xQ - Kp_Q (Q_ref - Q) ~ 1.7 > limit
    runs = {}
    for knob in (0, 1, "rk4"):
        env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, seed=77, auto_reset=False,
                       integrator="rk4" if knob == "rk4" else "etdrk4")
        env._debug_knob = 0 if knob == "rk4" else knob
        env.reset()
        env.fstate[23].copy_(torch.as_tensor(qref, dtype=env.fstate.dtype, device=gpu_device))
        a = torch.zeros(N, dtype=torch.uint8, device=gpu_device)
        outs = []
        for t in range(steps):
            obs, rew, done, _ = env.step(a)
            outs.append(obs.cpu().numpy().astype(np.float64))
            assert env.step_stats()["n_nonfinite"] == 0
        runs[knob] = np.stack(outs)
        vg, si = env.ev_vg.cpu().numpy().astype(np.float64), env.ev_si.cpu().numpy().astype(np.float64)
        ev = (env.ev_s0, env.ev_ds)
    want = {}
    over = 0.0
    for name, kw in (("rk45", dict(method="rk45", rtol=1e-8, atol=1e-11)), ("rk4", dict(method="rk4", micro=4096))):
        ref = BridgeVecEnv(N, n=n_sub, **kw)
        ref.reset(vg, si, ev_s0=ev[0], ev_ds=ev[1])
        ref.qref[:] = qref
        outs = []
        for t in range(steps):
            o2, _, _ = ref.step(np.zeros(N, dtype=np.int64))
            outs.append(o2)
            Y = ref.Y[::2]                      # iref = xDC + Kp_DC (Vdc_ref - Vdc) + j (xQ - Kp_Q (Q_ref - Q))
            over = max(over, float(np.hypot(Y[:, 19] + m.Kp_DC * (ref.vdcref[::2] - Y[:, 18]),
                                            Y[:, 20] - m.Kp_Q * (qref[::2] - o2[::2, 5])).max()))
        want[name] = np.stack(outs)
    assert over > m.iref_limit, "scenario must exceed the current limit (|iref| %.3f of %.3f)" % (over, m.iref_limit)
    fast, full = runs[0], runs[1]
    assert np.array_equal(fast[:, ::2], full[:, ::2])                       # deferred: the very same code
    assert rel_l2(fast[:, 1::2].reshape(-1, 11), full[:, 1::2].reshape(-1, 11)) < 1e-5

    def env_err(a, b):
        return rel_l2(a.transpose(1, 0, 2).reshape(N, -1), b.transpose(1, 0, 2).reshape(N, -1), axis=1)
    # With the limit active the right-hand side is discontinuous (sign tests freeze the PI integrators)
    # and the result depends on the integrator at the 1e-3 level -- fp64 on the CPU: RK4 at h and h/2
    # differ by 2.8e-4, RK4 and RK45 by 6.9e-4.  So: the device logic against the SAME scheme in fp64 ...
    e4 = env_err(runs["rk4"], want["rk4"])
    assert e4.max() < FP32_TOL, "classic RK4 vs fp64 classic RK4: %.3e" % e4.max()
    # ... and ETDRK4 against the converged RK45 at that level (un-limited envs: full accuracy).  This
    # 1000-fold action (dQ_ref = -8 pu; an env step moves Q_ref by 0.0075 pu) is far outside PVDER-v0.
    err = env_err(fast, want["rk45"])
    assert err[1::2].max() < FP32_TOL and err[::2].max() < 3e-3, "max rel-L2 over envs %.3e / %.3e" % (
        err[1::2].max(), err[::2].max())


def test_full_episode_drift_vs_converged_bridge(gpu_device):
    """A complete PVDER-v0 episode (160 env steps = 40 s, 38 voltage/insolation events, random
    actions): the fp32 ETDRK4 kernel must not drift away from the converged fp64 solution (the committed
    golden tests/golden/bridge_episode_random.npz, shared with the etdlin test)."""
    import bridge_cases
    g = bridge_cases.load("episode_random")
    c = bridge_cases.CASES["episode_random"]
    N, steps, actions = c["N"], c["steps"], g["actions"]
    env = make_vec(N, gpu_device, integrator="etdrk4", seed=c["seed"], events_spec=both_events_spec(), auto_reset=False)
    env.reset()
    assert np.array_equal(env.ev_vg.cpu().numpy(), g["vgrid"].astype(np.float32)) and \
        np.array_equal(env.ev_si.cpu().numpy(), g["sinsol"].astype(np.float32))
    want, want_r, want_d = g["obs"], g["reward"], g["done"]
    got, rmis = [], 0
    for t in range(steps):
        obs, rew, done, _ = env.step(torch.as_tensor(actions[t], device=gpu_device))
        got.append(obs.cpu().numpy().astype(np.float64))
        rmis += assert_discrete_rewards(rew.cpu().numpy(), want_r[t], want[t], what="step %d" % t)
        assert np.array_equal(done.cpu().numpy(), want_d[t])
    assert bool(done.all()) and want_d[-1].all()         # 160 * 0.25 s = max_sim_time
    got = np.stack(got)
    err = rel_l2(got.transpose(1, 0, 2).reshape(N, -1), want.transpose(1, 0, 2).reshape(N, -1), axis=1)
    assert err.max() < FP32_TOL, "whole-episode rel-L2 %.3e" % err.max()
    last = rel_l2(got[-20:].transpose(1, 0, 2).reshape(N, -1), want[-20:].transpose(1, 0, 2).reshape(N, -1), axis=1)
    assert last.max() < FP32_TOL, "last 5 s rel-L2 %.3e (drift)" % last.max()
    assert rmis <= 0.01 * N * steps


def test_step_is_graph_capturable(gpu_device):
    """The C entry points are asynchronous, allocation-free and never synchronise, so a step can be
    captured into a HIP graph (torch.cuda.CUDAGraph) and replayed: bit-identical to eager launches."""
    N, steps = 1024, 6
    gen = torch.Generator(device=gpu_device); gen.manual_seed(9)
    acts = torch.randint(0, 5, (steps, N), generator=gen, device=gpu_device, dtype=torch.uint8)
    eager = make_vec(N, gpu_device, seed=12, events_spec=both_events_spec(), auto_reset=True)
    eager.reset()
    want = []
    for t in range(steps):
        eager.step(acts[t], return_tensors=False)
        want.append(eager.out.clone())
    env = make_vec(N, gpu_device, seed=12, events_spec=both_events_spec(), auto_reset=True)
    env.reset()
    a0, a1 = acts[0].clone(), acts[1].clone()          # static inputs of the captured launches
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    state = env.state_dict()
    with torch.cuda.graph(g):                          # two steps per graph: the statistics banks ping-pong
        env.step(a0, return_tensors=False)
        o0 = env.out.clone()
        env.step(a1, return_tensors=False)
        o1 = env.out.clone()
    env.load_state_dict(state)                         # capture does not execute; start from the same state anyway
    got = []
    for t in range(0, steps, 2):
        a0.copy_(acts[t]); a1.copy_(acts[t + 1])
        g.replay()
        got += [o0.clone(), o1.clone()]
    torch.cuda.synchronize()
    for t in range(steps):
        assert torch.equal(got[t], want[t]), t


def test_direct_rccl_all_gather_world1(gpu_device):
    """RcclAllGather (ctypes on torch's librccl.so, own communicator and stream) at world size 1:
    the plumbing bench.py uses for N > 1 -- unique id through the process group, communicator,
    launch ordered behind the producer, completion event -- must reproduce the block."""
    import os
    import torch.distributed as dist
    from gym_PVDER_amd.distributed import RcclAllGather
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        env = make_vec(4096, gpu_device, seed=3, auto_reset=True)
        env.reset()
        g = RcclAllGather(gpu_device)
        outs = [env.out, torch.empty_like(env.out)]
        recv = [torch.zeros((13, 4096), device=gpu_device) for _ in range(2)]
        done = [None, None]
        want = []
        gen = torch.Generator(device=gpu_device); gen.manual_seed(5)
        for t in range(6):
            b = t & 1
            if done[b] is not None:
                torch.cuda.current_stream().wait_event(done[b])
            env.use_out(outs[b])
            env.step(torch.randint(0, 5, (4096,), generator=gen, device=gpu_device, dtype=torch.uint8), return_tensors=False)
            done[b] = g.launch(outs[b], recv[b])
            done[b].synchronize()
            assert torch.equal(recv[b], outs[b]), t
            want.append(outs[b].clone())
        assert not torch.equal(want[-1], want[-2])
        with pytest.raises(ValueError):
            env.use_out(torch.empty((13, 8), device=gpu_device))
        g.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_off_range_policy(gpu_device):
    """An env driven out of the range the fast integrators cover (Vdc_ref walked down 6 %: the duty cycle reaches
    its limit and the reference model's anti-windup switches the stiff dynamics).  Default ('integrate'): it is
    integrated through, like the reference; 'terminate': its episode ends there (done, reward -100, counted), the
    other envs of the batch never notice, and with auto-reset it starts over."""
    import time
    N, n_sub = 128, 15
    down = torch.full((N,), 4, dtype=torch.uint8, device=gpu_device)
    down[64:] = 0                                          # envs 64.. hold their references: they stay in range
    # (a) terminate
    env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, seed=2, auto_reset=False, off_range="terminate")
    env.reset()
    done_at = torch.full((N,), -1, dtype=torch.long, device=gpu_device)
    rew_at = torch.zeros(N, dtype=torch.float32, device=gpu_device)
    for t in range(150):
        obs, rew, done, _ = env.step(down)
        new = done & (done_at < 0)
        done_at = torch.where(new, torch.full_like(done_at, t), done_at)
        rew_at = torch.where(new, rew, rew_at)
    # each env on its own voltage events: they reach the limit -- a duty cycle that would cross it within a step --
    # between Vdc_ref -6 % and -9 % (an env whose grid voltage sits at 0.98 pu may get through the 150 steps)
    hit = done_at[:64] >= 0
    assert int(hit.sum()) >= 56 and bool((done_at[:64][hit] >= 60).all()), done_at[:64]
    assert bool((rew_at[:64][hit] == -100.0).all()) and bool((done_at[64:] < 0).all())
    first_done = int(done_at[0])                          # env 0 (global id 0, seed 2: the env of part (b))
    assert first_done >= 60
    assert bool((env.istate[3, :64][hit] == 2).all()) and bool((env.istate[3, 64:] == 0).all())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        env.step(down, return_tensors=False)
    torch.cuda.synchronize()
    assert (time.perf_counter() - t0) / 20 < 2e-3                                 # no 50 ms stalls
    # with auto-reset the terminated envs start over (episode counter) and the batch keeps its pace
    env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, seed=2, auto_reset=True, off_range="terminate")
    env.reset()
    for t in range(150):
        env.step(down, return_tensors=False)
    assert int((env.istate[2, :64] >= 2).sum()) >= 56 and bool((env.istate[2, 64:] == 1).all())
    # (b) integrate (default): the same drive is integrated through -- round 6: by the hybrid kernel of the duty-cycle limit
    # (step_sat_hybrid_kernel: event-driven, fp64; accurate and SLOW: ~0.1 s per batched step with 64 envs at the limit, a wave's
    # lanes each on their own events) -- or, "held_modes", on rounds 3-5's held operator sets (a few hundred microseconds per step).
    # Accuracy up to and behaviour beyond the limit: test_fixed_action_ramps_vs_converged_bridge[4]
    for mode, per_step in (("integrate", 0.5), ("held_modes", 1e-3)):
        env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, seed=2, auto_reset=False, off_range=mode)
        env.reset()
        for t in range(150):
            obs, rew, done, _ = env.step(down)
        assert not bool(done.any()) and env.step_stats()["n_nonfinite"] == 0 and bool(torch.isfinite(env.fstate).all())
        f = env.fstate
        m_abs = (torch.complex(f[2], f[3]) + 6000.0 * torch.complex(f[4], f[5])).abs()
        assert float(m_abs[:64].max()) > 0.999 and float(m_abs[:64].max()) < 1.03 and float(m_abs[64:].max()) < 0.96
        assert bool(env._lib.pvder_plan_hybrid_active(env._plan)) == (mode == "integrate")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            env.step(down, return_tensors=False)
        torch.cuda.synchronize()
        assert (time.perf_counter() - t0) / 8 < per_step, (mode, (time.perf_counter() - t0) / 8)


@pytest.mark.parametrize("integrator", ["etdrk4", "etdlin"])
def test_23_state_path_and_etdrk4_schedule_are_per_lane(gpu_device, integrator):
    """VERDICT r2 #4 / ADVICE r1: the ETDRK4 schedule used to take the fine rate for a whole wave if ANY of its envs had
    an input jump in a sim time step, so an env off the event grid -- or deferred to the 23-state path -- depended on
    its wave neighbours.  Every second lane is reset mid-episode (off the event grid: neighbours then meet their events
    at different sim time steps) AND every fourth env is knocked off balance (23-state path); every env must come out
    bit-identical to the same env stepped among lanes that all share its schedule."""
    n_sub, N, T, t_reset = 15, 128, 10, 4
    spec = both_events_spec()
    spec["voltage"]["t_events_start"] = spec["insolation"]["t_events_start"] = 0.2
    spec["voltage"]["t_events_step"] = 0.2          # events every 12 sim time steps, env step = 15
    rng = np.random.default_rng(6)
    acts = torch.as_tensor(rng.integers(0, 5, size=(T, N)), dtype=torch.uint8, device=gpu_device)
    idx = torch.arange(N, device=gpu_device)
    knock = (idx % 4) >= 2                          # lanes 2, 3 of every four: both reset parities are hit

    def run(src, reset_mask):
        env = make_vec(N, gpu_device, n_sim_time_steps_per_env_step=n_sub, seed=9, events_spec=spec, auto_reset=False,
                       integrator=integrator)
        env.reset()
        env.set_event_tables(env.ev_vg[:, src].contiguous(), env.ev_si[:, src].contiguous())
        outs = []
        for t in range(T):
            if t == t_reset:
                env.reset(reset_mask)
            if t in (2, t_reset + 2):
                env.fstate[6, knock[src]] *= 1.002   # phase b current: off balance -> the 23-state path takes these envs
            env.step(acts[t][src].contiguous(), return_tensors=False)
            outs.append(env.out.clone())
        return torch.stack(outs)

    mask = torch.zeros(N, dtype=torch.uint8, device=gpu_device)
    mask[::2] = 1
    mixed = run(idx, mask)
    for parity in (0, 1):
        for k in (0, 1):                            # lanes (parity, knocked or not): four classes, each among its own kind
            lane = 2 * k + parity
            src = (idx // 4) * 4 + lane
            alike = run(src, torch.full((N,), 1 - parity, dtype=torch.uint8, device=gpu_device))
            assert torch.equal(alike[:, :, lane::4], mixed[:, :, lane::4]), (parity, k)
    assert not torch.equal(mixed[-1][:, 0::4], mixed[-1][:, 1::4])

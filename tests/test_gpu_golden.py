"""-m gpu: HIP kernels against the committed golden trajectories (tests/golden/*.npz), the
single-env API contract of the reference's own tests, and size-independent properties at
BASELINE.json's full size (65 536 envs)."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import assert_discrete_rewards, both_events_spec, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(f)[5:-4] for f in glob.glob(os.path.join(GOLD, "traj_*.npz")))

FP32_TOL = 1e-4      # north_star tolerance (fp64 -> fp32): rel-L2 of the observation trajectory
FP64_TOL = 2e-6


def run_case(name, device, integrator, copies=64, **kw):
    from gym_PVDER_amd.vec_env import PVDERVec
    g = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    n, goal, discrete = int(g["n"]), str(g["goal"]), bool(g["discrete"])
    env = PVDERVec(copies, device=device, n_sim_time_steps_per_env_step=n, goals_list=[goal],
                   DISCRETE_REWARD=discrete, integrator=integrator, auto_reset=False, **kw)
    K1 = env.ev_vg.shape[0]
    assert K1 == len(g["vgrid"])
    env.set_event_tables(torch.as_tensor(g["vgrid"], dtype=torch.float32, device=device)[:, None].expand(K1, copies).contiguous(),
                         torch.as_tensor(g["sinsol"], dtype=torch.float32, device=device)[:, None].expand(K1, copies).contiguous())
    obs = [env.reset().cpu().numpy().astype(np.float64)]
    rew, done = [], []
    for a in g["actions"]:
        o, r, d, _ = env.step(torch.full((copies,), int(a), dtype=torch.uint8, device=device))
        obs.append(o.cpu().numpy().astype(np.float64)); rew.append(r.cpu().numpy()); done.append(d.cpu().numpy())
    return g, np.stack(obs), np.stack(rew), np.stack(done), env


@pytest.mark.parametrize("integrator", ["etdrk4", "rk4"])
@pytest.mark.parametrize("name", CASES)
def test_f32_kernels_vs_golden(gpu_device, name, integrator):
    g, obs, rew, done, env = run_case(name, gpu_device, integrator)
    assert np.array_equal(obs[:, :1].repeat(obs.shape[1], 1), obs)      # identical envs -> identical outputs
    err = rel_l2(obs[:, 0], g["obs"])
    assert err < FP32_TOL, "rel-L2 %.3e" % err
    assert np.array_equal(done[:, 0], g["done"])
    if bool(g["discrete"]):
        # exact except where the golden sits on a reward threshold -- asserted
        assert_discrete_rewards(rew[:, 0], g["reward"], g["obs"][1:], goal=str(g["goal"]), what=name)
    else:
        assert np.abs(rew[:, 0] - g["reward"]).max() < 1e-6


@pytest.mark.parametrize("name", ["random_both_n15", "vdcalt_both_n15"])
def test_classic_rk4_at_3750_micro_steps_vs_golden(gpu_device, name):
    """The literal "fixed-step HIP RK4" at the step count bench.py's classic_rk4 leg runs: 3750 per 1/60 s, h = 4.44 us,
    |h lambda|max = 2.82 at the nominal operating point against the scheme's bound of 2.95 along that eigen-direction (CPU:
    the oracle's Jacobian; the fast poles go like sqrt(Vdc): 2.94 at Vdc_ref +9 %, which is why the DEFAULT stays 4096)."""
    g, obs, rew, done, env = run_case(name, gpu_device, "rk4", micro_steps=3750)
    err = rel_l2(obs[:, 0], g["obs"])
    assert err < FP32_TOL, "rel-L2 %.3e" % err
    assert np.array_equal(done[:, 0], g["done"]) and env.step_stats()["n_nonfinite"] == 0


@pytest.mark.parametrize("name", ["random_both_n4", "qup_volt_n15"])
def test_rk45_f64_vs_golden(gpu_device, name):
    g, obs, rew, done, env = run_case(name, gpu_device, "rk45", copies=64, rtol=1e-8, atol=1e-11)
    err = rel_l2(obs[:, 0], g["obs"])
    assert err < FP64_TOL, "rel-L2 %.3e" % err
    assert np.array_equal(rew[:, 0], g["reward"]) and np.array_equal(done[:, 0], g["done"])


# ---- the reference's own tests (gym_PVDER/tests/test_gym_PVDER.py), on the HIP-backed single env
def test_make_and_env_contract(gpu_device):
    import gym_PVDER_amd as gp
    from gym_PVDER_amd.envs import PVDER
    env = gp.make('PVDER-v0')
    assert env.spec.id == 'PVDER-v0' and isinstance(env.unwrapped, PVDER)          # test :11-14
    ob = env.reset()
    assert env.observation_space.contains(ob.astype(np.float32)) and ob.shape == (11,)
    a = env.action_space.sample()
    ob, reward, done, info = env.step(a)
    assert env.observation_space.contains(ob.astype(np.float32))
    assert np.isscalar(reward) and isinstance(done, bool) and info == {}            # test :34-35
    for mode in env.metadata.get('render.modes', []):
        env.render(mode=mode)
    assert isinstance(env.env.sim.PV_model.Q_ref, float)                             # examples/...import_test.py:38
    with pytest.raises(AssertionError):
        env.step(7)                                                                  # PVDER_env.py:201
    env.close()


def test_render_and_recorded_trajectory(gpu_device, capsys):
    """render('vector') prints the reference's eleven fields in physical units (PVDER_env.py:339-353),
    render('human') returns the six figures of :358-364 drawn from the on-device trajectory ring
    buffer; the recorded quantities equal the oracle's algebraic outputs of the recorded states."""
    import gym_PVDER_amd as gp
    from gym_PVDER_amd import der_model
    from oracle import pvder_model as pm
    env = gp.make('PVDER-v0')
    env.reset()
    acts = [1, 3, 0, 2, 4, 1]
    for a in acts:
        env.step(a)
    line = env.render(mode='vector')
    out = capsys.readouterr().out
    for key in ('ma:', 'ia:', 'Vdc:', 'vta:', 'va:', 'Ppv:', 'S_inverter:', 'S_PCC:', 'Q_ref:', 'Vdc_ref:', 'tStart:'):
        assert key in line and key in out
    assert 'Reward:' in out and 'Time for step:' in out
    figs = env.render(mode='human')
    assert len(figs) == 6 and all(len(f.axes[0].lines) >= 2 for f in figs)
    t = env._vec.trajectory(0)
    assert len(t['t']) == len(acts) + 1 and t['t'][0] == 0.0
    assert np.allclose(np.diff(t['t']), env.n_sim_time_steps_per_env_step / 60.0)
    # oracle check of the last snapshot: same state -> same algebraic outputs (both fp64 on the host)
    m = pm.build_model("50")
    v = env._vec
    y = v.fstate[:23, 0].to(torch.float64).cpu().numpy()
    o = pm.algebraic(m, y, 0.0, float(t['vgrid'][-1]), 100.0, float(v.Q_ref[0]), float(v.Vdc_ref[0]))
    assert abs(t['ma'][-1] - o.m[0]) < 1e-12 and abs(t['mc'][-1] - o.m[2]) < 1e-12
    assert abs(t['vta'][-1] / der_model.VBASE - o.vt[0]) < 1e-12
    assert abs(t['vb'][-1] / der_model.VBASE - o.v[1]) < 1e-12
    assert abs(t['S_inverter'][-1] / der_model.SBASE - o.S) < 1e-12
    assert abs(t['S_PCC'][-1] / der_model.SBASE - o.S_PCC) < 1e-12
    assert abs(t['Vdc'][-1] - y[18] * der_model.VDCBASE) < 1e-9
    assert abs(t['ia'][-1] - (y[0] + 1j * y[1]) * der_model.IBASE) < 1e-9
    # the PCC quantities of the record agree with the kernel's own observation (obs rows 2-5, pu)
    ob = v.out[:11, 0].to(torch.float64).cpu().numpy()
    assert abs(t['va'][-1] / der_model.VBASE - (ob[2] + 1j * ob[3])) < 5e-6
    assert abs(t['S_PCC'][-1] / der_model.SBASE - (ob[4] + 1j * ob[5])) < 5e-6
    assert abs(t['Ppv'][-1] / der_model.SBASE - ob[7]) < 1e-12
    env.close()


def test_reward_python_types(gpu_device):
    import gym_PVDER_amd as gp
    env = gp.make('PVDER-v0', DISCRETE_REWARD=True, goals_list=['voltage_regulation'])
    env.reset()
    for _ in range(3):
        _, r, _, _ = env.step(env.action_space.sample())
        assert isinstance(r, int)                                                    # test :66
    env = gp.make('PVDER-v0', DISCRETE_REWARD=False, goals_list=['voltage_regulation'])
    env.reset()
    for _ in range(3):
        _, r, _, _ = env.step(env.action_space.sample())
        assert isinstance(r, float)                                                  # test :81


def test_time_steps_known_answer(gpu_device):
    """test_gym_PVDER.py:87-111 -- n=10, max_sim_time=25: exactly 150 steps, last obs[10] == 1."""
    import gym_PVDER_amd as gp
    env = gp.make('PVDER-v0', n_sim_time_steps_per_env_step=10, max_sim_time=25.0)
    ob = env.reset()
    assert env.n_sim_time_steps_per_env_step == 10 and env.max_sim_time == 25.0
    done, steps = False, 0
    while not done:
        ob, r, done, _ = env.step(env.action_space.sample())
        steps += 1
        assert steps <= 500
    assert steps == 150
    assert round(env.steps * env.n_sim_time_steps_per_env_step * env.sim.tInc, 6) == round(env.max_sim_time, 6)
    assert round(ob[-1] * env.max_sim_time, 6) == round(env.max_sim_time, 6)
    ob2, r2, d2, _ = env.step(0)
    assert d2 and np.array_equal(ob2, ob) and r2 == r                                # stale tuple (:145-152)


def test_update_env_events(gpu_device):
    """test_gym_PVDER.py:114-133."""
    import gym_PVDER_amd as gp
    env = gp.make('PVDER-v0')
    env.update_env_events(event_spec_list=[{'voltage': {'min': 0.95}}])
    env.reset()
    assert env.env_events_spec['voltage']['min'] == 0.95
    assert env.sim.simulation_events._events_spec['voltage']['min'] == 0.95
    vg = env._vec.ev_vg.cpu().numpy()
    assert vg[1:].min() < 0.98 and vg.min() >= 0.95
    # with a simulation attached the events are redrawn at once (PVDER_env.py:435-438), state untouched
    ob1, st1 = env._vec.out.clone(), env._vec.fstate.clone()
    env.update_env_events([{'voltage': {'min': 0.9, 'max': 0.91}}])
    vg2 = env._vec.ev_vg.cpu().numpy()
    assert vg2[1:].max() <= 0.91 + 1e-6 and vg2[1:].min() >= 0.9 - 1e-6
    assert torch.equal(env._vec.out, ob1) and torch.equal(env._vec.fstate, st1) and env.steps == 0
    with pytest.raises(ValueError):
        env.update_env_events([{'voltage': {'bogus': 1}}])
    with pytest.raises(ValueError):
        env.update_env_events([{'wind': {'min': 1}}])


# ---- full-size properties (BASELINE configs[2]: 65 536 envs, events on)
def test_full_size_properties(gpu_device):
    from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec
    spec = default_events_spec()
    spec['insolation']['ENABLE'] = True
    spec['voltage']['t_events_start'] = spec['insolation']['t_events_start'] = 0.1
    N, n_sub = 65536, 6
    kw = dict(n_sim_time_steps_per_env_step=n_sub, seed=99, events_spec=spec, max_sim_time=1.0)
    full = PVDERVec(N, device=gpu_device, **kw)
    half = [PVDERVec(N // 2, device=gpu_device, env_id_base=b, **kw) for b in (0, N // 2)]
    full.reset(); [h.reset() for h in half]
    gen = torch.Generator(device=gpu_device); gen.manual_seed(5)
    n_done = 0
    for t in range(12):
        a = torch.randint(0, 5, (N,), generator=gen, device=gpu_device, dtype=torch.uint8)
        obs, rew, done, _ = full.step(a)
        outs = [h.step(a[i * N // 2:(i + 1) * N // 2].contiguous()) for i, h in enumerate(half)]
        # sharding invariance: two shards keyed by global env id == the unsharded batch, bit for bit
        assert torch.equal(obs, torch.cat([o[0] for o in outs]))
        assert torch.equal(rew, torch.cat([o[1] for o in outs])) and torch.equal(done, torch.cat([o[2] for o in outs]))
        st = full.step_stats()
        # wave-level reduction == plain reduction (integers: exact in fp32)
        assert st["reward_sum"] == float(rew.sum()) and st["n_done"] == int(done.sum())
        assert st["n_nonfinite"] == 0 and st["n_bad_action"] == 0
        assert bool(torch.isfinite(obs).all()) and float(obs.abs().max()) < 10.0     # Box(-10, 10)
        n_done += int(done.sum())
    # max_sim_time = 1 s = 60 sim steps = 10 env steps: every env finished exactly once and was auto-reset
    assert n_done == N
    assert int(full.istate[2].min()) == 2 and int(full.istate[2].max()) == 2         # episode counters
    assert int(full.steps.max()) == 2
    # envs differ (independent event streams) ...
    assert float(obs[:, 2].std()) > 0
    # ... and an invalid action is counted, not executed
    a = torch.zeros(N, dtype=torch.uint8, device=gpu_device); a[:7] = 9
    full.step(a)
    assert full.step_stats()["n_bad_action"] == 7


# ---- single-phase model_1 (derId "10", PVDER_env.py:56, config_der.json:66-84)
def test_single_phase_model_vs_bridge(gpu_device):
    from gym_PVDER_amd.vec_env import PVDERVec
    from oracle.bridge_env import BridgeVecEnv
    g1 = np.load(os.path.join(GOLD, "steady_state.npz"))
    N, n_sub, steps = 64, 4, 4
    rng = np.random.default_rng(21)
    actions = rng.integers(0, 5, size=(steps, N))
    for integrator, kw, tol in (("etdrk4", {}, FP32_TOL), ("rk4", {}, FP32_TOL),
                                ("rk45", dict(rtol=1e-8, atol=1e-11), FP64_TOL)):
        env = PVDERVec(N, device=gpu_device, model_type='model_1', integrator=integrator,
                       n_sim_time_steps_per_env_step=n_sub, seed=8, auto_reset=False,
                       goals_list=['power_regulation'], **kw)
        assert env.phases == 1 and env.ns == 11 and env.fstate.shape == (13, N)
        ob0 = env.reset().cpu().numpy().astype(np.float64)
        assert np.abs(ob0 - g1["obs0_10"][None, :]).max() < 2e-6
        ref = BridgeVecEnv(N, n=n_sub, der_id="10", goal="power_regulation",
                           **(dict(method="rk4", micro=env.micro_steps) if integrator == "rk4"
                              else dict(method="rk45", rtol=1e-8, atol=1e-11)))      # etdrk4: vs converged
        ref.reset(env.ev_vg.cpu().numpy().astype(np.float64), env.ev_si.cpu().numpy().astype(np.float64),
                  ev_s0=env.ev_s0, ev_ds=env.ev_ds)
        got, want = [], []
        for t in range(steps):
            obs, rew, done, _ = env.step(torch.as_tensor(actions[t], device=gpu_device))
            o2, r2, d2 = ref.step(actions[t])
            got.append(obs.cpu().numpy().astype(np.float64)); want.append(o2)
            assert np.array_equal(done.cpu().numpy(), d2)
            # (single phase: Vrms is not in the observation's |v_a| / sqrt 2 form only for three phases; P_PCC is what counts here)
            assert_discrete_rewards(rew.cpu().numpy(), r2, o2, goal="power_regulation", der_id="10", what=integrator)
        got, want = np.stack(got), np.stack(want)
        err = rel_l2(got.transpose(1, 0, 2).reshape(N, -1), want.transpose(1, 0, 2).reshape(N, -1), axis=1)
        assert err.max() < tol, (integrator, err.max())


# ---- calc_returns sweep (PVDER_env.py:458-497) on the batched env
def test_calc_returns_sweep(gpu_device):
    from gym_PVDER_amd.evaluate import calc_returns
    res = calc_returns(episodes=8, device=gpu_device, goals=['voltage_regulation', 'Q_regulation'],
                       n_sim_time_steps_per_env_step=60, max_sim_time=5.0)
    assert set(res) == {'voltage_regulation', 'Q_regulation'}
    for goal in res:
        assert set(res[goal]) == {'random', 'inc', 'dec', 'no_change'}
        for spec in res[goal].values():
            assert -5 * 5 <= spec['return'] <= 5                 # 5 one-second steps, rewards in {-5,-1,+1}
    # 'inc' is a no-op: the references stay at their reset values (550 V, 0 VAR)
    assert res['voltage_regulation']['inc']['ref'] == pytest.approx([550.0, 0.0], abs=1e-3)
    # 'dec' raises Q_ref by 25*60 VAR per step (PVDER_env.py:75, :617)
    assert res['voltage_regulation']['dec']['ref'][1] == pytest.approx(5 * 25 * 60, rel=1e-5)
    assert res['voltage_regulation']['no_change']['ref'][1] == pytest.approx(-5 * 25 * 60, rel=1e-5)


def test_ragged_env_counts(gpu_device):
    """n_env not a multiple of the wave / workgroup size: tail lanes must neither store nor disturb."""
    from gym_PVDER_amd.vec_env import PVDERVec
    outs = {}
    for N in (1, 63, 65, 1000, 1024):
        env = PVDERVec(N, device=gpu_device, n_sim_time_steps_per_env_step=4, seed=31)
        env.reset()
        gen = torch.Generator(device=gpu_device); gen.manual_seed(1)
        a = torch.randint(0, 5, (3, 1024), generator=gen, device=gpu_device, dtype=torch.uint8)
        for t in range(3):
            obs, rew, done, _ = env.step(a[t, :N].contiguous())
        outs[N] = (obs.clone(), rew.clone(), env.step_stats())
        assert outs[N][2]["reward_sum"] == float(rew.sum()) and outs[N][2]["n_bad_action"] == 0
    for N in (1, 63, 65, 1000):
        assert torch.equal(outs[N][0], outs[1024][0][:N]) and torch.equal(outs[N][1], outs[1024][1][:N])


@pytest.mark.parametrize("integrator", ["etdrk4", "rk4", "rk45"])
def test_same_step_auto_reset_every_integrator(gpu_device, integrator):
    """max_sim_time = 1 s, n = 6: every env finishes at step 10, is reset in the same step call
    (obs = reset observation, reward/done = the ended transition) and runs on with new events."""
    from gym_PVDER_amd.vec_env import PVDERVec
    N = 128
    env = PVDERVec(N, device=gpu_device, integrator=integrator, n_sim_time_steps_per_env_step=6,
                   max_sim_time=1.0, seed=77, auto_reset=True)
    ob0 = env.reset().clone()
    ev1 = env.ev_vg.clone()
    a = torch.zeros(N, dtype=torch.uint8, device=gpu_device)
    for t in range(1, 13):
        obs, rew, done, _ = env.step(a)
        if t == 10:
            assert bool(done.all()) and env.step_stats()["n_done"] == N
            assert torch.allclose(obs.double(), ob0.double(), atol=2e-6)        # reset observation
            assert bool((rew != 0).all())                                        # reward of the ended transition
            assert int(env.istate[2].min()) == 2 and int(env.istate[0].max()) == 0
            assert not torch.equal(env.ev_vg, ev1)                               # fresh event draw (episode 2)
        else:
            assert not bool(done.any())
    assert int(env.steps.min()) == 2 and int(env.steps.max()) == 2
    assert float(obs[:, 10].min()) == pytest.approx(2 * 6 / 60.0, rel=1e-6)    # tStart/max_sim_time


# ---- device-resident batched Q-learning (gym_PVDER/agents/qagent.py)
def test_batched_q_agent_learns_on_device(gpu_device):
    from gym_PVDER_amd.agents import BatchedQAgent
    from gym_PVDER_amd.vec_env import PVDERVec
    env = PVDERVec(256, device=gpu_device, n_sim_time_steps_per_env_step=60, max_sim_time=8.0, auto_reset=False, seed=4)
    agent = BatchedQAgent(env, bins=3, seed=1)                  # 3^11 x 5 table (the reference uses 6 bins)
    assert agent.q_table.shape == (3 ** 11, 5) and float(agent.q_table.min()) == -2.0
    ob = env.reset()
    st = agent.get_discrete_state(ob)
    # reset observation: every component sits in the middle bin except none below -10/3 ... check by hand
    want = ((ob.double() + 10.0) / (20.0 / 3)).long().clamp(0, 2)
    assert torch.equal(st, (want * agent._strides).sum(1))
    stats = agent.train(episodes=6)
    assert len(stats['episode']) == 2 and stats['episode'] == [0, 5]
    assert agent.epsilon == pytest.approx(0.98 ** 6)
    touched = (agent.q_table != -2.0).any(dim=1).sum()
    assert int(touched) >= 1                                     # the table was written on the device
    assert -5 * 8 <= stats['min'][-1] <= stats['max'][-1] <= 8   # 8 one-second steps, rewards in {-5,-1,+1}
    a = agent.agent_action(env.obs)
    assert a.dtype == torch.uint8 and int(a.max()) <= 4


def test_checkpoint_resume_is_bit_exact(gpu_device, tmp_path):
    from gym_PVDER_amd.vec_env import PVDERVec
    kw = dict(device=gpu_device, n_sim_time_steps_per_env_step=6, max_sim_time=1.0, seed=13)
    env = PVDERVec(300, **kw)
    env.reset()
    gen = torch.Generator(device=gpu_device); gen.manual_seed(2)
    a = torch.randint(0, 5, (14, 300), generator=gen, device=gpu_device, dtype=torch.uint8)
    for t in range(6):
        env.step(a[t])
    torch.save(env.state_dict(), tmp_path / "ckpt.pt")
    ref = [tuple(x.clone() for x in env.step(a[t])[:3]) for t in range(6, 14)]      # crosses an auto-reset
    env2 = PVDERVec(300, **kw)
    env2.load_state_dict(torch.load(tmp_path / "ckpt.pt"))
    for t in range(6, 14):
        obs, rew, done, _ = env2.step(a[t])
        assert torch.equal(obs, ref[t - 6][0]) and torch.equal(rew, ref[t - 6][1]) and torch.equal(done, ref[t - 6][2])
    assert env2.step_stats() == env.step_stats()
    with pytest.raises(ValueError):
        PVDERVec(299, **kw).load_state_dict(env.state_dict())


def test_calc_returns_vs_oracle(gpu_device):
    """SURVEY.md 8(f) rank 2 against the oracle: the returns of the reference's four fixed policies
    (PVDER_env.py:458-497) from the batched rollout == the fp64 bridge (converged RK45) stepping the same
    event tables (counter generator keyed by (seed, env id, episode)) and the same action sequences,
    with the reference's reward rules (oracle/bridge_env.py) -- the bridge's figures are the committed golden
    tests/golden/bridge_calc_returns.npz (tests/golden/make_bridge_goldens.py calc_returns)."""
    import bridge_cases
    from gym_PVDER_amd import der_model
    from gym_PVDER_amd.evaluate import ACTION_SPECS, calc_returns
    g = bridge_cases.load_calc_returns()
    c = bridge_cases.CALC_RETURNS
    seed, episodes, T = c["seed"], c["episodes"], c["T"]
    rand = g["rand"]
    for goal, discrete in c["cells"]:
        got = calc_returns(episodes=episodes, device=gpu_device, goals=[goal], seed=seed, max_sim_time=T,
                           DISCRETE_REWARD=discrete, random_actions=torch.as_tensor(rand))
        for name, fixed in ACTION_SPECS.items():
            want, vdcref, qref = g["%s/%s" % (goal, name)]
            have = got[goal][name]['return']
            if discrete:
                assert abs(have - want) <= 4.0 / episodes + 1e-9, (goal, name, have, want)     # at most one threshold flip
            else:
                assert abs(have - want) <= 1e-3 * abs(want) + 1e-9, (goal, name, have, want)
            assert abs(got[goal][name]['ref'][0] - vdcref * der_model.VBASE) < 1e-2
            assert abs(got[goal][name]['ref'][1] - qref * der_model.SBASE) < 1e-1


def test_q_agent_update_vs_oracle(gpu_device):
    """SURVEY.md 8(f) rank 3 against the oracle: the device agent's discretisation and table update on the
    transitions of two scripted episodes == the reference's update rule (oracle/qagent_oracle.py, a
    line-cited restatement of qagent.py:37-45, :83-103) on the same transitions, entry for entry."""
    from gym_PVDER_amd.agents import BatchedQAgent
    from gym_PVDER_amd.vec_env import PVDERVec
    from oracle.qagent_oracle import OracleQTable
    env = PVDERVec(1, device=gpu_device, n_sim_time_steps_per_env_step=15, max_sim_time=6.0, auto_reset=False, seed=4)
    agent = BatchedQAgent(env, bins=6, seed=1)
    ref = OracleQTable(bins=6)
    rng = np.random.default_rng(0)
    for episode in range(2):
        ob = env.reset()
        s_dev = agent.get_discrete_state(ob)
        s_ref = ref.discrete(ob[0].cpu().numpy())
        alive = torch.ones(1, dtype=torch.bool, device=gpu_device)
        done = False
        while not done:
            a = int(rng.integers(0, 5))
            a_dev = torch.tensor([a], device=gpu_device)
            obs, r, d, _ = env.step(a_dev.to(torch.uint8))
            s2_dev = agent.get_discrete_state(obs)
            s2_ref = ref.discrete(obs[0].cpu().numpy())
            assert int(s2_dev) == int(np.ravel_multi_index(s2_ref, (6,) * 11))       # same table row
            agent.update(s_dev, a_dev, r.to(torch.float32), s2_dev, d, alive)
            done = bool(d[0])
            ref.update(s_ref, a, float(r[0]), s2_ref, done)
            s_dev, s_ref = s2_dev, s2_ref
        agent.update_epsilon(); ref.update_epsilon()
    assert agent.epsilon == pytest.approx(ref.epsilon)
    assert len(ref.q) >= 1
    for key, row in ref.q.items():
        got = agent.q_table[int(np.ravel_multi_index(key, (6,) * 11))].cpu().numpy()
        assert np.allclose(got, row, rtol=2e-6, atol=1e-6), (key, got, row)
    touched = int((agent.q_table != -2.0).any(dim=1).sum())
    assert touched == sum(1 for row in ref.q.values() if (row != -2.0).any())


def test_fused_q_agent_step_vs_oracle(gpu_device):
    """pvder_qagent_step_f32 (discretise + table update + action choice in ONE launch per env step) against the
    oracle's restatement of qagent.py:43-45, :63-68, :83-103 on the same transitions: table entries, table rows,
    episode returns, and the epsilon-greedy choice reproduced from the counter generator."""
    from gym_PVDER_amd.agents import BatchedQAgent
    from gym_PVDER_amd.vec_env import PVDERVec
    from oracle.qagent_oracle import OracleQTable
    from oracle.pvder_model import counter_uniform
    env = PVDERVec(1, device=gpu_device, n_sim_time_steps_per_env_step=15, max_sim_time=6.0, auto_reset=False, seed=4)
    agent = BatchedQAgent(env, bins=6, seed=21)
    ref = OracleQTable(bins=6)
    state = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    action = torch.zeros(1, dtype=torch.uint8, device=gpu_device)
    alive = torch.zeros(1, dtype=torch.uint8, device=gpu_device)
    ret = torch.zeros(1, dtype=torch.float64, device=gpu_device)
    counter = 0
    for episode in range(3):
        agent.epsilon = [1.0, 0.5, 0.0][episode]                 # all exploration, mixed, all greedy
        ob = env.reset()
        agent.fused_step(state, action, alive, ret, first=True, counter=counter)
        s_ref = ref.discrete(ob[0].cpu().numpy())
        want_ret, done, steps = 0.0, False, 0
        while not done:
            # the action the kernel chose == the reference's rule on the oracle's table with the same draws
            u0, u1 = counter_uniform(21, 0, counter, 0, 0), counter_uniform(21, 0, counter, 0, 1)
            a_ref = min(int(np.float32(u1) * np.float32(5)), 4) if np.float32(u0) <= np.float32(agent.epsilon) \
                else int(np.argmax(ref.q[s_ref]))
            assert int(action[0]) == a_ref and int(state[0]) == int(np.ravel_multi_index(s_ref, (6,) * 11))
            counter += 1
            obs, r, d, _ = env.step(action)
            a_taken = int(action[0])
            agent.fused_step(state, action, alive, ret, first=False, counter=counter)
            s2_ref = ref.discrete(obs[0].cpu().numpy())
            done = bool(d[0])
            ref.update(s_ref, a_taken, float(r[0]), s2_ref, done)
            want_ret += float(r[0])
            s_ref = s2_ref
            steps += 1
        assert steps == 24 and int(alive[0]) == 0 and float(ret[0]) == want_ret
        # a finished env returns stale tuples: further agent steps must not learn from them
        before = agent.q_table.clone()
        env.step(action); agent.fused_step(state, action, alive, ret, first=False, counter=counter + 1000)
        assert torch.equal(before, agent.q_table) and float(ret[0]) == want_ret
        agent.update_epsilon(); ref.update_epsilon()
    for key, row in ref.q.items():
        got = agent.q_table[int(np.ravel_multi_index(key, (6,) * 11))].cpu().numpy()
        assert np.allclose(got, row, rtol=2e-6, atol=1e-6), (key, got, row)
    touched = int((agent.q_table != -2.0).any(dim=1).sum())
    assert touched == sum(1 for row in ref.q.values() if (row != -2.0).any())


def test_fused_q_agent_trains_like_the_torch_loop(gpu_device):
    """train_fused (two launches per env step, no host synchronisation inside an episode): same bookkeeping as
    train() -- statistics every 5th episode, epsilon decay, a table that has learnt something -- and a
    reproducible exploration stream."""
    from gym_PVDER_amd.agents import BatchedQAgent
    from gym_PVDER_amd.vec_env import PVDERVec
    runs = []
    for rep in range(2):
        env = PVDERVec(256, device=gpu_device, n_sim_time_steps_per_env_step=60, max_sim_time=8.0, auto_reset=False, seed=4)
        agent = BatchedQAgent(env, bins=3, seed=1)
        stats = agent.train_fused(episodes=6)
        assert stats['episode'] == [0, 5] and agent.epsilon == pytest.approx(0.98 ** 6)
        assert -5 * 8 <= stats['min'][-1] <= stats['max'][-1] <= 8          # 8 one-second steps, rewards in {-5,-1,+1}
        assert int((agent.q_table != -2.0).any(dim=1).sum()) >= 1
        runs.append((stats['average'], stats['min'], stats['max']))
    # episode 0 explores with epsilon = 1: every action comes from the counter generator -> identical returns; later
    # episodes read a table in which, of several envs that hit one entry in the same step, one update wins
    assert runs[0][0][0] == runs[1][0][0] and runs[0][1][0] == runs[1][1][0]
    a = agent.agent_action(env.obs)
    assert a.dtype == torch.uint8 and int(a.max()) <= 4
    # the reference's agent utilities (agent_utilities.py): greedy evaluation, statistics bookkeeping, plot
    means = agent.evaluate(n_episodes=2)
    assert len(means) == 2 and all(-5 * 8 <= m <= 8 for m in means)
    agent.initialize_aggregate_rewards()
    agent.collect_aggregate_rewards(3, average=1.0, min=-2.0, max=4.0)
    assert agent.aggregate_episode_rewards == {'episode': [3], 'average': [1.0], 'max': [4.0], 'min': [-2.0]}
    import matplotlib
    matplotlib.use("Agg")
    assert agent.show_plots(show=False) is not None


def test_reference_helper_surface_of_the_single_env(gpu_device, capsys):
    """The rest of the reference env's public surface: episode statistics and their printers
    (env_utilities.py:12-101), the configuration printers, update_env_goal and calc_returns as METHODS of the env
    (PVDER_env.py:445-529) -- same names, same printed lines, same result table."""
    import gym_PVDER_amd as gp
    env = gp.make('PVDER-v0', max_sim_time=2.0, device=gpu_device)
    env.reset()
    for a in (1, 1, 3, 0):
        env.step(a)
    assert env._action_stats == {0: 1, 1: 2, 2: 0, 3: 1, 4: 0}
    assert env._reward_stats['step'] == [1, 2, 3, 4] and len(env._time_stats['step_time']) == 4
    env.show_action_stats(); env.show_reward_stats(); env.show_time_stats(); env.show_step_time()
    env.show_env_config()
    out = capsys.readouterr().out
    assert 'Environment name:GymDER_' in out and 'Goal type: voltage_regulation' in out
    assert "Reward type:['voltage_error'],Discrete:True" in out
    assert 'time step:0.2500,del Qref:375.00,del Vdcref:0.30' in out               # :508 with n = 15 (:616-618)
    assert 'Voltage events:True,Insolation events:False' in out                     # :60-61 defaults
    assert 'Time for step:' in out
    env.initialize_stats()
    assert env._action_stats == {i: 0 for i in range(5)} and env._reward_stats == {'step': [], 'reward': []}
    env.update_action_stats(2); env.update_reward_stats(); env.update_time_stats()
    assert env._action_stats[2] == 1 and env._reward_stats['step'] == [4] and env._time_stats['step'] == [4]
    env.update_env_goal(None, None)
    assert env.env_goal_spec['power_regulation']['reward']['my_spec'] == env.env_goal_spec['power_regulation']['reward']['required']
    env.update_env_goal('Q_regulation', {'Q_regulation': {}})
    assert 'under construction' in capsys.readouterr().out                          # :456
    table = env.calc_returns()
    assert table is env.env_average_return
    assert sorted(table) == ['Q_regulation', 'power_regulation', 'voltage_regulation']
    for goal, row in table.items():
        assert sorted(row) == ['dec', 'inc', 'no_change', 'random']
        for cell in row.values():
            assert -5 * 8 <= cell['return'] <= 8 and len(cell['ref']) == 2          # 8 steps of 0.25 s, rewards in {-5,-1,+1}
    # 'dec' sends action 1 (+Q_ref, the reference's labels :482-487): 8 steps x 375 VAR
    assert table['voltage_regulation']['dec']['ref'][1] == pytest.approx(8 * 375.0, rel=1e-5)
    assert "'return'" in capsys.readouterr().out                                    # show_env_returns printed the table


@pytest.mark.parametrize("N,bins", [(1, 6), (8, 6), (64, 4)])
def test_closed_loop_fused_episode_equals_the_two_launch_loop(gpu_device, N, bins):
    """the agent step INSIDE the fused rollout (pvder_plan_rollout_fused_agent: env step, discretise,
    table update, epsilon-greedy choice, next env step -- the env state and the agent's registers in registers, the
    episode-ending step by the finish kernel) against train_fused's two launches per env step: the same table, bit
    for bit, the same returns, states, actions, env state -- for one env, and for the envs of one wave (which step
    in lockstep in both forms; across waves the two-launch form itself is a race for shared table entries)."""
    from gym_PVDER_amd.agents import BatchedQAgent
    from gym_PVDER_amd.vec_env import PVDERVec
    res = []
    for closed in (False, True):
        env = PVDERVec(N, device=gpu_device, n_sim_time_steps_per_env_step=15, max_sim_time=6.0, auto_reset=False, seed=4,
                       events_spec=both_events_spec_())
        agent = BatchedQAgent(env, bins=bins, seed=21)
        agent.explore_spec = dict(agent.explore_spec, EPSILON_DECAY=0.5)        # exploration -> mixed -> mostly greedy
        stats = agent.train_fused(episodes=4, closed_loop=closed)
        torch.cuda.synchronize()
        res.append((agent.q_table, env.fstate.clone(), env.istate.clone(), env.out.clone(), stats, agent._counter))
    a, b = res
    assert a[5] == b[5] == 4 * (24 + 1)
    assert torch.equal(a[0], b[0])                                   # the whole table
    assert int((a[0] != -2.0).any(dim=1).sum()) >= 2                 # ... which has learnt something
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert a[4] == b[4]                                              # episode statistics (returns)


def test_closed_loop_fused_training_statistics_at_many_workgroups(gpu_device):
    """beyond one workgroup the closed loop inside the fused rollout is ASYNCHRONOUS Q-learning (workgroups
    advance through their env steps independently, table updates of other workgroups arrive late or lose a race) -- not the
    two-launch loop bit for bit.  What must hold: the same learning STATISTICS.  2048 envs (8 workgroups) on the reference's
    6-bin table (its grid is coarse: the whole batch lives in a handful of rows, every entry shared by hundreds of envs --
    the worst case for staleness), 6 episodes of 24 steps."""
    from gym_PVDER_amd.agents import BatchedQAgent
    from gym_PVDER_amd.vec_env import PVDERVec
    N, res = 2048, []
    for closed in (False, True):
        env = PVDERVec(N, device=gpu_device, n_sim_time_steps_per_env_step=15, max_sim_time=6.0, auto_reset=False, seed=4,
                       events_spec=both_events_spec_())
        agent = BatchedQAgent(env, bins=6, seed=21)
        agent.explore_spec = dict(agent.explore_spec, EPSILON_DECAY=0.6)
        agent.train_spec = dict(agent.train_spec, STATS_EVERY=1)
        stats = agent.train_fused(episodes=6, closed_loop=closed)
        torch.cuda.synchronize()
        res.append((agent.q_table.clone(), stats, env.istate.clone()))
    (qa, sa, ia), (qb, sb, ib) = res
    assert torch.equal(ia, ib)                                        # same clocks, same episode ends
    va, vb = (qa != -2.0).any(dim=1), (qb != -2.0).any(dim=1)         # rows that were ever updated
    na, nb, both = int(va.sum()), int(vb.sum()), int((va & vb).sum())
    assert na >= 2 and abs(na - nb) <= max(1, 0.25 * na), (na, nb)    # the same part of the table is visited
    assert both >= min(na, nb) - 1, (na, nb, both)
    # episode returns, episode by episode: the same learning curve (the reward is the discrete +1 / -1 / -5 ladder, 24 steps)
    ra, rb = np.array(sa['average']), np.array(sb['average'])
    assert ra.shape == rb.shape == (6,)
    assert np.abs(ra - rb).max() < 0.08 * 24, (ra, rb)
    # (entry VALUES are not compared: with hundreds of envs per row "one of their updates wins" in BOTH forms -- the
    # transition that ends an episode writes q_done = 0 over whatever the row held, and whether it lands before or after the
    # other envs' updates of that launch is an accident of either form; measured: rows that differ by their whole size)
    assert bool(torch.isfinite(qa).all()) and bool(torch.isfinite(qb).all())
    assert float(qa[va].abs().max()) < 5.0 * 24 and float(qb[vb].abs().max()) < 5.0 * 24       # |return-to-go| <= 5 per step


def both_events_spec_():
    from helpers import both_events_spec
    return both_events_spec()


# ---- the Q-network policy on the device + the batched DQN loop (second half; the reference's
#      examples/gym_PVDER_environment_tf_agents_DQN_demo.ipynb cells 12-28) ------------------------------------------------
def test_qnet_policy_on_the_device_matches_a_torch_forward_pass(gpu_device):
    """PVDER_POLICY_QNET (include/pvder_agent.h): observation(11) -> Dense(100, relu) -> Dense(5) evaluated inside the agent
    launch for every env -- against the same network in torch fp32: the same greedy action wherever the two best outputs are
    more than 1e-5 apart; with epsilon = 1 the actions are the exploration stream's (uniform over the five, reproducible)."""
    from gym_PVDER_amd.vec_env import PVDERVec
    from gym_PVDER_amd.agents import BatchedDQNAgent
    N = 4096
    env = PVDERVec(N, device=gpu_device, seed=5, events_spec=both_events_spec(), auto_reset=True)
    ag = BatchedDQNAgent(env, seed=3)
    with torch.no_grad():                                   # a network whose outputs differ visibly between envs
        ag.q_net.flat.copy_(torch.randn(ag.q_net.flat.shape, generator=torch.Generator().manual_seed(1)).to(gpu_device) * 0.5)
    env.reset()
    gen = torch.Generator(device=gpu_device); gen.manual_seed(2)
    seen = set()
    for t in range(12):
        env.step(torch.randint(0, 5, (N,), generator=gen, device=gpu_device, dtype=torch.uint8), return_tensors=False)
        if t >= 4:          # (PVDER-v0 observations of one batch differ in the third digit: spread them over the Box to see every action)
            env.out[:11].copy_((torch.rand((11, N), generator=gen, device=gpu_device) * 2 - 1) * (0.5 * t))
        ag.policy_step(first=(t == 0), epsilon=0.0)
        obs = env.obs.clone()
        q = ag.q_net(obs).detach()
        top2 = q.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        want = q.argmax(dim=1).to(torch.uint8)
        assert clear.float().mean() > 0.99
        assert torch.equal(ag.action[clear], want[clear]), t
        seen |= set(ag.action.unique().tolist())
    assert len(seen) >= 3                                   # (not a constant policy)
    ag.policy_step(first=False, epsilon=1.0)
    counts = torch.bincount(ag.action.long(), minlength=5).float() / N
    assert counts.min() > 0.15 and counts.max() < 0.25
    a1 = ag.action.clone()
    ag._counter -= 1
    ag.policy_step(first=False, epsilon=1.0)                # the same key: the same draw
    assert torch.equal(a1, ag.action)


def test_qnet_closed_loop_fused_episode_equals_the_two_launch_loop(gpu_device):
    """A greedy evaluation episode with the Q-network as the action source INSIDE the fused rollout
    (pvder_plan_rollout_fused_agent, PVDER_POLICY_QNET) against env launch + policy launch per step: returns, final state and
    counters bit for bit (64 envs: one workgroup; nothing is learned on the device, so larger batches agree too)."""
    from gym_PVDER_amd.vec_env import PVDERVec
    from gym_PVDER_amd.agents import BatchedDQNAgent
    for N in (64, 1000):
        train_env = PVDERVec(N, device=gpu_device, seed=8, events_spec=both_events_spec(), auto_reset=True)
        ag = BatchedDQNAgent(train_env, seed=4)
        with torch.no_grad():
            ag.q_net.flat.copy_(torch.randn(ag.q_net.flat.shape, generator=torch.Generator().manual_seed(7)).to(gpu_device) * 0.3)
        res = {}
        for fused in (False, True):
            ev = PVDERVec(N, device=gpu_device, seed=9, events_spec=both_events_spec(), auto_reset=False, max_sim_time=10.0)
            res[fused] = (ag.evaluate(ev, fused=fused), ev.fstate.clone(), ev.istate.clone(), ev.out.clone())
        a, b = res[False], res[True]
        assert a[0]["steps"] == 40 and torch.equal(a[0]["returns"], b[0]["returns"])
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert float(a[0]["returns"].abs().max()) > 0


def test_batched_dqn_loop_trains_on_device_collected_transitions(gpu_device):
    """The loop of the reference's DQN demo (cells 25-28: collect with the epsilon-greedy network, uniform replay, squared TD loss,
    Adam 1e-3, batch 64) with collection on the device: the replay buffer holds what the env produced (rows equal the output
    blocks), the loss is finite and falls on a fixed buffer, the weights the device policy reads are the ones torch trains."""
    from gym_PVDER_amd.vec_env import PVDERVec
    from gym_PVDER_amd.agents import BatchedDQNAgent
    N = 512
    env = PVDERVec(N, device=gpu_device, seed=11, events_spec=both_events_spec(), auto_reset=True, max_sim_time=5.0)
    ag = BatchedDQNAgent(env, seed=1, replay_buffer_capacity=N * 64)
    w0 = ag.q_net.flat.detach().clone()
    losses = ag.train(num_iterations=60, initial_collect_steps=8)
    assert ag.buf_n == min(ag.capacity, N * 68) and all(np.isfinite(losses))
    assert not torch.equal(w0, ag.q_net.flat.detach())                      # trained ...
    ag.policy_step(first=False, epsilon=0.0)
    assert torch.equal(ag.action, ag.greedy_actions(env.obs)) or \
        (ag.action != ag.greedy_actions(env.obs)).float().mean() < 0.01      # ... and the device policy follows the new weights
    # the buffer's rows are real transitions: rewards in the discrete ladder, done flags at the episode ends (20 steps)
    assert set(ag.buf_rew[:ag.buf_n].unique().tolist()) <= {1.0, -1.0, -5.0, -100.0}
    assert 0.02 < float(ag.buf_done[:ag.buf_n].mean()) < 0.1
    # a fixed buffer and a FIXED target network, more gradient steps: the squared TD error against that target falls (with the
    # demo's target_update_period = 1 the target moves with every step and the loss follows the growing returns instead)
    ag.spec['target_update_period'] = 10 ** 9
    first = np.mean([ag.train_step() for _ in range(20)])
    for _ in range(300):
        ag.train_step()
    last = np.mean([ag.train_step() for _ in range(20)])
    assert last < first, (first, last)
    ev = PVDERVec(256, device=gpu_device, seed=12, events_spec=both_events_spec(), auto_reset=False, max_sim_time=5.0)
    r = ag.evaluate(ev)
    assert np.isfinite(r["average_return"]) and r["steps"] == 20


def test_single_env_step_mirror_equals_the_copy_read_back(gpu_device):
    """pvder_plan_step_mirror (ABI 16): the single env's step whose read-back is written by the device itself into pinned host
    memory -- output rows, counters, the trajectory record, a sequence word the host spins on -- against the copy-packet
    read-back of ABI 15 (pvder_plan_step_fetch): a whole episode of the same random actions, every returned tuple, the
    counters behind `steps` and the recorded trajectory bit for bit; a stale tuple behind `done`; reset() in between."""
    import gym_PVDER_amd as gp
    rng = np.random.default_rng(11)
    acts = rng.integers(0, 5, size=200)
    from gym_PVDER_amd.envs.pvder_env import PVDER
    runs, count0 = [], PVDER.count
    for mirror in (True, False):
        PVDER.count = count0                                         # (the instance counter seeds the env's events: the same env twice)
        env = gp.make('PVDER-v0', max_sim_time=10.0, device=gpu_device)
        env.unwrapped._use_mirror = mirror
        rows = []
        for ep in range(2):
            rows.append(env.reset().tolist())
            for a in acts[:60]:
                obs, r, done, _ = env.step(int(a))
                rows.append(obs.tolist() + [float(r), float(done), float(env.unwrapped.steps)])
        tr = env.unwrapped._vec.trajectory(0)
        runs.append((rows, {k: np.asarray(v) for k, v in tr.items()}, env.unwrapped.done))
    (ra, ta, da), (rb, tb, db) = runs
    assert da and db                                                 # 10 s = 40 env steps: the episode ended, 20 stale tuples follow
    assert ra == rb
    assert sorted(ta) == sorted(tb)
    for k in ta:
        assert np.array_equal(ta[k], tb[k]), k
    # the block itself: a stale sequence word is refused before anything is launched
    from gym_PVDER_amd import _native
    e = gp.make('PVDER-v0', device=gpu_device).unwrapped
    e.reset()
    e.step(0)
    v = e._vec
    rc = v._lib.pvder_plan_step_mirror(v._plan, v.fstate.data_ptr(), v.istate.data_ptr(), v.ev_vg.data_ptr(), v.ev_si.data_ptr(),
                                       e._actions[0].data_ptr(), v.out.data_ptr(), None, v.stats.data_ptr(), None, None, 0,
                                       e._pin_mirror.data_ptr(), e._mirror_seq, v._stream())
    assert rc == -10001                                              # PVDER_EINVAL: seq == the word the block already holds
    pageable = torch.zeros(_native.MIRROR_WORDS, dtype=torch.float64)
    rc = v._lib.pvder_plan_step_mirror(v._plan, v.fstate.data_ptr(), v.istate.data_ptr(), v.ev_vg.data_ptr(), v.ev_si.data_ptr(),
                                       e._actions[0].data_ptr(), v.out.data_ptr(), None, v.stats.data_ptr(), None, None, 0,
                                       pageable.data_ptr(), 7, v._stream())
    assert rc == -10001                                              # ... and so is a block the device cannot address
    assert int(e.steps) == 1

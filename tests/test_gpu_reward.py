"""-m gpu: every branch of the device's reward / done logic (csrc `reward_of`, `epilogue`) against the
oracle's restatement of ``reward_calc`` / the done rule (PVDER_env.py:231-301, :183-191, TimeLimit of
gym_PVDER/__init__.py:10) evaluated on the SAME end-of-step states: 3 goals x {discrete: +1, -1, -5 on
both sides of both thresholds; continuous}, the optional Vdc_error term, time-limit and step-limit
done.  A discrete mismatch is tolerated only within 1e-6 of a threshold -- and asserted to be there."""
import math

import numpy as np
import pytest
import torch

from oracle import pvder_model as pm
from oracle.pvder_oracle_env import P_REF, Q_REF_TARGET, reward_terms

pytestmark = pytest.mark.gpu
N = 4096


def make_vec(device, **kw):
    from gym_PVDER_amd.vec_env import PVDERVec
    return PVDERVec(N, device=device, auto_reset=False, seed=5, **kw)


def tables(env, vg, si):
    K1 = env.ev_vg.shape[0]
    dev = env.device
    env.set_event_tables(torch.as_tensor(vg, dtype=torch.float32, device=dev)[None, :].expand(K1, N).contiguous(),
                         torch.as_tensor(si, dtype=torch.float32, device=dev)[None, :].expand(K1, N).contiguous())


def check(env, goal, discrete, steps, my_spec=None, vdc_row=18):
    """step with action 0 and compare every env's reward with reward_terms() on the device's own state"""
    m = pm.build_model("50")
    a = torch.zeros(N, dtype=torch.uint8, device=env.device)
    thresholds = {"voltage_regulation": (0.01, 0.05), "power_regulation": (0.01, 0.03), "Q_regulation": (0.01, 0.05)}[goal]
    seen, near = set(), 0
    for _ in range(steps):
        obs, rew, done, _ = env.step(a)
        o = obs.cpu().numpy().astype(np.float64)
        r = rew.cpu().numpy().astype(np.float64)
        vdc = env.fstate[vdc_row].cpu().numpy().astype(np.float64)
        vdc_ref = env.Vdc_ref.cpu().numpy().astype(np.float64)
        q_ref = env.Q_ref.cpu().numpy().astype(np.float64)
        assert np.array_equal(vdc.astype(np.float32), o[:, 6].astype(np.float32))
        for e in range(N):
            vrms = abs(complex(o[e, 2], o[e, 3])) / math.sqrt(2.0)       # balanced: sqrt(sum |v_p|^2 / 3) / sqrt 2
            want = reward_terms(goal, discrete, vrms, m.Vrms_ref, complex(o[e, 4], o[e, 5]), vdc[e], vdc_ref[e], my_spec,
                                Q_ref=q_ref[e])
            if discrete:
                seen.add(int(r[e]))
                if r[e] != want:
                    err = {"voltage_regulation": abs(vrms - m.Vrms_ref) / m.Vrms_ref,
                           "power_regulation": abs(o[e, 4] - P_REF / pm.SBASE) / (P_REF / pm.SBASE),
                           "Q_regulation": abs(o[e, 5] - Q_REF_TARGET / pm.SBASE) / (Q_REF_TARGET / pm.SBASE)}[goal]
                    evdc = abs(vdc[e] - vdc_ref[e]) / vdc_ref[e]
                    qt = (q_ref[e] if q_ref[e] != 0.0 else 1e-6) if goal == "voltage_regulation" else Q_REF_TARGET / pm.SBASE
                    eq = abs(o[e, 5] - qt) / abs(qt)
                    ev = abs(vrms - m.Vrms_ref) / m.Vrms_ref
                    extra = my_spec or []
                    close = min(abs(err - t) for t in thresholds) < 1e-6 or \
                        ("Vdc_error" in extra and min(abs(evdc - t) for t in (0.02, 0.05)) < 1e-6) or \
                        ("Q_error" in extra and min(abs(eq - t) for t in (0.01, 0.05)) < 2e-6 * max(1.0, eq)) or \
                        ("voltage_error" in extra and min(abs(ev - t) for t in (0.01, 0.05)) < 1e-6)
                    assert close, (goal, e, r[e], want, err, evdc, eq, ev)
                    near += 1
            else:
                assert abs(r[e] - want) <= 2e-6 * max(1.0, abs(want)) + 1e-9, (goal, e, r[e], want)
    return seen, near


@pytest.mark.parametrize("discrete", [True, False])
def test_voltage_error_branches(gpu_device, discrete):
    env = make_vec(gpu_device, goals_list=["voltage_regulation"], DISCRETE_REWARD=discrete)
    env.reset()
    tables(env, np.linspace(0.93, 1.07, N), np.full(N, 100.0))            # Vrms error 0 .. 7 %: both thresholds, both sides
    env.reset()
    seen, near = check(env, "voltage_regulation", discrete, steps=3)
    if discrete:
        assert seen == {1, -1, -5}


@pytest.mark.parametrize("discrete", [True, False])
def test_power_error_branches(gpu_device, discrete):
    env = make_vec(gpu_device, goals_list=["power_regulation"], DISCRETE_REWARD=discrete)
    env.reset()
    tables(env, np.ones(N), np.linspace(88.0, 106.0, N))                  # P_PCC 0.80 .. 0.97 pu around the 0.908 target
    env.reset()
    seen, near = check(env, "power_regulation", discrete, steps=14)
    if discrete:
        assert seen == {1, -1, -5}


@pytest.mark.parametrize("discrete", [True, False])
def test_q_error_branches(gpu_device, discrete):
    env = make_vec(gpu_device, goals_list=["Q_regulation"], DISCRETE_REWARD=discrete)
    env.reset()
    env.fstate[23].copy_(torch.linspace(0.09, 0.13, N, device=gpu_device))      # Q_ref around the 0.11 pu target
    seen, near = check(env, "Q_regulation", discrete, steps=8)
    if discrete:
        assert seen == {1, -1, -5}


@pytest.mark.parametrize("discrete", [True, False])
def test_vdc_error_term_of_my_spec(gpu_device, discrete):
    """power_regulation with my_spec = [power_error, Vdc_error] (PVDER_env.py:249-262): the two terms
    are summed (:301); DC-link voltages from -8 % to +8 % of Vdc_ref exercise the 2 % / 5 % ladder."""
    env = make_vec(gpu_device, goals_list=["power_regulation"], DISCRETE_REWARD=discrete)
    env.set_reward_spec(["power_error", "Vdc_error"])
    env.reset()
    env.fstate[18].mul_(torch.linspace(0.92, 1.08, N, device=gpu_device))       # Vdc
    seen, near = check(env, "power_regulation", discrete, steps=4, my_spec=["power_error", "Vdc_error"])
    if discrete:
        assert {2, 0, -4, -6, -10} & seen and len(seen) >= 4          # sums of two ladders
    with pytest.raises(ValueError):
        env.set_reward_spec(["Vdc_error"])                            # the required term must stay
    other = make_vec(gpu_device, goals_list=["voltage_regulation"])
    with pytest.raises(ValueError):
        other.set_reward_spec(["voltage_error", "Vdc_error"])         # Vdctarget exists for power_regulation only


@pytest.mark.parametrize("discrete", [True, False])
def test_q_error_term_under_voltage_regulation(gpu_device, discrete):
    """voltage_regulation with my_spec = [voltage_error, Q_error] (PVDER_env.py:249, :263-275): Qtarget is the env's
    own PV_model.Q_ref (:238), and 1e-6 where that is 0 in the discrete branch (:266-267).  Q_ref from -0.3 to
    +0.3 pu with every 16th env at exactly 0; a few steps after the jump Q_PCC is still on its way to the target,
    so the 1 % / 5 % ladder sees both thresholds from both sides; the grid voltage sweep moves the other term."""
    env = make_vec(gpu_device, goals_list=["voltage_regulation"], DISCRETE_REWARD=discrete)
    env.set_reward_spec(["voltage_error", "Q_error"])
    env.reset()
    tables(env, np.linspace(0.93, 1.07, N), np.full(N, 100.0))
    env.reset()
    q = torch.linspace(-0.3, 0.3, N, device=gpu_device)
    q[::16] = 0.0
    env.fstate[23].copy_(q)
    seen, near = check(env, "voltage_regulation", discrete, steps=6, my_spec=["voltage_error", "Q_error"])
    if discrete:
        assert {2, 0, -4, -6, -10} & seen and len(seen) >= 4          # sums of two ladders
    with pytest.raises(ValueError):
        env.set_reward_spec(["Q_error"])                              # the required term must stay
    with pytest.raises(ValueError):
        env.set_reward_spec(["voltage_error", "power_error"])         # Ptarget exists for power_regulation only


@pytest.mark.parametrize("goal", ["power_regulation", "Q_regulation"])
def test_voltage_error_term_under_the_other_goals(gpu_device, goal):
    """my_spec = [required, voltage_error]: voltage_error needs no target, so reward_calc evaluates it under any goal"""
    req = {"power_regulation": "power_error", "Q_regulation": "Q_error"}[goal]
    env = make_vec(gpu_device, goals_list=[goal], DISCRETE_REWARD=True)
    env.set_reward_spec([req, "voltage_error"])
    env.reset()
    tables(env, np.linspace(0.93, 1.07, N), np.full(N, 100.0))
    env.reset()
    seen, near = check(env, goal, True, steps=3, my_spec=[req, "voltage_error"])
    assert len(seen) >= 3


def test_done_by_max_sim_time_and_by_time_limit(gpu_device):
    """done = tStop >= max_sim_time (PVDER_env.py:183) or TimeLimit's step count (gym_PVDER/__init__.py:10);
    a finished env returns the stale tuple (:145-152)."""
    from gym_PVDER_amd.vec_env import PVDERVec
    a = torch.zeros(64, dtype=torch.uint8, device=gpu_device)
    env = PVDERVec(64, device=gpu_device, n_sim_time_steps_per_env_step=15, max_sim_time=1.0, auto_reset=False)
    env.reset()
    for k in range(1, 5):
        obs, rew, done, _ = env.step(a)
        assert bool(done.all()) == (k * 15 >= 60) and bool(done.any()) == bool(done.all())
    assert abs(float(obs[0, 10]) - 1.0) < 1e-6                        # tStart / max_sim_time
    stale = (obs.clone(), rew.clone())
    obs, rew, done, _ = env.step(a)
    assert torch.equal(obs, stale[0]) and torch.equal(rew, stale[1]) and bool(done.all())
    assert int(env.steps.max()) == 4
    # step limit alone: max_sim_time far away, 7 steps allowed
    env = PVDERVec(64, device=gpu_device, n_sim_time_steps_per_env_step=15, max_sim_time=40.0, auto_reset=False)
    env.max_episode_steps = 7
    env.reset()
    for k in range(1, 9):
        obs, rew, done, _ = env.step(a)
        assert bool(done.all()) == (k >= 7)
    assert int(env.steps.max()) == 7 and abs(float(obs[0, 10]) - 7 * 15 / 60.0 / 40.0) < 1e-6
    # the reference's own limit: 500 steps, and max_sim_time clamped to 500 n / 60 s (:569-571)
    env = PVDERVec(64, device=gpu_device, n_sim_time_steps_per_env_step=1, max_sim_time=40.0, auto_reset=False)
    assert abs(env.max_sim_time - 500 / 60.0) < 1e-12
    env.reset()
    acts = torch.zeros((499, 64), dtype=torch.uint8, device=gpu_device)
    env.rollout(acts)
    assert not bool((env.out[12] != 0).any()) and int(env.steps.min()) == 499
    obs, rew, done, _ = env.step(a)
    assert bool(done.all()) and int(env.steps.max()) == 500

"""-m gpu: per-LANE parameter override of the ETDRK4 integrator (SURVEY.md a10 "optional per-env override [P][N] for domain
randomisation"; config_der.json:2-21) -- one launch with 64 distinct parameter sets against the fp64 bridge integrating
each set with a model built from its own config block (tests/golden/bridge_lane_params.npz, tests/bridge_cases.py)."""
import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


def make_vec(N, device, **kw):
    from gym_PVDER_amd.vec_env import PVDERVec
    return PVDERVec(N, device=device, **kw)


def _replay(env, acts, device):
    got = []
    for t in range(acts.shape[0]):
        obs, rew, done, _ = env.step(torch.as_tensor(acts[t], dtype=torch.uint8, device=device))
        got.append(np.concatenate([obs.cpu().numpy().astype(np.float64), rew.cpu().numpy()[:, None].astype(np.float64)], axis=1))
        assert env.step_stats()["n_nonfinite"] == 0
    return np.stack(got)


def test_64_distinct_parameter_sets_in_one_launch_vs_the_bridge(gpu_device):
    import bridge_cases
    name = "lane_params"
    g = bridge_cases.load(name)
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    sets = bridge_cases.lane_param_sets(name)
    env = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4",
                   lane_params=sets)
    env.reset()
    assert np.array_equal(env.ev_vg.cpu().numpy(), vg.astype(np.float32)) and np.array_equal(env.ev_si.cpu().numpy(), si.astype(np.float32))
    got = _replay(env, acts, gpu_device)
    want = g["obs"]
    per_step = rel_l2(got[:, :, :10], want[:, :, :10], axis=2)
    traj = rel_l2(got[:, :, :10].transpose(1, 0, 2).reshape(c["N"], -1), want[:, :, :10].transpose(1, 0, 2).reshape(c["N"], -1), axis=1)
    # the bar of VERDICT r3 #8: < 1e-4 (measured: see the assertion messages / DESIGN.md)
    assert per_step.max() < 1e-4, (np.unravel_index(per_step.argmax(), per_step.shape), per_step.max())
    assert traj.max() < 3e-5, traj.max()
    assert np.array_equal(got[:, :, 11], g["reward"])                 # discrete rewards: exact
    # ... and the sets DO matter: the same batch without the override is far from these goldens
    ref = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4")
    ref.reset()
    plain = _replay(ref, acts, gpu_device)
    assert rel_l2(plain[:, :, :10], want[:, :, :10], axis=2).max() > 1e-2


def test_every_env_starts_at_its_own_steady_state(gpu_device):
    """VERDICT r4 #5 / ADVICE r4 #1 (PVDER_env.py:374-378: each DER is built with steadyStateInitialization=True; config_der.json:2-21):
    with 64 distinct parameter sets the reset observation equals the bridge's per-set steady state to 1e-6; the first env step
    under action 0 before any event shows NO transient (|d obs| <= 1e-5); and reset() returns the very observation the same-step
    auto-reset of the lane kernels returns for the same env."""
    import bridge_cases
    name = "lane_params"
    g = bridge_cases.load(name)
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    sets = bridge_cases.lane_param_sets(name)
    N = c["N"]
    env = make_vec(N, gpu_device, seed=c["seed"], auto_reset=True, events_spec=cc["spec_dict"], integrator="etdrk4", lane_params=sets,
                   max_sim_time=1.0)                                   # (4 env steps per episode: the auto-reset comes at once)
    obs0 = env.reset().cpu().numpy().astype(np.float64).copy()
    want0 = g["obs_reset"]
    assert np.abs(obs0[:, :10] - want0[:, :10]).max() < 1e-6, np.abs(obs0[:, :10] - want0[:, :10]).max()
    assert np.ptp(obs0[:, 0]) > 0.05 and np.ptp(obs0[:, 7]) > 0.05       # the operating points DO differ (Np 10 / 11, 90-100 % insolation)
    zero = torch.zeros(N, dtype=torch.uint8, device=gpu_device)
    obs1, _, _, _ = env.step(zero)
    d = np.abs(obs1.cpu().numpy().astype(np.float64)[:, :8] - obs0[:, :8])
    assert d.max() < 1e-5, d.max()                                      # (from the group's steady state: 3e-2 in Vdc within one env step)
    for _ in range(3):
        obs, rew, done, _ = env.step(zero)
    assert bool(done.all())                                             # the step that ended the episode returns the RESET observation
    auto = obs.cpu().numpy().astype(np.float64)
    env2 = make_vec(N, gpu_device, seed=c["seed"], auto_reset=True, events_spec=cc["spec_dict"], integrator="etdrk4", lane_params=sets,
                    max_sim_time=1.0)
    env2.reset(); again = env2.reset().cpu().numpy().astype(np.float64)     # (episode 2's tables, as the auto-reset drew them)
    assert np.array_equal(auto[:, :10], again[:, :10])


def test_both_kernel_forms_of_the_override_vs_the_bridge(gpu_device):
    """The override has two kernels -- a PHASE per lane (the latency form, batches up to 32 768 envs: what the 64-env test above
    runs) and a LANE per env (the throughput form beyond): a 49 152-env batch that tiles the golden's 64 parameter sets, event
    tables and actions takes the second; every copy of a set against the bridge, and the copies against each other bit for bit."""
    import bridge_cases
    name = "lane_params"
    g = bridge_cases.load(name)
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    sets = bridge_cases.lane_param_sets(name)
    reps = 768
    N = c["N"] * reps
    env = make_vec(N, gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4",
                   lane_params={k: np.tile(v, reps) for k, v in sets.items()})
    env.reset()
    env.set_event_tables(torch.from_numpy(np.tile(vg.astype(np.float32), (1, reps))).to(gpu_device),
                         torch.from_numpy(np.tile(si.astype(np.float32), (1, reps))).to(gpu_device))
    env.reset()
    steps = 16
    got = _replay(env, np.tile(acts[:steps], (1, reps)), gpu_device)                 # [steps, N, 12]
    want = g["obs"][:steps, :, :10]
    blocks = got[:, :, :10].reshape(steps, reps, c["N"], 10)
    assert rel_l2(blocks[:, 0], want, axis=2).max() < 1e-4
    assert np.array_equal(blocks, np.broadcast_to(blocks[:, :1], blocks.shape))     # a lane's result does not depend on where it sits
    # ... nor on which of the two kernel forms advanced it: the 64-env batch (a phase per lane) gives the same bits
    small = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4",
                     lane_params=sets)
    small.reset()
    got64 = _replay(small, acts[:steps], gpu_device)
    assert np.array_equal(got64[:, :, :10], blocks[:, 0])
    assert np.array_equal(got[:, :c["N"], 11], g["reward"][:steps])
    assert env.step_stats()["n_nonfinite"] == 0


def test_lanes_left_at_the_sets_values_match_the_plain_integrator(gpu_device):
    """NaN rows / missing keys keep the group's constants: the 23-state path with the override installed but empty gives
    what the same path gives without it (force-full ETDRK4 == lane kernel, bit for bit), and the balanced fast path's
    results to its tolerance."""
    import bridge_cases
    vg, si, acts, cc = bridge_cases.case_inputs("lane_params")
    N = 64
    kw = dict(seed=21, auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4")
    a = make_vec(N, gpu_device, lane_params={"Np": np.full(N, np.nan)}, **kw)
    b = make_vec(N, gpu_device, **kw)
    a.reset(); b.reset()
    ga, gb = _replay(a, acts[:12], gpu_device), _replay(b, acts[:12], gpu_device)
    assert rel_l2(ga[:, :, :10], gb[:, :, :10], axis=2).max() < 2e-5
    assert np.array_equal(ga[:, :, 11], gb[:, :, 11])


@pytest.mark.parametrize("N", [192, 33000])
def test_lane_params_through_episode_ends_with_auto_reset(gpu_device, N):
    """Both kernel forms (192 envs: a phase per lane; 33 000: a lane per env, a batch that is no multiple of anything) through
    two episode ends with same-step auto-reset: every env finishes and restarts exactly twice, nothing non-finite, statistics
    = plain sums over the output block."""
    rng = np.random.default_rng(5)
    env = make_vec(N, gpu_device, seed=7, auto_reset=True, integrator="etdrk4",
                   lane_params={"Np": rng.integers(10, 12, N).astype(np.float64), "C_actual": 300e-6 * rng.uniform(0.8, 1.3, N)})
    env.reset()
    done_count = torch.zeros(N, device=gpu_device)
    for t in range(330):
        a = torch.randint(0, 5, (N,), device=gpu_device, dtype=torch.uint8)
        obs, rew, done, _ = env.step(a)
        done_count += done.float()
        if t % 80 == 79 or t in (159, 160, 319, 320):
            st = env.step_stats()
            assert st["n_nonfinite"] == 0
            assert st["n_done"] == int(done.sum().item())
            assert abs(st["reward_sum"] - float(rew.sum().item())) < 1e-3 * max(1.0, abs(float(rew.sum().item())))
    assert bool((done_count == 2).all())
    assert bool((env.steps == 330 - 320).all())
    assert bool(torch.isfinite(env.fstate).all())


def test_lane_params_are_refused_where_they_cannot_apply(gpu_device):
    with pytest.raises(ValueError):
        make_vec(8, gpu_device, integrator="rk4", lane_params={"Np": np.full(8, 10.0)})    # classic RK4: one parameter set per launch
    with pytest.raises(ValueError):
        make_vec(8, gpu_device, model_type="model_1", lane_params={"Np": np.full(8, 10.0)})   # the single-phase model
    with pytest.raises(ValueError):
        make_vec(8, gpu_device, integrator="etdrk4", lane_params={"Rf_actual": np.full(8, 0.002)})   # inside the operator
    with pytest.raises(ValueError):
        make_vec(8, gpu_device, integrator="etdrk4", lane_params={"Np": np.full(7, 10.0)})
    with pytest.raises(ValueError):
        make_vec(8, gpu_device, integrator="etdrk4", lane_params={"C_actual": np.zeros(8)})  # pvder_fold_lane_params: EINVAL


# ---- round 5: the same override on the ETDLIN path (the default integrator): the group's operator, the env's own constants in f ----
def test_64_distinct_parameter_sets_on_the_matrix_core_path_vs_the_bridge(gpu_device):
    """pvder_plan_lane_params on an etdlin plan (ABI 16, step_etdlin_lane_kernel): the exponential operator stays the GROUP's (one
    coefficient block), each env's own 1/C, outer PI gains and PV array enter the right-hand side alone, two Krogstad steps per env
    step.  Same golden, same bars as the ETDRK4 lane kernels (worst env step < 1e-4, trajectory < 3e-5, discrete rewards exact; fp64
    prototype 5.9e-5 / 2.0e-5: tools/lane_etdlin_proto.py) -- at a sixth of their cost (bench.py `lane_params`)."""
    import bridge_cases
    name = "lane_params"
    g = bridge_cases.load(name)
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    sets = bridge_cases.lane_param_sets(name)
    env = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdlin",
                   lane_params=sets)
    assert env.integrator == "etdlin"
    obs0 = env.reset().cpu().numpy().astype(np.float64)
    assert np.abs(obs0[:, :10] - g["obs_reset"][:, :10]).max() < 1e-6          # every env at ITS OWN steady state
    assert np.array_equal(env.ev_vg.cpu().numpy(), vg.astype(np.float32)) and np.array_equal(env.ev_si.cpu().numpy(), si.astype(np.float32))
    got = _replay(env, acts, gpu_device)
    want = g["obs"]
    per_step = rel_l2(got[:, :, :10], want[:, :, :10], axis=2)
    traj = rel_l2(got[:, :, :10].transpose(1, 0, 2).reshape(c["N"], -1), want[:, :, :10].transpose(1, 0, 2).reshape(c["N"], -1), axis=1)
    assert per_step.max() < 1e-4, (np.unravel_index(per_step.argmax(), per_step.shape), per_step.max())
    assert traj.max() < 3e-5, traj.max()
    assert np.array_equal(got[:, :, 11], g["reward"])
    # the sets matter here too, and the two lane paths agree with each other far inside their distance to a batch without sets
    plain = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdlin")
    plain.reset()
    assert rel_l2(_replay(plain, acts, gpu_device)[:, :, :10], want[:, :, :10], axis=2).max() > 1e-2
    other = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4", lane_params=sets)
    other.reset()
    assert rel_l2(_replay(other, acts, gpu_device)[:, :, :10], got[:, :, :10], axis=2).max() < 1.5e-4
    # a lane's result does not depend on where it sits: the batch tiled 40 x (2 560 envs), every copy bit for bit
    reps = 40
    big = make_vec(c["N"] * reps, gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdlin",
                   lane_params={k: np.tile(v, reps) for k, v in sets.items()})
    big.reset()
    big.set_event_tables(torch.from_numpy(np.tile(vg.astype(np.float32), (1, reps))).to(gpu_device),
                         torch.from_numpy(np.tile(si.astype(np.float32), (1, reps))).to(gpu_device))
    big.reset()
    gb = _replay(big, np.tile(acts[:12], (1, reps)), gpu_device)[:, :, :10].reshape(12, reps, c["N"], 10)
    assert np.array_equal(gb, np.broadcast_to(gb[:, :1], gb.shape)) and np.array_equal(gb[:, 0], got[:12, :, :10])
    # fused rollouts refuse a plan with per-lane sets (the per-step launches carry them)
    with pytest.raises(Exception):
        env.rollout(torch.zeros((4, c["N"]), dtype=torch.uint8, device=gpu_device), fused=True)


def test_lane_sets_on_the_matrix_core_path_through_limits_and_episode_ends(gpu_device):
    """... and what the lane kernel does NOT integrate itself -- envs off balance, at the current limit (Q_ref pushed to it), with a
    sagging DC link -- goes to the 23-state lane path in the same launch: against the ETDRK4 lane kernel on the same batch to the
    two paths' tolerance for the envs that stay, and to 2e-5 / bit-comparable rewards for the envs both send down the 23-state path;
    same-step auto-reset through two episode ends, every env restarts twice at its own steady state, statistics = plain sums."""
    N = 1000
    rng = np.random.default_rng(5)
    lp = {"Np": rng.integers(10, 12, N).astype(np.float64), "C_actual": 300e-6 * rng.uniform(0.8, 1.3, N),
          "Kp_Q": 0.2 * rng.uniform(0.8, 1.25, N)}
    runs = {}
    for integ in ("etdlin", "etdrk4"):
        env = make_vec(N, gpu_device, seed=7, auto_reset=True, integrator=integ, lane_params=lp, events_spec=None, max_sim_time=10.0)
        env.reset()
        env.fstate[6, 1::50] *= 1.02                      # off balance
        env.fstate[env.ns, 7::40] = 1.0                   # Q_ref 1 pu: the current limit within a few env steps
        gen = torch.Generator(device=gpu_device); gen.manual_seed(3)
        rows, done_count = [], torch.zeros(N, device=gpu_device)
        for t in range(85):
            a = torch.randint(0, 5, (N,), generator=gen, device=gpu_device, dtype=torch.uint8)
            obs, rew, done, _ = env.step(a)
            done_count += done.float()
            st = env.step_stats()
            assert st["n_nonfinite"] == 0 and st["n_done"] == int(done.sum().item())
            assert abs(st["reward_sum"] - float(rew.sum().item())) < 1e-3 * max(1.0, abs(float(rew.sum().item())))
            rows.append(obs.cpu().numpy().astype(np.float64))
        assert bool((done_count == 2).all()) and bool(torch.isfinite(env.fstate).all())
        runs[integ] = np.stack(rows)
    d = rel_l2(runs["etdlin"][:, :, :10], runs["etdrk4"][:, :, :10], axis=2)
    hard = np.zeros(N, dtype=bool)
    hard[1::50] = True; hard[7::40] = True               # (a 1 pu Q_ref step, a 2 % imbalance: two integrators' own errors meet here)
    assert d[:, ~hard].max() < 2e-4, (np.unravel_index(d.argmax(), d.shape), d[:, ~hard].max())
    assert d[:, hard].max() < 2e-3, d[:, hard].max()
    assert np.median(d) < 3e-5


def test_lane_sets_beyond_the_validated_spread_fail_safe_to_etdrk4(gpu_device, caplog):
    """VERDICT r5 #4 / ADVICE r5 #2 (config_der.json:2-21): the etdlin lane path keeps the group's exponential operator and is validated
    up to C x 0.33-3, outer PI gains x 0.5-2 and PV array x insolation scale 0.8-1.0 (profiles/r05_lane_etdlin_envelope.txt).  Sets
    beyond EITHER envelope do not step with a scheme known to be 1e-3 off: the default constructor selects the ETDRK4 lane kernels
    by itself and says so in one INFO line; removing the sets restores the integrator the user asked for."""
    import logging
    N = 8
    base_c = 300e-6
    with caplog.at_level(logging.INFO, logger="gym_PVDER_amd"):
        env = make_vec(N, gpu_device, lane_params={"C_actual": base_c * np.linspace(0.5, 2.0, N), "Kp_Q": 0.2 * np.linspace(0.8, 1.25, N)})
    assert "validated" not in caplog.text and env._lane_steps_auto == 6 and env.integrator == "etdlin"
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="gym_PVDER_amd"):
        env = make_vec(N, gpu_device, lane_params={"C_actual": base_c * np.linspace(0.2, 1.0, N), "Ki_DC": -10.0 * np.linspace(0.9, 1.1, N)})
    assert "validated" in caplog.text and "x 5.00" in caplog.text and "etdrk4" in caplog.text
    assert len([r for r in caplog.records if "validated" in r.getMessage()]) == 1 and caplog.records[-1].levelno == logging.INFO
    assert env.integrator == "etdrk4"
    env.reset()
    env.step(torch.zeros(N, dtype=torch.uint8, device=gpu_device))
    assert env.step_stats()["n_nonfinite"] == 0 and env._plan_integrator == "etdrk4"
    env.set_lane_params(None)
    assert env.integrator == "etdlin"
    # the operating point counts too: 5 strings of 11, or 30 % insolation, is nowhere near where the shared Jacobian was taken
    for lp in ({"Np": np.full(N, 5.0)}, {"Sinsol_scale": np.linspace(0.3, 1.0, N)}):
        caplog.clear()
        with caplog.at_level(logging.INFO, logger="gym_PVDER_amd"):
            env = make_vec(N, gpu_device, lane_params=lp)
        assert env.integrator == "etdrk4" and "PV array x insolation scale" in caplog.text
    env = make_vec(N, gpu_device, lane_params={"Np": np.full(N, 10.0), "Sinsol_scale": np.linspace(0.9, 1.0, N)})
    assert env.integrator == "etdlin" and env._lane_steps_auto == 2


def test_four_times_the_spread_through_the_default_constructor_vs_the_bridge(gpu_device):
    """VERDICT r5 #4 "done": a batch at FOUR times the first golden's spread (C x 0.25-4, outer PI gains x 0.4-2.5;
    tests/golden/bridge_lane_params_x4.npz: 16 sets, each integrated by the bridge from its own steady state) through the DEFAULT
    constructor -- no integrator argument -- matches the per-set bridge < 1e-4 per env step."""
    import bridge_cases
    name = "lane_params_x4"
    g = bridge_cases.load(name)
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    sets = bridge_cases.lane_param_sets(name)
    env = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], lane_params=sets)
    assert env.integrator == "etdrk4"
    obs0 = env.reset().cpu().numpy().astype(np.float64)
    assert np.abs(obs0[:, :10] - g["obs_reset"][:, :10]).max() < 1e-6
    got = _replay(env, acts, gpu_device)
    err = rel_l2(got[:, :, :10], g["obs"][:, :, :10], axis=2)
    assert err.max() < 1e-4, (np.unravel_index(err.argmax(), err.shape), err.max())
    assert np.array_equal(got[:, :, 11], g["reward"])


def test_wider_parameter_spreads_take_more_krogstad_steps(gpu_device):
    """The etdlin lane path picks its number of Krogstad steps per env step from the spread of the sets around the group's values
    (profiles/r05_lane_etdlin_envelope.txt): 2 inside the first golden's spread, 6 at twice that (C x 0.5-2, outer PI gains
    x 0.67-1.5; tests/golden/bridge_lane_params_wide.npz: 24 sets, each integrated by the bridge from its own steady state) --
    where 6 steps hold 1e-4 and 2 steps do not; `lane_steps` overrides the choice."""
    import bridge_cases
    narrow = bridge_cases.lane_param_sets("lane_params")
    env = make_vec(64, gpu_device, lane_params=narrow)
    assert env._lane_steps_auto == 2
    name = "lane_params_wide"
    g = bridge_cases.load(name)
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    sets = bridge_cases.lane_param_sets(name)
    errs = {}
    for steps in (None, 2):
        env = make_vec(c["N"], gpu_device, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], lane_params=sets)
        assert env._lane_steps_auto == 6
        if steps:
            env.lane_steps = steps
        obs0 = env.reset().cpu().numpy().astype(np.float64)
        assert np.abs(obs0[:, :10] - g["obs_reset"][:, :10]).max() < 1e-6
        got = _replay(env, acts, gpu_device)
        errs[steps] = rel_l2(got[:, :, :10], g["obs"][:, :, :10], axis=2).max()
        if not steps:
            assert np.array_equal(got[:, :, 11], g["reward"])
    assert errs[None] < 1e-4, errs
    assert errs[2] > 1.5 * errs[None] and errs[2] > 1e-4, errs

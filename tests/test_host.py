"""CPU tests (-m "not gpu"): host logic of the env mirror, the C ABI surface, the registry."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_vector_matches_oracle():
    from gym_PVDER_amd import der_model
    from oracle import bridge, pvder_model as pm
    for der in ("50", "10"):
        m = pm.build_model(der)
        par = der_model.param_vector(der)
        assert np.array_equal(par[:27], bridge.pack_params(m)[:27])
        assert par[der_model.IDX["VRMS_REF"]] == m.Vrms_ref and par[der_model.IDX["VDC_REF0"]] == m.Vdc_ref0
        assert par[der_model.IDX["PPV_AT_MPP"]] == 0.0 and bridge.pack_params(m)[-1] == 0.0      # model option U5: off by default
        mo = pm.build_model(der, pm.ModelOptions(ppv_uses_vdc=False))
        po = der_model.param_vector(der, options={"ppv_at_mpp": True})
        assert po[der_model.IDX["PPV_AT_MPP"]] == 1.0 and bridge.pack_params(mo)[-1] == 1.0 and np.array_equal(po[:27], par[:27])
        assert np.abs(der_model.steady_state(po, der) - pm.steady_state(mo)).max() < 1e-12      # (the same steady state: Vdc = Vdcmpp0 there)
        y0 = der_model.steady_state(par, der)
        assert np.abs(y0 - pm.steady_state(m)).max() < 1e-12
    with pytest.raises(ValueError):
        der_model.param_vector("999")
    with pytest.raises(ValueError):
        der_model.param_vector("50", options={"no_such_option": True})


def header_functions():
    txt = open(os.path.join(ROOT, "include", "pvder_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pvder_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what include/pvder_hip.h declares."""
    import torch  # noqa: F401  (same load order as the product: torch's HIP runtime first)
    from gym_PVDER_amd import _native
    path = _native.build_native()
    L = ctypes.CDLL(path)
    declared = header_functions()
    assert declared == sorted(_native.SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    agent = sorted(set(re.findall(r"\b(pvder_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(os.path.join(
        ROOT, "include", "pvder_agent.h")).read(), flags=re.S))))
    assert agent == sorted(_native.AGENT_SYMBOLS)
    for sym in agent:
        assert hasattr(L, sym), sym
    assert L.pvder_version() == _native.ABI_VERSION == 18
    from gym_PVDER_amd import der_model
    assert L.pvder_par_count() == len(der_model.PAR)
    assert L.pvder_block_size() % 64 == 0


def test_ctypes_structs_match_header_layout(tmp_path):
    """sizeof/offsetof of the two host structs as gcc sees the header == the ctypes mirrors."""
    from gym_PVDER_amd import _native
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pvder_hip.h"\n#include "pvder_agent.h"\nint main(){'
                   'printf("%zu %zu %zu %zu %zu %zu %d %zu %zu %zu %zu\\n", sizeof(pvder_step_cfg), offsetof(pvder_step_cfg, del_qref),'
                   'offsetof(pvder_step_cfg, atol), sizeof(pvder_event_spec), offsetof(pvder_event_spec, v_min),'
                   'offsetof(pvder_event_spec, s_max), (int)PVDER_PAR_COUNT, sizeof(pvder_autoreset),'
                   'offsetof(pvder_autoreset, seed), offsetof(pvder_autoreset, spec), offsetof(pvder_autoreset, y0));'
                   'printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(pvder_step_cfg, micro_coarse), offsetof(pvder_step_cfg, graded),'
                   'offsetof(pvder_step_cfg, extra_rewards), sizeof(pvder_lin_coef), offsetof(pvder_lin_coef, sched),'
                   'offsetof(pvder_lin_coef, est_guard));'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(pvder_qagent_cfg), offsetof(pvder_qagent_cfg, lr), offsetof(pvder_qagent_cfg, low),'
                   'offsetof(pvder_qagent_cfg, seed), offsetof(pvder_qagent_cfg, env_id_base), offsetof(pvder_qagent_cfg, policy),'
                   'offsetof(pvder_qagent_cfg, hidden));return 0;}')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    S, E, A = _native.StepCfg, _native.EventSpec, _native.AutoReset
    assert vals[:6] == [ctypes.sizeof(S), S.del_qref.offset, S.atol.offset, ctypes.sizeof(E), E.v_min.offset, E.s_max.offset]
    assert vals[7:11] == [ctypes.sizeof(A), A.seed.offset, A.spec.offset, A.y0.offset]
    assert vals[11:13] == [S.micro_coarse.offset, S.graded.offset]
    Lc = _native.LinCoef
    assert vals[13:17] == [S.extra_rewards.offset, ctypes.sizeof(Lc), Lc.sched.offset, Lc.est_guard.offset]
    Qc = _native.QAgentCfg
    assert vals[17:24] == [ctypes.sizeof(Qc), Qc.lr.offset, Qc.low.offset, Qc.seed.offset, Qc.env_id_base.offset, Qc.policy.offset,
                           Qc.hidden.offset]
    from gym_PVDER_amd import der_model
    assert vals[6] == len(der_model.PAR)


def test_argument_validation_mirrors_reference_setters():
    """PVDER_env.py:561-620: ValueError for bad kwargs, clamps with the same bounds."""
    from gym_PVDER_amd.vec_env import EnvSpecMixin, PVDERVec

    class E(EnvSpecMixin):
        pass
    e = E()
    e._set_n(15)
    assert e._sim_time_per_env_step == pytest.approx(0.25) and e._delQref == 375 and e._delVdcref == pytest.approx(0.3)
    e._set_max_sim_time(40.0)
    assert e._max_sim_time == 40.0
    e._set_max_sim_time(1e9)
    assert e._max_sim_time == pytest.approx(500 * 0.25)          # :569-571
    e._set_max_sim_time(0.1)
    assert e._max_sim_time == 1.0                                 # :567-568
    e._set_max_sim_time(None)
    assert e._max_sim_time == pytest.approx(125.0)               # :565
    e._set_n(0)
    assert e._n == 1                                              # clamped with a warning (:608-612)
    e._set_n(None)
    assert e._n == 60                                             # class default (:64, :606)
    for bad in (1.5, "3", True):
        with pytest.raises(ValueError):
            e._set_n(bad)                                         # :614
    with pytest.raises(ValueError):
        e._set_max_sim_time("40")                                 # :575
    with pytest.raises(ValueError):
        e._set_discrete(1)                                        # :598
    with pytest.raises(ValueError):
        e._set_goals(["voltage_regulation", "fly"])               # :586
    e._set_goals(None)
    assert e._goals_list == ["voltage_regulation"]
    # PVDERVec validates BEFORE touching the device; without a GPU it then refuses to fall back
    with pytest.raises(ValueError):
        PVDERVec(0)
    with pytest.raises(ValueError):
        PVDERVec(4, integrator="euler")
    with pytest.raises(ValueError):
        PVDERVec(4, micro_steps=0)
    with pytest.raises(ValueError):
        PVDERVec(4, DISCRETE_REWARD="yes")
    with pytest.raises(RuntimeError):
        PVDERVec(4, device="cpu")                                 # no CPU fallback for the step path


def test_registry_mirrors_gym_registration():
    import gym_PVDER_amd as g
    s = g.spec("PVDER-v0")
    assert s.id == "PVDER-v0" and s.max_episode_steps == 500      # gym_PVDER/__init__.py:4,10
    assert s.kwargs == {'n_sim_time_steps_per_env_step': 15, 'max_sim_time': 40.0, 'DISCRETE_REWARD': True,
                        'goals_list': ['voltage_regulation']}       # gym_PVDER/__init__.py:6-9
    with pytest.raises(KeyError):
        g.spec("PVDER-v1")


def test_spaces_fallback():
    from gym_PVDER_amd.spaces import Box, Discrete
    d, b = Discrete(5, seed=0), Box(-10, 10, (11,), np.float32)
    assert all(d.contains(d.sample()) for _ in range(20)) and not d.contains(5) and not d.contains(1.0) and 3 in d
    assert b.contains(np.zeros(11, dtype=np.float32)) and not b.contains(np.full(11, 11.0)) and not b.contains(np.zeros(10))
    assert not b.contains(np.full(11, np.nan))


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "gym-solarpvder-environment_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "scipy" not in txt, f


def test_no_step_kernel_reads_the_dispatch_packet(tmp_path):
    """Kernel descriptors of the built gfx950 code object: no kernel may request the AQL dispatch
    pointer or use scratch (spills in a kernel that lives on one wave per SIMD).  (A kernel that reads the dispatch packet -- hipcc does
    that silently when it promotes a private array to LDS -- fetches it from host memory at every
    wave start: measured 25 us of workgroup start-up skew per launch, tools/trace_timeline.py.)"""
    import struct
    from gym_PVDER_amd import _native
    path = _native.build_native()
    llvm = "/opt/rocm/lib/llvm/bin"
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([llvm + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    # one offload bundle per .hip source of the library
    whole = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), whole)]
    assert starts
    kds, blobs = [], {}
    for n, st in enumerate(starts):
        part, co = str(tmp_path / ("fat%d.bin" % n)), str(tmp_path / ("lib%d.co" % n))
        open(part, "wb").write(whole[st:starts[n + 1] if n + 1 < len(starts) else len(whole)])
        subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        secs = subprocess.check_output([llvm + "/llvm-readelf", "-SW", co], text=True)
        m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
        blobs[n] = (open(co, "rb").read(), int(m.group(1), 16), int(m.group(2), 16))
        syms = subprocess.check_output([llvm + "/llvm-readelf", "-sW", co], text=True)
        kds += sorted(set((ln.split()[-1], int(ln.split()[1], 16), n) for ln in syms.splitlines()
                          if ln.rstrip().endswith(".kd")))
    assert len(kds) >= 10
    for name, addr, n in kds:
        blob, ro_addr, ro_off = blobs[n]
        kd = blob[ro_off + addr - ro_addr: ro_off + addr - ro_addr + 64]
        private_seg = struct.unpack_from("<I", kd, 4)[0]
        props = struct.unpack_from("<H", kd, 56)[0]          # kernel_code_properties
        assert props & 0x8, (name, "descriptor layout self-check: every kernel here takes kernel arguments")
        # no real scratch use: hipcc sometimes leaves a dead 20-byte frame object behind SGPR spills that it
        # turned into VGPR lanes (no scratch instruction in the kernel; measured: no effect on throughput)
        if "rollout_kernel" in name or "finish_kernel" in name or "rollout_agent_kernel" in name:
            continue            # the fused rollouts: one launch per K env steps, a few spilled loop-carried values
        if "etdlin64" in name or "sat_hybrid_kernel" in name:
            assert not props & 0x2, (name, "ENABLE_SGPR_DISPATCH_PTR")
            continue            # the fp64 tolerance-match kernel calls its step as a real function (0.5 ms kernel: not latency-bound);
            #                     the hybrid kernel of the duty-cycle limit (round 6) keeps its fp64 vectors in private memory -- a kernel
            #                     of its OWN, behind the step kernels, exactly so that THEIR descriptors stay free of scratch (below)
        if private_seg > 32:
            # (step_etdlin_kernel_w1: 68 bytes of frame objects behind SGPR spills that ended up in VGPR lanes / AGPRs) -- the
            # ISA of the kernel must not hold a single scratch access
            body = subprocess.check_output([llvm + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + name[:-3],
                                            str(tmp_path / ("lib%d.co" % n))], text=True)
            assert "v_mfma" in body or "v_fma" in body or "v_pk_fma" in body, (name, "disassembly self-check")
            assert not re.search(r"\b(scratch_|buffer_load|buffer_store)", body), (name, "uses scratch", private_seg)
            assert private_seg <= 128, (name, private_seg)
        assert not props & 0x2, (name, "ENABLE_SGPR_DISPATCH_PTR")


def test_etdrk4_coefficient_block_matches_independent_phi_functions():
    """pvder_etdrk4_coefficients (host code, no GPU): the 3 x 13 complex 2x2 matrices of the 5-stage
    Hochbruck-Ostermann scheme against an independent evaluation -- phi functions of h*Lz from
    scipy.linalg.expm on the augmented matrix -- plus the scheme's own consistency conditions."""
    import torch  # noqa: F401
    from scipy.linalg import expm
    from gym_PVDER_amd import _native, der_model
    L_ = _native.lib() if os.path.exists(_native.LIB_PATH) else ctypes.CDLL(_native.build_native())
    par = der_model.param_vector("50")
    P = der_model.IDX
    iLf = 1.0 / par[P["LF"]]
    kvi = complex(par[P["KVI_RE"]], par[P["KVI_IM"]])
    Z0 = iLf * (kvi + par[P["RF"]]) + 1j
    c0 = iLf * par[P["VDC_REF0"]] / 2.0
    Lz = np.array([[-Z0, c0 * par[P["KP_GCC"]]], [-par[P["WP"]], -par[P["WP"]]]], dtype=complex)

    def phis(h):                       # phi_0..phi_3 of h*Lz
        out = [expm(h * Lz)]
        for k in (1, 2, 3):
            M = np.zeros((2 * (k + 1), 2 * (k + 1)), dtype=complex)
            M[:2, :2] = h * Lz
            for j in range(k):
                M[2 * j:2 * j + 2, 2 * j + 2:2 * j + 4] = np.eye(2)
            out.append(expm(M)[:2, 2 * k:2 * k + 2])
        return out

    micro, coarse = 32, 1
    cfg = _native.StepCfg()
    cfg.n_sub, cfg.micro, cfg.micro_coarse, cfg.graded = 15, micro, coarse, 1
    n = L_.pvder_etdrk4_coef_count()
    assert n == 3 * 13 * 8 + 3 * 4 * 13 * 16           # complex 2x2 block + the switched block (4 anti-windup modes, real 4x4)
    host = (ctypes.c_float * n)()
    par_c = (ctypes.c_double * len(par))(*par)
    assert L_.pvder_etdrk4_coefficients(par_c, ctypes.byref(cfg), host) == 0
    full = np.array(list(host), dtype=np.float64)
    got = full[:3 * 13 * 8].reshape(3, 13, 4, 2)
    got = got[..., 0] + 1j * got[..., 1]                                   # [set, matrix, entry 11,12,21,22]
    # ---- the switched block: [set][mode][matrix][(M0k, M1k) k = 0..3, (M2k, M3k) k = 0..3] on (iR, iI, uR, uI)
    sw = full[3 * 13 * 8:].reshape(3, 4, 13, 2, 4, 2)                      # [set, mode, matrix, row pair, column, row in pair]
    sw = sw.transpose(0, 1, 2, 3, 5, 4).reshape(3, 4, 13, 4, 4)            # -> [.., row, column]
    emb = lambda z: np.array([[z.real, -z.imag], [z.imag, z.real]])        # complex scalar as a real 2x2
    for s in range(3):
        for k in range(13):
            G = got[s, k].reshape(2, 2)
            W = np.block([[emb(G[0, 0]), emb(G[0, 1])], [emb(G[1, 0]), emb(G[1, 1])]])
            assert np.abs(sw[s, 0, k] - W).max() <= 3e-6 * max(np.abs(W).max(), 1e-12) + 1e-12, (s, k)   # mode 0 = the complex block
    # a frozen component of u: its row of Lz zeroed -> phi_0 keeps that component, the other matrices never feed it
    Lr = np.block([[emb(-Z0), emb(complex(c0 * par[P["KP_GCC"]]))], [emb(complex(-par[P["WP"]])), emb(complex(-par[P["WP"]]))]])
    for mode in (1, 2, 3):
        Lm = Lr.copy()
        for bit, row in ((1, 2), (2, 3)):
            if mode & bit:
                Lm[row] = 0.0
        for s, h in enumerate((1.0 / (60 * micro), 1.0 / (60 * coarse), (1.0 - 1.0 / micro) / 60.0)):
            E = expm(h * Lm)
            assert np.abs(sw[s, mode, 1] - E).max() <= 3e-6 * np.abs(E).max()            # EM_E = phi_0(h L_mode)
            M = np.zeros((8, 8)); M[:4, :4] = h * Lm; M[:4, 4:] = np.eye(4)
            f1 = expm(M)[:4, 4:]
            B = sw[s, mode, 6] + sw[s, mode, 11] + sw[s, mode, 12]                       # b1 + b4 + b5 = h phi_1(h L_mode)
            assert np.abs(B - h * f1).max() <= 1e-5 * np.abs(h * f1).max()
            for row in ([2] if mode & 1 else []) + ([3] if mode & 2 else []):
                assert np.allclose(sw[s, mode, 1][row], np.eye(4)[row], atol=1e-7)
                for k in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
                    off = np.delete(sw[s, mode, k][row], row)
                    assert np.abs(off).max() < 1e-7, (mode, s, k)        # a frozen row only sees its own N (which is blanked)
    for s, h in enumerate((1.0 / (60 * micro), 1.0 / (60 * coarse), (1.0 - 1.0 / micro) / 60.0)):
        f, g = phis(h), phis(h / 2)                                        # phi_k(hL), phi_k(hL/2)
        a52 = 0.5 * g[2] - f[3] + 0.25 * f[2] - 0.5 * g[3]
        a54 = 0.25 * g[2] - a52
        a51 = 0.5 * g[1] - 2 * a52 - a54
        want = [g[0], f[0], 0.5 * h * g[1], h * (0.5 * g[1] - g[2]), h * (f[1] - 2 * f[2]), h * a51,
                h * (f[1] - 3 * f[2] + 4 * f[3]), h * g[2], h * f[2], h * a52, h * a54,
                h * (-f[2] + 4 * f[3]), h * (4 * f[2] - 8 * f[3])]      # order of the EM_* enum in pvder_kernels.hip
        for k, W in enumerate(want):
            w = W.reshape(4)
            assert np.abs(got[s, k] - w).max() <= 2e-6 * max(np.abs(w).max(), 1e-12) + 1e-12, (s, k)
        # consistency of the tableau: sum_j a_ij = c_i phi_1(c_i h L) h, sum_j b_j = h phi_1(h L)
        G = got[s].reshape(13, 2, 2)
        assert np.abs(G[3] + G[7] - 0.5 * h * g[1]).max() < 1e-6 * h                     # a31 + a32
        assert np.abs(G[4] + 2 * G[8] - h * f[1]).max() < 1e-6 * h                       # a41 + a42 + a43
        assert np.abs(G[5] + 2 * G[9] + G[10] - 0.5 * h * g[1]).max() < 1e-6 * h         # a51 + 2 a52 + a54
        assert np.abs(G[6] + G[11] + G[12] - h * f[1]).max() < 1e-6 * h                  # b1 + b4 + b5


def test_plan_and_rollout_argument_validation_without_a_gpu():
    """pvder_plan_create folds on the host (no device call); the rollout entry points reject bad arguments
    with PVDER_EINVAL before anything is launched -- fused rollouts exist for etdlin / three-phase only."""
    import torch  # noqa: F401
    from gym_PVDER_amd import _native, der_model
    L = _native.lib()
    C = ctypes
    par = der_model.param_vector("50")
    par_c = (C.c_double * len(par))(*par)
    y0 = der_model.steady_state(par, "50")
    y0_c = (C.c_double * len(y0))(*y0)
    cfg = _native.StepCfg()
    cfg.n_sub, cfg.micro, cfg.max_ksub, cfg.max_steps = 15, 4096, 2400, 500
    cfg.ev_s0, cfg.ev_ds, cfg.ev_count = 60, 60, 38
    cfg.del_qref, cfg.del_vdcref, cfg.inv_max_sim_time = 0.0075, 6e-4, 1.0 / 40.0
    cfg.p_target, cfg.q_target, cfg.rtol, cfg.atol = 0.9, 0.1, 1e-6, 1e-8
    ar = _native.AutoReset()
    ar.enabled, ar.y0 = 1, C.cast(y0_c, C.POINTER(C.c_double))
    ar.spec.ev_count = 38
    h = C.c_void_p()
    rc = L.pvder_plan_create(C.byref(h), _native.INTEGRATOR_IDS["rk4"], par_c, C.byref(cfg), C.byref(ar), None, None, 1024)
    assert rc == 0 and h.value
    one = C.c_void_p(8)               # any non-null value: validation fails before a pointer is touched
    assert L.pvder_plan_rollout_fused(h, one, one, one, one, one, 4, one, 0, None, one, None) == _native.EINVAL   # not etdlin
    assert L.pvder_plan_rollout(h, one, one, one, one, None, 4, one, 0, None, None, None) == _native.EINVAL        # no actions
    assert L.pvder_plan_rollout(h, one, one, one, one, one, -1, one, 0, None, None, None) == _native.EINVAL
    assert L.pvder_plan_rollout(h, one, one, one, one, one, 0, one, 0, None, None, None) == 0                      # nothing to do
    L.pvder_plan_destroy(h)
    assert L.pvder_plan_create(C.byref(h), 9, par_c, C.byref(cfg), C.byref(ar), None, None, 1024) == _native.EINVAL
    assert L.pvder_plan_create(C.byref(h), _native.INTEGRATOR_IDS["etdlin"], par_c, C.byref(cfg), C.byref(ar), None, None, 1024) \
        == _native.EINVAL             # etdlin without coefficient sets
    assert L.pvder_plan_rollout_fused(None, one, one, one, one, one, 4, one, 0, None, one, None) == _native.EINVAL
    # etdlin with coefficient sets but without the propagator of a small imbalance (a caller written against ABI 10)
    from gym_PVDER_amd import etd_linear
    cfg.micro = 32
    lin = _native.LinCoef()
    lin.sets, lin.sched, lin.n_sets, lin.n_sched_rows, lin.est_guard = 8, 8, 3, 30, etd_linear.EST_GUARD
    for k in range(3):
        lin.len_ticks[k] = 1
    assert L.pvder_plan_create(C.byref(h), _native.INTEGRATOR_IDS["etdlin"], par_c, C.byref(cfg), C.byref(ar), one, C.byref(lin), 1024) \
        == _native.EINVAL
    E = etd_linear.seq_propagator(par, "50", cfg.n_sub)
    for r in range(3):
        for k in range(3):
            lin.seq_prop[6 * r + 2 * k], lin.seq_prop[6 * r + 2 * k + 1] = E[r, k].real, E[r, k].imag
    assert abs(lin.seq_prop[8] - np.exp(-0.25 / 3.0)) < 1e-6
    assert L.pvder_plan_create(C.byref(h), _native.INTEGRATOR_IDS["etdlin"], par_c, C.byref(cfg), C.byref(ar), one, C.byref(lin), 1024) == 0
    L.pvder_plan_destroy(h)


def test_parameter_set_overrides_match_the_oracle_and_are_validated():
    """der_config overrides (a parameter set of a domain-randomised batch): the product's parameter vector and
    steady state against the oracle's model built with the same overrides; unknown keys / bad values raise."""
    from gym_PVDER_amd import der_model
    from oracle import bridge, pvder_model as pm
    ov = {"C_actual": 360e-6, "Kp_DC": -2.5, "Ki_Q": 12.0, "Np": 10, "Rf_actual": 0.003}
    par = der_model.param_vector("50", ov)
    m = pm.build_model("50", overrides=ov)
    assert np.array_equal(par[:27], bridge.pack_params(m)[:27])
    assert np.abs(der_model.steady_state(par, "50") - pm.steady_state(m)).max() < 1e-12
    assert not np.array_equal(par, der_model.param_vector("50"))
    assert np.array_equal(der_model.param_vector("50", {}), der_model.param_vector("50"))
    for bad in ({"nope": 1.0}, {"phases": 1}, {"C_actual": float("nan")}, {"C_actual": "big"}, {"Np": True}):
        with pytest.raises(ValueError):
            der_model.param_vector("50", bad)


def test_lane_parameter_folding_on_the_host():
    """pvder_fold_lane_params (no GPU): per-env values of the parameters outside the exponential operator, folded by the
    same code as a plan's own constants -- against der_model.param_vector / etd_linear.folded of each env's config block."""
    import ctypes as C
    from gym_PVDER_amd import _native, der_model, etd_linear
    L = _native.lib()
    N = 5
    rng = np.random.default_rng(0)
    base = der_model.DER_CONFIG["50"]
    sets = {"C_actual": base["C_actual"] * rng.uniform(0.5, 2.0, N), "Kp_DC": base["Kp_DC"] * rng.uniform(0.5, 2.0, N),
            "Ki_Q": base["Ki_Q"] * rng.uniform(0.5, 2.0, N), "Np": np.array([8.0, 9.0, np.nan, 11.0, 10.0]),
            "Sinsol_scale": np.array([1.0, 0.5, 0.9, np.nan, 0.8])}
    rows = np.ascontiguousarray(der_model.lane_param_rows(sets, N))
    assert rows.shape == (7, N) and np.isnan(rows[2]).all() and np.isnan(rows[5, 2])
    for opts in (None, {"ppv_at_mpp": True}):
        par = der_model.param_vector("50", None, opts)
        out = np.empty((_native.LPF_ROWS, N), dtype=np.float32)
        assert L.pvder_fold_lane_params((C.c_double * len(par))(*par), rows.ctypes.data, N, out.ctypes.data) == 0
        for e in range(N):
            ov = {k: float(v[e]) for k, v in sets.items() if k != "Sinsol_scale" and not np.isnan(v[e])}
            c = etd_linear.folded(der_model.param_vector("50", ov, opts))
            sc = 1.0 if np.isnan(sets["Sinsol_scale"][e]) else sets["Sinsol_scale"][e]
            want = [c["iC"], c["KpDC"], c["KiDC"], c["KpQ"], c["KiQ"]]
            np.testing.assert_allclose(out[:5, e], want, rtol=2e-7)
            if not opts:
                np.testing.assert_allclose(out[5:7, e], [c["pvA"] * sc, c["pvB"]], rtol=2e-7)
            else:       # model option U5: pv1 = pvA * Sinsol + pvB = -Ppv(Vdcmpp0) -- linear in the photo current, hence in the scale
                v0, e0 = c["vdc0"], np.exp(c["pvK"] * c["vdc0"])
                np.testing.assert_allclose(out[5:7, e], [-v0 * c["pvA"] * sc, -v0 * c["pvB"] * (1.0 - e0)], rtol=2e-6)
            # rows 7-10 (ABI 15): the env's OWN steady state -- the Newton root of der_model.steady_state for its config block
            # at the insolation it sees (the reference: steadyStateInitialization=True, PVDER_env.py:374-378)
            pe = der_model.param_vector("50", ov, opts)
            pe[der_model.IDX["SINSOL0"]] *= sc
            y0 = der_model.steady_state(pe, "50")
            np.testing.assert_allclose(out[7:, e], y0[:4], rtol=1e-6)
    bad = rows.copy(); bad[0, 1] = 0.0
    out = np.empty((_native.LPF_ROWS, N), dtype=np.float32)
    assert L.pvder_fold_lane_params((C.c_double * len(par))(*par), bad.ctypes.data, N, out.ctypes.data) == _native.EINVAL
    bad = rows.copy(); bad[6, 0] = np.inf                        # (an infinite insolation scale: non-finite PV constants)
    assert L.pvder_fold_lane_params((C.c_double * len(par))(*par), bad.ctypes.data, N, out.ctypes.data) == _native.EINVAL
    with pytest.raises(ValueError):
        der_model.lane_param_rows({"Rf_actual": np.ones(N)}, N)
    with pytest.raises(ValueError):
        der_model.lane_param_rows({"Np": np.ones(N + 1)}, N)


def test_lane_envelope_decision_on_the_host():
    """PVDERVec._lane_envelope (no GPU): the number of Krogstad steps inside the validated envelope of the etdlin lane path, None
    (= fail safe to the ETDRK4 lane kernels) beyond it -- spread of C / the outer PI gains AND of the operating point
    (Np x Sinsol_scale)."""
    from gym_PVDER_amd import der_model
    from gym_PVDER_amd.vec_env import PVDERVec
    N = 4
    v = PVDERVec.__new__(PVDERVec)
    object.__setattr__(v, "num_envs", N); object.__setattr__(v, "der_id", "50"); object.__setattr__(v, "der_config", None)
    rows = lambda lp: der_model.lane_param_rows(lp, N)
    assert v._lane_envelope(rows({"C_actual": 300e-6 * np.array([0.7, 1.0, 1.2, 1.4])})) == (2, None)
    assert v._lane_envelope(rows({"C_actual": 300e-6 * np.array([0.5, 1.0, 1.2, 2.0]), "Kp_Q": 0.2 * np.array([0.67, 1, 1, 1.5])}))[0] == 6
    assert v._lane_envelope(rows({"Ki_DC": -10.0 * np.array([0.5, 1.0, 1.2, 2.0])}))[0] == 8
    k, why = v._lane_envelope(rows({"Ki_DC": -10.0 * np.array([0.4, 1.0, 1.2, 2.0])}))
    assert k is None and "x 2.50" in why
    assert v._lane_envelope(rows({"Np": np.array([10.0, 11, 11, 10]), "Sinsol_scale": np.array([0.9, 1.0, 0.95, 1.0])})) == (2, None)
    for lp in ({"Np": np.array([5.0, 11, 11, 11])}, {"Sinsol_scale": np.array([0.3, 1, 1, 1])}, {"Np": np.array([12.0, 11, 11, 11])},
               {"Np": np.array([10.0, 11, 11, 11]), "Sinsol_scale": np.array([0.85, 1, 1, 1])}):
        k, why = v._lane_envelope(rows(lp))
        assert k is None and "PV array" in why, lp
    assert v._lane_envelope(rows({"Kp_DC": np.array([2.0, -2, -2, -2])}))[0] is None       # a sign flip is no "spread"
    assert v._lane_envelope(rows({"C_actual": np.array([np.nan, 300e-6, 300e-6, 300e-6])})) == (2, None)   # NaN = the group's value

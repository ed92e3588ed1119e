"""CPU checks of the C-ABI boundary: the library loads, exports every symbol the header declares,
and the host-side structs/layout agree with it. No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from avddpg_amd import _hip, config, dynamics, params


def _declared():
    text = open(_hip.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avd_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _hip.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/avddpg_hip.h but not exported"
    assert lib.avd_version() >= 1
    # every int-returning entry point has a ctypes prototype in the binding
    assert set(names) - {"avd_last_error", "avd_version", "avd_diagnostics_enabled"} == set(_hip._PROTOS)


def test_dynamic_symbol_table_is_exactly_the_declared_c_abi():
    """built with -fvisibility=hidden + a linker version script, the library's dynamic symbol table holds the
    `avd_*` entry points of include/avddpg_hip.h and nothing else -- no mangled C++ helper, launcher or kernel handle."""
    import subprocess

    for path in (_hip.LIB_PATH, _hip.DIAG_LIB_PATH):
        if not os.path.exists(path):
            continue
        nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True)
        syms = sorted(line.split()[-1] for line in nm.stdout.splitlines() if line.strip())
        assert syms and all(s.startswith("avd_") for s in syms), [s for s in syms if not s.startswith("avd_")]
        extra = set(syms) - set(_declared())
        assert all(s.startswith("avd_debug_") for s in extra), extra  # (phase-timing read-outs of diagnostic builds only)
        if path == _hip.LIB_PATH:
            assert set(syms) == set(_declared())
    # the f16x3 name is the real one; r03's bf16x3 stays as an alias with the same prototype
    assert _hip._PROTOS["avd_learn_set_split_bf16x3"] == _hip._PROTOS["avd_learn_set_split_f16x3"]


def test_shipped_library_reads_no_environment_switch():
    """every diagnostic switch (kernel / tile choices, work-skipping ablations) is compiled in only under
    -DAVD_DIAG (`make diag` -> libavddpg_hip_diag.so). The product library holds no "AVD_" string at all, does not import
    getenv, and says so through the C ABI; the diagnostic build holds the switches and says THAT."""
    import subprocess

    data = open(_hip.LIB_PATH, "rb").read()
    assert b"AVD_" not in data, sorted(set(re.findall(rb"AVD_[A-Z0-9_]+", data)))
    assert _hip.lib().avd_diagnostics_enabled() == 0
    nm = subprocess.run(["nm", "-D", "--undefined-only", _hip.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert not re.search(r"\bgetenv\b", nm.stdout), "the shipped library imports getenv"
    if os.path.exists(_hip.DIAG_LIB_PATH):
        diag = open(_hip.DIAG_LIB_PATH, "rb").read()
        assert b"AVD_FSPLIT_J" in diag and b"AVD_LEARN_KERNEL" in diag
        with _hip.diag_library() as d:
            assert d.avd_diagnostics_enabled() == 1
        assert _hip.lib().avd_diagnostics_enabled() == 0  # (the context manager restored the product library)


def test_error_reporting_without_gpu():
    with pytest.raises(_hip.AvdError, match="non-positive"):
        _hip.make_layout(4, 1, 0, 128, 48, 64)
    # argument validation happens before any HIP call
    with pytest.raises(_hip.AvdError, match="L must be"):
        _hip.call("avd_env_step_f32", None, 4, 99, None, None, None, None, None, None, None, None, None, None, None, None)
    lay = _hip.make_layout(12, 3, 307, 153, 57, 64)  # unpadded centralized widths: refused, with the remedy named
    with pytest.raises(_hip.AvdError, match="pad the widths"):
        _hip.call("avd_learn_f32", ctypes.byref(lay), 1, 0, *([None] * 8), 0.99, 2.5, None, None, None)


def test_layout_matches_reference_parameter_counts():
    lay = _hip.make_layout(4, 1, 256, 128, 48, 64)
    # SURVEY a-9/a-10: 35 073 actor + 41 409 critic trainables, 768 + 864 moving stats
    d = params.logical_dims(lay)
    a = sum(int(np.prod(shp(d))) for _, k, shp in params.ACTOR_WEIGHTS if k == "t")
    c = sum(int(np.prod(shp(d))) for _, k, shp in params.CRITIC_WEIGHTS if k == "t")
    assert (a, c) == (35073, 41409)
    assert lay.actor_size == 35076 and lay.theta_size == 35076 + 41412 and lay.stats_size == 1632
    assert len(params.ACTOR_WEIGHTS) == 14 and len(params.CRITIC_WEIGHTS) == 20
    for f in _hip._LAYOUT_FIELDS[6:]:
        assert getattr(lay, f) % 4 == 0


def test_pack_unpack_roundtrip_and_init_bounds():
    lay = _hip.make_layout(3, 1, 64, 32, 16, 64)
    th, st = params.init_weights(lay, np.random.RandomState(0))
    aw = params.unpack(lay, th, st, "actor")
    cw = params.unpack(lay, th, st, "critic")
    assert [w.shape for w in aw][:2] == [(3, 64), (64,)] and cw[12].shape == (64 + 16, 32)
    assert np.abs(aw[0]).max() <= 1 / np.sqrt(64) and np.abs(aw[6]).max() <= 1 / np.sqrt(32)
    assert np.abs(aw[12]).max() <= 0.003 and np.abs(cw[18]).max() <= 0.0003
    assert np.abs(cw[2]).max() <= 1 / np.sqrt(32)  # action layer uses layer2_init (agent/model.py:70)
    assert np.all(aw[2] == 1) and np.all(aw[5] == 1) and np.all(aw[4] == 0) and np.all(aw[1] == 0)
    th2, st2 = np.zeros_like(th), np.zeros_like(st)
    params.pack(lay, aw, th2, st2, "actor")
    params.pack(lay, cw, th2, st2, "critic")
    assert np.array_equal(th, th2) and np.array_equal(st, st2)
    with pytest.raises(ValueError):
        params.pack(lay, aw[:-1], th2, st2, "actor")


def test_env_consts_match_oracle_matrices():
    from oracle import platoon

    for method in ("euler", "exact"):
        conf = config.Config(method=method, dyn_coeff=0.25, pl_leader_tau=0.15, timegap=0.8, sample_rate=0.05)
        c = dynamics.env_consts(conf, 5)
        ep = platoon.EnvParams(method=method, dyn_coeff=0.25, pl_leader_tau=0.15, timegap=0.8, sample_rate=0.05)
        A, B, Cm = platoon.platoon_matrices(ep, 5)
        for i in range(5):
            assert np.array_equal(np.array(list(c.A[i]), dtype=np.float32), A[i].astype(np.float32).ravel())
            assert np.array_equal(np.array(list(c.B[i]), dtype=np.float32), B[i].astype(np.float32))
            assert np.array_equal(np.array(list(c.C[i]), dtype=np.float32), Cm[i].astype(np.float32))
    assert ctypes.sizeof(_hip.EnvConsts) == 4 * 4 + 20 * 4 + 16 * 24 * 4
    with pytest.raises(ValueError):
        dynamics.env_consts(config.Config(), 17)


def test_config_defaults_and_derived_counts():
    c = config.Config()
    assert (c.steps_per_episode, c.number_of_episodes, c.fed_update_delay_steps) == (600, 1666, 1)
    assert c.fed_enabled is False and config.Config(fed_method="interfrl").fed_enabled is True
    with pytest.raises(AttributeError):
        config.Config(not_a_field=1)


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "avddpg_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_philox_known_answers():
    """Random123 Philox4x32-10 known-answer vectors for the oracle's restatement of the device RNG."""
    from oracle import philox
    z = np.array([0])
    assert [int(v[0]) for v in philox.philox4x32_10(z, z, z, z, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    o = np.array([0xffffffff])
    assert [int(v[0]) for v in philox.philox4x32_10(o, o, o, o, 0xffffffff, 0xffffffff)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    idx = philox.replay_indices(3, 64, 1000, seed=1, counter=0)
    assert idx.shape == (3, 64) and idx.dtype == np.int32 and idx.min() >= 0 and idx.max() < 1000


def test_padded_pack_unpack_roundtrip():
    """Centralized widths (int(256*1.2), int(128*1.2), int(48*1.2) = 307/153/57) live in slabs padded to 320/160/64:
    pack -> unpack is the identity on the logical tensors and the padding holds its neutral values."""
    dims = params.Dims(12, 3, 307, 153, 57)
    H1p, H2p, Hap = params.padded_widths(dims.H1, dims.H2, dims.Ha)
    assert (H1p, H2p, Hap) == (320, 160, 64)
    lay = _hip.make_layout(12, 3, H1p, H2p, Hap, 64)
    th, st = params.init_weights(lay, np.random.RandomState(0), nominal=(256, 128), dims=dims)
    for which, n_all, n_tr in (("actor", 14, 10), ("critic", 20, 14)):
        w = params.unpack(lay, th, st, which, dims=dims)
        assert len(w) == n_all
        th2, st2 = np.full_like(th, 7.0), np.full_like(st, 7.0)
        params.pack(lay, w, th2, st2, which, dims=dims)
        w2 = params.unpack(lay, th2, st2, which, dims=dims)
        for x, y in zip(w, w2):
            assert x.shape == y.shape and np.array_equal(x, y)
        assert len(params.unpack(lay, th, st, which, trainable_only=True, dims=dims)) == n_tr
    cw = params.unpack(lay, th, st, "critic", dims=dims)
    assert cw[12].shape == (307 + 57, 153) and cw[0].shape == (12, 307) and cw[18].shape == (153, 3)
    # padded columns of W1 are zero, padded gamma/var are one
    aW1 = th[lay.aW1:lay.aW1 + 12 * 320].reshape(12, 320)
    assert np.all(aW1[:, 307:] == 0) and np.all(aW1[:, :307] != 0)
    assert np.all(th[lay.ag1 + 307:lay.ag1 + 320] == 1) and np.all(st[lay.amv1 + 307:lay.amv1 + 320] == 1)
    cW2 = th[lay.actor_size + lay.cW2:lay.actor_size + lay.cW2 + 384 * 160].reshape(384, 160)
    assert np.all(cW2[307:320] == 0) and np.all(cW2[320 + 57:] == 0) and np.all(cW2[:, 153:] == 0)
    assert np.array_equal(cW2[320:377, :153], cw[12][307:])


def test_shared_learner_argument_validation_without_gpu():
    """check_wide runs before any HIP call: unsupported widths and bad agent counts are refused with the rule named."""
    need = ctypes.c_size_t(0)
    lay = _hip.make_layout(4, 1, 320, 160, 48, 64)
    with pytest.raises(_hip.AvdError, match="multiples of 64"):
        _hip.call("avd_learn_shared_workspace", ctypes.byref(lay), 10, 5, ctypes.byref(need))
    lay = _hip.make_layout(4, 1, 1024, 1024, 48, 64)
    with pytest.raises(_hip.AvdError, match="multiple of n_sets"):
        _hip.call("avd_learn_shared_workspace", ctypes.byref(lay), 11, 5, ctypes.byref(need))
    _hip.call("avd_learn_shared_workspace", ctypes.byref(lay), 20480, 5, ctypes.byref(need))
    assert 20 * 2**30 < need.value < 40 * 2**30  # BASELINE config 5: ~26 GiB of scratch, fits 288 GB beside the replay ring
    with pytest.raises(_hip.AvdError, match="K %"):
        _hip.call("avd_gemm_bt_bf16", 128, 128, 100, None, 100, None, 100, None, 128, None)


def test_federated_server_keeps_the_reference_class_api():
    """src/server/federated.py:13-16, 18, 69: Server(name, debug_enabled), get_avg_params(system_params),
    get_weighted_avg_params(system_params, weight_sums). No compute here (no GPU): signatures only."""
    import inspect

    from avddpg_amd import federated

    assert list(inspect.signature(federated.Server.__init__).parameters)[:3] == ["self", "name", "debug_enabled"]
    assert list(inspect.signature(federated.Server.get_avg_params).parameters) == ["self", "system_params"]
    assert list(inspect.signature(federated.Server.get_weighted_avg_params).parameters) == ["self", "system_params", "weight_sums"]
    assert "oracle" not in open(federated.__file__).read().split('"""', 2)[2]


def test_learn_update_plan_describes_the_centralized_pipeline_and_single_launch_shapes():
    """avd_learn_update_plan (bench.py described the centralized pipeline with literals of the profiled box): chunks of one
    learn workgroup per CU, at most 32 chunks (the chunk grows instead); every other shape is ONE launch. Without a GPU the CU count
    falls back to 256."""
    import ctypes as C

    cen = _hip.make_layout(20, 5, 320, 160, 64, 64)   # centralized L = 5: S = 4 L, A = L, widths x 1.2 padded
    dec = _hip.make_layout(4, 1, 256, 128, 48, 64)
    ch, n, g = C.c_int(0), C.c_int(0), C.c_int(0)
    _hip.call("avd_learn_update_plan", C.byref(cen), 4096, C.byref(ch), C.byref(n), C.byref(g))
    assert ch.value >= 1 and n.value == (4096 + ch.value - 1) // ch.value and n.value <= 32 and g.value >= 1
    _hip.call("avd_learn_update_plan", C.byref(cen), 100000, C.byref(ch), C.byref(n), C.byref(g))
    assert n.value <= 32 and ch.value * n.value >= 100000
    _hip.call("avd_learn_update_plan", C.byref(dec), 20480, C.byref(ch), C.byref(n), C.byref(g))
    assert (ch.value, n.value, g.value) == (20480, 1, 0)
    with pytest.raises(_hip.AvdError):
        _hip.call("avd_learn_update_plan", C.byref(dec), 0, C.byref(ch), C.byref(n), C.byref(g))


def test_bench_default_run_declares_every_single_gpu_baseline_config():
    """bench.py's plain 1-GPU run must carry BASELINE configs[2] (both federated modes) and configs[4] beside the two configs[1]
    workloads: the keys of also_measured are a module constant, each naming the BASELINE.json entry it measures,
    and the run that emits them is the flag-less one (tests/test_gpu_bench_ranks.py runs it on the GPU). No GPU, no torch here."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert set(bench.EXTRA_CONFIGS) == {"config3_interfrl", "config3_nofrl", "config5"}
    configs = json.load(open(os.path.join(root, "BASELINE.json")))["configs"]
    assert "4096 platoons × 10 vehicles" in configs[2] and "hidden=1024" in configs[4]
    assert all("configs[2]" in bench.EXTRA_CONFIGS[k] and "10 vehicles" in bench.EXTRA_CONFIGS[k] for k in ("config3_interfrl", "config3_nofrl"))
    assert "configs[4]" in bench.EXTRA_CONFIGS["config5"] and "1024" in bench.EXTRA_CONFIGS["config5"]
    src = open(os.path.join(root, "bench.py")).read()
    assert "default_run" in src and "--no-extra-configs" in src and 'out.setdefault("also_measured", {}).update(' in src
    assert bench.PRIMARY_MODE == "interfrl"

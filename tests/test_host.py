"""CPU: host logic of the product library (no compute kernels are called)."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

from tests import helpers as H

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    """The C-ABI library loads and exports every function include/fjsp_amd.h declares."""
    from deep_reinforcement_learning_for_fjsp_amd import _capi
    from deep_reinforcement_learning_for_fjsp_amd._build import LIB_PATH
    hdr = open(os.path.join(REPO, "include", "fjsp_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fjsp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = C.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _capi.lib().fjsp_abi_version() == 1


def test_oracle_is_not_linked_into_the_product(built):
    """The product library must not contain the oracle (no CPU fallback in the product path)."""
    from deep_reinforcement_learning_for_fjsp_amd._build import LIB_PATH, PKG_DIR
    blob = open(LIB_PATH, "rb").read()
    assert b"fjo_step" not in blob and b"fjo_reset" not in blob
    for root, _, files in os.walk(PKG_DIR):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "pyoracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_missing_library_fails_loudly(built, monkeypatch):
    from deep_reinforcement_learning_for_fjsp_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libfjsp_amd.so")
    with pytest.raises(ImportError):
        _capi.lib()


def test_env_create_without_gpu_or_bad_args_errors(built):
    """No device / bad arguments -> an error code, never a silent CPU path."""
    import torch
    from deep_reinforcement_learning_for_fjsp_amd import _capi, instances as fi
    s = fi.InstanceSet(1).generate(0, 1, fi.bench_10x5_params())
    h = C.c_void_p()
    lib = _capi.lib()
    # fluid solution missing
    assert lib.fjsp_env_create(s.handle, 0, 1, 4, 0, 0, 0, C.byref(h)) == -7
    s.solve_fluid()
    if not torch.cuda.is_available():
        assert lib.fjsp_env_create(s.handle, 0, 1, 4, 0, 0, 0, C.byref(h)) == -6
        assert b"no HIP device" in lib.fjsp_last_error()
        from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
        with pytest.raises(RuntimeError):
            EnvBatch(s, 4)


def test_subclass_variants_refuse_multi_order_instances(built):
    """SO_SFJSP / MO_FJSSP_discretes are single-order in the reference (they only ever add order_dict[0]);
    the library refuses to build them on an instance with several orders instead of silently dropping orders."""
    from deep_reinforcement_learning_for_fjsp_amd import _capi
    insts, _, _ = H.load_suite("multiorder")
    s = H.instance_set_from(insts[1:2])
    h = C.c_void_p()
    for variant in (1, 2):
        assert _capi.lib().fjsp_env_create(s.handle, 0, 1, 2, variant, 0, 0, C.byref(h)) == -5
        assert b"single-order" in _capi.lib().fjsp_last_error()


def _write_csv(arr, folder):
    import csv
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "based_data.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["kind_count", "machine_count", "order_count", "DDT"])
        w.writerow([arr.R, arr.M, arr.S, arr.ddt])
    koff = np.concatenate(([0], np.cumsum(arr.Jr)))
    with open(os.path.join(folder, "process_data.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["kind", "task", "machine_selectable", "process_time"])
        for r in range(arr.R):
            for j in range(int(arr.Jr[r])):
                k = int(koff[r]) + j
                ms = tuple(int(m) for m in arr.elig_list[k, :arr.elig_n[k]])
                w.writerow([r, j, ms, tuple(int(arr.p[k, m]) for m in ms)])
    with open(os.path.join(folder, "order_data.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["order", "time_arrive", "time_delivery", "kind_number"])
        for s in range(arr.S):
            w.writerow([s, int(arr.arrive[s]), int(arr.delivery[s]), tuple(int(c) for c in arr.count[s])])


def test_csv_loader_roundtrip(built, tmp_path):
    """Arrays -> reference CSV folder format -> loader gives the arrays back (incl. `\\d+` DDT parsing)."""
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    insts, _, _ = H.load_suite("multijob")
    for i, a in enumerate(insts):
        _write_csv(a, str(tmp_path / ("I%d" % i)))
        b = fi.InstanceSet(1).load_csv(0, str(tmp_path), "I%d" % i).arrays(0)
        for key in ("Jr", "p", "elig_n", "count", "arrive", "delivery"):
            assert np.array_equal(getattr(a, key), getattr(b, key)), key
        for k in range(a.K):
            assert np.array_equal(a.elig_list[k, :a.elig_n[k]], b.elig_list[k, :b.elig_n[k]])
        assert b.ddt == float(int(a.ddt))      # "0.5" -> 0, "1.0" -> 1  (SO_DFJSP_instance_read.py:36-40)


def test_csv_loader_errors(built, tmp_path):
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd._capi import FjspError
    with pytest.raises(FjspError) as ei:
        fi.InstanceSet(1).load_csv(0, str(tmp_path), "missing")
    assert ei.value.code == -2
    insts, _, _ = H.load_suite("mk01")
    _write_csv(insts[0], str(tmp_path / "bad"))
    with open(str(tmp_path / "bad" / "process_data.csv"), "a") as f:
        f.write('0,9,"(0,)","(3,)"\n')          # operation numbering gap
    with pytest.raises(FjspError) as ei:
        fi.InstanceSet(1).load_csv(0, str(tmp_path), "bad")
    assert ei.value.code == -3


def test_generator_is_seeded_and_in_range(built):
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    prm = fi.bench_10x5_params()
    s = fi.InstanceSet(3).generate(0, 1000, prm).generate(1, 1000, prm).generate(2, 1001, prm)
    a, b, c = s.arrays(0), s.arrays(1), s.arrays(2)
    assert np.array_equal(a.p, b.p) and np.array_equal(a.elig_list, b.elig_list) and not np.array_equal(a.p, c.p)
    assert a.R == 10 and a.M == 5 and 30 <= a.K <= 50 and (a.count == 1).all()
    assert a.p.max() <= 20 and ((a.p > 0).sum(1) == a.elig_n).all() and (a.elig_n >= 1).all()
    # delivery = int(sum_k mean_m p * count * DDT / (2M))  (Instance_generate.py:81-88)
    mean = np.array([a.p[k, a.elig_list[k, :a.elig_n[k]]].sum() / a.elig_n[k] for k in range(a.K)])
    acc = 0.0
    for v in mean:
        acc = acc + v * 1
    assert a.delivery[0] == int(acc * 1.0 / 10)
    # stored synthetic fixtures are exactly generator(seed 1000 + i)
    insts, _, _ = H.load_suite("synth10x5")
    g = fi.InstanceSet(len(insts)).generate_range(1000, prm)
    for i, want in enumerate(insts):
        got = g.arrays(i)
        assert np.array_equal(got.p, want.p) and np.array_equal(got.elig_list, want.elig_list), i
    big = fi.InstanceSet(1).generate(0, 7, fi.reference_generator_params(1.0, 15, 3)).arrays(0)
    assert 3 <= big.R <= 12 and big.S == 3 and 40 <= big.p[big.p > 0].min() and big.p.max() <= 400
    assert (big.count >= 5).all() and (big.count <= 50).all() and big.arrive[0] == 0


def _lp_highs(a, Q, now):
    """Independent formulation of class_FJSSP.py:246-280 solved by scipy/HiGHS."""
    from scipy.optimize import linprog
    K, M = a.p.shape
    cols = [(k, m) for k in range(K) for m in range(M) if a.p[k, m] > 0]
    idx = {km: i for i, km in enumerate(cols)}
    n = len(cols) + 1
    A, b = [], []
    for k in range(K):
        row = np.zeros(n); row[-1] = 1.0
        for m in range(M):
            if a.p[k, m] > 0:
                row[idx[(k, m)]] = -(1.0 / a.p[k, m]) / Q[k]
        A.append(row); b.append(0.0)
    for m in range(M):
        row = np.zeros(n)
        for k in range(K):
            if a.p[k, m] > 0:
                row[idx[(k, m)]] = 1.0
        A.append(row); b.append(1.0)
    koff = np.concatenate(([0], np.cumsum(a.Jr)))
    for r in range(len(a.Jr)):
        for j in range(int(a.Jr[r]) - 1):
            k = int(koff[r]) + j
            if now[k + 1] == 0:
                row = np.zeros(n)
                for m in range(M):
                    if a.p[k + 1, m] > 0:
                        row[idx[(k + 1, m)]] += 1.0 / a.p[k + 1, m]
                    if a.p[k, m] > 0:
                        row[idx[(k, m)]] -= 1.0 / a.p[k, m]
                A.append(row); b.append(0.0)
    c = np.zeros(n); c[-1] = -1.0
    res = linprog(c, A_ub=np.array(A), b_ub=np.array(b), bounds=[(0, 1)] * len(cols) + [(None, None)], method="highs")
    assert res.status == 0
    return -res.fun, np.array(A), np.array(b), idx


def test_fluid_lp_is_optimal_and_deterministic(built):
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    cases = []
    for suite in ("mk01", "multijob"):
        cases += H.load_suite(suite)[0]
    cases += H.load_suite("synth10x5")[0][:6]
    rng = np.random.RandomState(5)
    for a in cases:
        koff = np.concatenate(([0], np.cumsum(a.Jr)))
        states = [(np.repeat(a.count[0], a.Jr).astype(np.int32),
                   np.array([a.count[0][r] if j == 0 else 0 for r in range(a.R) for j in range(a.Jr[r])], np.int32))]
        # a mid-episode state: some later stages already hold jobs (constraint :271 switched off there)
        Q = states[0][0].copy(); now = states[0][1].copy()
        for r in range(a.R):
            j = rng.randint(0, a.Jr[r])
            now[koff[r] + j] += 1
        states.append((Q, now))
        for Q, now in states:
            x, obj = fi.fluid_lp(a.Jr, a.p, Q, now)
            x2, obj2 = fi.fluid_lp(a.Jr, a.p, Q, now)
            assert np.array_equal(x, x2) and obj == obj2
            best, A, b, idx = _lp_highs(a, Q, now)
            assert abs(best - obj) <= 1e-9 * max(1.0, abs(best)), (a.name, best, obj)
            v = np.zeros(A.shape[1]); v[-1] = obj
            for (k, m), i in idx.items():
                v[i] = x[k, m]
            assert (A @ v - b).max() < 1e-9 and x.min() >= 0.0 and x.max() <= 1.0
            assert (x[a.p == 0] == 0).all()
    # the stored fixture x is what the solver returns today (x is an input of every parity test)
    a = H.load_suite("mk01")[0][0]
    s = H.instance_set_from([a]).solve_fluid()
    assert np.array_equal(s.arrays(0).x, a.x)


def test_pyset_emulation_matches_cpython(built):
    """list(set(a) & set(b)) order for small ints: the oracle's emulation vs this interpreter."""
    from oracle import pyoracle
    rnd = random.Random(3)
    for M in (3, 5, 6, 8, 10, 15, 20, 32, 40, 64):
        for _ in range(400):
            idle = sorted(rnd.sample(range(M), rnd.randint(0, M)))
            elig = rnd.sample(range(M), rnd.randint(1, M))
            assert pyoracle.pyset_and_list(idle, elig) == list(set(idle) & set(elig)), (idle, elig)
    # exhaustive over idle subsets for one shuffled eligibility tuple at M = 10
    elig = [7, 2, 9, 0, 5, 8]
    for mask in range(1 << 10):
        idle = [m for m in range(10) if mask >> m & 1]
        assert pyoracle.pyset_and_list(idle, elig) == list(set(idle) & set(elig))


def test_device_pyset_order_matches_cpython(built):
    """csrc/fjsp_pyset.h (host build of the code machine_select runs on the GPU) vs this interpreter."""
    from deep_reinforcement_learning_for_fjsp_amd import _capi
    lib = _capi.lib()
    out = np.zeros(32, np.int32)

    def order(idle, machines, ascending):
        mask = 0
        for m in idle:
            mask |= 1 << m
        arr = np.ascontiguousarray(machines, dtype=np.int32)
        n = lib.fjsp_pyset_and_order(mask, arr.ctypes.data_as(C.c_void_p), len(arr), int(ascending),
                                     out.ctypes.data_as(C.c_void_p))
        assert n >= 0
        return out[:n].tolist()

    rnd = random.Random(7)
    for M in (2, 5, 6, 8, 9, 10, 15, 20, 32):
        for _ in range(1500):
            idle = sorted(rnd.sample(range(M), rnd.randint(0, M)))
            elig = rnd.sample(range(M), rnd.randint(1, M))
            assert order(idle, elig, False) == list(set(idle) & set(elig)), (idle, elig)
            fl = sorted(elig)
            assert order(idle, fl, True) == list(set(idle) & set(fl)), (idle, fl)
    for elig in ([7, 2, 9, 0], [8, 0, 16, 24], [9, 1], [31, 7, 15, 23], [12, 4, 20]):
        for mask in range(1 << 10):
            idle = [m for m in range(10) if mask >> m & 1] + ([12, 15, 16, 20, 23, 24, 31] if mask & 1 else [])
            idle = sorted(set(idle))
            assert order(idle, elig, False) == list(set(idle) & set(elig)), (idle, elig)


def test_mo_dfjsp_machine_data_loader_and_create_checks(built, tmp_path):
    """MO_DFJSP_instance_read.py:56-93 through the product loader: the power column of process_data.csv, idle power
    and breakdown windows of machine_data.csv (one row per window, first row of a machine sets its idle power),
    the set/get ABI round trip, and the create-time checks of the dynamic variant."""
    import csv
    from deep_reinforcement_learning_for_fjsp_amd import _capi, instances as fi
    insts, _, _ = H.load_suite("mo_dfjsp")
    a = next(x for x in insts if x.name.startswith("gen"))
    folder = tmp_path / "D0"
    _write_csv(a, str(folder))
    koff = np.concatenate(([0], np.cumsum(a.Jr)))
    with open(folder / "process_data.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["kind", "task", "machine_selectable", "process_time", "power"])
        for r in range(a.R):
            for j in range(int(a.Jr[r])):
                k = int(koff[r]) + j
                ms = tuple(int(m) for m in a.elig_list[k, :a.elig_n[k]])
                w.writerow([r, j, ms, tuple(int(a.p[k, m]) for m in ms), tuple(int(a.power[k, m]) for m in ms)])
    off = np.concatenate(([0], np.cumsum(a.bk_n)))
    with open(folder / "machine_data.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["machine", "idle_power", "breakdown_start", "breakdown_end"])
        for m in range(a.M):
            if a.bk_n[m] == 0:
                w.writerow([m, int(a.idle_power[m])])
            for q in range(int(off[m]), int(off[m + 1])):
                w.writerow([m, int(a.idle_power[m]) + (7 if q > off[m] else 0), int(a.bk[q, 0]), int(a.bk[q, 1])])   # later rows ignored
    s = fi.InstanceSet(2).load_csv(0, str(tmp_path), "D0")
    got = s.arrays(0)
    assert np.array_equal(got.power, a.power) and np.array_equal(got.idle_power, a.idle_power)
    assert np.array_equal(got.bk_n, a.bk_n) and np.array_equal(got.bk, a.bk)
    assert np.array_equal(got.p, a.p) and got.ddt == float(int(a.ddt))
    # ABI round trip on a raw instance
    s.set_raw(1, a.Jr, a.p, a.elig_n, a.elig_list, a.count, a.arrive, a.delivery, a.ddt)
    assert not hasattr(s.arrays(1), "power")
    s.set_dynamic(1, a.power, a.idle_power, a.bk_n, a.bk)
    rt = s.arrays(1)
    assert np.array_equal(rt.power, a.power) and np.array_equal(rt.bk, a.bk) and np.array_equal(rt.idle_power, a.idle_power)
    # create-time checks run before any device is touched
    lib, h = _capi.lib(), C.c_void_p()
    plain = H.instance_set_from(H.load_suite("mk01")[0])
    assert lib.fjsp_env_create(plain.handle, 0, 1, 2, 4, 0, 0, C.byref(h)) == -7          # no machine data
    assert b"machine data" in lib.fjsp_last_error()
    s.set_x(1, a.x)
    p2 = a.p.copy(); el_n = a.elig_n.copy(); el = a.elig_list.copy()
    m_dead = a.M - 1                                   # strip the last machine from every operation that has another one
    ok = True
    for k in range(a.K):
        if p2[k, m_dead] > 0:
            if el_n[k] == 1:
                ok = False
                break
            lst = [m for m in el[k, :el_n[k]] if m != m_dead]
            p2[k, m_dead] = 0; el_n[k] = len(lst); el[k, :] = 0; el[k, :len(lst)] = lst
    if ok:
        bad = fi.InstanceSet(1).set_raw(0, a.Jr, p2, el_n, el, a.count, a.arrive, a.delivery, a.ddt)
        bad.set_dynamic(0, np.where(p2 > 0, a.power, 0), a.idle_power, a.bk_n, a.bk).solve_fluid()
        assert lib.fjsp_env_create(bad.handle, 0, 1, 2, 4, 0, 0, C.byref(h)) == -5         # Machine.gap_ave would divide by zero
        assert b"no eligible operation" in lib.fjsp_last_error()


def test_generated_machine_data_ranges(built):
    """InstanceSet.generate_machine_data: the reference generator's power ranges (Instance_generate.py:61-66),
    reproducible from the seed, windows sorted and disjoint per machine."""
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    s = fi.InstanceSet(2)
    prm = fi.reference_generator_params(1.0, 12, 3)
    for i in range(2):
        s.generate(i, 99, prm).generate_machine_data(i, 5, max_windows=3)
    a, b = s.arrays(0), s.arrays(1)
    assert np.array_equal(a.power, b.power) and np.array_equal(a.bk, b.bk) and np.array_equal(a.idle_power, b.idle_power)
    assert ((a.power > 0) == (a.p > 0)).all() and a.power[a.p > 0].min() >= 10 and a.power.max() <= 200
    assert a.idle_power.min() >= 1 and a.idle_power.max() <= 9 and a.bk_n.max() <= 3 and len(a.bk) == a.bk_n.sum()
    off = np.concatenate(([0], np.cumsum(a.bk_n)))
    for m in range(a.M):
        w = a.bk[off[m]:off[m + 1]]
        assert (w[:, 0] < w[:, 1]).all() and (w[1:, 0] > w[:-1, 1]).all()


def test_global_actions_and_seeds_do_not_depend_on_the_sharding():
    """bench.py / train_ppo.py shard env ids over ranks; what an environment plays is a function of its GLOBAL id."""
    from deep_reinforcement_learning_for_fjsp_amd.batch import ENV_SEED_STRIDE, global_actions
    whole = global_actions(4242, 0, 100, 64, 6, 5)
    assert whole.dtype == np.uint8 and whole.shape == (64, 100, 2)
    assert whole[..., 0].max() == 5 and whole[..., 1].max() == 4 and whole.min() == 0
    for lo, hi in ((0, 33), (33, 64), (64, 100)):
        assert np.array_equal(global_actions(4242, lo, hi - lo, 64, 6, 5), whole[:, lo:hi])
    assert not np.array_equal(global_actions(4243, 0, 100, 64, 6, 5), whole)
    # all six task rules and all five machine rules get a fair share
    h0 = np.bincount(whole[..., 0].ravel(), minlength=6) / whole[..., 0].size
    h1 = np.bincount(whole[..., 1].ravel(), minlength=5) / whole[..., 1].size
    assert np.abs(h0 - 1 / 6).max() < 0.02 and np.abs(h1 - 1 / 5).max() < 0.02
    assert ENV_SEED_STRIDE == 1000003           # (the constant of open_env() in csrc/fjsp_kernels.hip)


def test_generated_instances_do_not_depend_on_the_sharding(built):
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    whole = fi.InstanceSet(12).generate_range(500, fi.bench_10x5_params())
    part = fi.InstanceSet(5).generate_range(507, fi.bench_10x5_params())
    for i in range(5):
        a, b = whole.arrays(7 + i), part.arrays(i)
        assert a.K == b.K and np.array_equal(a.p, b.p) and np.array_equal(a.Jr, b.Jr) and a.delivery == b.delivery


def test_tensor_argument_checks():
    """EnvBatch's guards in front of the raw-pointer C ABI (the negative cases need no GPU)."""
    import torch
    from deep_reinforcement_learning_for_fjsp_amd.batch import _as_input, _check_output
    cpu = torch.device("cpu")
    a = _as_input("actions", torch.zeros(4, 2, dtype=torch.int64), (4, 2), torch.uint8, cpu)
    assert a.dtype == torch.uint8 and a.is_contiguous()
    a = _as_input("actions", torch.zeros(2, 4, dtype=torch.uint8).t(), (4, 2), torch.uint8, cpu)
    assert a.is_contiguous()
    assert _as_input("mo", None, (4, 4), torch.float64, cpu) is None
    with pytest.raises(ValueError):
        _as_input("actions", torch.zeros(4, dtype=torch.uint8), (4, 2), torch.uint8, cpu)
    with pytest.raises(ValueError):
        _as_input("mask", torch.zeros(3, dtype=torch.uint8), (4,), torch.uint8, cpu)
    ok = torch.zeros(4, 20, dtype=torch.float64)
    assert _check_output("state_out", ok, (4, 20), torch.float64, cpu) is ok
    for bad in (torch.zeros(4, 20, dtype=torch.float32), torch.zeros(4, 21, dtype=torch.float64),
                torch.zeros(20, 4, dtype=torch.float64).t(), [0.0] * 80):
        with pytest.raises(ValueError):
            _check_output("state_out", bad, (4, 20), torch.float64, cpu)


def test_create_refuses_instances_beyond_the_32_bit_clock(built):
    """Clocks / per-kind tardiness sums are 32-bit in the kernels; Python integers do not wrap.  An instance whose
    worst-case schedule could pass 2^31 is refused at create (before any device is touched)."""
    from deep_reinforcement_learning_for_fjsp_amd import _capi, instances as fi
    lib = _capi.lib()
    s = fi.InstanceSet(1)
    R, M = 2, 2
    Jr = np.array([120, 120], np.int32)
    K = int(Jr.sum())
    p = np.full((K, M), 65535, np.int32)
    elig_n = np.full(K, M, np.int32)
    elig_list = np.tile(np.arange(M, dtype=np.int32), (K, 1))
    count = np.array([[130, 130]], np.int32)                    # 31 200 operations x 65 535 x 130 jobs of a kind > 2^31
    s.set_raw(0, Jr, p, elig_n, elig_list, count, np.array([0], np.int32), np.array([10 ** 6], np.int32))
    s.set_x(0, np.full((K, M), 1.0 / K))
    h = C.c_void_p()
    rc = lib.fjsp_env_create(s.handle, 0, 1, 1, 0, 0, 0, C.byref(h))
    assert rc == -5, rc                                         # FJSP_E_UNSUPPORTED
    assert b"32-bit clocks" in lib.fjsp_last_error()

"""CPU: the C-ABI library loads and exports every symbol include/nanomod_hip.h declares (no compute
without a GPU), struct layouts match the header, and the host-side logic mirrors the reference."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nanomod_hip.h')


def test_library_exports_every_declared_symbol():
    import nanomod_amd._lib as L
    lib = L.load()
    with open(HEADER) as f:
        text = f.read()
    declared = set(re.findall(r'\b(nmod_[a-z_]+)\s*\(', text))
    assert declared == set(L._SIGNATURES), declared ^ set(L._SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nmod_abi_version() == 4
    assert b'invalid' in lib.nmod_strerror(-1) and b'65535' in lib.nmod_strerror(-3)


def test_shipped_binary_is_a_product_build():
    """nmod_build_info lists the experiment macros of every translation unit: the shipped library skips no phase
    (NMOD_SKIP / NMOD_EXP are phase-skip switches of the kernel headers) and runs the default variants."""
    import nanomod_amd._lib as L
    info = L.load().nmod_build_info().decode()
    parts = info.split(' | ')
    assert parts[0].startswith('arch=gfx950 abi=4 ')
    names = [p.split(':')[0] for p in parts[1:]]
    assert names == ['abi_tu', 'k1_f32_ks', 'k1_f32_all', 'k1_i16_ks', 'k1_i16_all']
    want = {'NMOD_SKIP': '0', 'NMOD_EXP': '0', 'NMOD_HIST_WAVES': '4', 'NMOD_WIDE_I16_WORDS': '2048',
            'NMOD_SWZ_MASK': '0', 'NMOD_PK_SELECT': '0', 'NMOD_CE_BUILTIN': '0', 'NMOD_XOR4_BANKS': '0', 'NMOD_NO_GRID': '0',
            'NMOD_CNT_SKIP': '0', 'NMOD_CNT_WAVES': '4', 'NMOD_KS_TOPS': '1', 'NMOD_WIDE_TOPS': '0', 'NMOD_CW_OR3': '1', 'NMOD_CNT_TAILS': '1', 'NMOD_WIDE_TAILS': '1'}
    for p in parts[1:]:
        got = dict(kv.split('=') for kv in p.split(': ', 1)[1].split())
        assert got == want, (p, got)
    # every NMOD_* switch the kernel sources test is reported (a new experiment macro must be added to build_info.hpp)
    src_dir = os.path.join(ROOT, 'nanomod_amd', 'csrc')
    used = set()
    for fn in os.listdir(src_dir):
        if fn.endswith(('.hpp', '.hip')) and fn != 'build_info.hpp':
            for ln in open(os.path.join(src_dir, fn)):
                if ln.lstrip().startswith(('#if', '#elif')):
                    used |= set(re.findall(r'\bNMOD_[A-Z0-9_]+\b', ln))
    structural = {'NMOD_INST_DTYPE', 'NMOD_INST_ALL', 'NMOD_HIP', 'NMOD_CAT', 'NMOD_CAT2', 'NMOD_LAUNCH_NAME', 'NMOD_FLAGS_NAME', 'NMOD_QP',
                  'NMOD_PK8', 'NMOD_ABI_VERSION', 'NMOD_BUILD_FLAGS'}
    used = {u for u in used if u not in structural and not u.startswith(('NMOD_BI_', 'NMOD_STR'))}
    assert used <= set(want), used - set(want)


def test_host_pipeline_config_validates():
    import nanomod_amd._lib as L
    lib = L.load()
    assert lib.nmod_host_pipeline_config(-1, 0, 0, 0) == -1
    assert lib.nmod_host_pipeline_config(0, 9, 0, 0) == -1
    assert lib.nmod_host_pipeline_config(0, 0, 0, 3) == -1
    assert lib.nmod_host_pipeline_config(1 << 20, 4, 2, 2) == 0
    assert lib.nmod_host_pipeline_config(0, 0, 0, 0) == 0
    st = L.NmodHostStats()
    assert lib.nmod_last_host_stats(C.byref(st)) == 0 and st.chunks == 0


def test_struct_layout_matches_header():
    import nanomod_amd._lib as L
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu", sizeof(nmod_params), ' \
          'offsetof(nmod_params, weights_dif), offsetof(nmod_params, max_n0), offsetof(nmod_params, timer), offsetof(nmod_params, flags), sizeof(nmod_out), ' \
          'sizeof(nmod_dispatch_stats), offsetof(nmod_dispatch_stats, rank_count_wide), offsetof(nmod_dispatch_stats, f64_redo));return 0;}' % HEADER
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 't.c')
        open(c, 'w').write(src)
        subprocess.check_call(['gcc', c, '-o', os.path.join(d, 't')])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, 't')]).split()]
    assert got == [C.sizeof(L.NmodParams), L.NmodParams.weights_dif.offset, L.NmodParams.max_n0.offset,
                   L.NmodParams.timer.offset, L.NmodParams.flags.offset, C.sizeof(L.NmodOut),
                   C.sizeof(L.NmodDispatchStats), L.NmodDispatchStats.rank_count_wide.offset, L.NmodDispatchStats.f64_redo.offset]


def test_dispatch_stats_without_a_call_and_form_flags():
    """nmod_last_dispatch_stats before the thread has run a batch: an error, not stale numbers; the kernel-form switches are
    per-call flags (the library reads no environment variable: SURVEY.md 8b, no process-wide state behind the ABI)"""
    import threading
    import nanomod_amd._lib as L
    lib = L.load()
    out = {}

    def fresh_thread():
        st = L.NmodDispatchStats()
        out['rc'] = lib.nmod_last_dispatch_stats(C.byref(st)); out['pos'] = st.positions
        out['null'] = lib.nmod_last_dispatch_stats(None)
    t = threading.Thread(target=fresh_thread); t.start(); t.join()
    assert out == {'rc': -1, 'pos': 0, 'null': -1}
    assert (L.FLAG_NO_COUNTING, L.FLAG_NO_COUNT_WIDE) == (4, 8)
    assert L.FLAG_NO_HOST_NARROW == 16
    prm = L.make_params(flags=64)                             # (an unknown flag bit)
    assert lib.nmod_workspace_bytes(C.byref(prm), 10) > 0 and lib.nmod_detect_batch(C.byref(prm), 1, None, None, None, None, None, None, 0, None) == -1
    src = open(os.path.join(ROOT, 'nanomod_amd', 'csrc', 'nanomod_hip.hip')).read() + open(os.path.join(ROOT, 'nanomod_amd', 'csrc', 'rank_stats_inst.hip')).read()
    assert 'getenv("NMOD_NO_COUNT' not in src


def test_cabi_collective_argument_plumbing_without_a_gpu():
    """nmod_comm_* / nmod_allgather_tracks (the torch-free all-gather, include/nanomod_hip.h): RCCL is bound at run time — the ids of
    two calls differ —, bad arguments are refused before any device or RCCL work, and a rank cannot be made without a device"""
    import nanomod_amd._lib as L
    from nanomod_amd import sharding
    lib = L.load()
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    rc = lib.nmod_comm_unique_id(buf)
    if rc == L.ERR_NO_RCCL:
        pytest.skip('no librccl.so on this machine')
    assert rc == 0 and sharding.CAbiComm.unique_id() != sharding.CAbiComm.unique_id() and len(sharding.CAbiComm.unique_id()) == 128
    h = C.c_void_p()
    assert lib.nmod_comm_unique_id(None) == -1
    assert lib.nmod_comm_init_rank(None, 1, 0, 0, C.byref(h)) == -1 and lib.nmod_comm_init_rank(buf, 0, 0, 0, C.byref(h)) == -1
    assert lib.nmod_comm_init_rank(buf, 2, 2, 0, C.byref(h)) == -1 and lib.nmod_comm_init_rank(buf, 2, -1, 0, C.byref(h)) == -1
    assert lib.nmod_comm_init_rank(buf, 1, 0, 0, None) == -1
    assert lib.nmod_allgather_tracks(None, None, 4, 1, None, None) == -1 and lib.nmod_comm_destroy(None) == -1
    if lib.nmod_device_count() == 0:
        assert lib.nmod_comm_init_rank(buf, 1, 0, 0, C.byref(h)) == -5 and not h.value
        with pytest.raises(L.NanomodLibraryError, match='no HIP device'):
            sharding.CAbiComm(buf.raw, 1, 0, 0)
    with pytest.raises(ValueError):
        sharding.CAbiComm(b'short', 1, 0, 0)
    assert b'librccl' in lib.nmod_strerror(-6)


def test_host_narrowing_accepts_exactly_the_int16_grid():
    """nmod_narrow_probe (the float64 -> int16 narrowing of the host-resident entry, host_pipeline.hpp): every k / 1000.0 with
    |k| <= 32 767 narrows to k — the division-free quotient test is exact for all 65 535 of them, through the vector loop and its
    scalar tail — and nothing else does: a neighbouring double, |k| = 32 768, NaN, infinities, huge values, half a milli-unit"""
    import nanomod_amd._lib as L
    lib = L.load()
    k = np.arange(-32767, 32768, dtype=np.int64)
    v = k.astype(np.float64) / 1000.0
    for n in (len(v), len(v) - 3, 13, 8, 7, 1, 0):            # (whole vectors, a ragged tail, only the tail, nothing)
        out = np.full(len(v), 77, np.int16)
        assert lib.nmod_narrow_probe(v.ctypes.data, n, out.ctypes.data) == 1
        assert np.array_equal(out[:n], k[:n]) and np.all(out[n:] == 77)
    rng = np.random.default_rng(3)
    big = rng.integers(-32767, 32768, 300_001)
    w = big / 1000.0
    out = np.empty(len(w), np.int16)
    assert lib.nmod_narrow_probe(w.ctypes.data, len(w), out.ctypes.data) == 1 and np.array_equal(out, big)
    for bad in (np.nextafter(7.233, 9.0), np.nextafter(-0.001, -9.0), 32.768, -32.768, np.nan, np.inf, -np.inf, 1e300, -4e9, 0.0005, 1e-320):
        for at in (0, 5, len(w) - 1, 123_456):
            x = w.copy(); x[at] = bad
            assert lib.nmod_narrow_probe(x.ctypes.data, len(x), out.ctypes.data) == 0, (bad, at)
    assert lib.nmod_narrow_probe(None, 4, out.ctypes.data) == -1 and lib.nmod_narrow_probe(w.ctypes.data, -1, out.ctypes.data) == -1
    z = np.array([0.0, -0.0, 0.001, -0.001])                   # a signed zero is the value zero
    assert lib.nmod_narrow_probe(z.ctypes.data, 4, out.ctypes.data) == 1 and list(out[:4]) == [0, 0, 1, -1]


def test_makefile_rebuilds_every_unit_when_a_header_changes():
    """every translation unit depends on $(HDRS): touching radix_sort.hpp (included by rank_order.hip only) must put
    rank_order.o into `make -n`'s plan (round 5 left it stale)"""
    csrc = os.path.join(ROOT, 'nanomod_amd', 'csrc')
    if not os.path.exists(os.path.join(csrc, 'build', 'rank_order.o')):
        pytest.skip('no objects built here')
    hdr = os.path.join(csrc, 'radix_sort.hpp')
    st = os.stat(hdr)
    try:
        os.utime(hdr, None)
        plan = subprocess.check_output(['make', '-n', '-C', csrc], text=True)
    finally:
        os.utime(hdr, ns=(st.st_atime_ns, st.st_mtime_ns))
    for obj in ('rank_order.o', 'nanomod_hip.o', 'rank_stats_d0_a0.o', 'rank_stats_d1_a1.o'):
        assert obj in plan, obj


def test_no_gpu_fails_loudly_not_silently():
    import nanomod_amd as nm
    L = nm._lib
    if L.load().nmod_device_count() > 0:
        pytest.skip('GPU present')
    with pytest.raises(L.NanomodLibraryError, match='no HIP device'):
        nm.detect_host(np.zeros(10, np.float32), np.array([0, 5, 10]), np.zeros(10, np.float32), np.array([0, 5, 10]),
                       np.zeros(2, np.int32))
    with pytest.raises(L.NanomodLibraryError):
        nm.getKStest({'coverages': [0, 0]}, [0.1, 0.2, 0.3], [0.2, 0.3, 0.4], '+')
    # invalid arguments are rejected before any device work
    prm = L.make_params(nb=1000)
    assert L.load().nmod_detect_batch(C.byref(prm), 1, None, None, None, None, None, None, 0, None) == -1


def test_missing_library_error(monkeypatch):
    import nanomod_amd._lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', '/nonexistent/libnanomod_hip.so')
    with pytest.raises(L.NanomodLibraryError, match='no CPU fallback'):
        L.load()


def test_run_ids_equal_pos_check():
    import nanomod_amd.detect as D
    rng = np.random.default_rng(0)
    chrom = np.array(['a'] * 40 + ['b'] * 30); strand = np.array((['+'] * 20 + ['-'] * 20) + ['+'] * 30)
    pos = np.cumsum(rng.choice([1, 1, 1, 2], 70))
    rid = D.run_ids(chrom, strand, pos)
    mlist = [((chrom[i], strand[i], int(pos[i]), 'A', 5, 5), []) for i in range(70)]
    for i in range(70):
        for j in range(max(0, i - 4), min(70, i + 5)):
            assert D.pos_check(mlist, i, j) == (rid[i] == rid[j]), (i, j)
    assert not D.pos_check(mlist, 0, -1) and not D.pos_check(mlist, 0, 70)


def test_encode_signals():
    import nanomod_amd.detect as D
    assert D.encode_signals([0.5, 0.25, -1.75]).dtype == np.float32
    v = np.round(np.random.default_rng(1).normal(0, 1, 1000), 3)
    k = D.encode_signals(v)
    assert k.dtype == np.int16 and np.array_equal(k / 1000.0, v)
    assert D.encode_signals([0.1234567]).dtype == np.float64          # neither: the fp64-key path (NMOD_DTYPE_F64)
    assert D.m_min_float(0.0) == 2.2250738585072014e-308 and D.m_max_float(float('inf')) == 1.7976931348623157e308
    assert np.isnan(D.m_min_float(float('nan')))


def test_filter_order_and_table_format_without_gpu():
    """mfilter_coverage + build_csr give the reference's position set and order; save_test formats like it"""
    import nanomod_amd.detect as D
    fx = H.load_inputs('ragged')
    exp, table = H.load_expected('ragged_stouffer')
    with tempfile.TemporaryDirectory() as out:
        mo = H.build_moptions(fx, out, 'x', 2, 2.0, 'stouffer')
        D.mfilter_coverage(mo)
        meta, sig0, off0, sig1, off1, rid = D.build_csr(mo)
        assert list(zip(meta['chrom'], meta['strand'], meta['pos'], meta['base'], meta['n0'], meta['n1'])) == list(zip(
            exp['chrom'], exp['strand'], exp['pos'], exp['base'], exp['n0'], exp['n1']))
        assert sig0.dtype == np.float32 and off0[-1] == sig0.shape[0]
        mo['sign_test'] = [(m, [(exp['mwu_u'][i], exp['mwu_p'][i]), (exp['t_t'][i], exp['t_p'][i]),
                                (exp['ks_d'][i], exp['ks_p'][i]), (exp['comb_st'][i], exp['comb_p'][i])])
                           for i, m in enumerate(zip(meta['chrom'], meta['strand'], meta['pos'], meta['base'], meta['n0'], meta['n1']))]
        D.save_test(mo)
        assert open(os.path.join(out, 'x_sign_test.txt')).read() == table


def test_shard_bounds_cover_exactly():
    from nanomod_amd import sharding
    for npos in (0, 1, 7, 64, 1000, 4600000):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_bounds(npos, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == npos
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
    assert sharding.halo_bounds(10, 20, 2, 100) == (8, 22) and sharding.halo_bounds(0, 20, 2, 21) == (0, 21)


def test_synth_generator_restatement_is_standard_normal_like():
    v = H.synth_ref(20240601, 0, 2000, 0, 200)
    assert abs(v.mean()) < 0.01 and abs(v.std() - 1.0) < 0.01 and v.dtype == np.float32
    b = H.synth_ref(20240601, 0, 20001, 1, 50, 10000, 0.8).reshape(20001, 50)
    assert abs(b[10000].mean() - 0.8) < 0.5 and abs(b[5000].mean()) < 0.5
    assert np.array_equal(H.synth_ref(1, 5, 10, 1, 7), H.synth_ref(1, 0, 15, 1, 7)[35:])      # counter-based


def _fixture_containers(name, tmp):
    """the golden fixture inputs as two per-group .npz containers (positions with no samples dropped)"""
    from nanomod_amd import container
    fx = H.load_inputs(name)
    paths = []
    for g in (0, 1):
        off = fx['off%d' % g]
        rows = np.nonzero(np.diff(off) > 0)[0]
        sig, noff = container.gather_rows(fx['sig%d' % g], off, rows)
        pth = os.path.join(tmp, 'g%d.npz' % g)
        container.save_group(pth, fx['chrom'][rows], fx['strand'][rows], fx['pos'][rows], fx['base%d' % g][rows], noff, sig)
        paths.append(pth)
    return paths


def test_cli_position_selection_and_table_writer_cpu():
    """array-native coverage filter / intersection / order == the reference's (golden), and the C table writer
    reproduces save_test's bytes (host-only entry point, no GPU needed)"""
    from nanomod_amd import cli, container
    for name, exp_name in (('ragged', 'ragged_stouffer'), ('g50', 'g50_fisher'), ('ties', 'ties_stouffer')):
        exp, table = H.load_expected(exp_name)
        with tempfile.TemporaryDirectory() as tmp:
            p0, p1 = _fixture_containers(name, tmp)
            meta, sig0, off0, sig1, off1, rid = cli.select_positions(container.load_group(p0), container.load_group(p1), 5,
                                                                     log=lambda *a: None)
            assert list(meta['chrom']) == list(exp['chrom']) and list(meta['pos']) == list(exp['pos'])
            assert list(meta['strand']) == list(exp['strand']) and list(meta['base']) == list(exp['base'])
            assert list(meta['n0']) == list(exp['n0']) and list(meta['n1']) == list(exp['n1'])
            out = os.path.join(tmp, 't.txt')
            cli.write_sign_test(out, meta, exp, True)
            assert open(out).read() == table


def test_cli_argument_validation():
    from nanomod_amd import cli
    from nanomod_amd import container as container_mod
    a = cli.build_parser().parse_args(['detect', '--wrkBase1', '/nonexistent1', '--wrkBase2', '/nonexistent2',
                                       '--MinCoverage', '2', '--WeightsDif', '0.5'])
    errs = cli.validate(a)
    assert any('MinCoverage' in e for e in errs) and sum('does not exist' in e for e in errs) == 2
    assert a.WeightsDif == 1.0                      # floor, NanoMod.py:76-78
    # --Pos chr:pos (NanoMod.py:117-129, myDetect.py:550-558): 0-based position, +-(window-1)/2 region; --plotType accepted
    a = cli.build_parser().parse_args(['detect', '--wrkBase1', 'x', '--wrkBase2', 'y', '--Pos', 'chrA:100', '--window', '21',
                                       '--plotType', 'Violin'])
    cli.validate(a)
    assert a.roi == {'Chr': 'chrA', 'Pos': 99, 'start_pos': 89, 'end_pos': 109}
    a = cli.build_parser().parse_args(['detect', '--wrkBase1', 'x', '--wrkBase2', 'y', '--Pos', 'chrA:100:90'])
    assert any('not larger than the start' in e for e in cli.validate(a))
    # the position-level part of the filter on a container
    with tempfile.TemporaryDirectory() as tmp:
        p0, _ = _fixture_containers('g50', tmp)
        g = container_mod.load_group(p0)
        a = cli.build_parser().parse_args(['detect', '--wrkBase1', p0, '--wrkBase2', p0, '--Pos', '%s:%d' % (g['chrom'][0], g['pos'][3] + 1),
                                           '--window', '5'])
        cli.validate(a)
        sub = cli.load_input(p0, a)
        assert set(sub['chrom']) == {g['chrom'][0]} and sub['pos'].min() >= g['pos'][3] - 2 and sub['pos'].max() <= g['pos'][3] + 2
        assert sub['off'][-1] == len(sub['sig']) and len(sub['pos']) > 0


def test_fast5_ingest_matches_reference_reader():
    """The array-native FAST5 ingest (walk order, min_lr filter, strand-aware position mapping, last-read base)
    builds what the reference's ReadAllFast5 builds from the same reads (golden, stub h5py); the HDF5 access
    itself sits behind `reader`, so no h5py is needed here."""
    from nanomod_amd import fast5_ingest
    z = np.load(os.path.join(H.GOLDEN, 'fast5_reads.npz'))
    with tempfile.TemporaryDirectory() as root:
        for i in range(len(z['rel'])):
            path = os.path.join(root, 'grp%d' % z['group'][i], str(z['rel'][i]))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            a, b = z['off'][i], z['off'][i + 1]
            with open(path, 'wb') as f:
                np.savez(f, chrom=z['chrom'][i], strand=z['strand'][i], start=z['start'][i], norm_mean=z['norm_mean'][a:b],
                         base=z['base'][a:b], has_align=z['has_align'][i])

        def reader(path):
            r = np.load(path)
            if not bool(r['has_align']):
                return None
            return str(r['chrom']), int(r['start']), str(r['strand']), r['norm_mean'], r['base']
        for g in (0, 1):
            got = fast5_ingest.ingest_folder(os.path.join(root, 'grp%d' % g), {'min_lr': 500, 'min_lr_nb': 0}, reader,
                                             log=lambda *a: None)
            exp = np.load(os.path.join(H.GOLDEN, 'fast5_expected_g%d.npz' % g))
            for k in ('chrom', 'strand', 'pos', 'base', 'off'):
                assert np.array_equal(got[k], exp[k]), k
            sig = np.concatenate([np.sort(got['sig'][got['off'][i]:got['off'][i + 1]]) for i in range(len(got['pos']))])
            assert np.array_equal(sig, exp['sig'])            # per position: the same multiset of samples
    # the filters
    assert not fast5_ingest.read_passes_filters(100, 'c', 0, '+', {'min_lr': 500}, log=lambda *a: None)
    assert fast5_ingest.read_passes_filters(600, 'c', 0, '+', {'min_lr': 500, 'Chr': 'c'})
    assert not fast5_ingest.read_passes_filters(600, 'c', 0, '+', {'min_lr': 500, 'Chr': 'd'})


def test_table_number_formats_equal_printf():
    """the table writer formats '%.3f' and '%.3E' itself (exact x87 arithmetic, snprintf for values near a rounding
    boundary): byte-identical to Python's formatting on boundary cases, halves, subnormals, random bit patterns"""
    import ctypes as C
    from nanomod_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(20241002)
    cases = [0.0, -0.0, 1.0, -1.0, 0.0005, 0.0015, 0.0025, 1.0005, 2.5, 12345.0, 1234.5, 0.5, 1e-300, 5e-324,
             2.2250738585072014e-308, 1.7976931348623157e308, 9.9995, 9.99949999, 99995.0, 999.95, 0.00099995, 1e15, 9e15,
             8.9999999e15, 1e16, 123456789.0005, 29750.5, 0.125, 0.0625, 1e-5, 9.9999e-5, 1e100, 1e-100, 4.35e-7,
             1.2345e-200, 9.9995e+99, 9.9995e-100, 1.0e23, 5e22, 9.9994999999999994e-05, 999949999.9999999, 0.9995, 0.99949999999999994]
    vals = np.array(cases + list(rng.normal(0, 100, 20000)) + list(10.0 ** rng.uniform(-320, 308, 20000))
                    + list(rng.integers(0, 10 ** 7, 20000) / 1000.0) + list(rng.integers(0, 10 ** 7, 20000) / 2000.0)
                    + list((rng.integers(10000, 100000, 20000) * 10.0 ** rng.integers(-200, 200, 20000)) / 2.0)
                    + list(np.frombuffer(rng.bytes(8 * 40000), dtype=np.float64)))
    vals = np.ascontiguousarray(vals[np.isfinite(vals)])
    for sci, fmt in ((0, '%.3f'), (1, '%.3E')):
        for lo in range(0, len(vals), 20000):
            v = np.ascontiguousarray(vals[lo:lo + 20000])
            buf = C.create_string_buffer(len(v) * 420 + 16)
            assert lib.nmod_format_probe(v.ctypes.data_as(C.POINTER(C.c_double)), len(v), sci, buf, len(buf)) == 0
            got = buf.raw.split(b'\0')[:len(v)]
            exp = [(fmt % x).encode() for x in v.tolist()]
            assert got == exp, [(x, g, e) for x, g, e in zip(v.tolist(), got, exp) if g != e][:5]
    special = np.array([np.nan, np.inf, -np.inf])
    buf = C.create_string_buffer(2000)
    assert lib.nmod_format_probe(special.ctypes.data_as(C.POINTER(C.c_double)), 3, 0, buf, 2000) == 0
    assert buf.raw.split(b'\0')[:3] == [b'nan', b'inf', b'-inf']
    assert lib.nmod_format_probe(special.ctypes.data_as(C.POINTER(C.c_double)), 3, 1, buf, 2000) == 0
    assert buf.raw.split(b'\0')[:3] == [b'NAN', b'INF', b'-INF']



def test_sign_test_records_behave_like_the_reference_list():
    """drop-in consumers call list methods on moptions['sign_test'] — mySimulate.getTopRank does
    `moptions['sign_test'].index(record)` (mySimulate.py:312) with records taken from 'sorted_sign_test'."""
    import pickle
    import nanomod_amd.detect as D
    n = 50
    rng = np.random.default_rng(1)
    meta = {'chrom': np.array(['chr1'] * n), 'strand': np.array(['+'] * 30 + ['-'] * 20), 'pos': np.arange(100, 100 + n),
            'base': np.array(list('ACGT' * 13)[:n]), 'n0': np.full(n, 7), 'n1': np.full(n, 9)}
    res = {k: rng.random(n) for k in ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p')}
    st = D.SignTestRecords(meta, res, True)
    as_list = [st._build(i) for i in range(n)]
    order = np.argsort(res['comb_p'], kind='stable')
    ranked = st.permuted(order)
    assert isinstance(st, __import__('collections').abc.Sequence)
    assert st == as_list and as_list == st.tolist() and not (st == as_list[:-1])
    # a reference-style getTopRank walk (mySimulate.py:300-316): records come from the ranked view, their index from sign_test
    for k, rec in enumerate(ranked):
        cur = st.index(rec)
        assert cur == int(order[k]) and st[cur] is rec
        assert D.pos_check(st, cur, cur) and ranked.index(rec) == k
        if k > 10:
            break
    # an equal record that is not one of ours is found by value; a foreign one raises like a list
    assert st.index(as_list[17]) == 17 and st.count(as_list[3]) == 1 and as_list[5] in st
    with pytest.raises(ValueError):
        st.index((('chrX', '+', 1, 'A', 1, 1), []))
    with pytest.raises(ValueError):
        st.index(st[40], 0, 10)
    assert (st + [1])[-1] == 1 and ([0] + st)[0] == 0 and list(reversed(st))[0] == as_list[-1]
    back = pickle.loads(pickle.dumps(ranked))
    assert type(back) is list and back == [as_list[i] for i in order]
    with pytest.raises(TypeError):
        hash(st)
    # records holding NaN statistics (zero variance, a flagged position) are not "edited" just because NaN != NaN; a real edit is
    assert not st.edited()
    res2 = {k: v.copy() for k, v in res.items()}
    res2['t_t'][4] = np.nan; res2['t_p'][4] = np.nan; res2['mwu_u'][9] = np.nan
    st2 = D.SignTestRecords(meta, res2, True)
    _ = [st2[i] for i in (3, 4, 9)]
    assert not st2.edited()
    st2[4][1].append((1.0, 0.5))                              # the reference's own way of adding a pair (myDetect.py:377)
    assert st2.edited()


def test_float_form_quotient_is_correctly_rounded_for_every_count():
    """The kernels form fl(c/n) as c*r corrected by one Newton step (hist_exact_quot, ks_rank.hpp) instead of an fp64
    division: exhaustively equal to c / n for every 0 <= c <= n <= 65 535 (the largest group K1 ranks)."""
    src = r"""
#include <stdio.h>
#include <math.h>
int main(void) {
  long bad = 0;
  #pragma omp parallel for schedule(dynamic, 64) reduction(+:bad)
  for (int n = 1; n <= 65535; n++) {
    double dn = n, r = 1.0 / dn;
    for (int c = 0; c <= n; c++) {
      double dc = c, q0 = dc * r, rem = fma(-q0, dn, dc), q = fma(rem, r, q0);
      if (q != dc / dn) bad++;
    }
  }
  printf("%ld", bad);
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 'q.c')
        open(c, 'w').write(src)
        subprocess.check_call(['gcc', '-O2', '-fopenmp', '-mfma', '-ffp-contract=off', c, '-o', os.path.join(d, 'q'), '-lm'])
        assert subprocess.check_output([os.path.join(d, 'q')]).decode() == '0'


def test_packed_sort_header_is_what_the_generator_emits():
    """nanomod_amd/csrc/packed_sort_i16.hpp is generated: tools/gen_packed_sort.py walks the bitonic network on packed int16
    keys, simulates the emitted program on random / tied / extreme keys for both lane-group sizes (its own assertion) and
    writes the header — the committed file must be its output."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'h.hpp')
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_packed_sort.py'), out], stdout=subprocess.DEVNULL)
        assert open(out).read() == open(os.path.join(ROOT, 'nanomod_amd', 'csrc', 'packed_sort_i16.hpp')).read()


def test_hostwalk_flatten_equals_numpy():
    """csrc/hostwalk.c (the row walk of detect.build_csr): float64 ndarrays, views, other dtypes, lists of numpy.float64 /
    floats / ints, tuples and empty rows flatten to what numpy.concatenate gives; a short output buffer raises"""
    hw = pytest.importorskip('nanomod_amd._hostwalk', reason='make -C nanomod_amd/csrc')
    rng = np.random.default_rng(4)
    big = rng.normal(0, 1, (50, 7))
    rows = [big[3], big[:, 2], np.arange(5, dtype=np.int32), np.float32([1.5, 2.5]), [np.float64(0.25), 1, 2.5], (3.0, 4.0), [],
            np.zeros(0), big[10].copy(), list(map(np.float64, big[11]))]
    exp = np.concatenate([np.asarray(r, dtype=np.float64).ravel() for r in rows])
    out = np.full(len(exp) + 3, -1.0)
    assert hw.flatten(rows, out) == len(exp)
    assert np.array_equal(out[:len(exp)], exp) and np.all(out[len(exp):] == -1.0)
    assert hw.flatten(tuple(rows), out) == len(exp)
    with pytest.raises(ValueError):
        hw.flatten(rows, np.empty(len(exp) - 1))
    with pytest.raises(TypeError):
        hw.flatten([['a']], np.empty(4))
    # through build_csr: the reference's dict-of-dict-of-list shape and the array shape give the same CSR
    import nanomod_amd.detect as D
    vals = np.round(rng.normal(0, 1, (300, 9)), 3)
    def ds(as_list):
        d = {i + 10: ([np.float64(v) for v in vals[i]] if as_list else vals[i].copy()) for i in range(300)}
        return {'norm_mean': {('chr1', '+'): d}, 'base': {('chr1', '+'): {i + 10: 'A' for i in range(300)}}, 'basedict': {}}
    got = []
    for as_list in (True, False):
        mo = {'ds2': ['a', 'b'], 'a': ds(as_list), 'b': ds(as_list), 'MinCoverage': 5, 'outLevel': 3}
        meta, s0, o0, s1, o1, rid = D.build_csr(mo)
        got.append((s0.copy(), o0.copy()))
        assert len(meta['pos']) == 300 and o0[-1] == 2700
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])


def test_hostwalk_dict_walks_equal_the_python_loops(capsys, monkeypatch):
    """csrc/hostwalk.c filter_coverage / join_strand (mfilter_coverage's inner loop, mtest2's loop header — myDetect.py:301-314,
    427-436) against the plain-Python statements of the same loops: dicts filled in scrambled order, positions missing on
    either side, rows as lists of numpy.float64 / arrays / tuples, base mismatches (printed like the reference), positions
    below MinCoverage, a strand that only one dataset holds, numpy-integer position keys (the general path)."""
    hw = pytest.importorskip('nanomod_amd._hostwalk', reason='make -C nanomod_amd/csrc')
    import copy
    import nanomod_amd.detect as D
    rng = np.random.default_rng(12)

    def dataset(seed, scramble, np_keys=False):
        r = np.random.default_rng(seed)
        ds = {'norm_mean': {}, 'base': {}, 'basedict': {}}
        for sk in (('chr2', '-'), ('chr1', '+'), ('chr1', '-'), ('chrX', '+') if seed % 2 else ('chrY', '+')):
            poss = np.unique(r.integers(0, 400, 250))
            if scramble:
                poss = r.permutation(poss)
            nm_, bs = {}, {}
            for p_ in poss.tolist():
                n = int(r.integers(1, 12))
                v = np.round(r.normal(0, 1, n), 3)
                row = [np.float64(x) for x in v] if p_ % 3 == 0 else (tuple(v.tolist()) if p_ % 3 == 1 else v)
                key = np.int64(p_) if np_keys else p_
                nm_[key] = row
                bs[key] = 'ACGT'[(p_ + (1 if (seed == 2 and p_ % 37 == 0) else 0)) % 4]
            ds['norm_mean'][sk] = nm_; ds['base'][sk] = bs
        return ds

    def reference_way(mo):
        mo = copy.deepcopy(mo)
        monkeypatch.setattr(D, '_hostwalk_module', lambda: None)
        D.mfilter_coverage(mo)
        out = D.build_csr(mo)
        monkeypatch.undo()
        return mo, out

    for scramble in (False, True):
        for np_keys in (False, True):
            mo = {'ds2': ['a', 'b'], 'a': dataset(1, scramble, np_keys), 'b': dataset(2, scramble, np_keys), 'MinCoverage': 4, 'outLevel': 3}
            ref_mo, ref = reference_way(mo)
            ref_print = capsys.readouterr().out
            got_mo = copy.deepcopy(mo)
            D.mfilter_coverage(got_mo)
            for dsn in ('a', 'b'):                           # the same positions left, the same strands left
                assert {sk: sorted(int(k) for k in d) for sk, d in got_mo[dsn]['norm_mean'].items()} == \
                       {sk: sorted(int(k) for k in d) for sk, d in ref_mo[dsn]['norm_mean'].items()}
                assert {sk: sorted(int(k) for k in d) for sk, d in got_mo[dsn]['base'].items()} == \
                       {sk: sorted(int(k) for k in d) for sk, d in ref_mo[dsn]['base'].items()}
            got = D.build_csr(got_mo)
            assert capsys.readouterr().out == ref_print and 'Error not equal' in ref_print
            for k in ('chrom', 'strand', 'pos', 'base', 'n0', 'n1', 'chrom_id'):
                assert list(got[0][k]) == list(ref[0][k]), k
            assert got[0]['names'] == ref[0]['names'] and len(got[0]['pos']) > 100
            for a, b in zip(got[1:], ref[1:]):
                assert a.dtype == b.dtype and np.array_equal(a, b)
    # the C functions themselves: errors
    with pytest.raises(TypeError):
        hw.join_strand({'x': [1.0]}, {'x': [1.0]}, {'x': 'A'}, {'x': 'A'})
    with pytest.raises(KeyError):
        hw.join_strand({5: [1.0]}, {5: [1.0]}, {}, {5: 'A'})
    with pytest.raises(ValueError):
        hw.join_strand({5: np.zeros((2, 2))}, {5: [1.0]}, {5: 'A'}, {5: 'A'})
    d, b = {1: [1.0, 2.0], 2: [1.0], 3: np.zeros(5)}, {1: 'A', 2: 'C', 3: 'G'}
    assert hw.filter_coverage(d, b, 2) == 1 and sorted(d) == [1, 3] and sorted(b) == [1, 3]
    p_, n0, n1, s0, s1, bases, mism, codes = hw.join_strand({}, {}, {}, {})
    assert len(p_) == 0 and len(s0) == 0 and bases == [] and mism == []


def test_hostwalk_two_step_join_gathers_and_narrows():
    """csrc/hostwalk.c join_strand_plan + copy_plan (round 6): the positions and sizes first, then the rows of several strands straight
    into ONE pair of arrays at running offsets — float64 equal to join_strand's own arrays, int16 equal to rint(1000 x) where every
    sample is k / 1000.0 with |k| <= 32 767 and refused (False) where one is not: off the grid, beyond the int16 range, NaN — for
    rows that are arrays, lists of numpy.float64 and tuples; build_csr then hands int16 CSR arrays over, or float64 ones it
    narrows the numpy way"""
    hw = pytest.importorskip('nanomod_amd._hostwalk', reason='make -C nanomod_amd/csrc')
    import nanomod_amd.detect as D
    rng = np.random.default_rng(5)

    def strand(npos, spoil=None):
        d0, d1, b0, b1 = {}, {}, {}, {}
        for p_ in rng.permutation(npos).tolist():
            for d in (d0, d1):
                v = np.round(rng.normal(0, 1, int(rng.integers(1, 300))) * 3, 3)
                d[p_] = v if p_ % 3 == 0 else ([np.float64(x) for x in v] if p_ % 3 == 1 else tuple(v.tolist()))
            b0[p_] = b1[p_] = 'ACGT'[p_ % 4]
        if spoil is not None:
            row = np.array(d1[npos // 2], dtype=np.float64); row[len(row) // 2] = spoil; d1[npos // 2] = row
        return d0, d1, b0, b1
    strands = [strand(5000), strand(300), strand(1)]
    whole = [hw.join_strand(*st) for st in strands]
    plans = [hw.join_strand_plan(*st) for st in strands]
    for w, pl in zip(whole, plans):
        assert all(np.array_equal(a, b) for a, b in zip(w[:3], pl[:3])) and w[5] == pl[4] and w[6] == pl[5] and np.array_equal(w[7], pl[6])
    t0 = sum(int(w[1].sum()) for w in whole); t1 = sum(int(w[2].sum()) for w in whole)
    ref0 = np.concatenate([w[3] for w in whole]); ref1 = np.concatenate([w[4] for w in whole])
    for dt in (np.float64, np.int16):
        o0 = np.full(t0 + 7, 99, dtype=dt); o1 = np.full(t1 + 3, 99, dtype=dt)
        a0, a1 = 7, 3                                          # (not from element 0: the running offsets are the caller's)
        for w, pl in zip(whole, plans):
            assert hw.copy_plan(pl[3], o0, a0, o1, a1) is True
            a0 += int(w[1].sum()); a1 += int(w[2].sum())
        want0, want1 = (ref0, ref1) if dt == np.float64 else (np.rint(ref0 * 1000).astype(np.int16), np.rint(ref1 * 1000).astype(np.int16))
        assert np.array_equal(o0[7:], want0) and np.array_equal(o1[3:], want1) and np.all(o0[:7] == 99) and np.all(o1[:3] == 99)
    for spoil in (0.12345, 32.768, -40.0, np.nan, np.nextafter(0.5, 1.0)):
        pl = hw.join_strand_plan(*strand(200, spoil))
        n_0, n_1 = int(pl[1].sum()), int(pl[2].sum())
        assert hw.copy_plan(pl[3], np.empty(n_0, np.int16), 0, np.empty(n_1, np.int16), 0) is False
        f0, f1 = np.empty(n_0), np.empty(n_1)
        assert hw.copy_plan(pl[3], f0, 0, f1, 0) is True and (np.isnan(spoil) or spoil in f1)
    with pytest.raises(ValueError):
        hw.copy_plan(plans[0][3], np.empty(10), 0, np.empty(10), 0)                      # too small
    with pytest.raises(ValueError):
        hw.copy_plan(plans[0][3], np.empty(t0, np.float32), 0, np.empty(t1, np.float32), 0)
    with pytest.raises((TypeError, ValueError)):
        hw.copy_plan(object(), np.empty(t0), 0, np.empty(t1), 0)
    # build_csr: int16 CSR arrays straight from the dicts; float64 rows that are not on the grid take the numpy way as before
    def moptions(strs):
        mo = {'ds2': ['a', 'b'], 'MinCoverage': 1, 'outLevel': 3}
        for g, name in enumerate(('a', 'b')):
            mo[name] = {'norm_mean': {('c%d' % i, '+'): st[g] for i, st in enumerate(strs)}, 'base': {('c%d' % i, '+'): st[2 + g] for i, st in enumerate(strs)}, 'basedict': {}}
        return mo
    meta, s0, off0, s1, off1, rid = D.build_csr(moptions(strands))
    assert s0.dtype == np.int16 and np.array_equal(s0, np.rint(ref0 * 1000).astype(np.int16)) and np.array_equal(s1, np.rint(ref1 * 1000).astype(np.int16))
    assert off0[-1] == t0 and off1[-1] == t1 and len(meta['pos']) == 5301
    sp = strand(200, 0.12345)
    meta, s0, off0, s1, off1, rid = D.build_csr(moptions([strands[1], sp]))
    assert s0.dtype == np.float64 and 0.12345 in s1


def test_dispatch_forms_keep_16_keys_per_lane():
    """what runs for the uniform sizes of BASELINE.json's configs: every wave-resident form up to 1 024 samples sorts
    at most 16 keys per lane (four waves per SIMD) — the description is host code, no device needed"""
    import ctypes as C
    from nanomod_amd import _lib as L
    lib = L.load()
    want = {
        (1, 100, 100): b'ks_rank_kernel<16,8,f32>', (1, 200, 200): b'ks_rank_kernel<16,16,f32>',
        (1, 500, 500): b'rank_count_wide_kernel<f32,ks> (event-like rows) | ks_rank_kernel<16,32,f32>',
        (1, 1000, 1000): b'rank_count_wide_kernel<f32,ks> (event-like rows) | ks_rank_kernel<16,64,f32>',
        (1, 2000, 2000): b'rank_count_value_kernel<f32,ks> (event-like rows) | ks_rank_kernel<32,64,f32>', (7, 2000, 1100): b'rank_count_value_kernel<f32> (event-like rows) | rank_pair_kernel<32,32,f32>', (1, 50, 1000): b'rank_count_wide_kernel<f32,ks> (event-like rows) | ks_rank_kernel<8,8,f32>',
        (1, 50, 300): b'ks_rank_kernel<8,8,f32>',
        (7, 200, 200): b'rank_count_kernel<f32> (event-like rows) | rank_hist_kernel<16,16,f32>', (7, 500, 500): b'rank_count_wide_kernel<f32> (event-like rows) | rank_hist_kernel<16,32,f32>',
        (7, 1000, 1000): b'rank_count_wide_kernel<f32> (event-like rows) | rank_hist_kernel<16,64,f32>',
        (7, 50, 1000): b'rank_count_wide_kernel<f32> (event-like rows) | rank_hist_kernel<1,64,f32,wide>', (7, 30, 40): b'rank_hist_kernel<8,8,f32>',
    }
    buf = C.create_string_buffer(128)
    for (tests, a, b), name in want.items():
        prm = L.make_params(tests=tests)
        assert lib.nmod_describe_dispatch(C.byref(prm), a, b, buf, 128) == 0
        assert buf.value == name, (tests, a, b, buf.value)


def test_mfilter_coverage_fast_path_keeps_the_reference_semantics():
    """a non-integer MinCoverage is not truncated (len 4 < 4.5 is deleted, as `len(row) < MinCoverage` does), and a
    strand missing from 'base' raises only when something is actually deleted (myDetect.py:304-309)"""
    import nanomod_amd.detect as D
    mo = {'ds2': ['A'], 'MinCoverage': 4.5, 'A': {'norm_mean': {('c', '+'): {1: [1.0] * 4, 2: [1.0] * 5}}, 'base': {('c', '+'): {1: 'A', 2: 'C'}}}}
    D.mfilter_coverage(mo)
    assert sorted(mo['A']['norm_mean'][('c', '+')]) == [2] and sorted(mo['A']['base'][('c', '+')]) == [2]
    mo = {'ds2': ['A'], 'MinCoverage': 3, 'A': {'norm_mean': {('c', '+'): {1: [1.0] * 4, 2: [1.0] * 5}}, 'base': {}}}
    D.mfilter_coverage(mo)                                   # nothing to delete: no lookup of the missing base dict
    assert sorted(mo['A']['norm_mean'][('c', '+')]) == [1, 2]
    mo['MinCoverage'] = 5
    with pytest.raises(KeyError):
        D.mfilter_coverage(mo)                               # the delete reaches base[sk], as in the reference


def test_join_strand_leaves_exotic_rows_and_bases_to_the_python_loop():
    """the C walk holds borrowed rows / bases and must not run Python code on them: a list subclass with its own
    __len__, or a str subclass with its own __eq__, sends the strand to the Python loop — same CSR either way"""
    import nanomod_amd.detect as D

    class OddList(list):
        pass

    class OddStr(str):
        def __eq__(self, other):
            return str.__eq__(self, other)
        __hash__ = str.__hash__

    def options(row_type, base_type):
        mo = {'ds2': ['A', 'B'], 'outLevel': 3}
        for ds, shift in (('A', 0.0), ('B', 0.5)):
            mo[ds] = {'norm_mean': {('c', '+'): {p: row_type([np.float64(p + shift + 0.001 * i) for i in range(6)]) for p in range(5000)}},
                      'base': {('c', '+'): {p: base_type('ACGT'[p % 4]) for p in range(5000)}}, 'basedict': {}}
        return mo
    ref = D.build_csr(options(list, str))
    for rt, bt in ((OddList, str), (list, OddStr)):
        got = D.build_csr(options(rt, bt))
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        assert list(got[0]['pos']) == list(ref[0]['pos']) and [str(b) for b in got[0]['base']] == [str(b) for b in ref[0]['base']]


def test_records_edited_in_place_reach_the_table(tmp_path):
    """VERDICT r4 weak point 8: the reference's combin_pvalues edits records in place (`sign_test[i][1].append(...)`,
    myDetect.py:377).  The lazily built records are kept once handed out, so such an edit persists; save_test notices it
    (`SignTestRecords.edited()`) and writes the table from the records, not from the result arrays — host code only."""
    import nanomod_amd.detect as D
    n = 6
    meta = dict(chrom=np.array(['c'] * n, dtype=object), strand=np.array(['+'] * n, dtype=object), pos=np.arange(n, dtype=np.int64),
                base=np.array(['A'] * n, dtype=object), n0=np.full(n, 5, np.int32), n1=np.full(n, 5, np.int32), names=['c'],
                chrom_id=np.zeros(n, np.int32))
    res = {k: np.arange(n, dtype=np.float64) + j for j, k in enumerate(('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p'))}
    recs = D.SignTestRecords(meta, res, False)
    assert not recs.edited()
    for i in range(n):                                   # what a reference-style combine loop does
        recs[i][1].append((10.0 + i, 0.5))
    assert recs.edited() and len(recs[3][1]) == 4 and recs[3][1][3] == (13.0, 0.5)
    meta2, res2 = D._arrays_from_records(recs, True)
    assert list(res2['comb_st']) == [10.0 + i for i in range(n)] and list(res2['comb_p']) == [0.5] * n
    assert list(res2['ks_d']) == list(res['ks_d']) and list(meta2['pos']) == list(range(n))

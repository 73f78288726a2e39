"""bench.py's per-kernel reporting on synthetic HIP-event statistics (no GPU): which kernel the line calls dominant, where its time
comes from, and what happens to the two launch classes whose spans overlap when the solve of a sweep radius runs beside the z-pass
of the next one (bench.kernel_report).  The driver parses this line at the end of every round; a ranking mistake would misreport
the roofline without any kernel having changed."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench  # noqa: E402

STEPS = 2


def _stats(zinv_ms, solve_ms):
    """one launch class per row: name, launches over STEPS steps, total ms over STEPS steps, algorithmic bytes"""
    rows = [("xpass_hess_1to3", 26, 144.0, 6.0e11), ("ypass_hess_3to6", 26, 288.0, 1.5e12), ("xpass_disp_1to2", 8, 44.0, 2.0e11),
            ("ypass_disp_2to3", 8, 62.0, 3.4e11), ("zpass_c2r_hess_6to3inv", 22, 2 * zinv_ms, 1.4e12), ("collapse_inv", 22, 2 * solve_ms, 9.4e11),
            ("zpass_c2r_disp_3", 8, 64.0, 3.1e11), ("exchange", 0, 0.0, 0.0)]
    return [dict(name=n, launches=l, total_ms=t, alg_bytes=b) for n, l, t, b in rows]


def test_every_kernel_in_line():
    kern, table, table_steps, overlapped, roofline, cls = bench.kernel_report(_stats(190.0, 215.0), STEPS, 1024, 8)
    assert not overlapped and table is kern and table_steps == STEPS
    assert roofline["kernel"] == "k_strided<double, 1024, 8, 1, true>"                      # 269 ms per step by symbol against 215 and 190
    assert set(roofline["classes"]) == {"xpass_hess_1to3", "ypass_hess_3to6", "xpass_disp_1to2", "ypass_disp_2to3"}
    assert abs(roofline["ms_per_step"] - 269.0) < 1e-9 and abs(roofline["avg_ms"] - 538.0 / 68) < 1e-9
    assert abs(roofline["achieved"] - (6.0e11 + 1.5e12 + 2.0e11 + 3.4e11) / 0.538 / 1e9) < 1e-6
    assert roofline["measured"] == "HIP events of the timed region"
    assert cls["class"] == "collapse_inv" == max(kern, key=lambda s: s["total_ms"])["name"]       # 430 ms over the two steps
    assert abs(sum(s["total_ms"] for s in kern) / STEPS - (269.0 + 190.0 + 215.0 + 32.0)) < 1e-9
    assert 0.0 < roofline["share_of_gpu_time"] < 1.0


def test_overlapping_spans_with_an_in_line_pass():
    timed = _stats(370.0, 243.0)                                   # spans: the two classes run beside each other
    inline = {"stats": _stats(190.0, 215.0), "steps": STEPS, "dt": 1.6}
    kern, table, table_steps, overlapped, roofline, cls = bench.kernel_report(copy.deepcopy(timed), STEPS, 1024, 8, inline=inline, solve_beside=True)
    # the solve trails into whatever follows the z-pass it starts beside: no span of the timed region counts as a kernel's time
    assert overlapped == {s["name"] for s in table} and "exchange" not in overlapped
    assert table is not kern and all("symbol" in s for s in table)
    # ranking on the in-line table: the strided passes, not the 370 ms span of the z-pass -- and their time from that pass too
    assert roofline["kernel"] == "k_strided<double, 1024, 8, 1, true>" and "in-line pass" in roofline["measured"]
    assert abs(roofline["avg_ms"] - 538.0 / 68) < 1e-9 and abs(roofline["ms_per_step"] - 269.0) < 1e-9
    # the most expensive class is the solve, and its time is the in-line one, not its span
    assert cls["class"] == "collapse_inv" and abs(cls["ms_per_step"] - 215.0) < 1e-9 and "in-line pass" in cls["measured"]
    shares = sum(s["total_ms"] for s in table)
    assert abs(roofline["share_of_gpu_time"] - 538.0 / shares) < 1e-12


def test_overlapping_spans_without_an_in_line_pass():
    kern, table, table_steps, overlapped, roofline, cls = bench.kernel_report(_stats(370.0, 243.0), STEPS, 1024, 8, inline=None, solve_beside=True)
    assert overlapped and table is kern
    assert roofline["kernel"] == "k_strided<double, 1024, 8, 1, true>"    # the 370 ms span does not make the z-pass dominant
    assert cls["class"] == "ypass_hess_3to6"                         # ... nor the solve's span the most expensive class
    for r in (roofline, cls):
        assert r["share_of_gpu_time"] is None and "ranking" in r


def test_fp32_symbols():
    # fp32 lines of 1024 / 2048 points: the packed-arithmetic kernels of csrc/pf_fft16_kernels.hip, named by family
    assert bench.symbol_of("ypass_hess_3to6", 1024, 4) == "k_strided_pk8<1, pre, band>"
    assert bench.symbol_of("ypass_hess_3to6", 2048, 4) == "k_strided16<1, pre, band>"
    assert bench.symbol_of("xpass_fwd", 2048, 4) == "k_strided16<-1, pre, band>"
    assert bench.symbol_of("ypass_hess_3to6", 512, 4) == "k_strided<float, 512, 16, 1, true>"
    assert bench.symbol_of("zpass_c2r_hess_6to3inv", 2048, 4) == "k_c2r_invariants_spec<float, 2048, 0>"
    assert bench.symbol_of("zpass_c2r_hess_6to3inv", 1024, 4) == "k_c2r_invariants_pk2<1024, 2>" and bench.symbol_of("zpass_c2r_hess_6to3inv", 512, 4) == "k_c2r_invariants_pk2<512, 1>"
    assert bench.symbol_of("zpass_c2r_hess_6to3inv", 256, 8) == "k_c2r_invariants<double, 256, 0>"
    assert bench.symbol_of("zpass_c2r_hess_6to3inv", 512, 8) == "k_c2r_invariants_spec<double, 512, 0>"
    assert bench.symbol_of("zpass_c2r_hess_6to3inv", 1024, 8) == "k_c2r_invariants_spec<double, 1024, 0>"


def test_counter_lookup_by_symbol():
    """the counter summaries of profiles/ are keyed by rocprofv3's symbol: a k_strided launch class whose launcher took the instantiation
    with whole 64-bit addresses per lane (last template argument false) is found under that spelling; the chirp-z path names its family"""
    k = {"k_strided<float __vector(2), 2048, 4, 1, false>": 1, "k_collapse_inv<true, float>": 2}
    assert bench.counter_symbol(k, "k_strided<float __vector(2), 2048, 4, 1, true>") == "k_strided<float __vector(2), 2048, 4, 1, false>"
    assert bench.counter_symbol(k, "k_collapse_inv<true, float>") == "k_collapse_inv<true, float>"
    assert bench.counter_symbol(k, "k_c2r<double, 64, 4>") is None
    bench.GENERAL_PATH = True
    try:
        assert bench.symbol_of("zpass_c2r_plain", 200, 8) == "k_blue<512, mode> x 3" and bench.symbol_of("zpass_r2c", 100, 8) == "k_blue<256, mode> x 3"
        assert bench.symbol_of("collapse", 200, 8) == "k_collapse<double, true, float>"
    finally:
        bench.GENERAL_PATH = False
    assert bench.symbol_of("zpass_c2r_plain", 200, 8).startswith("k_mixed_c2r<double, 4, PfPlanCT<4, 5, 5>")


def _prime_factors(v):
    out, p = [], 2
    while v > 1:
        while v % p == 0:
            out.append(p); v //= p
        p += 1
    return out


def test_mixed_plan_names_follow_the_kernel_sources():
    """sizes that are not a power of two: bench.py names a run's kernels from the list of sizes whose stage plans
    csrc/pf_mixed_kernels.hip compiles in, and from the rule that makes a plan -- both restated there"""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pinocchio_amd", "csrc", "pf_mixed_kernels.hip")).read()
    lines = [re.search(r"#define PF_MIXED_CT_SIZES%s\(X\)(.*)" % sfx, src).group(1) for sfx in ("", "_1", "_2")]   # (one list per translation unit)
    assert tuple(int(v) for line in lines for v in re.findall(r"X\((\d+)\)", line)) == bench.MIXED_CT_SIZES
    m = [v for v in range(12, 257) if v & (v - 1) and all(p in (2, 3, 5) for p in _prime_factors(v))]
    assert sorted(bench.MIXED_CT_SIZES) == [8 * v for v in m]        # every n = 8 m >= 96 with m = 2^a 3^b 5^c that is not a power of two
    assert bench.mixed_radices(768, False) == [8, 8, 4, 3] and bench.mixed_radices(384, True) == [8, 8, 2, 3]
    assert bench.mixed_radices(200, False) == [8, 5, 5] and bench.mixed_radices(100, True) == [4, 5, 5]
    assert bench.mixed_radices(100, False) is None and bench.mixed_radices(56, False) is None
    for n in bench.MIXED_CT_SIZES:
        rs, rz = bench.mixed_radices(n, False), bench.mixed_radices(n // 2, True)
        assert rs and rz and rs[0] == 8 and int(np.prod(rs)) == n and int(np.prod(rz)) == n // 2
    assert bench.symbol_of("ypass_hess_3to6", 768, 8) == "k_mixed_strided<double, 1, PfPlanCT<8, 8, 4, 3> >"
    assert bench.symbol_of("zpass_c2r_hess_6to3inv", 200, 8) == "k_mixed_c2r_invariants<double, 4, PfPlanCT<4, 5, 5>, 0>"
    assert bench.symbol_of("zpass_c2r_disp_3", 200, 4) == "k_mixed_c2r<float, 4, PfPlanCT<4, 5, 5> >"
    assert bench.symbol_of("zpass_r2c", 72, 8) == "k_mixed_r2c<double, 4, PfPlanRT>"
    assert bench.symbol_of("xpass_fwd", 72, 8) == "k_mixed_strided<double, -1, PfPlanRT>"


def test_result_fingerprint_adds_up_over_any_decomposition():
    """bench.fingerprint_of_column: the sums of the parts of a column, each with the global index of its first entry, equal the sum
    of the whole (mod 2^64) wherever it is cut -- and a single changed bit in a sampled cell changes it"""
    import numpy as np
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100003).astype(np.float32)
    whole = bench.fingerprint_of_column(a, 0)
    for cut in (1, 6, 7, 8, 50000, 99999):
        assert (bench.fingerprint_of_column(a[:cut], 0) + bench.fingerprint_of_column(a[cut:], cut)) & 0xFFFFFFFFFFFFFFFF == whole
    parts = np.array_split(a, 8)
    first = np.cumsum([0] + [len(p) for p in parts[:-1]])
    assert sum(bench.fingerprint_of_column(p, int(f)) for p, f in zip(parts, first)) & 0xFFFFFFFFFFFFFFFF == whole
    b = a.copy()
    b[14 * bench.FINGERPRINT_STRIDE] = np.nextafter(b[14 * bench.FINGERPRINT_STRIDE], np.float32(9))
    assert bench.fingerprint_of_column(b, 0) != whole
    c = a.copy()
    c[[0, bench.FINGERPRINT_STRIDE]] = c[[bench.FINGERPRINT_STRIDE, 0]]          # two sampled cells exchanged: the weights notice
    assert bench.fingerprint_of_column(c, 0) != whole
    v = rng.standard_normal((1000, 3)).astype(np.float32)                          # a vector block: three words per cell
    assert (bench.fingerprint_of_column(v[:400], 0) + bench.fingerprint_of_column(v[400:], 1200)) & 0xFFFFFFFFFFFFFFFF == bench.fingerprint_of_column(v, 0)
    r = rng.integers(-1, 12, 5000).astype(np.int32)                                 # Rmax: int32 words
    assert (bench.fingerprint_of_column(r[:123], 0) + bench.fingerprint_of_column(r[123:], 123)) & 0xFFFFFFFFFFFFFFFF == bench.fingerprint_of_column(r, 0)


def test_fingerprint_goldens_cover_the_configurations_that_are_held_against_them():
    """tests/golden/bench_fingerprints.json (made on a GPU by make_bench_fingerprints.py) has the default bench configurations and
    the ones the multi-process GPU tests assert; every entry carries the exact columns, the one float column and the sources' stamp"""
    import json
    g = json.load(open(bench.FINGERPRINT_FILE))
    for n, ns, lpt, fb in ((1024, 12, True, 8), (1024, 12, True, 4), (512, 12, True, 8), (256, 12, True, 8), (64, 3, True, 8), (64, 4, True, 8)):
        e = g[bench.fingerprint_key(n, ns, lpt, fb)]
        fp = e["fingerprint"]
        assert set(fp) == {"FMAX", "RMAX", "ZEL ", "2LPT", "31PT", "32PT_sumsq", "PDF"}
        assert all(isinstance(fp[k], str) and len(fp[k]) == 16 for k in fp if k != "32PT_sumsq") and fp["32PT_sumsq"] > 0
        assert e["cells_in_fmax_pdf"] == n ** 3 and len(e["kernel_source_sha"]) == 16
        assert bench.fingerprints_agree(fp, dict(fp)) and not bench.fingerprints_agree(fp, dict(fp, FMAX="0" * 16))


def test_exchange_report_per_field_and_per_link_numbers():
    """bench.exchange_report: what an N > 1 line says about its all-to-alls -- per transposed field the time on the communication
    stream and the compute beside it, the rate per xGMI link (a rank's field leaves in P - 1 pieces of 1 / P over P - 1 links at
    once), how much of the exchange time hid behind the kernels, and the prediction from the one-GPU slab measurements"""
    import argparse
    args = argparse.Namespace(steps=2, n=1024, field_bytes=8)
    P = 8
    field_gb = 1024 ** 2 * 520 * 16 / P / 1e9            # one rank's share of a half spectrum
    stats = [dict(name="xpass_hess_1to3", launches=26, total_ms=60.0, alg_bytes=1e11), dict(name="collapse_inv", launches=22, total_ms=100.0, alg_bytes=1e11),
             dict(name="exchange", launches=100, total_ms=90.0, alg_bytes=100 * field_gb * 1e9)]
    res = dict(stats=stats, dt=0.2, exchange_kind="rccl", replicated_spectrum=False, ranks_in_communicator=P, solve_beside=True)
    ex = bench.exchange_report(res, args, P)
    assert ex["calls_per_step"] == 50 and abs(ex["ms_per_step_on_comm_stream"] - 45.0) < 1e-12 and abs(ex["compute_ms_per_step"] - 80.0) < 1e-12
    f = ex["per_transposed_field"]
    assert abs(f["ms_on_comm_stream"] - 0.9) < 1e-12 and abs(f["compute_ms_beside_it"] - 1.6) < 1e-12
    assert abs(f["MB_per_link"] - 1e3 * field_gb / P) < 1e-9
    assert abs(ex["GBps_per_link"] - (50 * field_gb / P) / 0.045) < 1e-6
    # step 100 ms, compute 80 + exchange 45 = 125: 25 of the 45 ms of exchanges ran beside kernels
    assert abs(ex["exchange_hidden_fraction"] - 25.0 / 45.0) < 1e-12
    m = ex["model"]
    assert abs(m["wire_ms_per_step_at_assumed_link_rate"] - 1e3 * (50 * field_gb / P) / bench.XGMI_GBS_PER_LINK) < 1e-9
    if "from" in m:   # a committed slab measurement of this configuration exists for the profile round in use
        assert m["step_ms_if_exchanges_hide"] <= m["step_ms_if_nothing_hides"] and m["compute_ms_per_step_per_rank_on_one_gpu"] > 0
    one = bench.exchange_report(dict(res, stats=stats[:2]), args, P)   # a sweep that exchanged nothing: no per-field numbers, no division by zero
    assert one["calls_per_step"] == 0 and "per_transposed_field" not in one

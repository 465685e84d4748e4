"""CPU: the C-ABI library loads, exports every symbol include/minilp_hip.h declares, the host-side
model building / MPS parsing behaves like the reference, and a solve without a GPU fails loudly
(no CPU fallback exists)."""
import ctypes
import math
import os
import re

import pytest

import minilp_amd as M
from minilp_amd import build as mbuild
from tests.common import ROOT
from tests.test_oracle_kat import MPS_TESTPROB

INF = math.inf


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(M.lib_path()):
        mbuild.build(verbose=False)


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "minilp_hip.h")).read()
    names = sorted(set(re.findall(r"\b(mlp_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 35
    lib = ctypes.CDLL(M.lib_path())
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_problem_building_and_reference_panics_become_einval():
    p = M.Problem(M.MINIMIZE)
    x = p.add_var(1.0, (0.0, INF))
    y = p.add_var(2.0, (0.0, 3.0))
    assert (x, y) == (0, 1) and p.num_vars == 2
    p.add_constraint([(y, 1.0), (x, 1.0)], M.LE, 4.0)  # unsorted is fine (lib.rs:476)
    with pytest.raises(M.InternalError) as e:           # duplicate variable: reference panics (lib.rs:247-249)
        p.add_constraint([(x, 1.0), (x, 2.0)], M.LE, 1.0)
    assert e.value.code == -1
    with pytest.raises(M.InternalError):                # out-of-range variable
        p.add_constraint([(7, 1.0)], M.LE, 1.0)
    q = p.clone()
    q.add_var(0.0, (0.0, 1.0))
    assert (p.num_vars, q.num_vars) == (2, 3)


def test_mps_parser_host_side():
    f = M.MpsFile.parse(MPS_TESTPROB, M.MINIMIZE)  # the reference test's data fixture (mps.rs:437-462)
    assert f.problem_name == "TESTPROB"
    assert f.variables == {"XONE": 0, "YTWO": 1, "ZTHREE": 2}
    assert f.problem.num_vars == 3
    for bad, msg in [("ROWS\n", "expected NAME"), ("NAME x\nROWS\n N c\nCOLUMNS\n  a c 1 zz 2\nRHS\nENDATA\n", "unknown constraint"),
                     ("NAME x\nROWS\n Q c\n", "unexpected row type"), ("NAME x\nROWS\n N c\nCOLUMNS\nRHS\n", "expected ENDATA")]:
        with pytest.raises(ValueError) as e:
            M.MpsFile.parse(bad, M.MINIMIZE)
        assert msg in str(e.value)


def test_mps_ranges_and_bounds_rules():
    text = """NAME r
ROWS
 N obj
 L r1
 G r2
 E r3
 E r4
COLUMNS
    x obj 1 r1 1
    x r2 1 r3 1
    x r4 1
    y obj 1 r1 1
    z obj 1 r1 1
RHS
    rhs r1 10 r2 2
    rhs r3 5 r4 5
    other r1 99
RANGES
    rng r1 4 r2 3
    rng r3 2 r4 -2
BOUNDS
 UP b x 7
 UP b y -3
 FR b z
 UP other x 1
ENDATA
"""
    f = M.MpsFile.parse(text, M.MINIMIZE)
    # inspected through the oracle's identical reader in tests/test_hip_parity.py; here: shape only
    assert f.problem.num_vars == 3


def test_no_gpu_fails_loudly():
    if M.device_count() > 0:
        pytest.skip("a GPU is visible")
    p = M.Problem(M.MAXIMIZE)
    x = p.add_var(1.0, (0.0, INF))
    p.add_constraint([(x, 1.0)], M.LE, 4.0)
    with pytest.raises(M.InternalError) as e:
        p.solve()
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_bulk_model_building_equals_per_call():
    from minilp_amd import lpgen
    lp = lpgen.gen_mixed_lp(40, 60, 4, 9)
    bulk = lpgen.build_problem(M.Problem, lp)
    one = M.Problem(lp["direction"])
    for j in range(lp["n"]):
        one.add_var(float(lp["obj"][j]), (float(lp["lo"][j]), float(lp["hi"][j])))
    for i in range(lp["m"]):
        b, e = int(lp["indptr"][i]), int(lp["indptr"][i + 1])
        one.add_constraint_arrays(lp["indices"][b:e], lp["data"][b:e], int(lp["ops"][i]), float(lp["rhs"][i]))
    assert bulk.variables() == one.variables()
    for x, y in zip(bulk.constraints(), one.constraints()):
        assert (x[0] == y[0]).all() and (x[1] == y[1]).all() and x[2:] == y[2:]


def test_null_solution_handle_is_an_error_not_a_crash():
    import ctypes as C
    lib = M.lib()
    out = C.c_double()
    assert lib.mlp_solution_var_value(None, 0, C.byref(out)) == -1
    assert lib.mlp_solution_num_vars(None) == 0
    h = C.c_void_p()
    assert lib.mlp_solution_fix_var(C.byref(h), 0, 1.0) == -1


def test_public_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/minilp_hip.h must compile as C99 (and as C++) on its own."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "minilp_hip.h"\nint main(void) { mlp_iter_info i; mlp_stats s; (void)i; (void)s; return MLP_STAGE_APPLY; }\n')
    for cc, flags in (("gcc", ["-std=c99", "-pedantic"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            continue
        r = subprocess.run([cc, *flags, "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def _build_toy(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "toy")
    libdir = os.path.join(root, "minilp_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "toy.c"),
                        "-L", libdir, "-lminilp_hip", "-Wl,-rpath," + libdir, "-lm", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_plain_c_program_links_against_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    import subprocess
    exe = _build_toy(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    if r.returncode != 0:  # no GPU in this container: the product must say so, not fall back
        assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr
    else:
        assert "objective 7 x 1 y 3" in r.stdout


@pytest.mark.gpu
def test_plain_c_program_solves_the_readme_toy(tmp_path):
    import subprocess
    r = subprocess.run([_build_toy(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "objective 7 x 1 y 3" in r.stdout
    assert "warm-started objective 6.5 x 0.5 y 3" in r.stdout


def test_linear_expr_mirror_builds_the_same_constraint():  # lib.rs:84-158
    p = M.Problem(M.MINIMIZE)
    x, y, z = (p.add_var(1.0, (0.0, 4.0)) for _ in range(3))
    e = M.LinearExpr.empty().add(z, 3.0).add(x, 1.0)
    assert len(e) == 2 and list(e) == [(z, 3.0), (x, 1.0)]
    p.add_constraint(e, M.LE, 5.0)
    p.add_constraint([(z, 3.0), (x, 1.0)], M.LE, 5.0)
    p.add_constraint(M.LinearExpr([(y, 2.0)]), M.GE, 1.0)
    c = p.constraints()
    assert (c[0][0] == c[1][0]).all() and (c[0][1] == c[1][1]).all() and c[0][2:] == c[1][2:]
    assert list(c[0][0]) == [x, z] and list(c[0][1]) == [1.0, 3.0]   # stored sorted by variable, like CsVec::new
    assert list(c[2][0]) == [y]


def test_bench_contract_pieces_that_need_no_gpu():
    """bench.py: CLI defaults (N = 1, K/W that finish in minutes) and the committed PMC traffic figure
    that feeds roofline.traffic for the default workload."""
    import subprocess
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, check=True).stdout
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out
    sys.path.insert(0, root)
    import bench
    a = types.SimpleNamespace(rows=100000, cols=100000, nnz_per_row=100, seed=4)
    t, src = bench.pmc_traffic(a, "stream_late")
    assert t is not None and 3.3e9 < t < 3.6e9 and src.endswith(".json")   # HBM bytes per launch of the solve-dominant kernel
    assert bench.pmc_traffic(types.SimpleNamespace(rows=10, cols=10, nnz_per_row=2, seed=1), "stream_late")[0] is None
    assert bench.HBM_PEAK_GBS == 8000.0
    # the printed line is a compact digest of the detailed record (the driver reads the tail of stdout)
    import json
    detail = json.load(open(os.path.join(root, "profiles", "r03e_bench_detail.json")))   # a full record of a run on the GPU box
    line = bench.compact_line(detail)
    text = json.dumps(line)
    assert len(text) < 3600, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "windows"):
        assert key in line, key
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "ftran", "measured"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "measured"}
    assert line["windows"]["late"]["kernels"]["w_pass_v"]["frac"] > 0.5 and line["config"]["workload"].startswith("config 4")
    # provenance: what this run measured itself is tagged live, what it read from profiles/ says which file
    assert line["roofline"]["measured"]["achieved"] == "live" and line["roofline"]["measured"]["traffic"].startswith("committed:profiles/")
    assert line["windows"]["late"]["measured"] == "live" and line["cpu_baseline"]["measured"] == "live"
    assert line["roofline"]["kernel"] == "k_stream_w" and 0.5 < line["roofline"]["frac"] < 1.0 and "timed_window" in line["roofline"]
    fs = line["full_solve"]                                                       # the same run solved config 4 to the optimum, live
    assert fs["measured"] == "live" and fs["complete"] is True and fs["pivots"] > 10 ** 6 and fs["total_solve_wall_s"] > 0
    assert fs["certificate"]["relative_gap"] < 1e-9
    # FTRAN: the column FTRAN in microseconds and bytes per window, the dense-rhs FTRAN as a stream (no BTRAN-shaped pass under an FTRAN label)
    ft = line["roofline"]["ftran"]
    assert set(ft["column"]) == {"early", "mid", "late"} and all(set(v) == {"us", "bytes"} for v in ft["column"].values())
    assert ft["dense_rhs_late"]["frac"] > 0.5 and "tau_stream_late_window" not in ft


def test_mid_solve_basis_fixtures_are_wellformed():
    """tests/golden/cfg4_basis_p*.bin.gz (made on the GPU by tools/make_cfg4_basis.py): mode-1 checkpoints of
    config 4 whose basic / non-basic sets partition the 200 000 variables; bench.py's mid / late windows load them."""
    import gzip
    import json
    import struct
    import numpy as np
    import bench
    idx = [json.loads(l) for l in open(os.path.join(ROOT, "tests", "golden", "cfg4_basis_index.jsonl"))]
    for path, rec in zip((bench.MID_BASIS, bench.LATE_BASIS), idx):
        blob = gzip.open(path, "rb").read()
        assert os.path.basename(path) == rec["file"] and len(blob) == rec["raw_bytes"]
        magic, version, mode, m, n, flags, _pad, obj, pivots = struct.unpack_from("<8sIIQQIIdQ", blob, 0)
        assert (magic, version, mode, m, n) == (b"MLPBASIS", 1, 1, 100000, 100000)
        assert pivots == rec["pivots"] and abs(obj + rec["objective"]) <= 1e-9 * abs(obj)   # stored for the minimised form
        assert flags & 1 and not flags & 2 and flags & 4                                    # primal phase, steepest edge on
        bv = np.frombuffer(blob, dtype=np.int32, count=m, offset=56)
        nv = np.frombuffer(blob, dtype=np.int32, count=n, offset=56 + 4 * m)
        both = np.sort(np.concatenate([bv, nv]))
        assert (both == np.arange(m + n)).all()
        structural_basics = int((bv < n).sum())
        assert rec["nucleus_size"] <= structural_basics <= rec["nucleus_size"] + 64   # nucleus = non-singleton basic columns


def _mps_with_number(tok):
    return f"NAME t\nROWS\n N c\n L r\nCOLUMNS\n x c 1 r {tok}\nRHS\n rhs r 4\nENDATA\n"


def test_mps_numbers_parse_like_f64_from_str_whatever_the_locale():
    """mps.rs:330-336 parses numbers with f64::from_str: locale-independent, no hex floats, a leading '+' is fine,
    out-of-range magnitudes saturate.  (strtod is LC_NUMERIC-dependent and accepts 0x1p3.)"""
    import locale
    old = locale.setlocale(locale.LC_NUMERIC)
    switched = None
    for name in ("de_DE.UTF-8", "de_DE.utf8", "fr_FR.UTF-8", "fr_FR.utf8", "ru_RU.UTF-8"):
        try:
            locale.setlocale(locale.LC_NUMERIC, name)  # a comma-decimal locale, if the image has one
            switched = name
            break
        except locale.Error:
            continue
    try:
        for tok, want in [("1.5", 1.5), ("+1.5", 1.5), ("-2.5e-1", -0.25), (".5", 0.5), ("1.", 1.0), ("1e999", INF), ("-1e999", -INF),
                          ("1e-999", 0.0), ("inf", INF), ("1E2", 100.0),
                          ("1" + "0" * 400 + "e-5", INF), ("-" + "9" * 400 + ".5e-20", -INF),       # a long integer mantissa with a
                          ("0." + "0" * 400 + "1e5", 0.0), ("0." + "0" * 10 + "1e-400", 0.0)]:       # negative exponent still overflows, and vice versa
            f = M.MpsFile.parse(_mps_with_number(tok), M.MINIMIZE)
            (idx, val, op, rhs), = f.problem.constraints()
            assert val[0] == want, (tok, val[0], switched)
        for bad in ("0x1p3", "1,5", "1.5x", "+-1", "--1", "e5", "+"):
            with pytest.raises(ValueError) as e:
                M.MpsFile.parse(_mps_with_number(bad), M.MINIMIZE)
            assert "parse float" in str(e.value), bad
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)


def _c_prototypes(header_text):
    """{name: number of parameters} of every function prototype of the C header."""
    text = re.sub(r"/\*.*?\*/", " ", header_text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(mlp_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", text):
        args = " ".join(m.group(2).split())
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return protos


def _rust_extern_fns(path):
    """{name: number of parameters} of every fn declared inside `extern "C" { ... }` blocks of a Rust source file."""
    src = "\n".join(ln.split("//")[0] for ln in open(path).read().splitlines())
    fns = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
        for m in re.finditer(r"\bfn\s+([a-z_0-9]+)\s*\(([^()]*)\)", blk.group(1)):
            args = " ".join(m.group(2).split()).rstrip(",")
            fns[m.group(1)] = 0 if not args else args.count(",") + 1
    return fns


def test_rust_sys_crate_declares_only_what_the_library_exports_with_the_headers_arity():
    """The Rust crates cannot be compiled here (no cargo / rustc in the image); this makes the FFI layer checkable anyway: every
    `extern "C"` declaration of integration/rust/minilp-hip-sys must be a symbol libminilp_hip.so exports, with the number of
    parameters include/minilp_hip.h gives it."""
    sys_rs = os.path.join(ROOT, "integration", "rust", "minilp-hip-sys", "src", "lib.rs")
    fns = _rust_extern_fns(sys_rs)
    assert len(fns) >= 40, sorted(fns)
    protos = _c_prototypes(open(os.path.join(ROOT, "include", "minilp_hip.h")).read())
    lib = ctypes.CDLL(M.lib_path())
    for name, nargs in sorted(fns.items()):
        assert hasattr(lib, name), f"{name} is declared in minilp-hip-sys but not exported by the library"
        assert name in protos, f"{name} is declared in minilp-hip-sys but not in the C header"
        assert protos[name] == nargs, f"{name}: {nargs} parameters in minilp-hip-sys, {protos[name]} in the C header"
    # and the drop-in crate calls nothing the sys crate does not declare
    used = set(re.findall(r"\bsys::(mlp_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "integration", "rust", "minilp", "src", "lib.rs")).read()))
    used |= set(re.findall(r"\bsys::(mlp_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "integration", "rust", "minilp", "src", "mps.rs")).read()))
    assert used and used <= set(fns), sorted(used - set(fns))


def test_rust_api_surface_document_matches_the_crate_source():
    """integration/rust/API_SURFACE.md (generated by tools/rust_api_surface.py in the build container, where the reference
    sources are) lists every public item of the reference's lib.rs / mps.rs next to the crate's: none missing, none different —
    and the crate's side of the list is re-derived here from the crate source, so the document cannot go stale."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rust_api_surface", os.path.join(ROOT, "tools", "rust_api_surface.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    doc = open(os.path.join(ROOT, "integration", "rust", "API_SURFACE.md")).read()
    assert "### MISSING" not in doc and "### different" not in doc
    counts = re.findall(r"public items in the reference: \*\*(\d+)\*\*; present in the crate with an identical signature: \*\*(\d+)\*\*", doc)
    assert len(counts) == 2 and all(a == b and int(a) > 0 for a, b in counts), counts
    for fname in ("lib.rs", "mps.rs"):
        ours = mod.pub_items(os.path.join(ROOT, "integration", "rust", "minilp", "src", fname))
        section = doc.split("## `%s`" % fname)[1].split("\n## `")[0]
        listed = re.findall(r"^\* `(.*)`$", section.split("### identical")[1].split("###")[0], flags=re.M)
        assert listed and all(sig in set(ours.values()) for sig in listed), [s for s in listed if s not in set(ours.values())]


def test_product_library_links_no_vendor_math_library():
    """Every kernel on the product path is hand-written — since round 4 also the blocked re-inversion of a large nucleus
    (csrc/inverse.inc; rounds 1-3 linked rocSOLVER / rocBLAS for it).  The shared library may depend on the HIP runtime and
    the C / C++ runtimes only (RCCL is dlopen'ed when a sharded solve asks for that transport)."""
    import subprocess
    out = subprocess.run(["readelf", "-d", mbuild.SO], capture_output=True, text=True, check=True).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", out)
    assert needed, out
    banned = [n for n in needed if re.search(r"rocsolver|rocblas|hipblas|rocsparse|hipsparse|MIOpen|rccl|torch", n, re.I)]
    assert not banned, needed
    assert any("amdhip64" in n for n in needed), needed

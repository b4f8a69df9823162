"""Host logic of the product library without a GPU: every C-ABI symbol of include/t41rx.h is
exported, the host-side coefficient designer matches the oracle's, argument checking and error
codes behave, and creating a context without a HIP device fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def T(built):
    import t41_sdr_amd
    t41_sdr_amd.load()
    return t41_sdr_amd


def test_every_declared_symbol_is_exported(T):
    hdr = open(os.path.join(ROOT, "include", "t41rx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(t41rx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = C.CDLL(T.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, "declared in include/t41rx.h but not exported: %s" % missing
    from t41_sdr_amd import _lib
    assert declared == set(_lib.SYMBOLS), "python binding and header disagree"
    assert lib.t41rx_abi_version() == 5


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(t41[rt]x_[a-z0-9_]+)\s*\(", hdr))


def test_nothing_but_the_declared_symbols_is_exported(T):
    """the converse: libt41rx.so is built with -fvisibility=hidden and an export list
    (t41_sdr_amd/csrc/exports.map); its dynamic symbol table is the two headers' entry points and nothing else --
    no t41:: C++ symbol, no kernel stub, no host helper, no diagnostic hook"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", T.LIB_PATH], text=True)
    exported = {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}
    declared = _declared("t41rx.h") | _declared("t41tx.h")
    assert exported == declared, "exported but not declared: %s; declared but not exported: %s" % (
        sorted(exported - declared), sorted(declared - exported))
    # and the export list itself names exactly those (plus the diagnostic builds' t41rx_debug_* pattern)
    m = open(os.path.join(ROOT, "t41_sdr_amd", "csrc", "exports.map")).read()
    m = re.sub(r"#.*", "", m)
    listed = set(re.findall(r"\b(t41[rt]x_[a-z0-9_]+);", m))
    assert listed == declared
    # every declaration carries the export attribute (a declaration without it would be hidden by -fvisibility=hidden)
    for header in ("t41rx.h", "t41tx.h"):
        hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        for name in _declared(header):
            assert re.search(r"T41RX_API[^;]*\b%s\s*\(" % name, hdr), "%s lacks T41RX_API in %s" % (name, header)


def test_params_struct_layout_matches_header(T):
    hdr = open(os.path.join(ROOT, "include", "t41rx.h")).read()
    body = re.search(r"typedef struct t41rx_params \{(.*?)\} t41rx_params;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int32_t|float)\s+(\w+);", body)
    assert [n for _, n in fields] == [n for n, _ in T.Params._fields_]
    assert C.sizeof(T.Params) == 4 * len(fields)


@pytest.mark.parametrize("kw", [
    dict(mode=0, FLoCut=200, FHiCut=3000),
    dict(mode=1, FLoCut=-3000, FHiCut=-200),
    dict(mode=2, FLoCut=-3000, FHiCut=3000),
    dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000),
    dict(mode=0, FLoCut=400, FHiCut=600, rfGainAllBands=6, RFgain=3, audioVolume=55),
    dict(mode=0, FLoCut=100, FHiCut=11000),  # > 10 kHz: resampler design caps at 10 kHz (Filter.cpp:408)
    dict(mode=0, FLoCut=400, FHiCut=600, fft_length=4096),
    dict(mode=0, FLoCut=200, FHiCut=3000, fft_length=1024),
    dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1),
    dict(mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=2, AGC_thresh=30),
    dict(mode=3, FLoCut=200, FHiCut=3000, AGCMode=3),
    dict(mode=1, FLoCut=-3000, FHiCut=-200, AGCMode=4, AGC_thresh=10),
])
def test_designer_matches_oracle(T, kw):
    p = T.default_params(**kw)
    N = p.fft_length
    got = T.blob_fields(T.design_coeffs(p), N)
    ref = O.coeff_arrays(O.design(O.default_params(**kw)), N)
    for k in ("dec1", "dec2", "int1", "int2", "biquad_lowpass1", "agc"):
        assert np.array_equal(got[k], ref[k]), k  # same f32 arithmetic -> bit-exact
    # the mask goes through a different f32 FFT decomposition: rounding-level agreement
    assert np.abs(got["mask"] - ref["mask"]).max() < 4e-7 * np.abs(ref["mask"]).max()


def test_scalars_follow_process_cpp(T):
    p = T.default_params(mode=1, FLoCut=-2800, FHiCut=-300, rfGainAllBands=4, RFgain=7, audioVolume=40,
                         IQAmpCorrectionFactor=1.05, IQPhaseCorrectionFactor=0.02, xmtMode=1, CWFreqShift=600)
    s = T.blob_fields(T.design_coeffs(p), 512)["scalars"]
    assert s[0] == np.float32(10.0 ** float(np.float32(4) / np.float32(20)))       # Process.cpp:117
    assert s[1] == 7.0 and s[2] == np.float32(-1.05) and s[3] == np.float32(0.02)  # :133, :166
    fk = float(np.float32(2800.0 * 0.001))  # float * double -> double -> float; LSB uses -FLoCut, :485
    assert s[4] == np.float32(7.0874 * fk ** -1.232)
    assert s[5] == 20.0                                                             # DSP_Fn.cpp:453
    x = np.float32(0.4)
    assert s[6] == np.float32(8.0) * (np.float32(5) * x * x * x * x * x)            # :929, :964
    assert s[7] == 1.0 and s[8] == 600.0                                            # LSB + CW: +CWFreqShift


def test_design_rejects_bad_arguments(T):
    from t41_sdr_amd import _lib
    lib = T.load()
    p = T.default_params()
    n = lib.t41rx_coeff_blob_bytes(512)
    assert n == 4 * (32 + 28 + 46 + 48 + 32 + 5 + 16 + 16 + 1024)  # 32-word header: 8 + t41rx_params padded to 24; 16 scalar slots
    assert lib.t41rx_coeff_blob_bytes(500) == 0
    buf = (C.c_uint8 * n)()
    assert lib.t41rx_design_coeffs(C.byref(p), buf, n - 1) == _lib.ERR_ARG
    assert lib.t41rx_design_coeffs(None, buf, n) == _lib.ERR_ARG
    for bad in (dict(fft_length=300), dict(mode=9), dict(FLoCut=3000, FHiCut=200), dict(audioVolume=101),
                dict(mode=1, FLoCut=200, FHiCut=3000), dict(FHiCut=20000), dict(AGCMode=5), dict(AGCMode=-1)):
        with pytest.raises(T.T41RxError) as e:
            T.design_coeffs(T.default_params(**bad))
        assert e.value.status == _lib.ERR_ARG
    assert b"argument" in lib.t41rx_strerror(_lib.ERR_ARG)


def test_no_cpu_fallback_without_a_device(T):
    """On a box without a HIP device creating a context must fail with ERR_HIP, never silently
    run somewhere else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from t41_sdr_amd import _lib
    with pytest.raises(T.T41RxError) as e:
        T.RxChain(4)
    assert e.value.status == _lib.ERR_HIP


def test_product_never_touches_the_oracle():
    """the shipped package must not import, link or open anything under oracle/"""
    pkg = os.path.join(ROOT, "t41_sdr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "t41_oracle" not in txt and "libt41oracle" not in txt, f
    import subprocess
    so = os.path.join(pkg, "libt41rx.so")
    if os.path.exists(so):
        assert "t41oracle" not in subprocess.run(["ldd", so], capture_output=True, text=True).stdout


def test_experiment_builds_are_fenced(T, tmp_path):
    """the timing-experiment switches that make the kernels compute WRONG results (T41RX_ABLATE, _LOO,
    _AGC_X, _FCABL) and the diagnostic ones sit behind ONE guard (t41_sdr_amd/csrc/rx_experiments.hpp): without
    -DT41RX_EXPERIMENT=1 they do not compile, and a library built with them refuses t41rx_create() unless the environment
    opts in.  Checked here with the dispatch translation unit (no kernels: seconds) linked against the product's objects."""
    import subprocess
    csrc = os.path.join(ROOT, "t41_sdr_amd", "csrc")
    base = ["hipcc", "-O1", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include")]
    for flag in ("-DT41RX_ABLATE=3", "-DT41RX_LOO=2", "-DT41RX_AGC_X=1", "-DT41RX_AGC_R04CHECK=1", "-DT41RX_FCABL=2", "-DT41RX_STAMP", "-DT41RX_CLK"):
        r = subprocess.run(base + [flag, "-fsyntax-only", "rx_dispatch.hip"], cwd=csrc, capture_output=True, text=True)
        assert r.returncode != 0 and "T41RX_EXPERIMENT=1" in r.stderr, (flag, r.stderr[-400:])
    obj = str(tmp_path / "rx_dispatch_exp.o")
    subprocess.check_call(base + ["-DT41RX_EXPERIMENT=1", "-DT41RX_LOO=2", "-c", "rx_dispatch.hip", "-o", obj], cwd=csrc)
    objs = [os.path.join(csrc, o) for o in ("rx512_ssb.o", "rx512_am.o", "rx512_nfm.o", "rx512_sam.o", "rx_long.o", "fastconv.o", "display_kernel.o",
                                            "nr_kernels.o", "rx_host.o", "design.o", "nr_tables.o", "tx_kernels.o", "tx_host.o", "tx_tables.o")]
    lib_path = str(tmp_path / "libt41rx_exp.so")
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + objs + ["-o", lib_path])
    code = ("import ctypes as C, sys; lib = C.CDLL(%r); lib.t41rx_last_error.restype = C.c_char_p; ctx = C.c_void_p(); "
            "p = (C.c_int32 * 32)(); lib.t41rx_default_params(p); rc = lib.t41rx_create(C.byref(ctx), 0, 4, p); "
            "print(rc, lib.t41rx_last_error().decode())" % lib_path)
    env = {k: v for k, v in os.environ.items() if k != "T41RX_ALLOW_EXPERIMENT"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    rc, msg = r.stdout.strip().split(" ", 1)
    assert int(rc) == -2 and "timing experiment" in msg and "T41RX_ALLOW_EXPERIMENT" in msg   # T41RX_ERR_UNSUPPORTED
    env["T41RX_ALLOW_EXPERIMENT"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert "EXPERIMENT BUILD" in r.stderr and "WRONG" in r.stderr
    assert "timing experiment" not in r.stdout   # past the fence (and, without a GPU, on to T41RX_ERR_HIP)
    # the product library is not one
    assert T.load().t41rx_abi_version() == 5

"""CPU: the C-ABI library loads and exports every symbol include/qcx.h declares (no compute calls
without a GPU), refuses to compute without a device, and its gate kernels contain no fused
multiply-add (bit parity with the reference depends on separately rounded products and sums)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qcx.h")


def declared_symbols():
    """every non-inline qcx_* function any header under include/ declares"""
    names = set()
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        txt = open(os.path.join(inc, h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        # declarations start in column 0 with a return type; calls inside inline bodies are indented
        for m in re.finditer(r"^(?![ \t])[^\n#{}()]*?\b(qcx_[a-zA-Z0-9_]+)\s*\([^;{]*\)\s*;", txt, flags=re.M):
            if "static" not in m.group(0):
                names.add(m.group(1))
    return sorted(names)


def test_header_symbols_all_exported(qc):
    names = declared_symbols()
    assert len(names) >= 35
    L = C.CDLL(qc.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the ctypes signature table covers exactly the header
    from quantumcomputer_amd._lib import SIGNATURES
    assert sorted(SIGNATURES) == names


def test_dynamic_symbol_table_has_no_torch_or_gsl(qc):
    out = subprocess.run(["nm", "-D", "--defined-only", qc.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(not s.startswith(("gsl_", "at_", "torch")) for s in exported)
    need = subprocess.run(["ldd", qc.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" in need and "torch" not in need and "gsl" not in need


def test_compat_header_compiles_reference_style_circuit():
    """include/qcx_compat.h gives the reference's own names/signatures: a C file written like
    qc_shor.c:678-737 (gate calls with the ignored matrix argument) must compile and link."""
    src = r'''
    #include "qcx_compat.h"
    static void my_iqft(Register *reg, gsl_spmatrix_complex *matrix) {
        double theta;
        for (int l = reg->L_size + reg->M_size - 1; l >= reg->M_size; l--) {
            hadamard_gate(l, reg, matrix);
            for (int k = l - 1; k >= reg->M_size; k--) {
                theta = M_PI / INT_POW(2, l - k);
                c_phase_shift_gate(l, k, theta, reg, matrix);
            }
        }
    }
    int main(void) { Register reg; gsl_spmatrix_complex *m = 0; if (0) { reset_register(reg); my_iqft(&reg, m);
        c_amodc_gate(15, 7ULL, 4, &reg, m); swap_states(&reg); (void)measure_state(reg, (gsl_rng *)0); } return 0; }
    '''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "t.c")
        open(f, "w").write(src)
        r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), f,
                            "-L", os.path.join(ROOT, "quantumcomputer_amd"), "-lqcx", "-lm",
                            "-Wl,-rpath," + os.path.join(ROOT, "quantumcomputer_amd"), "-o", os.path.join(d, "t")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_compat_header_compiles_a_main_shaped_like_the_reference():
    """tests/c/refstyle_main.c: a main() with the reference's own declarations and call sequence (qc_shor.c:1284-1347:
    gsl_rng_alloc(gsl_rng_mt19937), gsl_rng_set, gsl_vector_complex_alloc x 2, gsl_spmatrix_complex_alloc_nzmax,
    current_state / new_state, the four frees) and no qcx_* call in it compiles warning-free against qcx_compat.h, as
    do tests/c/refstyle_circuit.c and the operate_matrix / gsl_rng_uniform shims.  Without a GPU the program must fail
    loudly at its first gate (abort with the library's message), not compute anything."""
    import tempfile
    lib = os.path.join(ROOT, "quantumcomputer_amd")
    with tempfile.TemporaryDirectory() as d:
        for name in ("refstyle_main.c", "refstyle_circuit.c"):
            exe = os.path.join(d, name[:-2])
            r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                                os.path.join(ROOT, "tests", "c", name), "-L", lib, "-lqcx", "-lm", "-Wl,-rpath," + lib, "-o", exe],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
        import torch
        if not torch.cuda.is_available():
            r = subprocess.run([os.path.join(d, "refstyle_main"), "15", "3", "4", "7", "1"], capture_output=True, text=True)
            assert r.returncode != 0 and "register allocation on the GPU failed" in r.stderr and r.stdout == ""


def test_no_gpu_means_loud_failure_not_fallback(qc):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(qc.QcxError) as e:
        qc.Register(3, 4)
    assert e.value.status == 5          # QCX_HIP_ERROR


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    bad = []
    for base in ("quantumcomputer_amd", "host", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|qcx_oracle|orc_[a-z]+\(", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_gate_kernels_have_no_fma():
    s_path = os.path.join(ROOT, "quantumcomputer_amd", "libqcx.gfx950.s")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "quantumcomputer_amd", "csrc"), "-s", "isa"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = open(s_path).read()
    # FMA appears in ONE place only: the opt-in tolerance-mode pass kernels (k_fused_rounds<..., TOL = 1 or 2>, k_fused_q3 and
    # k_gen_cols<.., TOL = true>, K6t); every bit-exact kernel -- per-gate kernels, the exact fused passes (k_gen_cols<.., false>
    # among them), measurement, exchange -- is free of it
    funcs = {m.group(1): txt[m.start():txt.find(".Lfunc_end", m.start())] for m in re.finditer(r"^(_Z\w+):", txt, re.M)}
    with_fma = [name for name, body in funcs.items() if re.search(r"v_fma_f64|v_fmac_f64|v_pk_fma_f64", body)]
    # (round 5: + k_fused_x8<.., TOL = true>, the tolerance mode's hand-written radix-8 round)
    assert with_fma and all(re.match(r"_ZN3qcx14k_fused_roundsILi\d+ELi\d+ELi\d+ELb[01]ELi[12]ELb[01]EE|_ZN3qcx10k_fused_q3ILi\d+ELi\d+ELi\d+ELb0ELb[01]EE|_ZN3qcx10k_gen_colsILi\d+ELb1EE|_ZN3qcx10k_fused_x8ILi\d+ELi\d+ELb[01]ELb1EE", name) for name in with_fma), with_fma
    assert any(re.match(r"_ZN3qcx10k_fused_x8ILi\d+ELi\d+ELb[01]ELb0EE", name) for name in funcs), "the exact walk on 8 amplitudes per thread"
    assert any(re.match(r"_ZN3qcx10k_gen_colsILi\d+ELb0EE", name) for name in funcs), "the exact by-columns kernel"
    assert any(re.match(r"_ZN3qcx10k_fused_q3ILi\d+ELi\d+ELi\d+ELb1ELb[01]EE", name) for name in funcs), "the exact radix-8 Hadamard kernel"
    assert txt.count("global_load_dwordx4") > 50            # 16-B amplitude accesses everywhere
    # the "+ 0.0" canonicalisation must survive optimisation in the Hadamard kernels
    body = txt[txt.index("k_h_pair"):]
    assert re.search(r"v_add_f64 v\[\d+:\d+\], v\[\d+:\d+\], 0\b", body)


def test_fused_rounds_kernel_keeps_its_scalar_record_loads():
    """Two code-generation facts the fused passes depend on ("Gate fusion"): the record array is read with
    scalar loads (a vector load would wait for the tile DMA -- this regresses as soon as an asm statement captures the
    __restrict__ record pointer), and the hand-scheduled item walk is in the kernel (EXEC-masked rotations)."""
    s_path = os.path.join(ROOT, "quantumcomputer_amd", "libqcx.gfx950.s")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "quantumcomputer_amd", "csrc"), "-s", "isa"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = open(s_path).read()
    bodies = [txt[m.start():txt.find(".Lfunc_end", m.start())] for m in re.finditer(r"^_ZN3qcx14k_fused_roundsILi512ELi11ELi7ELb[01]ELi0ELb0EE\w*:", txt, re.M)]
    assert len(bodies) == 2
    for b in bodies:
        assert len(re.findall(r"global_load_dword ", b)) == 0 and len(re.findall(r"flat_load", b)) == 0
        assert b.count("s_load_dword") > 20 and b.count("v_cmpx_eq_u32") >= 15 * 4
        assert not re.search(r"v_fma_f64|v_fmac_f64|v_pk_fma_f64", b)
        m = re.search(r"; ScratchSize: (\d+)", txt[txt.find(b[:60]) + len(b):])
        assert m and int(m.group(1)) <= 64          # at most a handful of spilled dwords


def test_walk_record_registers_are_inside_the_allocators_budget():
    """The item walk keeps its two gate records in the named blocks s[72:79] / s[80:87] (clobbers of the asm statement).  Every
    kernel that holds the walk is built for at most 7 waves per SIMD (round 5), where those registers lie inside the
    allocator's budget: the build must not warn about "reserved registers" on a clobber list, and the kernel descriptor must
    count them (SGPR count above 87)."""
    s_path = os.path.join(ROOT, "quantumcomputer_amd", "libqcx.gfx950.s")
    r = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "quantumcomputer_amd", "csrc"), "-s", "isa"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "reserved registers" not in r.stderr, r.stderr[-2000:]
    txt = open(s_path).read()
    funcs = {m.group(1): txt[m.start():txt.find(".Lfunc_end", m.start())] for m in re.finditer(r"^(_Z\w+):", txt, re.M)}
    sgprs = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s*(\S+)\s.*?\.sgpr_count:\s*(\d+)", txt, re.S)}
    seen = 0
    for name, body in funcs.items():
        if "s_load_dwordx8 s[72:79]" not in body:
            continue
        seen += 1
        assert sgprs[name] >= 88 + 6, (name, sgprs[name])
        assert not re.match(r"_ZN3qcx14k_fused_roundsILi\d+ELi\d+ELi8ELb[01]ELi[01]E", name), ("an 8-wave build of the walk", name)
    assert seen >= 10, seen


def test_exact_walk_on_8_amplitudes_declares_its_fixed_registers():
    """k_fused_x8 (round 5) runs a whole round as ONE asm statement on fixed registers -- v[18:63], s[72:98] on its clobber
    list.  They must lie INSIDE the kernel's declared budget (no "reserved register" games: the kernel is built for 4 waves
    per SIMD, where the allocator's own budget reaches past them), the kernel must not spill, and it must stay free of FMA."""
    s_path = os.path.join(ROOT, "quantumcomputer_amd", "libqcx.gfx950.s")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "quantumcomputer_amd", "csrc"), "-s", "isa"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not re.search(r"k_fused_x8[^\n]*\n[^\n]*reserved registers", r.stderr), "the walk's registers must be inside the allocator's budget"
    txt = open(s_path).read()
    meta = {m.group(1): (int(m.group(2)), int(m.group(3)), int(m.group(4)))
            for m in re.finditer(r"\.name:\s*(\S*k_fused_x8\S*)\s.*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S)}
    assert len(meta) == 7, sorted(meta)                       # 2^10, 2^11, 2^12 tiles, each read or generated; + the tolerance mode's (2^12)
    funcs = {m.group(1): txt[m.start():txt.find(".Lfunc_end", m.start())] for m in re.finditer(r"^(_Z\w+):", txt, re.M)}
    for name, (scratch, sgpr, vgpr) in meta.items():
        if "ELb0ELb1EEEv" in name:                           # k_fused_x8<.., TOL = true>: FMA allowed, another register plan; no spills
            assert scratch == 0 and vgpr <= 128, (name, scratch, vgpr)
            continue
        assert scratch == 0 and sgpr >= 99 and 64 <= vgpr <= 128, (name, scratch, sgpr, vgpr)
        body = funcs[name]
        assert not re.search(r"v_fma_f64|v_fmac_f64", body)
        assert body.count("ds_read_b128 v[32:35]") >= 1 and body.count("s_load_dwordx8 s[72:79], s[88:89]") >= 14
        assert "v_xor_b32" in body and ("global_load_lds_dwordx4" in body or "ELb1ELb0EEEv" in name)     # (GEN = true: the tiles are generated, not read)


def test_product_rng_matches_oracle_and_known_answers(qc, ob):
    r = qc.Rng(5489)
    v = [r.get() for _ in range(10000)]
    assert v[0] == 3499211612 and v[-1] == 4123659995
    for seed in (0, 1, 12345, 4357, 2 ** 32 - 1):
        a, b = qc.Rng(seed), ob.Rng(seed)
        assert [a.get() for _ in range(1300)] == [b.get() for _ in range(1300)]
        assert a.uniform() == b.uniform()


def test_product_int_pow_matches_reference_wrap(qc, ob):
    for b, p in [(2, 32), (2, 31), (7, 16), (7, 32), (2, 64), (3, 20), (7, 2), (2, 0), (10, 9), (10, 10)]:
        assert qc.lib().qcx_ref_int_pow(float(b), float(p)) == ob.ref_intpow(b, p)
    assert qc.lib().qcx_ref_int_pow(7.0, 16.0) == 2768600449


def test_polar_is_glibc_sincos_in_product_and_oracle(qc, ob):
    """the phase factor is gsl_complex_polar(1, theta) as gcc -O2 + glibc evaluate it: ONE sincos call.  glibc's
    sincos and sin() differ in the last bit for some arguments; product, oracle and libm sincos must agree there."""
    import math
    m = C.CDLL("libm.so.6")
    m.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    th = 0.20966817126512538
    sn, cs = C.c_double(), C.c_double()
    m.sincos(th, C.byref(sn), C.byref(cs))
    assert qc.polar(th) == ob.polar(th) == (cs.value, sn.value)
    assert sn.value != math.sin(th)                       # the witness: stand-alone sin() rounds the other way here
    import random
    rnd = random.Random(5)
    for _ in range(2000):
        t = rnd.uniform(-8, 8) if rnd.random() < 0.5 else math.pi / (1 << rnd.randrange(1, 40))
        assert qc.polar(t) == ob.polar(t)

"""The drop-in boundary as artefacts, checked mechanically:

  integration/hip_ffi.rs                          complete `extern "C"` block for every export of include/phmm.h
  integration/lorikeet-hip.patch                  AVXMode::Hip + the arm in PairHMM::compute_likelihoods + --pairhmm-backend
  integration/lorikeet-hip-shared-handle.patch    the phmm_submit / phmm_wait worker variant, on top of the first

No Rust toolchain exists in this image, so the A/B run of `lorikeet call` itself stays open; what CAN be checked is:
the patches apply to the reference tree, the Rust declarations agree with the C header name by name (arity, pointer
depth, constness, scalar width), and a plain C99 translation unit that calls every export compiles against the
header and links against the library."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phmm.h")
FFI = os.path.join(ROOT, "integration", "hip_ffi.rs")
PATCH = os.path.join(ROOT, "integration", "lorikeet-hip.patch")
PATCH2 = os.path.join(ROOT, "integration", "lorikeet-hip-shared-handle.patch")
REFERENCE = "/root/reference"

C_SCALARS = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64",
             "int32_t": "i32", "int64_t": "i64", "double": "f64", "size_t": "usize", "char": "c_char", "void": "void",
             "phmm_handle": "phmm_handle", "phmm_batch": "phmm_batch", "phmm_engine_config": "phmm_engine_config", "phmm_sw_parameters": "phmm_sw_parameters",
             "phmm_realign_config": "phmm_realign_config", "phmm_plan_info": "phmm_plan_info"}
RS_SCALARS = {"c_int": "i32", "c_uint": "u32", "u8": "u8", "u32": "u32", "u64": "u64", "i32": "i32", "i64": "i64", "f64": "f64",
              "usize": "usize", "c_char": "c_char", "c_void": "void", "phmm_handle": "phmm_handle",
              "phmm_batch": "phmm_batch", "phmm_engine_config": "phmm_engine_config",
              "phmm_sw_parameters": "phmm_sw_parameters", "phmm_realign_config": "phmm_realign_config",
              "phmm_plan_info": "phmm_plan_info"}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))


def _c_type(tok):
    """'const uint32_t *' -> ('u32', [const-ness per pointer level...]) as a canonical string."""
    tok = tok.strip()
    stars = tok.count("*")
    words = tok.replace("*", " * ").split()
    # constness of the pointee of the outermost pointer level(s): read left to right
    base_words, levels, cur_const = [], [], False
    for w in words:
        if w == "const":
            cur_const = True
        elif w == "*":
            levels.append("const" if cur_const else "mut")
            cur_const = False
        else:
            base_words.append(w)
    base = C_SCALARS[" ".join(base_words)]
    assert len(levels) == stars
    # levels[0] describes what the first '*' points to (the base type), levels[1] the next, ...; Rust writes it outside-in
    return base, tuple(reversed(levels))


def parse_header():
    text = _strip_c(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(phmm_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                params.append((mm.group(2), _c_type(mm.group(1))))
        protos[name] = (_c_type(ret), params)
    return protos


def _rs_type(tok):
    tok = tok.strip()
    levels = []
    while tok.startswith("*"):
        mm = re.match(r"\*(const|mut)\s+(.*)$", tok)
        levels.append(mm.group(1))
        tok = mm.group(2).strip()
    return RS_SCALARS[tok], tuple(levels)


def parse_rust(path=FFI):
    text = re.sub(r"//[^\n]*", " ", open(path).read())
    block = re.search(r'extern "C" \{(.*)\}', text, flags=re.S).group(1)
    protos = {}
    for m in re.finditer(r"pub fn (phmm_[a-z0-9_]+)\s*\(([^)]*)\)\s*(->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(4)
        params = []
        for a in [x for x in args.split(",") if x.strip()]:
            pn, pt = a.split(":", 1)
            params.append((pn.strip(), _rs_type(pt)))
        protos[name] = (_rs_type(ret) if ret else ("void", ()), params)
    return protos


def test_rust_declarations_mirror_the_header_one_to_one():
    c, rs = parse_header(), parse_rust()
    assert len(c) >= 28, sorted(c)
    assert sorted(c) == sorted(rs), (sorted(set(c) - set(rs)), sorted(set(rs) - set(c)))
    for name in c:
        (cret, cparams), (rret, rparams) = c[name], rs[name]
        assert cret == rret, (name, "return", cret, rret)
        assert len(cparams) == len(rparams), (name, "arity", len(cparams), len(rparams))
        for (cn, ct), (rn, rt) in zip(cparams, rparams):
            assert ct == rt, (name, cn, ct, rt)
            assert cn == rn or (cn, rn) == ("stream_v", "stream"), (name, cn, rn)


def test_rust_constants_and_struct_mirror_the_header():
    h, r = open(HEADER).read(), open(FFI).read()
    for name, val in re.findall(r"#define (PHMM_[A-Z0-9_]+) (\d+)u?\b", h):
        m = re.search(r"pub const %s: c_(?:int|uint) = (\d+);" % name, r)
        assert m and m.group(1) == val, name
    fields_c = re.findall(r"^\s+(uint8_t|double)\s+(\w+)(\[\d+\])?;", re.search(r"typedef struct phmm_engine_config \{(.*?)\}", h, re.S).group(1), re.M)
    fields_r = re.findall(r"pub (\w+): (\[u8; \d+\]|u8|f64),", re.search(r"pub struct phmm_engine_config \{(.*?)\}", r, re.S).group(1))
    assert [(n, {"uint8_t": "u8", "double": "f64"}[t] if not arr else "[u8; %s]" % arr[1:-1]) for t, n, arr in fields_c] == \
           [(n, t) for n, t in fields_r]
    # phmm_realign_config: same fields, same order, same widths
    fc = re.findall(r"^\s+(phmm_sw_parameters|int32_t|uint32_t|double)\s+(\w+);", re.search(r"typedef struct phmm_realign_config \{(.*?)\}", h, re.S).group(1), re.M)
    fr = re.findall(r"pub (\w+): (\w+),", re.search(r"pub struct phmm_realign_config \{(.*?)\}", r, re.S).group(1))
    assert [(n, {"phmm_sw_parameters": "phmm_sw_parameters", "int32_t": "i32", "uint32_t": "u32", "double": "f64"}[t]) for t, n in fc] == fr and len(fr) == 4


def test_every_export_is_callable_from_plain_c99():
    """A C translation unit that takes the address of and calls every export with arguments of the declared types,
    compiled -std=c99 -pedantic -Werror against include/phmm.h and linked against libphmm.so (not run: no device here)."""
    protos = parse_header()
    lib = os.path.join(ROOT, "lorikeet_amd", "libphmm.so")
    if not os.path.exists(lib):
        pytest.skip("libphmm.so not built")
    ctype = {("i32", ()): "int"}
    body = ['#include "phmm.h"', "#include <stddef.h>", "int main(int argc, char **argv) {", "  (void)argv;",
            "  if (argc < 1000) return 0; /* compile + link check only */"]
    text = _strip_c(open(HEADER).read())
    for name, (ret, params) in protos.items():
        m = re.search(r"\b%s\s*\(([^;{}]*?)\)\s*;" % name, text, flags=re.S)
        args = []
        if m.group(1).strip() != "void":
            for i, a in enumerate(m.group(1).split(",")):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                decl = mm.group(1).strip()
                var = "%s_a%d" % (name, i)
                body.append("  %s %s = (%s)0;" % (decl, var, decl))
                args.append(var)
        body.append("  (void)%s(%s);" % (name, ", ".join(args)) if ret[0] != "void" or ret[1] else "  %s(%s);" % (name, ", ".join(args)))
    body += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "caller.c")
        open(src, "w").write("\n".join(body) + "\n")
        r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                            src, "-o", os.path.join(d, "caller"), "-L", os.path.dirname(lib), "-lphmm",
                            "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("name", ["hip_ffi.rs", "hip_backend.rs"])
def test_the_patch_carries_the_rust_files_verbatim(name):
    want = open(os.path.join(ROOT, "integration", name)).read().splitlines()
    patch = open(PATCH).read()
    seg = patch.split("diff --git a/src/pair_hmm/%s b/src/pair_hmm/%s" % (name, name), 1)[1].split("\ndiff --git", 1)[0]
    got = [l[1:] for l in seg.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert got == want


def test_the_safe_wrapper_calls_what_the_ffi_declares():
    """Every phmm_* call in hip_backend.rs passes as many arguments as hip_ffi.rs declares (checked textually: a Rust
    toolchain is not available here), and its brackets balance."""
    ffi = open(FFI).read()
    backend = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "integration", "hip_backend.rs")).read())
    arity = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
             for m in re.finditer(r"pub fn (phmm_\w+)\(([^)]*)\)", ffi, flags=re.S)}
    calls = 0
    for m in re.finditer(r"\b(phmm_\w+)\(", backend):
        name = m.group(1)
        if name not in arity:
            continue
        depth, i, args, cur = 1, m.end(), [], ""
        while depth:
            ch = backend[i]
            depth += ch in "([{"
            depth -= ch in ")]}"
            if ch == "," and depth == 1:
                args.append(cur)
                cur = ""
            elif depth:
                cur += ch
            i += 1
        args.append(cur)
        n = len([a for a in args if a.strip()])
        assert n == arity[name], (name, n, arity[name])
        calls += 1
    assert calls >= 8
    for name in ("phmm_compute", "phmm_realign_reads", "phmm_region_submit", "phmm_wait", "phmm_calculate_cigar"):
        assert re.search(r"\b%s\(" % name, backend), name
    for o, c in ("()", "[]", "{}"):
        assert backend.count(o) == backend.count(c), (o, c)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree only exists in the build container")
def test_patches_apply_to_the_reference_tree():
    r = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=REFERENCE, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the files the reviews name (pair_hmm.rs:345-375, ...engine.rs:654-672, cli.rs:1928/2706/3485, assembly_based_caller_utils.rs:960-964)
    touched = set(re.findall(r"^diff --git a/(\S+)", open(PATCH).read(), flags=re.M))
    assert {"src/pair_hmm/pair_hmm.rs", "src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs", "src/cli.rs",
            "src/assembly/assembly_based_caller_utils.rs", "src/smith_waterman/smith_waterman_aligner.rs",
            "src/pair_hmm/hip_ffi.rs", "src/pair_hmm/hip_backend.rs", "src/pair_hmm/mod.rs", "build.rs", "Cargo.toml",
            # round 3: the call sites of the whole per-region path (compute_read_likelihoods ...engine.rs:195-242,
            # realign_reads_to_their_best_haplotype assembly_based_caller_utils.rs:208-246 with its caller
            # haplotype_caller_engine.rs:1345-1355, create_read_aligned_to_ref's tail, calculate_cigar cigar_utils.rs:358-457)
            "src/haplotype/haplotype_caller_engine.rs", "src/model/allele_likelihoods.rs", "src/reads/alignment_utils.rs",
            "src/reads/cigar_utils.rs"} <= touched
    assert open(PATCH).read().count('Arg::new("pairhmm-backend")') == 3   # one per subcommand that has --disable-avx
    # second patch: on a copy of the touched files with the first one applied
    with tempfile.TemporaryDirectory() as d:
        for f in touched:
            src = os.path.join(REFERENCE, f)
            if os.path.exists(src):
                os.makedirs(os.path.dirname(os.path.join(d, f)), exist_ok=True)
                shutil.copy(src, os.path.join(d, f))
        subprocess.run(["git", "init", "-q", "."], cwd=d, check=True)
        r = subprocess.run(["git", "apply", PATCH], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run(["git", "apply", "--check", PATCH2], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run(["git", "apply", PATCH2], cwd=d, capture_output=True, text=True)
        backend = open(os.path.join(d, "src/pair_hmm/hip_backend.rs")).read()
        assert "phmm_submit(" in backend and "phmm_wait(h, ticket)" in backend and "phmm_compute(" not in backend
        # the patches add balanced delimiters to every Rust file they produce or touch (the cheapest syntax check there
        # is without rustc): open-minus-close of each bracket kind is what it was before (0 for new files)
        def surplus(path):
            if not os.path.exists(path):
                return (0, 0, 0)
            text = re.sub(r"//[^\n]*", "", open(path).read())
            text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
            text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)
            return tuple(text.count(o) - text.count(c) for o, c in ("()", "[]", "{}"))
        for f in touched:
            if f.endswith(".rs"):
                assert surplus(os.path.join(d, f)) == surplus(os.path.join(REFERENCE, f)), f


def test_the_patch_binds_every_call_site_of_the_path():
    """What the hunks must contain (textual: no Rust toolchain here): the engine-level call and the realignment are routed
    to the device under AVXMode::Hip, the realignment picks up what the engine-level call left for it, and calculate_cigar
    has its device arm; every cfg(feature = "hip") statement guard is a block (attributes on `if` expressions do not compile)."""
    patch = open(PATCH).read()
    def hunk(path):
        return patch.split("diff --git a/%s b/%s" % (path, path), 1)[1].split("\ndiff --git", 1)[0]
    eng = hunk("src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs")
    assert "fn compute_read_likelihoods_hip" in eng and "hip_backend::region_compute(" in eng and "hip_backend::stash_realignment(" in eng
    assert "remove_poorly_modeled_evidence(s, removed)" in eng and "fn remove_poorly_modeled_evidence" in hunk("src/model/allele_likelihoods.rs")
    asm = hunk("src/assembly/assembly_based_caller_utils.rs")
    assert "fn realign_reads_to_their_best_haplotype_hip" in asm and "hip_backend::take_realignment(" in asm and "hip_backend::realign_reads(" in asm
    assert "AlignmentUtils::apply_realignment(" in asm and "pub fn apply_realignment" in hunk("src/reads/alignment_utils.rs")
    assert 'args.get_one::<String>("pairhmm-backend")' in hunk("src/haplotype/haplotype_caller_engine.rs")
    assert "hip_backend::calculate_cigars(" in hunk("src/reads/cigar_utils.rs")
    added = [l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++")]
    for i, l in enumerate(added[:-1]):
        if l.strip() == '#[cfg(feature = "hip")]':
            assert not added[i + 1].strip().startswith("if "), added[i + 1]


def test_the_f32_first_engine_is_reachable_from_the_command_line():
    """the reference's production arm (gkl, pair_hmm.rs:348-366) is f32-first; `--pairhmm-backend hip-f32` (or
    LORIKEET_HIP_F32_FIRST=1) creates every engine with PHMM_FLAG_F32_FIRST, `--pairhmm-backend hip` stays f64."""
    patch = open(PATCH).read()
    backend = open(os.path.join(ROOT, "integration", "hip_backend.rs")).read()
    header = open(os.path.join(ROOT, "include", "phmm.h")).read()
    flag = int(re.search(r"#define PHMM_FLAG_F32_FIRST (\d+)u", header).group(1))
    assert re.search(r"pub const PHMM_FLAG_F32_FIRST: c_uint = %d;" % flag, open(FFI).read())
    # no engine is created with a literal flag word any more: both creation sites ask engine_flags()
    assert backend.count("phmm_create(device, engine_flags())") == 2 and "phmm_create(device, 0)" not in backend
    assert "pub fn set_f32_first(on: bool)" in backend and 'std::env::var("LORIKEET_HIP_F32_FIRST")' in backend
    assert patch.count('.value_parser(["auto", "avx", "hip", "hip-f32", "scalar"])') == 3
    assert 'Some("hip-f32") => Self::hip_or_panic(true)' in patch and 'Some("hip") => Self::hip_or_panic(false)' in patch
    assert "crate::pair_hmm::hip_backend::set_f32_first(f32_first);" in patch

"""VERDICT r4 item 7 (rows a21 / f3): the Rust side of the drop-in -- integration/hip_backend.rs, hip_ffi.rs and the hunks of
integration/lorikeet-hip.patch -- has never met rustc (no toolchain in the image).  Short of compiling it, every name it takes
from the reference tree is resolved here against that tree with a small tokenizer (tests/rust_index.py):

* every call `name(...)`, `.name(...)`, `Type::name(...)` whose name the reference defines is made with a number of arguments
  one of the reference's definitions takes (methods: without `self`); where the path names a type the reference implements
  (`AlignmentUtils::`, `CigarUtils::`, `AlleleLikelihoods::` ...) the function must be in THAT type's impl blocks (or in the
  hunks the patch adds to them);
* every other called name is one of ours (defined in the added code, a local closure / binding), or belongs to the short,
  reviewed lists below (std, rust-htslib's bam::Record and Cigar, clap, rayon) -- an unknown name fails;
* field accesses resolve to a field of a reference struct or of ours.
Runs in the build container only (needs /root/reference)."""
import glob
import os
import re

import pytest

import rust_index as R
from conftest import ROOT

REFERENCE = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "lorikeet-hip.patch")
pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree only exists in the build container")

STD = set("""and_then any as_mut_ptr as_ptr as_ref as_slice as_str borrow_mut by_ref cloned collect entry enumerate expect
extend_from_slice fetch_and fetch_or flat_map flatten for_each from_ptr get_mut insert into_iter into_owned is_none iter iter_mut keys
load map max null null_mut ok or_insert parse pop position push sum take to_string_lossy to_vec unwrap unwrap_or unwrap_or_else var
var_os with_capacity with filter get clone len is_empty is_some is_null next values new set min store lock contains push_str
to_string to_owned from try_from into zip rev chain copied all find sort_by last first retain remove clear resize fill swap
FnOnce FnMut Fn concat join starts_with ends_with contains_key or_insert_with drain truncate split_at chunks windows abs powi sqrt floor ceil""".split())
HTSLIB = set("cigar mapq qname qual seq set_pos push_aux pos set U32 CigarString Equal Diff HardClip Ins Pad RefSkip Match Del SoftClip".split())
CLAP = set("arg get_flag get_one try_get_one value_parser default_value help long option".split())
RAYON = set("current_thread_index num_threads".split())
EXTERNAL = STD | HTSLIB | CLAP | RAYON


def _reference_index():
    fns, items, fields, variants, impls = {}, set(), set(), set(), {}
    for f in glob.glob(os.path.join(REFERENCE, "src", "**", "*.rs"), recursive=True):
        t = R.strip(open(f).read())
        d = R.definitions(t)
        for k, v in d[0].items():
            fns.setdefault(k, set()).update(v)
        items |= d[1]
        fields |= d[2]
        variants |= d[3]
        for m in re.finditer(r"\bimpl\b\s*(?:<[^{]*?>\s*)?(?:[A-Za-z_][\w:]*(?:<[^{]*?>)?\s+for\s+)?([A-Za-z_]\w*)[^{;]*\{", t):
            end = R._match(t, m.end() - 1, "{", "}")
            for k, v in R.definitions(t[m.end():end - 1])[0].items():
                impls.setdefault(m.group(1), {}).setdefault(k, set()).update(v)
    return fns, items, fields, variants, impls


def _our_code():
    """{file: added text} of the patch (the two new files are in it verbatim: test_integration_artifacts.py checks that)"""
    added, cur = {}, None
    for ln in open(PATCH).read().splitlines():
        if ln.startswith("diff --git"):
            cur = ln.split(" b/")[1]
        elif ln.startswith("+") and not ln.startswith("+++") and cur and cur.endswith(".rs"):
            added.setdefault(cur, []).append(ln[1:])
    return {f: R.strip("\n".join(lines)) for f, lines in added.items()}


def test_the_tokenizer_on_rust_it_must_get_right():
    t = R.strip('fn f<\'a>(x: &\'a str, c: char) -> u8 { let s = "a(b\\"c"; let q = \'(\'; /* ( */ g(x, h(1, 2)) // (\n}')
    assert t.count("(") == t.count(")") == 3
    fns, _, _, _ = R.definitions(R.strip("impl X { pub fn a(&self, b: Vec<(u8, u8)>, c: HashMap<u8, Vec<u8>>) {} fn b<T: Into<u8>>(mut self) {} fn c() {} }"))
    assert fns == {"a": {2}, "b": {0}, "c": {0}}
    calls, fields, paths = R.uses(R.strip("let y = a.b(c, d(e, f)).g; X::h::<u8>(1, (2, 3)); k!(1); if (a) {}"))
    assert ("method", "b", 2) in calls and ("free", "d", 2) in calls and ("path", "h", 2) in calls and "g" in fields and ("X", "h") in paths
    assert not any(n == "k" or n == "if" for _, n, _ in calls)


def test_every_name_the_rust_side_takes_from_the_reference_exists_there_with_that_arity():
    ref_fns, ref_items, ref_fields, ref_variants, ref_impls = _reference_index()
    assert len(ref_fns) > 1000 and "compute_read_likelihoods" in ref_fns and "AlleleLikelihoods" in ref_impls
    ours = _our_code()
    assert {"src/pair_hmm/hip_backend.rs", "src/pair_hmm/hip_ffi.rs", "src/bin/lorikeet.rs"} <= set(ours)
    text = "\n".join(ours.values())
    text = re.sub(r"#!?\[[^\]]*\]", " ", text)                      # attributes: cfg(...), derive(...), repr(C), allow(...)
    our_fns, our_items, our_fields, our_variants = R.definitions(text)
    # what the patch ADDS to the reference's own types (new methods in existing impl blocks) counts for those types
    locals_ = set(re.findall(r"\blet\s+(?:mut\s+)?([a-z_]\w*)", text)) | set(re.findall(r"[(,|]\s*(?:mut\s+)?([a-z_]\w*)\s*:", text)) | \
        set(re.findall(r"\|\s*(?:mut\s+)?([a-z_]\w*)\s*[|,]", text))
    calls, field_uses, paths = R.uses(text)
    assert len(calls) > 500
    wrong, unknown, checked = [], set(), 0
    for kind, name, n in calls:
        if name in our_fns:
            if n not in our_fns[name] and not (kind == "path" and n - 1 in our_fns[name]) and not (name in ref_fns and n in ref_fns[name]):
                wrong.append((kind, name, n, "ours take", sorted(our_fns[name])))
        elif name in ref_fns and name not in EXTERNAL:
            checked += 1
            if n not in ref_fns[name] and not (kind == "path" and n - 1 in ref_fns[name]):
                wrong.append((kind, name, n, "the reference takes", sorted(ref_fns[name])))
        elif name in EXTERNAL or name in locals_ or name in ref_items or name in our_items or name in ref_variants or name in our_variants:
            continue
        else:
            unknown.add(name)
    assert not wrong, wrong
    assert not unknown, sorted(unknown)
    assert checked >= 20, checked          # (calls whose name only the reference defines: get_bases, get_allele, set_transient_attribute ...)
    # Type::function paths into the reference's own types: the function is in that type's impl blocks (or added to them here)
    missing = []
    for ty, name in sorted(paths):
        if ty in ref_impls and ty not in ("Self",) and name[0].islower():
            if name not in ref_impls[ty] and name not in our_fns:
                missing.append("%s::%s" % (ty, name))
    assert not missing, missing
    # fields: of a reference struct, of ours, or a method used as a value
    stray = sorted(f for f in field_uses if f not in ref_fields and f not in our_fields and f not in ref_fns and f not in our_fns
                   and f not in EXTERNAL and f not in locals_ and not f.isdigit() and f not in ("await",))
    assert not stray, stray


def test_the_reference_call_sites_the_patch_touches_are_where_the_survey_says():
    """rows a21 / f3: the hunks sit on the functions SURVEY 8a names (the context lines are the reference's own)."""
    patch = open(PATCH).read()
    def hunk(path):
        return patch.split("diff --git a/%s b/%s" % (path, path), 1)[1].split("\ndiff --git", 1)[0]
    assert "impl<'a> PairHMM<'a>" in hunk("src/pair_hmm/pair_hmm.rs")
    assert "impl AVXMode" in hunk("src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs")
    assert "impl HaplotypeCallerEngine" in hunk("src/haplotype/haplotype_caller_engine.rs")
    assert "fn prepare_pileup" in hunk("src/bin/lorikeet.rs") and "hip_pool_threads(m, threads)" in hunk("src/bin/lorikeet.rs")

"""CPU: the Rust shim crate (rust/whisper-hip, uncompiled here: no Rust toolchain in the build environment)
declares exactly the C ABI: every function of its `extern "C"` block exists in include/whisper_hip.h with the
same number of arguments AND, argument by argument, the Rust type the C type maps to (a `c_int` / `i64` swap or a
dropped `const` is an ABI break that arity cannot see), the same for the return type; the #[repr(C)] structs list the
header's fields in the header's order with matching field types, and the
crate exposes the reference's public surface (lib.rs:1-6, transcribe.rs:23-29, mod.rs:47-71, audio.rs:34)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "whisper-hip")


def _strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def _split_args(a):
    a = a.strip()
    if a in ("", "void"):
        return []
    return [x for x in a.split(",") if x.strip()]


def header_functions():
    h = _strip_comments(open(os.path.join(ROOT, "include", "whisper_hip.h")).read())
    out = {}
    for m in re.finditer(r"\b(wb_\w+)\s*\(([^;{()]*?)\)\s*;", h, flags=re.S):
        out[m.group(1)] = len(_split_args(m.group(2)))
    return out, h


def rust_functions():
    r = _strip_comments(open(os.path.join(CRATE, "src", "ffi.rs")).read())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', r, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(wb_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        out[m.group(1)] = len(_split_args(m.group(2)))
    return out, r


def struct_fields_c(h, name):
    body = re.search(r"typedef\s+struct(?:\s+\w+)?\s*\{([^}]*)\}\s*%s\s*;" % name, h, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            first, *rest = [x.strip() for x in decl.split(",")]
            fields.append(first.split()[-1])
            fields += rest
    return fields


def struct_fields_rust(r, name):
    body = re.search(r"pub\s+struct\s+%s\s*\{(.*?)\}" % name, r, flags=re.S).group(1)
    return re.findall(r"pub\s+(\w+)\s*:", body)


# ---- C type -> Rust type, per argument -------------------------------------------------------------------------
C_SCALARS = {"int": "c_int", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32",
             "uint64_t": "u64", "float": "c_float", "double": "c_double", "char": "c_char", "void": "c_void",
             "size_t": "usize", "wb_model": "wb_model", "wb_session": "wb_session", "wb_dims": "wb_dims",
             "wb_decode_params": "wb_decode_params", "wb_kernel_stat": "wb_kernel_stat", "wb_comm": "wb_comm",
             "wb_allgather_fn": "wb_allgather_fn"}
RUST_EQUIV = {"c_int": {"c_int", "i32"}, "i32": {"i32", "c_int"}, "c_float": {"c_float", "f32"}, "c_double": {"c_double", "f64"}}


def c_type_to_rust(decl, is_return=False):
    """'const float* mel' / 'wb_model** out' / 'int64_t n' -> '*const c_float' / '*mut *mut wb_model' / 'i64'."""
    d = decl.strip()
    if not is_return:
        d = re.sub(r"\s*\b[A-Za-z_]\w*\s*(\[\s*\])?$", lambda m: "*" if m.group(1) else "", d)     # drop the parameter name
    d = d.replace("struct ", "")
    n_ptr = d.count("*")
    base = d.replace("*", " ").split()
    is_const = "const" in base
    base = [b for b in base if b != "const"]
    assert len(base) == 1, decl
    rust = C_SCALARS[base[0]]
    for level in range(n_ptr):
        # only the pointee of the innermost pointer carries the C const in this header ("const T*", "const char* const*")
        rust = ("*const " if (is_const and level == 0) else "*mut ") + rust
    return rust


def normalise_rust(t):
    return re.sub(r"\s+", " ", t.strip())


def same_type(c_as_rust, rust):
    if c_as_rust == rust:
        return True
    # pointers: compare level by level with the scalar equivalences (c_int == i32 on every target the crate supports)
    ca, ra = c_as_rust.split(" "), rust.split(" ")
    if len(ca) != len(ra) or ca[:-1] != ra[:-1]:
        return False
    return ra[-1] in RUST_EQUIV.get(ca[-1], {ca[-1]})


def header_signatures():
    h = _strip_comments(open(os.path.join(ROOT, "include", "whisper_hip.h")).read())
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(wb_\w+)\s*\(([^;{()]*?)\)\s*;", h, flags=re.S):
        ret = m.group(1).strip()
        if not ret or ret.startswith("typedef") or ret.endswith(","):
            continue
        out[m.group(2)] = (ret, _split_args(m.group(3)))
    return out


def rust_signatures():
    r = _strip_comments(open(os.path.join(CRATE, "src", "ffi.rs")).read())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', r, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(wb_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = [normalise_rust(a.split(":", 1)[1]) for a in _split_args(m.group(2))]
        out[m.group(1)] = (normalise_rust(m.group(3)) if m.group(3) else "()", args)
    return out


def test_extern_block_types_match_the_header_argument_by_argument():
    hs, rs = header_signatures(), rust_signatures()
    checked = 0
    for name, (rret, rargs) in rs.items():
        cret, cargs = hs[name]
        want_ret = "()" if cret.strip() == "void" else c_type_to_rust(cret, is_return=True)
        assert same_type(want_ret, rret), f"{name}: returns {cret!r} in the header ({want_ret}), {rret} in ffi.rs"
        assert len(cargs) == len(rargs), name
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            want = c_type_to_rust(ca)
            assert same_type(want, ra), f"{name} argument {i}: header {ca.strip()!r} = {want}, ffi.rs has {ra}"
            checked += 1
    assert checked >= 100, checked
    # the check has teeth: an i64 / c_int swap or a lost const is caught
    assert not same_type(c_type_to_rust("int64_t n"), "c_int") and not same_type(c_type_to_rust("const float* p"), "*mut c_float")
    assert same_type(c_type_to_rust("const char* const* names"), "*const *const c_char") or \
        same_type(c_type_to_rust("const char* const* names"), "*mut *const c_char")


def test_repr_c_struct_field_types_match_the_header():
    h = _strip_comments(open(os.path.join(ROOT, "include", "whisper_hip.h")).read())
    r = _strip_comments(open(os.path.join(CRATE, "src", "ffi.rs")).read())
    for name in ("wb_dims", "wb_decode_params"):
        body = re.search(r"typedef\s+struct(?:\s+\w+)?\s*\{([^}]*)\}\s*%s\s*;" % name, h, flags=re.S).group(1)
        c_fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                ty = " ".join(decl.split(",")[0].split()[:-1])
                for f in [decl.split(",")[0].split()[-1]] + [x.strip() for x in decl.split(",")[1:]]:
                    c_fields.append((f, C_SCALARS[ty]))
        rbody = re.search(r"pub\s+struct\s+%s\s*\{(.*?)\}" % name, r, flags=re.S).group(1)
        r_fields = re.findall(r"pub\s+(\w+)\s*:\s*([\w:]+)", rbody)
        assert [f for f, _ in c_fields] == [f for f, _ in r_fields], name
        for (f, ct), (_, rt) in zip(c_fields, r_fields):
            assert rt in RUST_EQUIV.get(ct, {ct}), f"{name}.{f}: header {ct}, ffi.rs {rt}"


def test_extern_block_matches_the_header():
    hf, h = header_functions()
    rf, r = rust_functions()
    assert len(rf) >= 20
    for name, nargs in rf.items():
        assert name in hf, f"{name} is not declared in include/whisper_hip.h"
        assert hf[name] == nargs, f"{name}: header takes {hf[name]} arguments, ffi.rs declares {nargs}"
    for must in ("wb_prep_audio", "wb_forward_encoder", "wb_forward_decoder", "wb_forward", "wb_waveform_to_tokens",
                 "wb_model_load_dump_dir", "wb_model_load_burn_record", "wb_model_dims", "wb_model_free",
                 "wb_session_begin", "wb_session_step", "wb_session_decode", "wb_last_error"):
        assert must in rf, must
    for s in ("wb_dims", "wb_decode_params"):
        assert struct_fields_rust(r, s) == struct_fields_c(h, s), s


def test_crate_exposes_the_reference_surface():
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    for sig in (r"pub fn prep_audio\(", r"pub fn max_waveform_samples\(", r"pub fn forward_encoder\(",
                r"pub fn forward_decoder\(", r"pub fn forward\(", r"pub fn encoder_ctx_size\(",
                r"pub fn decoder_ctx_size\(", r"pub fn waveform_to_text<"):
        assert re.search(sig, lib), sig
    # transcribe.rs:23-29: (whisper, bpe, lang, waveform: Vec<f32>, sample_rate: usize) -> Result<(String, Vec<usize>)>
    m = re.search(r"pub fn waveform_to_text<[^>]*>\((.*?)\)\s*->\s*Result<\(String, Vec<usize>\)>", lib, flags=re.S)
    assert m, "waveform_to_text must keep the reference's result type"
    args = [a.split(":")[0].strip() for a in m.group(1).split(",")]
    assert args == ["whisper", "bpe", "lang", "waveform", "sample_rate"], args
    toml = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'links = "whisper_hip"' in toml and os.path.exists(os.path.join(CRATE, "build.rs"))


def test_integration_md_excerpt_is_the_crate_s_ffi_rs():
    """INTEGRATION.md section 2 shows the binding a maintainer would add: its Rust excerpt must be lines of the real
    src/ffi.rs (round 2's excerpt still showed an earlier draft's type names)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```rust\n(.*?)```", md[md.index("## 2. The Rust binding"):], flags=re.S).group(1)
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    lines = [l.rstrip() for l in block.splitlines() if l.strip() and "see src/ffi.rs" not in l]
    assert len(lines) >= 40
    missing = [l for l in lines if l not in ffi and l.strip() != "}"]
    assert not missing, missing[:5]
    assert "WbModel" not in md and "src/hip.rs" not in md

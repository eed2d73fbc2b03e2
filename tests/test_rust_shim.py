"""CPU: the Rust shim crate (rust/whisper-hip, uncompiled here: no Rust toolchain in the build environment)
declares exactly the C ABI: every function of its `extern "C"` block exists in include/whisper_hip.h with the
same number of arguments, the #[repr(C)] structs list the header's fields in the header's order, and the
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

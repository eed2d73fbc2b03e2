//! whisper-hip: the reference's public surface on the MI355X engine.
//!
//! Mirrors Gadersd/whisper-burn so that `src/transcribe.rs` can switch over without touching its callers:
//!
//! | whisper-burn                                            | here                                  |
//! |---------------------------------------------------------|---------------------------------------|
//! | `audio::prep_audio(waveform, sample_rate)` audio.rs:34  | [`prep_audio`]                        |
//! | `audio::max_waveform_samples` audio.rs:12-17            | [`max_waveform_samples`]              |
//! | `Whisper::forward_encoder` mod.rs:52                    | [`Whisper::forward_encoder`]          |
//! | `Whisper::forward_decoder` mod.rs:56                    | [`Whisper::forward_decoder`]          |
//! | `Whisper::forward` mod.rs:48                            | [`Whisper::forward`]                  |
//! | `Whisper::{encoder,decoder}_ctx_size` mod.rs:64-70      | same names                            |
//! | `transcribe::waveform_to_text` transcribe.rs:23-29      | [`waveform_to_text`] (same signature) |
//!
//! Tensors cross as [`Tensor`] (shape + row-major `Vec<f32>`): there is no Burn backend underneath.
//! UNCOMPILED in this repository's build environment (no Rust toolchain there).
pub mod ffi;

use std::error::Error;
use std::ffi::{CStr, CString};
use std::fmt;
use std::os::raw::c_int;

pub type Result<T> = std::result::Result<T, Box<dyn Error + Send + Sync + 'static>>; // token.rs:6

/// Error carrying the `wb_status` and the engine's message (the reference panics or returns Box<dyn Error>).
#[derive(Debug)]
pub struct HipError {
    pub status: i32,
    pub message: String,
}
impl fmt::Display for HipError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "whisper_hip status {}: {}", self.status, self.message)
    }
}
impl Error for HipError {}

pub fn last_error() -> String {
    unsafe {
        let p = ffi::wb_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

fn check(status: c_int) -> Result<()> {
    if status == ffi::WB_OK { Ok(()) } else { Err(Box::new(HipError { status, message: last_error() })) }
}

/// Row-major f32 tensor (what a `Tensor<B, D>` of the reference holds).
#[derive(Clone, Debug, PartialEq)]
pub struct Tensor {
    pub shape: Vec<usize>,
    pub data: Vec<f32>,
}
impl Tensor {
    pub fn zeros(shape: &[usize]) -> Self {
        Tensor { shape: shape.to_vec(), data: vec![0.0; shape.iter().product()] }
    }
    pub fn dims(&self) -> &[usize] { &self.shape }
}

/// audio.rs:12-17.
pub fn max_waveform_samples(n_frame_max: usize) -> usize {
    unsafe { ffi::wb_max_waveform_samples(n_frame_max as i64) as usize }
}

/// audio.rs:34-56: `[1, N]` waveform -> `[1, 80, N / 160]` log-mel on `device`.  N < 400 is the reference's
/// assert (audio.rs:292) and comes back as `WB_ERR_SHAPE`.
pub fn prep_audio(waveform: &Tensor, sample_rate: f64, device: i32) -> Result<Tensor> {
    let n = *waveform.shape.last().unwrap_or(&0);
    let mut mel = Tensor::zeros(&[1, 80, n / 160]);
    let mut n_frames: i64 = 0;
    check(unsafe { ffi::wb_prep_audio(device, waveform.data.as_ptr(), n as i64, sample_rate, mel.data.as_mut_ptr(), &mut n_frames) })?;
    debug_assert_eq!(n_frames as usize, n / 160);
    Ok(mel)
}

/// mod.rs:41-71 `Whisper<B>`: the model handle (weights resident in HBM, immutable after load).
pub struct Whisper {
    raw: *mut ffi::wb_model,
    dims: ffi::wb_dims,
    device: i32,
}
// whisper_hip.h, "Conventions": the model is immutable after load and may be shared by threads; calls from different
// threads return the single-threaded results and take turns on the GPU (one process-wide turn inside the library).
unsafe impl Send for Whisper {}
unsafe impl Sync for Whisper {}

impl Whisper {
    /// `load_whisper` (load.rs:295-310): a dump directory written by python/dump.py.
    pub fn load_dump_dir(dir: &str, device: i32) -> Result<Self> {
        let c = CString::new(dir)?;
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::wb_model_load_dump_dir(c.as_ptr(), device, ffi::WB_F32, &mut raw) })?;
        Self::from_raw(raw, device)
    }
    /// `load_whisper_model_file` (bin/transcribe/main.rs:63-70): `<name>.mpk.gz` + `<name>.cfg`.
    pub fn load_record(mpk_gz: &str, cfg: &str, device: i32) -> Result<Self> {
        let (a, b) = (CString::new(mpk_gz)?, CString::new(cfg)?);
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::wb_model_load_burn_record(a.as_ptr(), b.as_ptr(), device, ffi::WB_F32, &mut raw) })?;
        Self::from_raw(raw, device)
    }
    fn from_raw(raw: *mut ffi::wb_model, device: i32) -> Result<Self> {
        let mut dims = ffi::wb_dims::default();
        check(unsafe { ffi::wb_model_dims(raw, &mut dims) })?;
        Ok(Whisper { raw, dims, device })
    }
    pub fn dims(&self) -> ffi::wb_dims { self.dims }
    pub fn device(&self) -> i32 { self.device }
    /// mod.rs:64-66.
    pub fn encoder_ctx_size(&self) -> usize { self.dims.n_audio_ctx as usize }
    /// mod.rs:68-70.
    pub fn decoder_ctx_size(&self) -> usize { self.dims.n_text_ctx as usize }
    /// Opt-in, not reference behaviour: bound the encoder POSITIONS (not the mel frames, mod.rs:236-241) by
    /// `n_audio_ctx`, i.e. Whisper's own 30 s window of 3000 frames; `false` restores the reference's 14.9 s windows.
    pub fn set_frame_limit(&mut self, whisper_geometry: bool) -> Result<()> {
        check(unsafe { ffi::wb_model_set_frame_limit(self.raw, whisper_geometry as c_int) })
    }

    /// Arithmetic of the encoder-side Linear layers: 0 = exact-f32 MFMA, 1 = split precision (three fp16 MFMAs per
    /// product, f32-grade; the default).
    pub fn encoder_gemm(&self) -> i32 { unsafe { ffi::wb_model_encoder_gemm(self.raw) as i32 } }
    /// 1 = split-precision fp16 MFMA decoder GEMMs in batch mode (default), 0 = exact f32 (also after a range-guard trip).
    pub fn decoder_gemm(&self) -> i32 { unsafe { ffi::wb_model_decoder_gemm(self.raw) as i32 } }

    /// mod.rs:52-54: `[B, 80, T]` -> `[B, C, d]`, `C = (T - 1) / 2 + 1`; T > n_audio_ctx is the reference's panic.
    pub fn forward_encoder(&self, mel: &Tensor) -> Result<Tensor> {
        let (b, t) = (mel.shape[0], mel.shape[2]);
        let mut out = Tensor::zeros(&[b, (t.max(1) - 1) / 2 + 1, self.dims.n_audio_state as usize]);
        check(unsafe { ffi::wb_forward_encoder(self.raw, mel.data.as_ptr(), b as c_int, t as c_int, out.data.as_mut_ptr()) })?;
        Ok(out)
    }
    /// mod.rs:56-62: tokens `[n, L]`, encoder_output `[n, C, d]` -> logits `[n, L, V]` (stateless).
    pub fn forward_decoder(&self, tokens: &[i32], n: usize, l: usize, encoder_output: &Tensor) -> Result<Tensor> {
        let c = encoder_output.shape[1];
        let mut logits = Tensor::zeros(&[n, l, self.dims.n_vocab as usize]);
        check(unsafe {
            ffi::wb_forward_decoder(self.raw, tokens.as_ptr(), n as c_int, l as c_int, encoder_output.data.as_ptr(),
                                    c as c_int, logits.data.as_mut_ptr())
        })?;
        Ok(logits)
    }
    /// mod.rs:48-50.
    pub fn forward(&self, mel: &Tensor, tokens: &[i32], l: usize) -> Result<Tensor> {
        let (b, t) = (mel.shape[0], mel.shape[2]);
        let mut logits = Tensor::zeros(&[b, l, self.dims.n_vocab as usize]);
        check(unsafe { ffi::wb_forward(self.raw, mel.data.as_ptr(), b as c_int, t as c_int, tokens.as_ptr(), l as c_int, logits.data.as_mut_ptr()) })?;
        Ok(logits)
    }
}
impl Drop for Whisper {
    fn drop(&mut self) { unsafe { ffi::wb_model_free(self.raw) } }
}

/// What `waveform_to_text` needs from the tokenizer: exactly the calls transcribe.rs makes on `Gpt2Tokenizer`
/// (token.rs:26-48).  whisper-burn's own `Gpt2Tokenizer` implements it verbatim.
pub trait Tokenizer {
    type Lang: Copy;
    fn special_token_id(&self, name: SpecialToken<Self::Lang>) -> Option<usize>;   // token.rs:26-30
    fn decode(&self, tokens: &[usize], skip_special: bool) -> Result<String>;      // token.rs:32-35
    fn is_special(&self, token: usize) -> bool;                                    // token.rs:37-43
    fn vocab_size(&self) -> usize;                                                 // token.rs:45-47
}
/// token.rs:267-295, the five names transcribe.rs:179-185 looks up.
#[derive(Clone, Copy)]
pub enum SpecialToken<L> {
    EndofText,
    StartofTranscript,
    Transcribe,
    NoTimeStamps,
    Language(L),
}

/// transcribe.rs:23-29, same argument order and result: `(text, tokens)`; tokens include each window's prompt and
/// EOT exactly as the reference's stitch leaves them (transcribe.rs:56-63).
///
/// One error is worth a retry: `WB_ERR_STATE` whose message says a decoder activation "left fp16's range" (the
/// split-precision decoder GEMM's range guard, include/whisper_hip.h at `wb_session_step`).  The library has by then
/// switched this model to its exact-f32 decoder kernels, so calling again succeeds.
pub fn waveform_to_text<T: Tokenizer>(whisper: &Whisper, bpe: &T, lang: T::Lang, waveform: Vec<f32>,
                                      sample_rate: usize) -> Result<(String, Vec<usize>)> {
    let mut p = ffi::wb_decode_params::default();
    unsafe { ffi::wb_decode_params_default(&mut p) }; // beam 5 x depth 100, padding 10, 3 s overlap, (40, 3) stitch
    let id = |t: SpecialToken<T::Lang>| -> Result<i32> {
        bpe.special_token_id(t).map(|v| v as i32).ok_or_else(|| "special token missing from the tokenizer".into())
    };
    p.tok_start_of_transcript = id(SpecialToken::StartofTranscript)?;
    p.tok_language = id(SpecialToken::Language(lang))?;
    p.tok_transcribe = id(SpecialToken::Transcribe)?;
    p.tok_no_timestamps = id(SpecialToken::NoTimeStamps)?;
    p.tok_end_of_text = id(SpecialToken::EndofText)?;
    // transcribe.rs:243-251 builds this mask per window (51 864 decodes each time); it is hoisted here.  The reference adds
    // a [vocab_size] mask to [.., n_vocab] logits and panics on a tokenizer / checkpoint mismatch: same contract
    if bpe.vocab_size() != whisper.dims.n_vocab as usize {
        return Err(format!("tokenizer vocabulary ({}) and model vocabulary ({}) differ", bpe.vocab_size(),
                           whisper.dims.n_vocab).into());
    }
    let is_special: Vec<u8> = (0..whisper.dims.n_vocab as usize).map(|t| bpe.is_special(t) as u8).collect();
    let wlen = max_waveform_samples(whisper.encoder_ctx_size() - p.padding as usize) as i64; // transcribe.rs:32-34
    let n_win = unsafe {
        ffi::wb_window_extents(waveform.len() as i64, sample_rate as c_int, wlen, p.overlap_seconds,
                               std::ptr::null_mut(), std::ptr::null_mut(), 0)
    } as usize;
    let stride = (4 + p.max_depth + 4) as usize;
    let mut rows = vec![0i32; n_win.max(1) * stride];
    let mut lens = vec![0i32; n_win.max(1)];
    let mut out = vec![0i32; n_win.max(1) * stride];
    let mut n_out: i64 = 0;
    check(unsafe {
        ffi::wb_waveform_to_tokens(whisper.raw, waveform.as_ptr(), waveform.len() as i64, sample_rate as c_int, &p,
                                   is_special.as_ptr(), 0, -1, rows.as_mut_ptr(), stride as i32, lens.as_mut_ptr(),
                                   out.as_mut_ptr(), out.len() as i64, &mut n_out)
    })?;
    let tokens: Vec<usize> = out[..n_out as usize].iter().map(|&t| t as usize).collect();
    let text = bpe.decode(&tokens[..], true)?; // transcribe.rs:67
    Ok((text, tokens))
}


/// One rank of the multi-GPU path (not in the reference, which is single-device): the reference's windows are
/// independent (transcribe.rs:195-201), so rank `rank` of `world` decodes its contiguous block of them, the ranks
/// exchange ONE buffer of token rows (RCCL over xGMI through `comm`) and every rank stitches all rows
/// (transcribe.rs:56-63) -- the result equals `waveform_to_text` on one GPU.
pub struct Comm { raw: *mut ffi::wb_comm }
impl Comm {
    /// Rank 0 makes the id and ships its 128 bytes to the other ranks (a file, a socket, MPI ...).
    pub fn unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::wb_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn new(id: &[u8; 128], rank: i32, world: i32, device: i32) -> Result<Comm> {
        let mut raw: *mut ffi::wb_comm = std::ptr::null_mut();
        check(unsafe { ffi::wb_comm_init(id.as_ptr(), rank as c_int, world as c_int, device as c_int, &mut raw) })?;
        Ok(Comm { raw })
    }
}
impl Drop for Comm {
    fn drop(&mut self) { unsafe { ffi::wb_comm_free(self.raw) } }
}

pub fn waveform_to_text_sharded<T: Tokenizer>(whisper: &Whisper, bpe: &T, lang: T::Lang, waveform: Vec<f32>,
                                              sample_rate: usize, rank: i32, world: i32, comm: &Comm)
                                              -> Result<(String, Vec<usize>)> {
    let mut p = ffi::wb_decode_params::default();
    unsafe { ffi::wb_decode_params_default(&mut p) };
    let id = |t: SpecialToken<T::Lang>| -> Result<i32> {
        bpe.special_token_id(t).map(|v| v as i32).ok_or_else(|| "special token missing from the tokenizer".into())
    };
    p.tok_start_of_transcript = id(SpecialToken::StartofTranscript)?;
    p.tok_language = id(SpecialToken::Language(lang))?;
    p.tok_transcribe = id(SpecialToken::Transcribe)?;
    p.tok_no_timestamps = id(SpecialToken::NoTimeStamps)?;
    p.tok_end_of_text = id(SpecialToken::EndofText)?;
    if bpe.vocab_size() != whisper.dims.n_vocab as usize {
        return Err(format!("tokenizer vocabulary ({}) and model vocabulary ({}) differ", bpe.vocab_size(),
                           whisper.dims.n_vocab).into());
    }
    let is_special: Vec<u8> = (0..whisper.dims.n_vocab as usize).map(|t| bpe.is_special(t) as u8).collect();
    let wlen = max_waveform_samples(whisper.encoder_ctx_size() - p.padding as usize) as i64;
    let n_win = unsafe {
        ffi::wb_window_extents(waveform.len() as i64, sample_rate as c_int, wlen, p.overlap_seconds,
                               std::ptr::null_mut(), std::ptr::null_mut(), 0)
    } as usize;
    let stride = (4 + p.max_depth + 4) as usize;
    let mut rows = vec![0i32; n_win.max(1) * stride];
    let mut lens = vec![0i32; n_win.max(1)];
    let mut out = vec![0i32; n_win.max(1) * stride];
    let mut n_out: i64 = 0;
    check(unsafe {
        ffi::wb_waveform_to_tokens_sharded(whisper.raw, waveform.as_ptr(), 0, waveform.len() as i64, sample_rate as c_int,
                                           &p, is_special.as_ptr(), rank as c_int, world as c_int,
                                           Some(ffi::wb_comm_allgather_thunk), comm.raw as *mut std::os::raw::c_void,
                                           rows.as_mut_ptr(), stride as i32, lens.as_mut_ptr(), n_win.max(1) as i64,
                                           out.as_mut_ptr(), out.len() as i64, &mut n_out)
    })?;
    let tokens: Vec<usize> = out[..n_out as usize].iter().map(|&t| t as usize).collect();
    let text = bpe.decode(&tokens[..], true)?;
    Ok((text, tokens))
}
